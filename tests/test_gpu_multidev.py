"""anh_set_devices: ONE process driving several replicas through the C ABI (include/annonet_hip.h) — the path a C++ host takes
(INTEGRATION.md).  A one-GPU box rehearses it with a device list that repeats device 0: every replica is a full engine of its
own, the exchange step runs on the rehearsal backend (fixed-order sum) instead of RCCL; with distinct devices the same code
issues one grouped ncclAllReduce.
  * training: the mini-batch is split along N, the loss scale is the whole batch, the bucket after the exchange is the SUM of
    the replicas' gradients (= a single process with per-replica batch-norm groups), every replica ends with identical weights;
  * inference: the tile list is split, the overlap sums are exchanged, ONE host label map comes back — equal to the oracle's
    except at exact near-ties of the blended planes (summation order inside 4-tile corners)."""
import numpy as np
import pytest
import torch

import annonet_amd as aa
from conftest import random_params
from oracle.oracle import OracleNet

pytestmark = pytest.mark.gpu


@pytest.fixture
def two_replicas():
    aa.set_devices([0, 0])
    yield
    aa.set_devices([])


def trainer(seed=3, precision=aa.ANH_FP32):
    t = aa.TrainingNet(1, 3, precision, seed=seed)
    t.Initialize()
    t.SetNetWidth(0.25, 4)
    t.SetClassCount(3)
    t.SetLearningRate(0.05)
    return t


def batch(n, d, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (n, d, d, 3), dtype=np.uint8)
    lab = rng.integers(0, 3, (n, d, d)).astype(np.uint16)
    lab[rng.random((n, d, d)) < 0.1] = aa.LABEL_IGNORE
    return img, lab, [aa.set_weights(l, 0.5, 0.5) for l in lab]


def test_two_replica_training_step_is_the_sum_of_the_shards(two_replicas):
    n, d = 5, 31          # odd batch: shards of 2 and 3 samples
    img, lab, wl = batch(n, d)
    m = trainer()
    assert m.replicas() == 2
    p0, r0 = m.get_params()
    m.StartTraining(list(img), wl)
    m.synchronize()
    got_bucket = m.get_grads()
    # the same two shards through single-device trainers (loss scale = the WHOLE batch), gradients summed on the host
    aa.set_devices([])
    shard_grads = []
    want_loss = 0.0
    for r in range(2):
        lo, hi = aa.shard_range(n, 2, r)
        s = trainer()
        assert s.replicas() == 1
        s.set_params(p0, r0)
        d_img = torch.from_numpy(img[lo:hi].copy()).cuda()
        d_lab = torch.from_numpy(lab[lo:hi].view(np.int16).copy()).cuda()
        d_w = torch.from_numpy(np.stack([w["weight"] for w in wl[lo:hi]])).cuda()
        s.forward_backward_device(d_img.data_ptr(), d_lab.data_ptr(), d_w.data_ptr(), hi - lo, d, d, n)
        s.synchronize()
        shard_grads.append(s.get_grads())
        want_loss += s.get_last_loss()
    np.testing.assert_allclose(got_bucket, shard_grads[0] + shard_grads[1], rtol=1e-6, atol=1e-9)   # a + b in fp32 (canonical vs tap-major order of the add: last bit)
    assert abs(m.get_last_loss() - want_loss) <= 1e-6 * abs(want_loss)
    np.testing.assert_array_equal(m.replica_params(0), m.replica_params(1))   # identical update on every replica
    assert np.abs(m.replica_params(0) - p0).max() > 0


def test_two_replica_training_runs_several_steps_and_stays_in_lockstep(two_replicas):
    m = trainer(precision=aa.ANH_BF16)
    losses = []
    for i in range(4):
        img, lab, wl = batch(4, 31, seed=i)
        m.StartTraining(list(img), wl)
        losses.append(m.get_last_loss())
    assert all(np.isfinite(losses))
    np.testing.assert_array_equal(m.replica_params(0), m.replica_params(1))
    rt = m.GetRuntimeNet(aa.ANH_FP32)      # a snapshot is ONE replica on the trainer's first device (the host only serializes it)
    assert rt.L.anh_handle_replicas(rt.h, 0) == 1
    p, _ = rt.get_params()
    np.testing.assert_array_equal(p, m.replica_params(0))


@pytest.mark.parametrize("world", [2, 3])
def test_multi_replica_annonet_infer_returns_one_label_map(world):
    o = OracleNet(1, 3, 3, 0.25, 4)
    p, r = random_params(o, 7)
    o.params[:], o.running[:] = p, r
    ov = o.required_input_dim()
    rng = np.random.default_rng(5)
    image = rng.integers(0, 256, (230, 301, 3), dtype=np.uint8)
    tp = aa.tiling.parameters(96, 96, ov, ov)
    single = aa.RuntimeNet(aa.net_config(1, 3, 3, 0.25, 4, aa.ANH_FP32))
    single.set_params(p, r)
    want_labels, want_planes = aa.annonet_infer(single, image, tiling_parameters=tp, want_blended=True)
    aa.set_devices([0] * world)
    try:
        net = aa.RuntimeNet(aa.net_config(1, 3, 3, 0.25, 4, aa.ANH_FP32))
        assert net.L.anh_handle_replicas(net.h, 0) == world
        net.set_params(p, r)
        labels, planes = aa.annonet_infer(net, image, tiling_parameters=tp, want_blended=True)
        labels_only = aa.annonet_infer(net, image, tiling_parameters=tp)
    finally:
        aa.set_devices([])
    np.testing.assert_array_equal(labels_only, labels)
    scale = np.abs(want_planes).max()
    assert np.abs(planes - want_planes).max() <= 2e-6 * scale          # summation order where four tiles meet
    differ = labels != want_labels
    assert differ.mean() < 1e-4
    top2 = np.sort(want_planes, axis=0)[-2:]
    assert np.all((top2[1] - top2[0])[differ] <= 4e-6 * scale)          # only at exact near-ties
    ref = o.infer(image, max_tile=(96, 96), overlap=ov)
    assert (labels != ref).mean() < 1e-4
