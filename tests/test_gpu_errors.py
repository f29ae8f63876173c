"""The error contract of the boundary: nothing aborts.  A mini-batch whose layer tensors cannot fit in HBM
must come back as ANH_ERR_OOM through the C ABI — and as a POSITIVE exit code from the training tool, because
find_max_mini-batch_size.cmd:43-53,66-67 bisects the largest `-b` on exactly that: positive = "too big, shrink", negative (a
crash) = "stop searching"; annonet_train_main.cpp:616-620 prints the in-loop error and leaves with 2."""
import os
import subprocess

import numpy as np
import pytest

import annonet_amd as aa
import png_util as pu

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = os.path.join(ROOT, "annonet_amd", "lib", "annonet_train_hip")
ANH_ERR_OOM = 2


def test_a_batch_that_cannot_fit_is_ANH_ERR_OOM_and_the_handle_trains_on():
    import torch
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=1)
    t.SetNetWidth(1.0, 1); t.SetClassCount(3); t.Initialize(); t.SetLearningRate(0.05)
    d = 227
    rng = np.random.default_rng(0)
    n_small = 2
    img = torch.from_numpy(rng.integers(0, 256, (n_small, d, d, 3), dtype=np.uint8)).cuda()
    lab = torch.from_numpy(rng.integers(0, 3, (n_small, d, d)).astype(np.int16)).cuda()
    w = torch.ones((n_small, d, d), dtype=torch.float32, device="cuda")
    # ~37 MB of layer tensors per 227^2 tile in bf16: 20,000 tiles would need ~740 GB of the 288 GB.  The pass is refused while its
    # tensors are planned, before any kernel is enqueued, so the (small) input arrays are never read beyond their end.
    with pytest.raises(aa.AnnonetHipError) as err:
        t.forward_backward_device(img.data_ptr(), lab.data_ptr(), w.data_ptr(), 20000, d, d, 20000)
    assert err.value.code == ANH_ERR_OOM and "memory" in str(err.value).lower()
    free_after, total = torch.cuda.mem_get_info()
    assert free_after > 0.8 * total                     # what the failed plan had grown was handed back
    losses = []
    for _ in range(3):                                   # the same handle trains a batch that fits
        t.forward_backward_device(img.data_ptr(), lab.data_ptr(), w.data_ptr(), n_small, d, d, n_small)
        t.apply_update(1.0)
        losses.append(t.get_last_loss())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_training_tool_leaves_with_a_positive_code_when_the_batch_is_too_big(tmp_path):
    rng = np.random.default_rng(3)
    d = tmp_path / "data"
    d.mkdir()
    lab = np.zeros((120, 130), np.uint16)
    lab[30:80, 40:90] = 1
    pu.write_png(d / "a.png", rng.integers(0, 256, (120, 130, 3), dtype=np.uint8))
    pu.write_png(str(d / "a.png") + "_mask.png", pu.labels_to_rgba(lab))
    # default net (width 1.0, crops of 107^2): ~8 MB of layer tensors per crop -> 50,000 crops cannot fit
    r = subprocess.run([TRAIN, str(d), "-b", "50000", "--max-total-steps", "3", "--data-loader-thread-count", "4"], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 2, (r.returncode, r.stdout[-600:], r.stderr[-600:])    # positive: "shrink the batch"; a signal would be negative
    assert "out of memory" in r.stdout.lower()
    r = subprocess.run([TRAIN, str(d), "-b", "8", "--max-total-steps", "3", "--data-loader-thread-count", "2"], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr        # the size the bisection would settle on trains
