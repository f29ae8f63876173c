"""A fresh trainer's FIRST step over HBM that held junk.

Round 3 found a stale cross-workgroup read by an accident of test order: the fused head kernel's last workgroup took `invstd` of the
last hidden layer with a plain load from an array a fold-job workgroup of the SAME launch writes (another XCD's L2 may still serve
what memory held before).  On zeroed memory that made every gradient below the last hidden layer exactly zero on a trainer's first
step; on recycled memory that held the right numbers it hid.  This case removes the accident: the allocator's free blocks are filled
with NaN patterns and random bits, released, and ONE fresh bf16 trainer of the benchmark net runs ONE step on them; every layer's
gradient is held to the bf16-restating oracle and must be finite and non-zero.  Run once — a fault found here is to be explained from
the failure, not re-rolled (DESIGN.md §5 lists every in-kernel "last workgroup" finish and what it reads)."""
import numpy as np
import pytest

import annonet_amd as aa
from conftest import random_params
from oracle.oracle import OracleNet, IGNORE

pytestmark = pytest.mark.gpu


def test_first_step_of_a_fresh_trainer_on_dirtied_memory_matches_the_oracle():
    import torch
    dev = torch.device("cuda:0")
    # dirty what the library's hipMalloc calls will be handed next: several block sizes, NaN patterns and random bits
    junk = []
    for i, words in enumerate((6e7, 3e7, 1e7, 5e6, 2e6, 1e6, 3e5, 1e5, 3e4)):
        if i % 2: junk.append(torch.full((int(words),), float("nan"), device=dev))
        else: junk.append(torch.randint(-2 ** 31, 2 ** 31 - 1, (int(words),), device=dev, dtype=torch.int32))
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()

    levels, classes, n, d = 2, 3, 4, 67
    o = OracleNet(levels, 3, classes, 1.0, 1)
    p, r = random_params(o, 19)
    o.params[:], o.running[:] = p, r
    o.set_hyper(lr=0.05, wd=0.0005, mom=0.9, bn_window=100)
    o.set_bf16_emulation(True)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (n, d, d, 3), dtype=np.uint8)
    lab = rng.integers(0, classes, (n, d, d)).astype(np.uint16)
    lab[rng.random(lab.shape) < 0.05] = IGNORE
    wl = [aa.set_weights(lab[i], 0.5, 0.5) for i in range(n)]
    w = np.stack([x["weight"] for x in wl])
    want_loss = o.train_step(img, lab, w, apply_update=False)

    t = aa.TrainingNet(levels, 3, aa.ANH_BF16)
    t.SetNetWidth(1.0, 1); t.SetClassCount(classes); t.Initialize(); t.SetLearningRate(0.05)
    t.set_params(p, r)
    t.StartTraining(list(img), wl)                       # the handle's first step
    got_loss = t.get_last_loss()
    assert abs(got_loss - want_loss) <= 2e-3 * max(1.0, abs(want_loss))
    g, gw = t.get_grads(), np.array(o.grads)
    assert np.isfinite(g).all()
    for li, L in enumerate(o.layers):
        nw = L.k * L.k * L.cin * L.cout
        a, b = g[L.w_off:L.w_off + nw], gw[L.w_off:L.w_off + nw]
        assert np.abs(a).max() > 0, f"layer {li}: filter gradient is exactly zero"
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-20)
        cos = float(a @ b) / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30)
        assert rel < 0.175 and cos > 0.985, (li, L.cin, L.cout, L.k, rel, cos)
        if L.has_bn:
            for off in (L.g_off, L.beta_off):
                ga, gb = g[off:off + L.cout], gw[off:off + L.cout]
                assert np.abs(ga).max() > 0, f"layer {li}: bn gradient is exactly zero"
                assert np.linalg.norm(ga - gb) <= 0.175 * max(np.linalg.norm(gb), 1e-20), (li, off)
