#!/usr/bin/env python3
"""Runs the same three bf16 training steps on fresh trainers over and over in ONE process, with the freed HBM in between filled with
garbage (NaN bit patterns / random bytes), and compares the parameters with the first run: a read of memory the step did not write,
or a race, shows as a mismatch.     usage: python tools/determinism_probe.py [repeats] [tile side] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import annonet_amd as aa

R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
D = int(sys.argv[2]) if len(sys.argv) > 2 else 99
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(12)
img = rng.integers(0, 256, (N, D, D, 3), dtype=np.uint8)
lab = rng.integers(0, 3, (N, D, D)).astype(np.uint16)
w = np.ones((N, D, D), np.float32)
dev = torch.device("cuda:0")
timg, tlab, tw = (torch.from_numpy(a).to(dev) for a in (img, lab.view(np.int16), w))


def one():
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=3)
    t.SetNetWidth(1.0, 1); t.SetClassCount(3); t.Initialize(); t.SetLearningRate(0.05)
    for _ in range(3):
        t.forward_backward_device(timg.data_ptr(), tlab.data_ptr(), tw.data_ptr(), N, D, D, N)
        t.apply_update(1.0)
    t.synchronize()
    p = t.get_params()
    del t
    return p


first = one()
bad = 0
for r in range(R):
    # dirty the allocator's free blocks: whatever the next trainer gets is not zero
    junk = [torch.full((int(s),), float("nan"), device=dev) if r % 2 else torch.randint(0, 2 ** 31 - 1, (int(s),), device=dev, dtype=torch.int32)
            for s in (3e7, 1e7, 5e6, 1e6, 3e5)]
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    got = one()
    same = np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1])
    if not same:
        bad += 1
        d = np.abs(got[0] - first[0])
        print(f"run {r}: parameters differ: {int((d > 0).sum())} of {d.size}, max {float(np.nanmax(d)):.3g}, nan {int(np.isnan(got[0]).sum())}", flush=True)
print(f"{bad} of {R} runs differ from the first")
sys.exit(1 if bad else 0)
