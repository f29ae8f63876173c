#!/usr/bin/env python3
"""Throughput of the HOST-buffer training path (NetPimpl::TrainingNet::StartTraining = anh_trainer_step): every step hands
32 images + 32 weighted-label images in host memory to the library, as annonet_train_main.cpp:583-614 does.
PCIe-inclusive; not the bench.py value (which has its inputs resident in HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import annonet_amd as aa  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(0)
    n, d = 32, 227
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=2)
    t.SetNetWidth(1.0, 1)
    t.SetClassCount(3)
    t.Initialize()
    t.SetLearningRate(0.1)
    img = list(rng.integers(0, 256, (n, d, d, 3), dtype=np.uint8))
    lab = rng.integers(0, 3, (n, d, d)).astype(np.uint16)
    wl = [aa.set_weights(l, 0.5, 0.5) for l in lab]
    for _ in range(5):
        t.StartTraining(img, wl)
    t.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t.StartTraining(img, wl)
    t.synchronize()
    dt = time.perf_counter() - t0
    print(f"host-buffer path: {1e3 * dt / steps:.3f} ms/step, {n * steps / dt:.0f} tiles/s (PCIe-inclusive)")


if __name__ == "__main__":
    main()
