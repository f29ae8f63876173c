#!/usr/bin/env python3
"""Throughput of training on crops cut on the device (anh_trainer_step_crops): the dataset's full images are resident in
HBM, every step hands 32 crop SPECS (image, rectangle, flips, brightness) to the library — no mini-batch on the host, no
per-step PCIe upload.  Also times the reference-shaped alternative on the same crops: cut + weighted on the host by the
same steps done with numpy + the library's host set_weights, then StartTraining."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import annonet_amd as aa  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(0)
    n, d, side, images = 32, 227, 1024, 48
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=2)
    t.SetNetWidth(1.0, 1); t.SetClassCount(3); t.Initialize(); t.SetLearningRate(0.1)
    ds = aa.Dataset(3)
    full = []
    for _ in range(images):
        img = rng.integers(0, 256, (side, side, 3), dtype=np.uint8)
        lab = np.kron(rng.integers(0, 3, (side // 32, side // 32)), np.ones((32, 32), dtype=np.int64)).astype(np.uint16)
        ds.add(img, lab)
        full.append((img, lab))

    def specs():
        return [(int(rng.integers(0, images)), int(rng.integers(-60, side - d + 60)), int(rng.integers(-60, side - d + 60)),
                 int(rng.random() < 0.5), int(rng.random() < 0.5), float(np.exp(rng.normal() * 0.1)) if rng.random() < 0.5 else 1.0) for _ in range(n)]

    batches = [specs() for _ in range(16)]
    for k in range(5):
        t.StartTrainingOnCrops(ds, batches[k % 16], d)
    t.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        t.StartTrainingOnCrops(ds, batches[k % 16], d)
    t.synchronize()
    dt = time.perf_counter() - t0
    print(f"device-crop path: {1e3 * dt / steps:.3f} ms/step, {n * steps / dt:.0f} tiles/s ({images} images of {side}x{side} resident)")

    # the host alternative: one thread cutting + weighting a batch with numpy + set_weights
    def host_batch(sp):
        out_i, out_l = [], []
        for (i, left, top, flr, fud, gain) in sp:
            ys = np.clip(np.arange(top, top + d), 0, side - 1); xs = np.clip(np.arange(left, left + d), 0, side - 1)
            chip = full[i][0][np.ix_(ys, xs)]
            lab = full[i][1][np.ix_(ys, xs)].copy()
            lab[(np.arange(top, top + d) < 0) | (np.arange(top, top + d) >= side), :] = aa.LABEL_IGNORE
            lab[:, (np.arange(left, left + d) < 0) | (np.arange(left, left + d) >= side)] = aa.LABEL_IGNORE
            wl = aa.set_weights(lab, 0.5, 0.5)
            if flr: chip, wl = chip[:, ::-1], wl[:, ::-1]
            if fud: chip, wl = chip[::-1], wl[::-1]
            if gain != 1.0: chip = np.floor(np.clip(chip * gain, 0, 255) + 0.5).astype(np.uint8)
            out_i.append(np.ascontiguousarray(chip)); out_l.append(np.ascontiguousarray(wl))
        return out_i, out_l
    t0 = time.perf_counter()
    for k in range(5):
        host_batch(batches[k])
    dt = time.perf_counter() - t0
    print(f"host cut + set_weights (1 thread, numpy): {1e3 * dt / 5:.2f} ms per batch of {n} = {n * 5 / dt:.0f} crops/s/core")


if __name__ == "__main__":
    main()
