#!/usr/bin/env python3
"""Fixed cost and per-tile cost of every conv / filter-gradient launch of the training step: the step is run at several batch sizes
(one-stream schedule, so that a launch is measured alone), and each profiler entry's time is fitted as  t(n) = F + c * n.
usage: ANH_CONCURRENT_WGRAD=0 python tools/launch_fit.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("ANH_CONCURRENT_WGRAD", "0")
import annonet_amd as aa  # noqa: E402

d = 227
rng = np.random.default_rng(0)
batches = (8, 16, 32, 64)
times = {}
for n in batches:
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=1)
    t.SetNetWidth(1.0, 1); t.SetClassCount(3); t.Initialize()
    img = rng.integers(0, 256, (n, d, d, 3), dtype=np.uint8)
    lab = rng.integers(0, 3, (n, d, d)).astype(np.uint16)
    wl = [aa.set_weights(l, 0.5, 0.5) for l in lab]
    for _ in range(3):
        t.StartTraining(list(img), wl)
    t.synchronize()
    t.profile_enable(True)
    for _ in range(6):
        t.StartTraining(list(img), wl)
    t.synchronize()
    for e in t.profile():
        if e["launches"]:
            times.setdefault(e["name"], {})[n] = 1e3 * e["total_ms"] / e["launches"]
    t.profile_enable(False)
    del t
print(f"{'entry':48s} " + " ".join(f"n={n:<4d}" for n in batches) + "   F us   c us/tile-of-227^2")
for name, v in times.items():
    if len(v) != len(batches) or not ("conv_mfma" in name or "wgrad_mfma" in name or name in ("bn_bwd_apply", "head_fused_fwd_loss_bwd")):
        continue
    x = np.array(batches, float); y = np.array([v[n] for n in batches])
    c, F = np.polyfit(x, y, 1)
    print(f"{name:48s} " + " ".join(f"{v[n]:6.1f}" for n in batches) + f"  {F:6.1f}  {c:6.3f}")
