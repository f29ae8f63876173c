#!/usr/bin/env python3
"""Rate of the HOST-buffer annonet_infer() (image and label map in pageable host memory, as annonet_infer_main.cpp calls it):
PCIe-inclusive; bench.py --mode infer has the image and the label map resident in HBM."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import annonet_amd as aa  # noqa: E402

t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=2)
t.SetNetWidth(1.0, 1); t.SetClassCount(3); t.Initialize()
net = t.GetRuntimeNet(aa.ANH_BF16)     # random-init weights of the BASELINE net
del t
rng = np.random.default_rng(3)
side = 4096
img = rng.integers(0, 256, (side, side, 3), dtype=np.uint8)
tp = aa.tiling.parameters(1024, 1024, 35, 35)
for _ in range(2):
    out = aa.annonet_infer(net, img, tiling_parameters=tp)
t0 = time.perf_counter()
for _ in range(5):
    out = aa.annonet_infer(net, img, tiling_parameters=tp)
dt = (time.perf_counter() - t0) / 5
print(f"annonet_infer host path 4096^2: {dt*1e3:.1f} ms, {side*side/dt/1e6:.0f} Mpx/s")
