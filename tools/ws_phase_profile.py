"""Per-phase wall-clock accounting of the warp-specialised conv kernels over one BASELINE training step.
Needs the library built with -DANH_WS_PROFILE (the phase counters are compiled out otherwise):
    make -C annonet_amd/csrc clean && make -C annonet_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DANH_WS_PROFILE"
    ANH_WS_PROF=1 python tools/ws_phase_profile.py 2> phases.txt
Each conv launch then prints one "[ws prof]" line (producer: commit / fetch / barrier, consumer: mfma / store / barrier, in us per wave).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import annonet_amd as aa  # noqa: E402

n, d = 32, 227
rng = np.random.default_rng(0)
t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=1)
t.SetNetWidth(1.0, 1)
t.SetClassCount(2)
t.Initialize()
img = rng.integers(0, 256, (n, d, d, 3), dtype=np.uint8)
lab = rng.integers(0, 2, (n, d, d)).astype(np.uint16)
wl = [aa.set_weights(l, 0.5, 0.5) for l in lab]
for i in range(3):
    print(f"--- step {i}", file=sys.stderr, flush=True)
    t.StartTraining(list(img), wl)
    t.synchronize()
