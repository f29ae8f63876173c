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
for i in range(0 if len(sys.argv) > 1 and sys.argv[1] == "infer" else 3):
    print(f"--- step {i}", file=sys.stderr, flush=True)
    t.StartTraining(list(img), wl)
    t.synchronize()


def infer_pass():
    """the same accounting for the inference forms: one 4096^2 image, 1024^2 tiles (batches of 8 windows of 851^2)"""
    tr = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=2)
    tr.SetNetWidth(1.0, 1); tr.SetClassCount(3); tr.Initialize()
    net = tr.GetRuntimeNet(aa.ANH_BF16)
    image = rng.integers(0, 256, (4096, 4096, 3), dtype=np.uint8)
    for i in range(2):
        print(f"--- infer pass {i}", file=sys.stderr, flush=True)
        aa.annonet_infer(net, image, tiling_parameters=aa.tiling.parameters(1024, 1024, 35, 35))


if len(sys.argv) > 1 and sys.argv[1] == "infer":
    infer_pass()
