"""One tiled annonet_infer() on a seeded image in a fresh process (the tile batch size is an environment switch read once per
process): writes label map and blended planes of a bf16 and an fp32 runtime net.
usage: run_tiled_infer.py out.npz [width scaler = 0.5] [min filters = 4] [classes = 3] [levels = 1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import annonet_amd as aa  # noqa: E402

out = sys.argv[1]
scaler = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
min_filters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
classes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
levels = int(sys.argv[5]) if len(sys.argv) > 5 else 1
rng = np.random.default_rng(11)
H, W = 470, 610
image = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
res = {}
for name, prec in (("bf16", aa.ANH_BF16), ("fp32", aa.ANH_FP32)):
    t = aa.TrainingNet(levels, 3, prec, seed=5)
    t.SetNetWidth(scaler, min_filters); t.SetClassCount(classes); t.Initialize()
    net = t.GetRuntimeNet(prec)
    ov = t.GetRequiredInputDimension()
    tp = aa.tiling.parameters(160, 208, ov, ov)      # 3 x 3 or more tiles of equal size
    gains = [0.0, 0.02, -0.01, 0.015][:classes]
    labels, blended = aa.annonet_infer(net, image, gains=gains, tiling_parameters=tp, want_blended=True)    # resident planes: tiles run in batches
    res[name + "_labels"] = labels
    res[name + "_blended"] = blended
    res[name + "_labels_streamed"] = aa.annonet_infer(net, image, gains=gains, tiling_parameters=tp)      # streamed host form: batches within a tile row
    res[name + "_tiles"] = np.int64(len(aa.tiling.get_tiles(W, H, tp)))
np.savez(out, **res)
