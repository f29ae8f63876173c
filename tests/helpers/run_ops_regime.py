"""Child process of tests/test_gpu_ops_regime.py: the MFMA cases of tests/test_gpu_ops.py, run in ONE process whose environment
shrinks the persistent kernels' grids (ANH_WS_WGS / ANH_WGRAD_WGS / ANH_STEM_BLOCKS / ANH_STEM_WGRAD_BLOCKS are read once per
process), so that every workgroup walks many (pixel tile, slab) items: ring buffers wrap, the XCD-band walk (grid 8) and the
grid-stride walk (grid 3) both run, the producer-issued stores / sums of "two items ago" and the tile-ahead old-value prefetch
reach their steady state.  The bars are those of test_gpu_ops.py (one bf16 ulp on <= 2 % of the elements against orc_op_*; filter
gradients rtol 2e-3; fused sums 2e-5 / 1e-4 of float64 sums over the stored tensor) — the functions themselves are reused.

usage: run_ops_regime.py [small|full]      prints one line per case, exits 1 if any case failed
"""
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

import annonet_amd as aa  # noqa: E402
import test_gpu_ops as ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402

BF = aa.ANH_BF16

# every geometry of the persistent kernels at sizes where a grid of 3 / 8 workgroups walks >= 4 items each
#            type k  s  p  cin cout    n   h   w
REGIME = [
    ((0, 3, 1, 1, 32, 32), 3, 40, 70),      # GeoS1, one slab, filter fragments in registers: 3 x 5 x 3 = 45 tiles
    ((0, 3, 1, 1, 64, 64), 2, 33, 65),      # GeoS1, resident filter slabs: 30 tiles x 2 slabs
    ((0, 3, 1, 1, 128, 128), 1, 30, 56),    # two workgroup groups of 64 channels, 4 slabs: the side-56 layer's row shape
    ((0, 3, 2, 0, 32, 64), 2, 47, 67),      # GeoDown forward / GeoUp backward-data (four accumulator groups, read-modify-write prefetch)
    ((0, 3, 2, 0, 64, 128), 1, 41, 71),
    ((1, 3, 2, 0, 128, 64), 1, 20, 35),     # cont: GeoUp forward / GeoDown backward-data
    ((1, 3, 2, 0, 64, 32), 2, 23, 33),
]
# (b) full-width shapes at the benchmark's real plane size (8 tiles across a 227-pixel row, ragged last tile, band walk over 464 tiles)
FULL = [
    ((0, 3, 1, 1, 32, 32), 2, 227, 227),    # dec0: skip-add prologue forward, backward-data with the fused sums, filter gradient
    ((1, 3, 2, 0, 64, 32), 2, 113, 113),    # up1: 113 -> 227; its backward-data is the stride-2 gather form at full width
    ((0, 3, 2, 0, 32, 64), 2, 227, 227),    # down1: its backward-data accumulates into the skip gradient (dx_init) at side 227
]


def run(name, fn, *args):
    try:
        fn(*args)
        print("ok   ", name, flush=True)
        return 0
    except Exception:   # an assertion of the reused test, or an error status of the library
        print("FAIL ", name, flush=True)
        traceback.print_exc()
        return 1


def must_be_mfma(desc, n, h, w):
    xa, kw, filters, rng = ops.make_inputs(desc, n, h, w, 4, 1, True)
    _, mf = aa.op_conv_forward(BF, desc, xa, filters=filters, **kw)
    assert mf, ("not on the MFMA kernels", desc)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    shapes = REGIME if which == "small" else FULL
    bad = 0
    for desc, n, h, w in shapes:
        tag = "%s n%d %dx%d" % (desc, n, h, w)
        bad += run("mfma path " + tag, must_be_mfma, desc, n, h, w)
        for prologue in (0, 1, 2):
            bad += run("forward p%d %s" % (prologue, tag), ops.test_conv_forward, desc, n, h, w, prologue, BF)
        bad += run("backward-data " + tag, ops.test_conv_backward_data, desc, n, h, w, BF)
        for prologue in (0, 2):
            bad += run("backward-filter p%d %s" % (prologue, tag), ops.test_conv_backward_filter, desc, n, h, w, prologue, BF)
        for prologue in (1, 2):
            bad += run("forward + bn statistics p%d %s" % (prologue, tag), ops.test_conv_forward_fused_bn_statistics, desc, n, h, w, prologue)
        for accumulate in (False, True):
            bad += run("backward-data + bn sums acc=%d %s" % (accumulate, tag), ops.test_conv_backward_data_fused_bn_reduction, desc, n, h, w, accumulate)
    if which == "small":
        # the stem kernels are persistent too (ANH_STEM_BLOCKS / ANH_STEM_WGRAD_BLOCKS)
        stem = ((0, 5, 1, 2, 3, 32), 3, 45, 70)
        for prologue in (0,):
            bad += run("stem forward", ops.test_conv_forward, *stem, prologue, BF)
        bad += run("stem filter gradient with dy in the kernel", ops.test_stem_filter_gradient_computes_dy_in_kernel)
        bad += run("stem filter gradient", ops.test_conv_backward_filter, *stem, 0, BF)
    print("failed cases: %d" % bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
