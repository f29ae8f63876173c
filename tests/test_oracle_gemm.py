"""The oracle's im2col + blocked-SGEMM path (conv_algo = 1: the CPU design SURVEY.md §8d names, dlib's cpu_dlib.cpp + BLAS,
/root/reference/annonet_train_cpu.vcxproj:93,113,230 — the form bench.py times as cpu_baseline) against the oracle's direct loops."""
import numpy as np
import pytest

from conftest import random_params
from oracle.oracle import OracleNet


def _pair(levels, in_ch, classes, scaler, min_filters, seed):
    a, b = OracleNet(levels, in_ch, classes, scaler, min_filters), OracleNet(levels, in_ch, classes, scaler, min_filters)
    p, r = random_params(a, seed)
    for o in (a, b):
        o.params[:], o.running[:] = p, r
        o.set_hyper(lr=0.05)
    b.set_conv_algorithm(1)
    return a, b


@pytest.mark.parametrize("levels,in_ch,classes,scaler,minf,side,n", [
    (2, 3, 3, 0.25, 4, 43, 2),     # every layer kind: 5x5 stem, stride-2 con, cont, skip adds, 1x1 head
    (1, 1, 2, 0.5, 8, 27, 3),      # grayscale, one level
    (2, 3, 4, 1.0, 1, 35, 1),      # the benchmark net's widths (32 / 64 / 128), one tile
    (0, 3, 3, 0.5, 5, 17, 2),      # no level; widths that are not a multiple of 8 (scalar GEMM remainders)
])
def test_forward_and_training_step_agree_with_the_direct_path(levels, in_ch, classes, scaler, minf, side, n):
    a, b = _pair(levels, in_ch, classes, scaler, minf, 3)
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (n, side, side, in_ch), dtype=np.uint8)
    fa, fb = a.forward(img), b.forward(img)
    scale = np.abs(fa).max()
    assert np.abs(fa - fb).max() <= 1e-5 * scale
    lab = rng.integers(0, classes, (n, side, side)).astype(np.uint16)
    lab[rng.random(lab.shape) < 0.05] = 65535
    w = rng.random((n, side, side), dtype=np.float32) + 0.5
    la, lb = a.train_step(img, lab, w), b.train_step(img, lab, w)
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(la))
    ga, gb = np.array(a.grads), np.array(b.grads)
    for L in a.layers:   # per parameter segment: <= 1e-5 of the segment's largest gradient
        for off, cnt in ((L.w_off, L.k * L.k * L.cin * L.cout), (L.g_off, L.cout if L.has_bn else 0), (L.beta_off, L.cout if L.has_bn else 0), (L.b_off, L.cout if L.has_bias else 0)):
            if cnt:
                seg_a, seg_b = ga[off:off + cnt], gb[off:off + cnt]
                assert np.abs(seg_a - seg_b).max() <= 2e-5 * max(np.abs(seg_a).max(), 1e-12), (L.k, L.cin, L.cout, off)
    assert np.abs(np.array(a.params) - np.array(b.params)).max() <= 1e-6
    assert np.abs(np.array(a.running) - np.array(b.running)).max() <= 1e-6


def test_con_forward_is_the_same_fmaf_chain():
    """A con layer's GEMM runs over (tap, input channel) in the direct loop's order, padding taps as exact +0 terms: bit-identical."""
    a, b = _pair(0, 3, 3, 1.0, 1, 5)
    img = np.random.default_rng(2).integers(0, 256, (1, 19, 23, 3), dtype=np.uint8)
    a.forward(img), b.forward(img)
    assert np.array_equal(a.layer_output(0), b.layer_output(0))
