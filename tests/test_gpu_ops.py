"""Kernel-level parity: one conv / cont op at a time, identical inputs on both sides (anh_op_* vs the oracle's orc_op_*).

  fp32 mode           forward and backward-data are the oracle's k-ordered fmaf chains: BIT-EXACT.
                      backward-filter sums over pixels in a different order: rtol 2e-4.
  bf16 mode, generic  same chains on bf16-rounded operands: forward / backward-data BIT-EXACT after rounding.
  bf16 mode, MFMA     fp32 accumulation in MFMA order: |diff| <= one bf16 ulp of the value (rounding-boundary flips),
                      on at most 2% of the elements; backward-filter rtol 2e-3 against double accumulation.
Shapes cover every layer kind of the net (stem 5x5 on 3 channels, 3x3 s1, 3x3 s2, transposed 3x3 s2, 1x1 head), odd
sizes, ragged tile edges, and the prologue kinds (raw / bn+relu / bn+relu with skip add).
"""
import numpy as np
import pytest

import annonet_amd as aa
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

#           type k  s  p  cin cout   n  h   w
SHAPES = [
    ((0, 5, 1, 2, 3, 32), 2, 19, 23),     # stem
    ((0, 3, 1, 1, 32, 32), 2, 21, 37),    # dec0 / enc
    ((0, 3, 1, 1, 64, 64), 1, 18, 33),
    ((0, 3, 1, 1, 128, 128), 1, 9, 35),
    ((0, 3, 2, 0, 32, 64), 2, 23, 31),    # down
    ((0, 3, 2, 0, 64, 128), 1, 15, 19),
    ((1, 3, 2, 0, 128, 64), 1, 7, 9),     # up (transposed)
    ((1, 3, 2, 0, 64, 32), 2, 11, 15),
    ((0, 3, 1, 1, 256, 256), 1, 9, 19),   # levels = 3: 256 channels (four workgroup groups of 64; filter gradient in two tile groups)
    ((0, 3, 2, 0, 128, 256), 1, 13, 17),
    ((1, 3, 2, 0, 256, 128), 1, 5, 7),
    ((0, 1, 1, 0, 32, 3), 2, 17, 13),     # head
    ((0, 3, 1, 1, 8, 16), 1, 12, 10),     # narrow nets (width scaler < 1)
    ((0, 3, 1, 1, 40, 24), 1, 10, 13),    # widths that are not multiples of 32
]


def make_inputs(desc, n, h, w, seed, prologue, bf16):
    rng = np.random.default_rng(seed)
    cin, cout = desc[4], desc[5]
    rnd = orc.bf16_round if bf16 else (lambda a: a.astype(np.float32))
    xa = rnd(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32))
    kw = {}
    if prologue >= 1:
        kw.update(sa=rng.uniform(0.5, 1.5, cin).astype(np.float32), ta=rng.uniform(-0.3, 0.3, cin).astype(np.float32))
    if prologue == 2:
        kw.update(xb=rnd(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32)),
                  sb=rng.uniform(0.5, 1.5, cin).astype(np.float32), tb=rng.uniform(-0.3, 0.3, cin).astype(np.float32))
    k = desc[1]
    lim = np.sqrt(6.0 / (k * k * (cin + cout)))
    filters = rng.uniform(-lim, lim, k * k * cin * cout).astype(np.float32)
    return xa, kw, filters, rng


def assert_bf16_close(got, want, what):
    ulp = np.abs(want) * 2.0 ** -7 + 1e-30
    diff = np.abs(got - want)
    assert (diff <= ulp * 1.001 + 1e-6 * np.abs(want).max()).all(), (what, float((diff / ulp).max()))
    assert (diff > 0).mean() <= 0.02, (what, float((diff > 0).mean()))


@pytest.mark.parametrize("precision", [aa.ANH_FP32, aa.ANH_BF16])
@pytest.mark.parametrize("prologue", [0, 1, 2])
@pytest.mark.parametrize("desc,n,h,w", SHAPES)
def test_conv_forward(desc, n, h, w, prologue, precision):
    bf16 = precision == aa.ANH_BF16
    xa, kw, filters, rng = make_inputs(desc, n, h, w, 1, prologue, bf16)
    bias = rng.uniform(-0.1, 0.1, desc[5]).astype(np.float32) if desc[1] == 1 else None
    want = orc.op_conv_forward(desc, xa, filters=filters, bias=bias, bf16=bf16, **kw)
    got, mfma = aa.op_conv_forward(precision, desc, xa, filters=filters, bias=bias, **kw)
    assert got.shape == want.shape
    if not mfma:
        np.testing.assert_array_equal(got, want)
    else:
        assert_bf16_close(got, want, ("fwd", desc))


@pytest.mark.parametrize("precision", [aa.ANH_FP32, aa.ANH_BF16])
@pytest.mark.parametrize("desc,n,h,w", SHAPES)
def test_conv_backward_data(desc, n, h, w, precision):
    bf16 = precision == aa.ANH_BF16
    _, _, filters, rng = make_inputs(desc, n, h, w, 2, 0, bf16)
    ho, wo = aa.netpimpl._out_dim(desc, h), aa.netpimpl._out_dim(desc, w)
    dy = rng.normal(0, 1e-3, (n, ho, wo, desc[5])).astype(np.float32)
    if bf16:
        dy = orc.bf16_round(dy)
    want = orc.op_conv_backward_data(desc, dy, filters, (h, w), bf16=bf16)
    got, mfma = aa.op_conv_backward_data(precision, desc, dy, filters, (h, w))
    if not mfma:
        np.testing.assert_array_equal(got, want)
    else:
        assert_bf16_close(got, want, ("dgrad", desc))


@pytest.mark.parametrize("precision", [aa.ANH_FP32, aa.ANH_BF16])
@pytest.mark.parametrize("prologue", [0, 2])
@pytest.mark.parametrize("desc,n,h,w", SHAPES)
def test_conv_backward_filter(desc, n, h, w, prologue, precision):
    bf16 = precision == aa.ANH_BF16
    xa, kw, _, rng = make_inputs(desc, n, h, w, 3, prologue, bf16)
    ho, wo = aa.netpimpl._out_dim(desc, h), aa.netpimpl._out_dim(desc, w)
    dy = rng.normal(0, 1e-3, (n, ho, wo, desc[5])).astype(np.float32)
    if bf16:
        dy = orc.bf16_round(dy)
    want = orc.op_conv_backward_filter(desc, xa, dy=dy, bf16=bf16, **kw)
    got, mfma = aa.op_conv_backward_filter(precision, desc, xa, dy=dy, **kw)
    tol = 2e-3 if mfma else 2e-4
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())


def test_ops_reject_bad_arguments():
    x = np.zeros((1, 4, 4, 8), np.float32)
    with pytest.raises(aa.AnnonetHipError):
        aa.op_conv_forward(aa.ANH_FP32, (0, 3, 0, 1, 8, 8), x, filters=np.zeros(9 * 64, np.float32))  # stride 0
    with pytest.raises(aa.AnnonetHipError):
        aa.op_conv_forward(aa.ANH_FP32, (0, 3, 2, 0, 8, 8), x[:, :2, :2], filters=np.zeros(9 * 64, np.float32))  # too small


# ---- the epilogues / prologues the training step fuses into the MFMA kernels -------------------------------------------------
# The statistics are defined on the STORED outputs, so the expected sums are computed (in float64) from the tensor the op
# returns: the check is independent of the conv's own rounding.  Tolerance: fp32 per-lane running sums over <= a few
# thousand values, then double: rtol 2e-5 of the sum of magnitudes.
FUSED_SHAPES = [
    ((0, 3, 1, 1, 32, 32), 3, 21, 37),    # GeoS1, one channel group, ragged edges
    ((0, 3, 1, 1, 64, 64), 2, 18, 33),    # GeoS1, two 32-channel tiles per workgroup
    ((0, 3, 1, 1, 128, 128), 1, 9, 35),   # two workgroup groups of 64 channels
    ((0, 3, 1, 1, 256, 256), 1, 9, 19),   # four groups
    ((0, 3, 2, 0, 32, 64), 2, 23, 31),    # GeoDown
    ((1, 3, 2, 0, 64, 32), 2, 11, 15),    # GeoUp, fused (one channel tile)
    ((1, 3, 2, 0, 128, 64), 1, 7, 9),     # GeoUp with two tiles: fused by the forward-only form of the kernel
    ((1, 3, 2, 0, 256, 128), 1, 6, 7),    # ... as two workgroup groups
]


@pytest.mark.parametrize("prologue", [1, 2])
@pytest.mark.parametrize("desc,n,h,w", FUSED_SHAPES)
def test_conv_forward_fused_bn_statistics(desc, n, h, w, prologue):
    xa, kw, filters, _ = make_inputs(desc, n, h, w, 11, prologue, True)
    y_ref, _ = aa.op_conv_forward(aa.ANH_BF16, desc, xa, filters=filters, **kw)
    y, sums, fused = aa.op_conv_forward_stats(aa.ANH_BF16, desc, xa, filters=filters, **kw)
    assert np.array_equal(y, y_ref)                        # the epilogue must not disturb the result
    y64 = y.astype(np.float64).reshape(-1, desc[5])
    want = np.stack([y64.sum(0), (y64 * y64).sum(0)], 1)
    scale = np.stack([np.abs(y64).sum(0), (y64 * y64).sum(0)], 1) + 1e-12
    assert (np.abs(sums - want) <= 2e-5 * scale).all(), float((np.abs(sums - want) / scale).max())
    assert fused                                           # every geometry carries the sums in its epilogue


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("desc,n,h,w", FUSED_SHAPES)
def test_conv_backward_data_fused_bn_reduction(desc, n, h, w, accumulate):
    rng = np.random.default_rng(5)
    cin, cout = desc[4], desc[5]
    ho, wo = aa.netpimpl._out_dim(desc, h), aa.netpimpl._out_dim(desc, w)
    k = desc[1]
    lim = np.sqrt(6.0 / (k * k * (cin + cout)))
    filters = rng.uniform(-lim, lim, k * k * cin * cout).astype(np.float32)
    dy = orc.bf16_round(rng.normal(0, 1, (n, ho, wo, cout)).astype(np.float32))
    y_prev = orc.bf16_round(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32))
    init = orc.bf16_round(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32)) if accumulate else None
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32); shift = rng.uniform(-0.3, 0.3, cin).astype(np.float32)
    mean = rng.uniform(-0.2, 0.2, cin).astype(np.float32); invstd = rng.uniform(0.7, 1.4, cin).astype(np.float32)
    dx_ref, _ = aa.op_conv_backward_data(aa.ANH_BF16, desc, dy, filters, (h, w))
    dx, sums, fused = aa.op_conv_backward_data_bn(aa.ANH_BF16, desc, dy, filters, (h, w), y_prev, scale, shift, mean, invstd, dx_init=init)
    if accumulate:
        assert np.array_equal(dx, orc.bf16_round(init + dx_ref))   # bf16(old + bf16(conv)): the stored skip-gradient sum
    else:
        assert np.array_equal(dx, dx_ref)
    # expected sums from the stored dx, with the kernels' fp32 mask and xhat expressions
    z = y_prev.astype(np.float64) * scale + shift                    # the sign of the kernels' single-rounding fmaf
    dz = np.where(z > 0, dx, 0).astype(np.float64)
    xhat = ((y_prev - mean).astype(np.float32) * invstd).astype(np.float64)
    want = np.stack([(dz * xhat).reshape(-1, cin).sum(0), dz.reshape(-1, cin).sum(0)], 1)
    mag = np.stack([np.abs(dz * xhat).reshape(-1, cin).sum(0), np.abs(dz).reshape(-1, cin).sum(0)], 1) + 1e-12
    assert (np.abs(sums - want) <= 1e-4 * mag).all(), float((np.abs(sums - want) / mag).max())
    assert isinstance(fused, bool)


def test_stem_filter_gradient_computes_dy_in_kernel():
    desc, n, h, w = (0, 5, 1, 2, 3, 32), 3, 21, 43
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    da = orc.bf16_round(rng.normal(0, 1, (n, h, w, 32)).astype(np.float32))
    y = orc.bf16_round(rng.normal(0, 1, (n, h, w, 32)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32); shift = rng.uniform(-0.3, 0.3, 32).astype(np.float32)
    mean = rng.uniform(-0.2, 0.2, 32).astype(np.float32); invstd = rng.uniform(0.7, 1.4, 32).astype(np.float32)
    coef = np.concatenate([rng.uniform(0.5, 1.5, 32), rng.uniform(-0.01, 0.01, 32), rng.uniform(-0.01, 0.01, 32)]).astype(np.float32)
    dw, in_kernel = aa.op_conv_backward_filter_bn(aa.ANH_BF16, desc, img, da, y, scale, shift, mean, invstd, coef)
    assert in_kernel
    # reference: dy materialised with the bn_bwd_apply expression (fp32, bf16-rounded as stored), then the plain stem wgrad
    z = y.astype(np.float64) * scale + shift
    dz = np.where(z > 0, da, 0).astype(np.float32)
    xhat = ((y - mean).astype(np.float32) * invstd).astype(np.float32)
    dyv = orc.bf16_round((coef[:32] * ((dz - coef[32:64]).astype(np.float32) - (xhat * coef[64:]).astype(np.float32))).astype(np.float32))
    x = img.astype(np.float32) / 256.0
    want = orc.op_conv_backward_filter(desc, x, dy=dyv)
    tol = 2e-3 * np.abs(want).max() + 1e-6
    assert np.abs(dw - want).max() <= tol, float(np.abs(dw - want).max() / tol)


def test_256_channel_layers_take_the_mfma_kernels():
    """levels = 3 at width 1.0 reaches 256 channels: the three layer kinds must not fall back to the generic kernels."""
    for desc, n, h, w in [((0, 3, 1, 1, 256, 256), 1, 9, 19), ((0, 3, 2, 0, 128, 256), 1, 13, 17), ((1, 3, 2, 0, 256, 128), 1, 5, 7)]:
        xa, kw, filters, rng = make_inputs(desc, n, h, w, 4, 1, True)
        y, mfma_f = aa.op_conv_forward(aa.ANH_BF16, desc, xa, filters=filters, **kw)
        dx, mfma_b = aa.op_conv_backward_data(aa.ANH_BF16, desc, orc.bf16_round(rng.normal(0, 1, y.shape).astype(np.float32)), filters, (h, w))
        dw, mfma_w = aa.op_conv_backward_filter(aa.ANH_BF16, desc, xa, dy=orc.bf16_round(rng.normal(0, 1e-3, y.shape).astype(np.float32)), **kw)
        assert mfma_f and mfma_b and mfma_w, (desc, mfma_f, mfma_b, mfma_w)
