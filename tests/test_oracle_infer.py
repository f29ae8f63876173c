"""Tiler contract, clamp-to-edge crop, blending and argmax of the oracle's annonet_infer() restatement
(annonet_infer.cpp:32-240, annonet.h:74-120).  Hand-computable cases; no reference fixture exists."""
import numpy as np
import pytest

from conftest import random_params
from oracle import oracle as orc
from oracle.oracle import OracleNet


def check_tiles(get_tiles, width, height, mw, mh, ox, oy):
    tiles = get_tiles(width, height, mw, mh, ox, oy)
    cover = np.zeros((height, width), np.int32)
    uniq = np.zeros((height, width), np.int32)
    for (fl, ft, fr, fb), (ul, ut, ur, ub) in tiles:
        assert 0 <= fl <= fr < width and 0 <= ft <= fb < height
        assert fr - fl + 1 <= mw and fb - ft + 1 <= mh
        assert fl <= ul <= ur <= fr and ft <= ut <= ub <= fb
        cover[ft:fb + 1, fl:fr + 1] += 1
        uniq[ut:ub + 1, ul:ur + 1] += 1
    assert cover.min() >= 1                      # full rects cover the image
    assert uniq.max() <= 1                       # unique rects are disjoint
    assert ((cover == 1) == (uniq == 1)).all()   # unique = covered by exactly one tile
    if len(tiles) == 1:
        assert tiles[0][0] == tiles[0][1] == (0, 0, width - 1, height - 1)
    return tiles


@pytest.mark.parametrize("w,h,mw,mh,ov", [
    (227, 227, 1024, 1024, 35), (1, 1, 64, 64, 5), (20, 300, 100, 100, 35), (4096, 4096, 1024, 1024, 35),
    (1025, 1024, 1024, 1024, 35), (1000, 777, 128, 96, 15), (300, 300, 120, 110, 35), (5, 5, 1024, 1024, 35),
])
def test_tiler_contract(w, h, mw, mh, ov):
    check_tiles(orc.get_tiles, w, h, mw, mh, ov, ov)


def test_tiler_overlap_at_least_requested():
    tiles = orc.get_tiles(4096, 100, 1024, 1024, 35, 35)
    xs = sorted({(t[0][0], t[0][2]) for t in tiles})
    for (l0, r0), (l1, r1) in zip(xs, xs[1:]):
        assert r0 - l1 + 1 >= 35


def test_tiler_rejects_tile_smaller_than_twice_overlap():
    with pytest.raises(RuntimeError):
        orc.get_tiles(500, 500, 60, 60, 35, 35)
    with pytest.raises(RuntimeError):
        orc.get_tiles(300, 300, 71, 71, 35, 35)  # would leave a tile without a unique part


def test_outpaint_replicates_edges():
    img = np.arange(6 * 7, dtype=np.uint8).reshape(6, 7)
    out = orc.outpaint(img, (2, 1, 4, 3))  # l,t,r,b
    ys = np.clip(np.arange(6), 1, 3)
    xs = np.clip(np.arange(7), 2, 4)
    np.testing.assert_array_equal(out, img[np.ix_(ys, xs)])
    np.testing.assert_array_equal(orc.outpaint(img, (9, 9, 12, 12)), img)  # empty intersection: untouched


def make_net(levels=1, seed=5):
    net = OracleNet(levels, 3, 3, 0.25, 4)
    p, r = random_params(net, seed)
    net.params[:] = p
    net.running[:] = r
    return net


def test_single_tile_is_plain_forward_plus_argmax():
    net = make_net(2)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)  # 23 valid, 31 valid (4m+3)
    labels, blended = net.infer(img, want_blended=True)
    logits = net.forward(img[None])[0]
    np.testing.assert_array_equal(blended, logits)
    np.testing.assert_array_equal(labels, logits.argmax(0).astype(np.uint16))


def test_tile_is_clamp_padded_when_size_not_valid():
    net = make_net(2)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (21, 30, 3), dtype=np.uint8)
    labels, blended = net.infer(img, want_blended=True)
    th, tw = net.recommended_input_dim(21), net.recommended_input_dim(30)
    cy, cx = 21 // 2, 30 // 2
    top, left = cy - th // 2, cx - tw // 2
    ys = np.clip(np.arange(top, top + th), 0, 20)
    xs = np.clip(np.arange(left, left + tw), 0, 29)
    padded = img[np.ix_(ys, xs)]
    logits = net.forward(padded[None])[0]
    np.testing.assert_array_equal(blended, logits[:, -top:-top + 21, -left:-left + 30])


def test_two_tile_blend_ramps():
    net = make_net(1)
    rng = np.random.default_rng(2)
    H, W = 15, 41
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    tiles = [((0, 0, 24, H - 1), (0, 0, 15, H - 1)), ((16, 0, W - 1, H - 1), (25, 0, W - 1, H - 1))]
    labels, blended = net.infer(img, tiles=tiles, want_blended=True)

    def tile_logits(l, r):
        fw = r - l + 1
        tw, th = net.recommended_input_dim(fw), net.recommended_input_dim(H)
        cx, cy = l + fw // 2, H // 2
        left, top = cx - tw // 2, cy - th // 2
        ys = np.clip(np.arange(top, top + th), 0, H - 1)
        xs = np.clip(np.arange(left, left + tw), 0, W - 1)
        lg = net.forward(img[np.ix_(ys, xs)][None])[0]
        return lg, left, top

    a, la, ta = tile_logits(0, 24)
    b, lb, tb = tile_logits(16, W - 1)
    want = np.zeros((3, H, W), np.float32)
    for x in range(W):
        for y in range(H):
            va = a[:, y - ta, x - la] if x <= 24 else None
            vb = b[:, y - tb, x - lb] if x >= 16 else None
            if x <= 15:
                want[:, y, x] = va
            elif x >= 25:
                want[:, y, x] = vb
            else:
                t0 = (24 - x) / float(24 - 15)
                t1 = (x - 16) / float(25 - 16)
                acc = np.zeros(3, np.float32)
                acc = (acc.astype(np.float64) + t0 * va.astype(np.float64)).astype(np.float32)
                acc = (acc.astype(np.float64) + t1 * vb.astype(np.float64)).astype(np.float32)
                want[:, y, x] = acc
    np.testing.assert_array_equal(blended, want)
    # the two ramps sum to (W-1)/W everywhere in the overlap
    ovw = 24 - 16 + 1
    for x in range(16, 25):
        assert abs((24 - x) / 9 + (x - 16) / 9 - (ovw - 1) / ovw) < 1e-12
    np.testing.assert_array_equal(labels, blended.argmax(0))


def test_argmax_gain_ties_and_nan():
    """find_label (annonet_infer.cpp:170-185): strict '>' keeps the lowest class on ties; an all-NaN pixel
    keeps 65535; gains are added in double and rounded to float."""
    net = OracleNet(0, 3, 3, 0.25, 4)
    net.params[:] = 0  # zero filters -> logits == head bias everywhere
    L = net.layers[-1]
    net.running[:] = 1
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (6, 6, 3), dtype=np.uint8)
    net.params[L.b_off:L.b_off + 3] = [1.0, 1.0, 0.5]
    assert (net.infer(img) == 0).all()                      # tie 0/1 -> 0
    assert (net.infer(img, gains=[0.0, 1e-3, 0.0]) == 1).all()
    assert (net.infer(img, gains=[0.0, 0.0, 0.6]) == 2).all()
    assert (net.infer(img, gains=[0.0, 1e-9, 0.0]) == 0).all()  # 1 + 1e-9 rounds to 1.0f: still a tie
    net.params[L.b_off:L.b_off + 3] = np.nan
    assert (net.infer(img) == 65535).all()


def test_detection_levels_filter_blobs():
    net = make_net(1, seed=9)
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (31, 31, 3), dtype=np.uint8)
    base, blended = net.infer(img, want_blended=True)
    assert (base > 0).any()
    # a detection level nobody reaches removes every non-zero blob
    out = net.infer(img, detection_levels=[0.0, 1e9, 1e9])
    assert (out == 0).all()
    # zero levels: filter is off (annonet_infer.cpp:187-191)
    np.testing.assert_array_equal(net.infer(img, detection_levels=[0.0, 0.0, 0.0]), base)
    # tiny positive level: every blob that has a pixel with margin over class 0 survives
    out = net.infer(img, detection_levels=[0.0, 1e-30, 1e-30])
    np.testing.assert_array_equal(out, base)
