"""CPU-only checks of the product library: it loads, exports every symbol include/annonet_hip.h declares, and its host
logic (spec, dimension maths, tiler, set_weights, crop rect, outpaint, LR-schedule helper) agrees with the oracle and
with the reference's own known answers.  No compute entry point is called here (there is no GPU in this container)."""
import os
import re

import numpy as np
import pytest

import annonet_amd as aa
from annonet_amd import _lib
from oracle import oracle as orc
from oracle.oracle import OracleNet
from test_oracle_infer import check_tiles
from test_reference_known_answers import check_random_rect, check_set_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "annonet_hip.h")).read()
    declared = set(re.findall(r"\b(anh_[a-z0-9_]+)\s*\(", header)) - {"anh_status"}
    L = aa.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in annonet_hip.h but not exported"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())


def test_no_product_file_touches_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "annonet_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f


@pytest.mark.parametrize("levels,in_ch,classes,scaler,minf", [(0, 3, 3, 1.0, 1), (1, 1, 2, 0.5, 8), (2, 3, 3, 1.0, 1), (3, 3, 5, 0.25, 12), (2, 3, 3, 2.0, 1)])
def test_spec_matches_oracle(levels, in_ch, classes, scaler, minf):
    cfg = aa.net_config(levels, in_ch, classes, scaler, minf, aa.ANH_FP32)
    mine = aa.net_layers(cfg)
    ref = OracleNet(levels, in_ch, classes, scaler, minf)
    assert len(mine) == len(ref.layers)
    for a, b in zip(mine, ref.layers):
        for f, _ in _lib.LayerDesc._fields_:
            assert getattr(a, f) == getattr(b, f), f
    L = aa.lib()
    import ctypes as C
    assert L.anh_net_param_count(C.byref(cfg)) == ref.n_params
    assert L.anh_net_running_count(C.byref(cfg)) == ref.n_running
    assert L.anh_required_input_dim(C.byref(cfg)) == ref.required_input_dim()
    for n in list(range(1, 70)) + [227, 1024, 4096]:
        assert L.anh_recommended_input_dim(levels, n) == ref.recommended_input_dim(n)


def test_bad_config_is_an_error_not_a_crash():
    import ctypes as C
    L = aa.lib()
    bad = aa.net_config(levels=7)
    assert L.anh_net_layer_count(C.byref(bad)) == -1
    assert b"level count" in L.anh_last_error()
    assert L.anh_recommended_input_dim(9, 10) == -1


def test_set_weights_known_answers_product():
    check_set_weights(lambda lab, cw, iw: aa.set_weights(lab, cw, iw)["weight"])
    out = aa.set_weights(np.array([[0, 65535, 1]], np.uint16), 0.5, 0.5)
    assert out["label"].tolist() == [[0, 65535, 1]]


def test_set_weights_matches_oracle_on_random_crops():
    rng = np.random.default_rng(0)
    for _ in range(20):
        lab = rng.integers(0, 5, (rng.integers(1, 40), rng.integers(1, 40))).astype(np.uint16)
        lab[rng.random(lab.shape) < 0.2] = 65535
        cw, iw = rng.random(2)
        np.testing.assert_array_equal(aa.set_weights(lab, cw, iw)["weight"], orc.set_weights(lab, cw, iw))


def test_random_rect_product():
    check_random_rect(aa.random_rect_containing_point)
    rng = np.random.default_rng(1)
    for _ in range(100):
        a = [int(v) for v in rng.integers(0, 2**32, 2)] + [int(v) for v in rng.integers(-50, 50, 2)] + [int(v) for v in rng.integers(1, 30, 2)]
        assert aa.random_rect_containing_point(*a) == orc.random_rect_containing_point(*a)


@pytest.mark.parametrize("w,h,mw,mh,ov", [
    (227, 227, 1024, 1024, 35), (1, 1, 64, 64, 5), (20, 300, 100, 100, 35), (4096, 4096, 1024, 1024, 35),
    (16384, 16384, 1024, 1024, 35), (1025, 1024, 1024, 1024, 35), (1000, 777, 128, 96, 15), (300, 300, 120, 110, 35),
])
def test_tiler_contract_and_oracle_agreement(w, h, mw, mh, ov):
    get = lambda *a: aa.tiling.get_tiles(a[0], a[1], aa.tiling.parameters(a[2], a[3], a[4], a[5]))
    if w * h <= 4096 * 4096:
        tiles = check_tiles(get, w, h, mw, mh, ov, ov)
    else:
        tiles = get(w, h, mw, mh, ov, ov)
    assert tiles == orc.get_tiles(w, h, mw, mh, ov, ov)


def test_tiler_contract_on_random_sizes():
    """The two shortcuts of tiled inference rest on the tiler's contract: the class planes are cleared in the blend frames only, and a batch's
    blend assigns inside unique rectangles without looking at the other tiles — both need "a unique-rectangle pixel is covered by exactly one
    full rectangle" (check_tiles) on every tiling, not just the benchmark's."""
    rng = np.random.default_rng(0)
    get = lambda *a: aa.tiling.get_tiles(a[0], a[1], aa.tiling.parameters(a[2], a[3], a[4], a[5]))
    done = 0
    while done < 150:
        w, h = int(rng.integers(1, 600)), int(rng.integers(1, 600))
        mw, mh, ov = int(rng.integers(40, 400)), int(rng.integers(40, 400)), int(rng.integers(0, 39))
        if mw <= 2 * ov or mh <= 2 * ov:
            continue
        tiles = check_tiles(get, w, h, mw, mh, ov, ov)
        assert tiles == orc.get_tiles(w, h, mw, mh, ov, ov)
        done += 1


def test_tiler_rejects_impossible_overlap():
    with pytest.raises(aa.AnnonetHipError):
        aa.tiling.get_tiles(500, 500, aa.tiling.parameters(60, 60, 35, 35))
    with pytest.raises(aa.AnnonetHipError):
        aa.tiling.get_tiles(300, 300, aa.tiling.parameters(71, 71, 35, 35))


def test_outpaint_matches_oracle():
    rng = np.random.default_rng(2)
    for _ in range(20):
        nr, nc = rng.integers(1, 20, 2)
        img = rng.integers(0, 256, (nr, nc, 3), dtype=np.uint8)
        l, t = rng.integers(-5, nc), rng.integers(-5, nr)
        inside = (int(l), int(t), int(l + rng.integers(0, 12)), int(t + rng.integers(0, 12)))
        np.testing.assert_array_equal(aa.outpaint(img, inside), orc.outpaint(img, inside))


def test_ignore_large_nonzero_regions_hand_cases():
    """annonet_train_main.cpp:434-502 on blobs whose fate can be read off the picture (rf = 4: area bar 16*by_area)."""
    lab = np.zeros((12, 16), np.uint16)
    lab[1:4, 1:9] = 2          # 3 x 8 = 24 pixels, width 8, height 3
    lab[6:8, 2:4] = 1          # 2 x 2
    lab[5:11, 12] = 3          # 1 x 6 column, height 6
    lab[9, 5] = 2; lab[10, 6] = 2   # diagonal pair: ONE blob under 8-connectivity
    lab[0, 15] = aa.LABEL_IGNORE
    same, n = aa.ignore_large_nonzero_regions(lab, 4)                      # CLI defaults: every test off
    assert n == 0 and (same == lab).all()
    out, n = aa.ignore_large_nonzero_regions(lab, 4, by_width=1.5)         # wider than 6
    assert n == 24 and (out[1:4, 1:9] == aa.LABEL_IGNORE).all() and (out[6:8, 2:4] == 1).all() and (out[5:11, 12] == 3).all()
    out, n = aa.ignore_large_nonzero_regions(lab, 4, by_height=1.25)       # taller than 5
    assert n == 6 and (out[5:11, 12] == aa.LABEL_IGNORE).all() and (out[1:4, 1:9] == 2).all()
    out, n = aa.ignore_large_nonzero_regions(lab, 4, by_area=0.25)         # more than 4 pixels
    assert n == 30 and (out[6:8, 2:4] == 1).all() and out[9, 5] == 2 and out[10, 6] == 2
    out, n = aa.ignore_large_nonzero_regions(lab, 4, by_area=1.0 / 16)     # more than 1 pixel: the diagonal pair goes as one blob
    assert out[9, 5] == aa.LABEL_IGNORE and out[10, 6] == aa.LABEL_IGNORE and n == 24 + 4 + 6 + 2
    # equal-size is kept (strict '>'), and touching blobs of DIFFERENT labels stay separate
    lab2 = np.zeros((4, 8), np.uint16); lab2[:, :4] = 1; lab2[:, 4:] = 2
    out, n = aa.ignore_large_nonzero_regions(lab2, 4, by_area=1.0, by_width=1.0, by_height=1.0)
    assert n == 0
    out, n = aa.ignore_large_nonzero_regions(lab2, 4, by_width=0.75)
    assert n == 32


def test_ignore_large_nonzero_regions_matches_oracle():
    rng = np.random.default_rng(8)
    for case in range(30):
        nr, nc = (int(v) for v in rng.integers(1, 60, 2))
        # blobby label images: a coarse random field upsampled, a few classes, some ignored pixels
        coarse = rng.integers(0, 4, ((nr + 4) // 5 + 1, (nc + 4) // 5 + 1))
        lab = np.kron(coarse, np.ones((5, 5), dtype=np.int64))[:nr, :nc].astype(np.uint16)
        lab[rng.random((nr, nc)) < 0.05] = aa.LABEL_IGNORE
        lab[rng.random((nr, nc)) < 0.05] = 0
        rf = int(rng.integers(3, 12))
        a, w, h = (float(v) if rng.random() < 0.7 else float("inf") for v in rng.uniform(0.2, 3.0, 3))
        got, n = aa.ignore_large_nonzero_regions(lab, rf, by_area=a, by_width=w, by_height=h)
        want, wn = orc.ignore_large_nonzero_regions(lab, rf, by_area=a, by_width=w, by_height=h)
        np.testing.assert_array_equal(got, want)
        assert n == wn
        assert ((got == lab) | (got == aa.LABEL_IGNORE)).all()       # pixels are only ever relabelled to "ignore"
        again, n2 = aa.ignore_large_nonzero_regions(got, rf, by_area=a, by_width=w, by_height=h)
        assert n2 == 0 and (again == got).all()                       # idempotent
    with pytest.raises(aa.AnnonetHipError):
        aa.ignore_large_nonzero_regions(np.zeros((3, 3), np.uint16), 0)
    with pytest.raises(aa.AnnonetHipError):
        aa.ignore_large_nonzero_regions(np.zeros((3, 3), np.uint16), 4, by_area=-1.0)


def test_count_steps_without_decrease():
    rng = np.random.default_rng(3)
    down = np.linspace(2.0, 1.0, 300) + rng.normal(0, 0.01, 300)
    flat = 1.0 + rng.normal(0, 0.01, 300)
    assert aa.count_steps_without_decrease(down) < 20
    assert aa.count_steps_without_decrease(np.concatenate([down, flat])) >= 250
    for v in (down, flat, np.concatenate([down, flat]), flat[:2], flat[:0]):
        assert aa.count_steps_without_decrease(v) == orc.count_steps_without_decrease(v)


def test_compute_entry_points_fail_loudly_without_a_gpu():
    if aa.lib().anh_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(aa.AnnonetHipError) as e:
        aa.RuntimeNet(aa.net_config())
    assert e.value.code == 3
    t = aa.TrainingNet()
    with pytest.raises(aa.AnnonetHipError):
        t.Initialize()
