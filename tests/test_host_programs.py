"""Host logic of the reference's mains above the drop-in boundary (annonet_amd/host/annonet_host.h, image_io.h),
checked through annonet_amd/lib/host_selftest against numpy restatements (tests/png_util.py) — CPU only:
the PNG codec (annonet.cpp:150,155; annonet_infer_main.cpp:413), parse_anno_classes (annonet_parse_anno_classes.cpp:22-83),
decode_rgba_label_image (annonet.cpp:41-58), resize_label_image (annonet.cpp:134-141), both confusion matrices
(annonet_infer_main.cpp:202-272,482-494) and the matrix printout (:101-194)."""
import json
import os
import subprocess

import numpy as np
import pytest

import png_util as pu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "annonet_amd", "lib", "host_selftest")


def run(*args, ok=True):
    r = subprocess.run([TOOL, *map(str, args)], capture_output=True, text=True, timeout=120)
    assert (r.returncode == 0) == ok, r.stderr
    return r.stdout if ok else r.stderr


@pytest.mark.parametrize("channels,filt", [(1, 0), (3, 1), (4, 2), (3, 0), (4, 0)])
def test_png_codec_round_trip(tmp_path, channels, filt):
    rng = np.random.default_rng(channels * 10 + filt)
    a = rng.integers(0, 256, (37, 53, channels), dtype=np.uint8)
    src, dst = tmp_path / "in.png", tmp_path / "out.png"
    pu.write_png(src, a, filter_type=filt)
    run("png-roundtrip", src, dst)
    got = pu.read_png(dst)
    np.testing.assert_array_equal(got.reshape(a.shape), a)


def test_png_codec_refuses_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.png"
    p.write_bytes(b"\xff\xd8\xff\xe0 not really a jpeg")
    assert "JPEG" in run("png-roundtrip", p, tmp_path / "y.png", ok=False)
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 5)
    run("png-roundtrip", p, tmp_path / "y.png", ok=False)


def test_parse_anno_classes(tmp_path):
    out = run("classes", "-").strip().splitlines()
    assert out == ["0 0 255 0 64 clean", "1 255 255 0 128 minor defect", "2 255 0 0 128 major defect"]   # annonet_parse_anno_classes.cpp:25-29
    doc = {"anno_classes": [{"name": "background", "color": {"red": 1, "green": 2, "blue": 3, "alpha": 4}},
                            {"name": "scär \"x\"", "color": {"alpha": 200, "blue": 0, "green": 17, "red": 255}, "extra": [1, 2.5, None, True]}]}
    f = tmp_path / "c.json"
    f.write_text(json.dumps(doc))
    out = run("classes", f).strip().splitlines()
    assert out == ["0 1 2 3 4 background", "1 255 17 0 200 scär \"x\""]
    f.write_text(json.dumps({"anno_classes": [{"name": "n", "color": {"red": 0, "green": 0, "blue": 0, "alpha": 0}}]}))
    assert "reserved for pixels to be ignored" in run("classes", f, ok=False)
    f.write_text("{\"anno_classes\": [")
    assert "Error parsing json" in run("classes", f, ok=False)
    f.write_text(json.dumps({"anno_classes": [{"name": "n"}]}))
    assert "no color found" in run("classes", f, ok=False)


def test_decode_rgba_label_image(tmp_path):
    rng = np.random.default_rng(1)
    labels = rng.integers(0, 3, (41, 29)).astype(np.uint16)
    labels[rng.random(labels.shape) < 0.2] = 65535
    pu.write_png(tmp_path / "m.png", pu.labels_to_rgba(labels))
    counts = run("decode-mask", tmp_path / "m.png", "-", tmp_path / "l.raw").split()
    got = np.fromfile(tmp_path / "l.raw", np.uint16).reshape(labels.shape)
    np.testing.assert_array_equal(got, labels)
    assert [int(c) for c in counts] == [int((labels == k).sum()) for k in range(3)]
    bad = pu.labels_to_rgba(labels)
    bad[3, 4] = (9, 9, 9, 9)
    pu.write_png(tmp_path / "bad.png", bad)
    assert "Unknown class: r = 9, g = 9, b = 9, alpha = 9" in run("decode-mask", tmp_path / "bad.png", "-", tmp_path / "l.raw", ok=False)


@pytest.mark.parametrize("shape,target", [((40, 60), (30, 20)), ((17, 23), (46, 34)), ((12, 12), (12, 12)), ((5, 7), (1, 1))])
def test_resize_label_image_nearest_neighbour(tmp_path, shape, target):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 5, shape).astype(np.uint16)
    a.tofile(tmp_path / "a.raw")
    run("resize-labels", tmp_path / "a.raw", shape[0], shape[1], target[0], target[1], tmp_path / "b.raw")
    got = np.fromfile(tmp_path / "b.raw", np.uint16).reshape(target[1], target[0])
    np.testing.assert_array_equal(got, pu.resize_nearest(a, target[0], target[1]))


def blobby(rng, shape, K, p_ignore=0.1):
    """label images with real regions: a few random rectangles of each class over background"""
    a = np.zeros(shape, np.uint16)
    for _ in range(14):
        y, x = rng.integers(0, shape[0]), rng.integers(0, shape[1])
        h, w = rng.integers(2, 12), rng.integers(2, 12)
        a[y:y + h, x:x + w] = rng.integers(0, K)
    if p_ignore:
        a[rng.random(shape) < p_ignore] = 65535
    return a


@pytest.mark.parametrize("seed", range(6))
def test_confusion_matrices_per_pixel_and_per_region(tmp_path, seed):
    rng = np.random.default_rng(seed)
    K = 3 + seed % 2
    gt = blobby(rng, (48, 64), K)
    res = blobby(rng, (48, 64), K, p_ignore=0)
    res[10:30, 10:40] = np.where(gt[10:30, 10:40] == 65535, 0, gt[10:30, 10:40])   # partly right, so that the diagonal is populated
    gt.tofile(tmp_path / "gt.raw"); res.tofile(tmp_path / "res.raw")
    out = np.array(run("confusion", tmp_path / "gt.raw", tmp_path / "res.raw", 48, 64, K).split(), dtype=np.int64).reshape(2, K, K)
    per_pixel, per_region = pu.confusion_matrices(gt, res, K)
    np.testing.assert_array_equal(out[0], per_pixel)
    np.testing.assert_array_equal(out[1], per_region)
    assert per_region.sum() > 4


def test_confusion_matrix_printout_layout(tmp_path):
    """print_confusion_matrix (annonet_infer_main.cpp:101-194): column widths, recall / precision / accuracy lines"""
    gt = np.array([[0, 0, 1, 1], [2, 2, 2, 65535]], np.uint16)
    res = np.array([[0, 1, 1, 1], [2, 2, 0, 0]], np.uint16)
    gt.tofile(tmp_path / "gt.raw"); res.tofile(tmp_path / "res.raw")
    text = run("print-confusion", tmp_path / "gt.raw", tmp_path / "res.raw", 2, 4, 3)
    lines = text.splitlines()
    assert lines[0].endswith("predicted") and lines[1].split() == ["0", "1", "2", "recall"]
    assert lines[2].split() == ["0", "1", "1", "0", "50.00", "%"]
    assert lines[3].split() == ["truth", "1", "0", "2", "0", "100.00", "%"]      # the "truth" label sits on the middle row
    assert lines[4].split() == ["2", "1", "0", "2", "66.67", "%"]
    assert lines[5].split() == ["precision", "50", "%", "67", "%", "100", "%"]
    assert lines[6].split() == ["accuracy", "71.43", "%"]


# ---------------------------------------------------------------------------------------------------------------- N2: training data path
from oracle import oracle as orc   # noqa: E402  (the checker of the host-cut crops)


@pytest.mark.parametrize("case", [
    dict(left=7, top=4, dim=35, flr=0, fud=0, gain=1.0, f=1.0, off=(0, 0, 0)),
    dict(left=-12, top=60, dim=35, flr=1, fud=0, gain=1.23, f=1.0, off=(0, 0, 0)),          # partly outside: outpaint + ignore labels
    dict(left=30, top=-9, dim=40, flr=0, fud=1, gain=0.8, f=1.5, off=(5, -17, 30)),          # further downscaling + colour offset
    dict(left=-200, top=10, dim=33, flr=1, fud=1, gain=1.0, f=2.37, off=(0, 0, 0)),          # the rectangle misses the image entirely
])
def test_host_cut_crop_matches_the_oracle(tmp_path, case):
    """cut_crop_on_host (= randomly_crop_image, annonet_train_main.cpp:129-231, with the draws given) bit for bit against the oracle's
    crop_sample — the same checker the device-cut crops are held to (tests/test_gpu_crops.py)"""
    rng = np.random.default_rng(7)
    h, w = 90, 120
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    coarse = rng.integers(0, 3, (h // 6 + 1, w // 6 + 1))
    lab = np.kron(coarse, np.ones((6, 6), dtype=np.int64))[:h, :w].astype(np.uint16)
    lab[rng.random((h, w)) < 0.1] = 65535
    pu.write_png(tmp_path / "i.png", img)
    pu.write_png(tmp_path / "i.png_mask.png", pu.labels_to_rgba(lab))
    c = case
    run("crop", tmp_path / "i.png", tmp_path / "i.png_mask.png", c["left"], c["top"], c["dim"], c["flr"], c["fud"], c["gain"], c["f"], *c["off"], tmp_path / "out")
    d = c["dim"]
    gi = np.fromfile(str(tmp_path / "out") + ".img.raw", np.uint8).reshape(d, d, 3)
    gl = np.fromfile(str(tmp_path / "out") + ".lab.raw", np.uint16).reshape(d, d)
    gw = np.fromfile(str(tmp_path / "out") + ".w.raw", np.float32).reshape(d, d)
    wi, wl, ww = orc.crop_sample(img, lab, c["left"], c["top"], d, c["flr"], c["fud"], c["gain"], 0.5, 0.5, c["f"], 0, 0, c["off"])
    np.testing.assert_array_equal(gl, wl)
    np.testing.assert_array_equal(gw, ww)
    np.testing.assert_array_equal(gi, wi)


def test_add_random_noise_is_uniform_in_the_rounded_level():
    lo, hi, mean = run("noise", 12.4, 5, 200).split()       # std::round(12.4) = 12 (annonet_train_main.cpp:75)
    assert int(lo) == -12 and int(hi) == 12 and abs(float(mean)) < 0.1
    lo, hi, mean = run("noise", 0.4, 5, 20).split()         # rounds to 0: untouched
    assert (int(lo), int(hi), float(mean)) == (0, 0, 0.0)


def test_color_offsets_follow_the_covariance_square_root():
    """apply_random_color_offset's offsets [UPSTREAM-UNVERIFIED]: round(0.1 * tform * gaussian) — strongly correlated channels, sigma ~ 7"""
    v = np.array(run("color-offsets", 3, 4000).split(), dtype=np.int64).reshape(-1, 3)
    tform = np.array([[-66.379, 25.094, 6.79698], [-68.0492, -0.302309, -13.9539], [-68.4907, -24.0199, 7.27653]])
    want_cov = 0.01 * tform @ tform.T
    got_cov = np.cov(v.T)
    assert np.abs(got_cov - want_cov).max() < 0.12 * np.abs(want_cov).max()
    assert np.abs(v.mean(0)).max() < 0.5


def test_shared_lru_cache_evicts_the_least_recently_used(tmp_path):
    from collections import OrderedDict
    rng = np.random.default_rng(0)
    keys = [f"k{'x' * int(i)}" for i in rng.integers(0, 9, 300)]
    for cap in (1, 3, 8, 20):
        d, hits, misses, ev = OrderedDict(), 0, 0, 0
        for k in keys:
            if k in d:
                d.move_to_end(k); hits += 1
            else:
                d[k] = 1; misses += 1
                if len(d) > cap:
                    d.popitem(last=False); ev += 1
        assert [int(x) for x in run("lru", cap, *keys).split()] == [hits, misses, ev, len(d)]
