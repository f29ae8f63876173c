"""Training crops cut on the device from HBM-resident full images: bit-exact against the oracle's
composition of chip + outpaint + ignore-outside + set_weights + flips + brightness (annonet_train_main.cpp:110-232), and a
training step driven by crop specs against StartTraining on the same crops made on the host."""
import numpy as np
import pytest

import annonet_amd as aa
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make_dataset(rng, channels, sizes, classes):
    ds = aa.Dataset(channels)
    full = []
    for (h, w) in sizes:
        img = rng.integers(0, 256, (h, w) if channels == 1 else (h, w, channels), dtype=np.uint8)
        coarse = rng.integers(0, classes, ((h + 6) // 7 + 1, (w + 6) // 7 + 1))
        lab = np.kron(coarse, np.ones((7, 7), dtype=np.int64))[:h, :w].astype(np.uint16)
        lab[rng.random((h, w)) < 0.05] = aa.LABEL_IGNORE
        assert ds.add(img, lab) == len(full)
        full.append((img, lab))
    return ds, full


def random_specs(rng, full, n, dim):
    specs = []
    for _ in range(n):
        i = int(rng.integers(0, len(full)))
        h, w = full[i][1].shape
        left = int(rng.integers(-dim + 1, w))      # anything from "one column inside" to far outside on either side
        top = int(rng.integers(-dim + 1, h))
        gain = 1.0 if rng.random() < 0.4 else float(np.exp(rng.normal() * 0.2))
        specs.append((i, left, top, int(rng.random() < 0.5), int(rng.random() < 0.5), gain))
    return specs


@pytest.mark.parametrize("channels,classes,dim", [(3, 3, 35), (1, 5, 48), (3, 40, 33)])
def test_crop_batch_matches_oracle(channels, classes, dim):
    rng = np.random.default_rng(channels * 100 + classes)
    ds, full = make_dataset(rng, channels, [(60, 83), (41, 37), (120, 64), (20, 150)], classes)
    specs = random_specs(rng, full, 24, dim)
    specs += [(0, -dim - 5, 3, 0, 0, 1.0), (1, 10, 100, 1, 1, 1.5), (2, 0, 0, 0, 0, 0.0), (3, 149, 19, 1, 0, 300.0)]   # outside entirely; zero / saturating gains
    img, lab, wgt = ds.crop_batch(specs, dim, classes, 0.5, 0.5)
    for k, (i, left, top, flr, fud, gain) in enumerate(specs):
        wi, wl, ww = orc.crop_sample(full[i][0], full[i][1], left, top, dim, flr, fud, gain, 0.5, 0.5)
        np.testing.assert_array_equal(img[k], wi, err_msg=f"crop {k} {specs[k]}")
        np.testing.assert_array_equal(lab[k], wl, err_msg=f"crop {k} {specs[k]}")
        np.testing.assert_array_equal(wgt[k], ww, err_msg=f"crop {k} {specs[k]}")   # bit-exact: the table is the host's own arithmetic
    # other class / image weights, and a histogram whose allocated length depends on which label comes first (annonet_train.h:29-34)
    img2, lab2, wgt2 = ds.crop_batch(specs[:6], dim, classes, 0.3, 0.9)
    for k, (i, left, top, flr, fud, gain) in enumerate(specs[:6]):
        np.testing.assert_array_equal(wgt2[k], orc.crop_sample(full[i][0], full[i][1], left, top, dim, flr, fud, gain, 0.3, 0.9)[2])


@pytest.mark.parametrize("channels,dim", [(3, 35), (1, 40)])
def test_crop_batch_with_downscaling_noise_and_colour_offset_matches_oracle(channels, dim):
    """further downscaling (annonet_train_main.cpp:124-127,160-171), add_random_noise (:73-105) and the colour offset (:226-231) on
    the device, bit for bit against the oracle's restatement (same float32 bilinear arithmetic, same counter-based noise draws)"""
    rng = np.random.default_rng(50 + channels)
    ds, full = make_dataset(rng, channels, [(130, 171), (64, 90)], 4)
    specs = []
    for k, base in enumerate(random_specs(rng, full, 18, dim)):
        f = [1.0, 1.5, 2.0, 2.37][k % 4]
        level = [0, 3, 40, 255][(k // 2) % 4]
        off = (0, 0, 0) if channels == 1 or k % 3 == 0 else tuple(int(v) for v in rng.integers(-40, 41, 3))
        i, left, top = base[0], base[1], base[2]
        if f != 1.0:   # keep part of the larger rectangle on the image
            h, w = full[i][1].shape
            left, top = int(rng.integers(-20, w - 10)), int(rng.integers(-20, h - 10))
        specs.append((i, left, top, base[3], base[4], base[5], f, level, int(rng.integers(0, 2 ** 62)), off))
    img, lab, wgt = ds.crop_batch(specs, dim, 4, 0.5, 0.5)
    for k, (i, left, top, flr, fud, gain, f, level, seed, off) in enumerate(specs):
        wi, wl, ww = orc.crop_sample(full[i][0], full[i][1], left, top, dim, flr, fud, gain, 0.5, 0.5, f, level, seed, off)
        np.testing.assert_array_equal(lab[k], wl, err_msg=f"crop {k} {specs[k]}")
        np.testing.assert_array_equal(wgt[k], ww, err_msg=f"crop {k} {specs[k]}")
        np.testing.assert_array_equal(img[k], wi, err_msg=f"crop {k} {specs[k]}")
    noisy = [k for k, s in enumerate(specs) if s[7] == 40 and s[6] == 1.0 and s[5] == 1.0]
    assert noisy, "the parametrisation lost its plain noise case"
    k = noisy[0]
    clean = ds.crop_batch([specs[k][:6]], dim, 4)[0][0].astype(int)
    delta = img[k].astype(int) - np.clip(clean + np.asarray(specs[k][9] if channels == 3 else 0), 0, 255)
    unclamped = (clean > 45) & (clean < 210)
    assert np.abs(delta[unclamped]).max() <= 40 and delta[unclamped].std() > 15        # uniform in [-40, 40]: sigma 23.4
    with pytest.raises(aa.AnnonetHipError):
        ds.crop_batch([(0, 0, 0, 0, 0, 1.0, 0.5, 0, 0, (0, 0, 0))], dim, 4)               # factor < 1


def test_crop_batch_errors():
    rng = np.random.default_rng(1)
    ds, full = make_dataset(rng, 3, [(30, 30)], 3)
    with pytest.raises(aa.AnnonetHipError):
        ds.crop_batch([(1, 0, 0, 0, 0, 1.0)], 16, 3)            # no such image
    with pytest.raises(aa.AnnonetHipError):
        ds.crop_batch([(0, 0, 0, 0, 0, 1.0)], 16, 2)            # label 2 with two classes
    with pytest.raises(aa.AnnonetHipError):
        ds.crop_batch([(0, 0, 0, 0, 0, -1.0)], 16, 3)
    img, lab, wgt = ds.crop_batch([(0, 0, 0, 0, 0, 1.0)], 16, 3)   # the handle still works after an error
    np.testing.assert_array_equal(img[0], full[0][0][:16, :16])


def test_training_on_device_crops_equals_training_on_host_crops():
    rng = np.random.default_rng(9)
    d = aa.RuntimeNet.GetRecommendedInputDimension(1, 40)
    ds, full = make_dataset(rng, 3, [(90, 120), (75, 64)], 3)

    def trainer():
        t = aa.TrainingNet(1, 3, aa.ANH_BF16, seed=3)
        t.SetNetWidth(0.5, 4); t.SetClassCount(3); t.Initialize(); t.SetLearningRate(0.05)
        return t

    a, b = trainer(), trainer()
    for step in range(3):
        specs = random_specs(rng, full, 6, d)
        a.StartTrainingOnCrops(ds, specs, d)
        crops = [orc.crop_sample(full[i][0], full[i][1], left, top, d, flr, fud, gain, 0.5, 0.5) for (i, left, top, flr, fud, gain) in specs]
        wl = []
        for (_, lab, wgt) in crops:
            x = np.zeros(lab.shape, dtype=aa.netpimpl.WLABEL)
            x["label"], x["weight"] = lab, wgt
            wl.append(x)
        b.StartTraining([c[0] for c in crops], wl)
        a.synchronize(); b.synchronize()
        assert a.get_last_loss() == b.get_last_loss()
    pa, ra = a.get_params()
    pb, rb = b.get_params()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ra, rb)


def test_dataset_remove_frees_the_image_and_reuses_its_index():
    rng = np.random.default_rng(4)
    ds = aa.Dataset(3)
    imgs = [rng.integers(0, 256, (40 + 10 * i, 50, 3), dtype=np.uint8) for i in range(3)]
    labs = [rng.integers(0, 3, im.shape[:2]).astype(np.uint16) for im in imgs]
    idx = [ds.add(im, lb) for im, lb in zip(imgs, labs)]
    assert idx == [0, 1, 2] and ds.resident_bytes() == sum(im.shape[0] * im.shape[1] * 5 for im in imgs)
    ds.remove(1)
    assert ds.resident_bytes() == sum(imgs[i].shape[0] * imgs[i].shape[1] * 5 for i in (0, 2))
    with pytest.raises(aa.AnnonetHipError, match="removed"):
        ds.crop_batch([(1, 0, 0, 0, 0, 1.0)], 35, 3, 0.5, 0.5)
    with pytest.raises(aa.AnnonetHipError, match="no such image"):
        ds.remove(1)
    again = ds.add(imgs[1], labs[1])
    assert again == 1                                     # the freed slot is handed out again
    got_img, _, _ = ds.crop_batch([(1, 3, 2, 0, 0, 1.0)], 35, 3, 0.5, 0.5)
    np.testing.assert_array_equal(got_img[0], imgs[1][2:37, 3:38])
