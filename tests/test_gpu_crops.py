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
