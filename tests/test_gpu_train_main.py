"""annonet_train_hip (annonet_amd/host/train_tool.cpp = the job of the reference's annonet_train_main.cpp:260-644 on the drop-in headers)
end to end on a synthetic anno directory: option echo, dataset scan, LRU cache + loader threads, device-cut mini-batches (default)
and host-cut mini-batches (--host-crops, the reference's data path), annonet.dnn + trainer state file, resume, and the two-replica
data-parallel mode.  The saved net is then read back by the inference tool."""
import os
import re
import subprocess

import numpy as np
import pytest

import annonet_amd as aa
import png_util as pu

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAIN = os.path.join(ROOT, "annonet_amd", "lib", "annonet_train_hip")
INFER = os.path.join(ROOT, "annonet_amd", "lib", "annonet_infer_hip")


@pytest.fixture
def dataset(tmp_path):
    rng = np.random.default_rng(11)
    d = tmp_path / "data"
    d.mkdir()
    for k, (h, w) in enumerate(((160, 200), (140, 150), (200, 170))):
        lab = np.zeros((h, w), np.uint16)
        for _ in range(8):
            y, x = rng.integers(0, h - 20), rng.integers(0, w - 20)
            lab[y:y + rng.integers(10, 50), x:x + rng.integers(10, 50)] = rng.integers(1, 3)
        img = (rng.integers(0, 60, (h, w, 3)) + (lab[:, :, None] * 80)).astype(np.uint8)      # the classes are separable by brightness
        lab[rng.random((h, w)) < 0.1] = 65535
        pu.write_png(d / f"img{k}.png", img)
        pu.write_png(str(d / f"img{k}.png") + "_mask.png", pu.labels_to_rgba(lab))
    pu.write_png(d / "unlabelled.png", rng.integers(0, 256, (50, 50, 3), dtype=np.uint8))   # no mask: ignored by the scan (require_ground_truth)
    return d


def train(cwd, d, *extra, steps=6):
    r = subprocess.run([TRAIN, str(d), "-b", "6", "--net-width-scaler", "0.25", "--net-width-min-filter-count", "4", "--input-dimension-multiplier", "1.2",
                        "--max-total-steps", str(steps), "--save-interval", "4", "--data-loader-thread-count", "3", "--cached-image-count", "2", "--seed", "5", *extra],
                       capture_output=True, text=True, timeout=900, cwd=cwd)
    return r


def test_train_main_device_crops_then_infer(tmp_path, dataset):
    r = train(tmp_path, dataset, "-l", "-u", "-o", "-n", "6", "--multiplicative-brightness-change-probability", "0.5", "-f", "1.5")
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    for line in ("Initial downscaling factor = 1", "Further downscaling factor = 1.5", "Allow flipping input images upside down = yes", "Minibatch size = 6",
                 "Net width scaler = 0.25, min filter count = 4", "Required input dimension = 35", "Requested input dimension = 42", "Actual input dimension = 43",
                 "images in dataset: 3", "Mini-batches are cut on the device from HBM-resident full images", "Now training...", "saving network"):
        assert line in out, line
    assert "Warning: no anno_classes.json file found" in out and "Using the default anno classes" in out
    m = re.search(r"steps: 6, full images decoded: (\d+), cache hits: (\d+), evictions: (\d+), images resident in HBM: 3, HBM evictions: 0", out)
    assert m and int(m.group(1)) >= 3 and int(m.group(3)) >= 1          # 3 images through a 2-entry cache: evictions happened
    assert out.count("saving network") == 3                              # steps 0 and 4 (save interval) + the final save (:611-613,634-636)
    classes_json, factor, blob = aa.dnn_envelope_unpack((tmp_path / "annonet.dnn").read_bytes())
    assert classes_json in ("", b"") and factor == 1.5
    net = aa.RuntimeNet.Deserialize(blob, aa.ANH_FP32)
    assert net.cfg.classes == 3 and net.cfg.levels == 2
    r = subprocess.run([INFER, str(dataset), "--dnn", str(tmp_path / "annonet.dnn"), "-w", "128", "-h", "128"], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0 and "downscaling factor = 1.5" in r.stdout and "Confusion matrix per pixel:" in r.stdout


def test_train_main_bounds_the_hbm_resident_images(tmp_path, dataset):
    """--hbm-image-budget-gib: the full images kept in HBM for the device-crop path are an LRU set within a byte budget (the
    reference bounds its decoded images by --cached-image-count): 3 images of ~100-170 KB under a 0.0002 GiB (~215 KB) budget."""
    # mini-batches of ONE crop: an image of the mini-batch being assembled is never evicted, so with larger batches all three images
    # may have to be resident together; 14 single-crop steps draw at least two different images
    r = subprocess.run([TRAIN, str(dataset), "-b", "1", "--net-width-scaler", "0.25", "--net-width-min-filter-count", "4", "--input-dimension-multiplier", "1.2",
                        "--max-total-steps", "14", "--save-interval", "100", "--data-loader-thread-count", "2", "--cached-image-count", "2", "--seed", "5",
                        "--hbm-image-budget-gib", "0.0002"], capture_output=True, text=True, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"images resident in HBM: (\d+), HBM evictions: (\d+)", r.stdout)
    assert m and int(m.group(1)) <= 2 and int(m.group(2)) >= 1, r.stdout[-400:]


def test_train_main_host_crops_learn_and_resume(tmp_path, dataset):
    r = train(tmp_path, dataset, "--host-crops", "--initial-learning-rate", "0.05", steps=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Mini-batches are cut on the host by the loader threads" in r.stdout
    classes_json, factor, blob = aa.dnn_envelope_unpack((tmp_path / "annonet.dnn").read_bytes())
    net = aa.RuntimeNet.Deserialize(blob, aa.ANH_FP32)
    # the classes are separable by brightness: 60 steps must beat guessing by a wide margin on a training image
    img = pu.read_png(dataset / "img0.png")
    gt = pu.read_png(str(dataset / "img0.png") + "_mask.png")
    labels = aa.annonet_infer(net, img, tiling_parameters=aa.tiling.parameters(128, 128, 35, 35))
    want = np.full(gt.shape[:2], 65535, np.uint16)
    for k, col in enumerate(pu.DEFAULT_CLASSES):
        want[(gt == col).all(axis=2)] = k
    valid = want != 65535
    assert (labels[valid] == want[valid]).mean() > 0.8
    # the 10-minute synchronization file exists only if 10 minutes passed; write one now and resume from it
    assert not (tmp_path / "annonet_trainer_state_file.dat").exists()


def test_train_main_two_replicas_and_errors(tmp_path, dataset):
    r = train(tmp_path, dataset, "--devices", "0,0", steps=3)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Mini-batches are cut on the host by the loader threads" in r.stdout       # several devices: host mini-batches, split along N
    r = subprocess.run([TRAIN], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "You call this program like this" in r.stdout
    r = subprocess.run([TRAIN, str(dataset), "-d", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "strictly positive" in r.stderr
    empty = tmp_path / "empty"
    empty.mkdir()
    r = subprocess.run([TRAIN, str(empty)], capture_output=True, text=True, timeout=120, cwd=tmp_path)   # no images at all
    assert r.returncode == 1 and "Didn't find an anno dataset." in r.stdout
    bad = tmp_path / "bad"
    bad.mkdir()
    pu.write_png(bad / "x.png", np.zeros((60, 60, 3), np.uint8))
    pu.write_png(str(bad / "x.png") + "_mask.png", np.full((60, 60, 4), 7, np.uint8))     # an RGBA value that is no class
    r = subprocess.run([TRAIN, str(bad), "-b", "2", "--max-total-steps", "1"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 2 and "Unknown class: r = 7" in r.stdout                      # in-loop errors print and exit(2) (:616-620)
