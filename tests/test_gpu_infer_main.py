"""annonet_infer_hip (annonet_amd/host/infer_tool.cpp = the job of the reference's annonet_infer_main.cpp:283-538 on the drop-in
headers) end to end on a synthetic anno directory: option parsing, annonet.dnn envelope, reader / writer pools, the tool's
timing lines, result PNGs and both confusion matrices — against a numpy restatement fed with the label maps that the Python
mirror of annonet_infer() produces from the same net in the bit-exact fp32 mode."""
import os
import re
import subprocess

import numpy as np
import pytest

import annonet_amd as aa
import png_util as pu
from conftest import random_params
from oracle.oracle import OracleNet

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "annonet_amd", "lib", "annonet_infer_hip")
K = 3


def parse_matrix(lines, at):
    """rows of the printed matrix that starts after the line index `at` (header 'predicted', class header, K rows)"""
    rows = []
    for line in lines[at + 3:at + 3 + K]:
        toks = [t for t in line.split() if t != "truth"]
        rows.append([int(v) for v in toks[1:1 + K]])
    return np.array(rows)


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = tmp_path_factory.mktemp("anno")
    o = OracleNet(2, 3, K, 0.25, 4)
    p, r = random_params(o, 21)
    net = aa.RuntimeNet(aa.net_config(2, 3, K, 0.25, 4, aa.ANH_FP32))
    net.set_params(p, r)
    (d / "annonet.dnn").write_bytes(aa.dnn_envelope_pack("", 1.0, net.Serialize()))
    rng = np.random.default_rng(4)
    images = {}
    (d / "sub").mkdir()
    for name, (h, w), with_mask in (("a.png", (150, 170), True), ("sub/b.png", (97, 131), True), ("c.png", (260, 190), False)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pu.write_png(d / name, img, filter_type=1)
        gt = None
        if with_mask:
            gt = np.zeros((h, w), np.uint16)
            for _ in range(10):
                y, x = rng.integers(0, h), rng.integers(0, w)
                gt[y:y + rng.integers(5, 40), x:x + rng.integers(5, 40)] = rng.integers(0, K)
            gt[rng.random((h, w)) < 0.3] = 65535
            pu.write_png(str(d / name) + "_mask.png", pu.labels_to_rgba(gt))
        images[name] = (img, gt)
    return d, net, images


def run_tool(d, *extra):
    r = subprocess.run([TOOL, str(d), "--dnn", str(d / "annonet.dnn"), "-w", "96", "-h", "96", *extra], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_infer_main_end_to_end(dataset):
    d, net, images = dataset
    gains = [0.0, 0.1, 0.0]
    out = run_tool(d, "--precision", "fp32", "-g", "1:0.1", "--full-image-reader-thread-count", "2", "--result-image-writer-thread-count", "2")
    assert "Deserializing annonet, downscaling factor = 1" in out
    assert "Using gains: 0:0 1:0.1 2:0" in out and "Using detection levels: 0:0 1:0 2:0" in out
    assert " found 3 candidates" in out                                     # _mask.png / _result.png files are not inputs
    assert re.search(r"All 3 images processed in [0-9.]+ seconds! \(actual inference: [0-9.e-]+ seconds\)", out)
    assert re.search(r"Processing time excluding the first image: average = [0-9.e+-]+ ms, max = [0-9.e+-]+ ms", out)   # annonet_infer_main.cpp:498-507
    assert "All result images written!" in out
    ov = 35   # TrainingNet::GetRequiredInputDimension() of the tool's build (2 levels)
    tp = aa.tiling.parameters(96, 96, ov, ov)
    per_pixel, per_region = np.zeros((K, K), np.int64), np.zeros((K, K), np.int64)
    for name, (img, gt) in images.items():
        labels = aa.annonet_infer(net, img, gains=gains, tiling_parameters=tp)
        got = pu.read_png(str(d / name) + "_result.png")
        np.testing.assert_array_equal(got, pu.labels_to_rgba(labels))      # the tool's result image = the label map in anno colours
        if gt is not None:
            pp, pr = pu.confusion_matrices(gt, labels, K)
            per_pixel += pp
            per_region += pr
    lines = out.splitlines()
    i_pix = lines.index("Confusion matrix per pixel:")
    i_reg = lines.index("Confusion matrix per region (two-way):")
    np.testing.assert_array_equal(parse_matrix(lines, i_pix), per_pixel)
    np.testing.assert_array_equal(parse_matrix(lines, i_reg), per_region)
    assert per_pixel.sum() > 1000 and per_region.sum() > 10


def test_infer_main_detection_levels_and_two_replicas(dataset):
    d, net, images = dataset
    out = run_tool(d, "--precision", "fp32", "-d", "1:0.05", "-d", "2:0.05")
    assert "Using detection levels: 0:0 1:0.05 2:0.05" in out
    tp = aa.tiling.parameters(96, 96, 35, 35)
    img = images["a.png"][0]
    want = aa.annonet_infer(net, img, detection_levels=[0, 0.05, 0.05], tiling_parameters=tp)
    np.testing.assert_array_equal(pu.read_png(str(d / "a.png") + "_result.png"), pu.labels_to_rgba(want))
    # one process, two replicas (device 0 twice on a one-GPU box): every image's tile list is sharded, ONE label map comes back
    out = run_tool(d, "--precision", "fp32", "--devices", "0,0")
    labels = aa.annonet_infer(net, img, tiling_parameters=tp)
    got = pu.read_png(str(d / "a.png") + "_result.png")
    assert (got != pu.labels_to_rgba(labels)).any(axis=2).mean() < 1e-4     # equal except exact near-ties where four tiles meet
    out = run_tool(d)                                                      # the default: bf16 MFMA path
    assert "All result images written!" in out


def test_infer_main_errors(tmp_path, dataset):
    d = dataset[0]
    r = subprocess.run([TOOL, str(d), "--dnn", str(d / "annonet.dnn"), "-g", "7:1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Can't define class-specific value for index 7 when there are only 3 classes" in r.stdout
    r = subprocess.run([TOOL, str(d), "--dnn", str(d / "annonet.dnn"), "-g", "nonsense"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "index:gain" in r.stdout
    r = subprocess.run([TOOL, str(d), "--dnn", str(tmp_path / "missing.dnn")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Unable to open file" in r.stdout
    r = subprocess.run([TOOL], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "You call this program like this" in r.stdout
    r = subprocess.run([TOOL, "--nope"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2       # option errors return 2 (annonet_infer_main.cpp:330-335)
