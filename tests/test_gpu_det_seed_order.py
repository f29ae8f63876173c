"""ANH_DET_SEED_REFERENCE_ORDER=1: the reference's detection-level filter as it is written — seeds are stored as point(r, c)
(annonet_infer.cpp:210) and the blob is looked up at (point.y(), point.x()) = (c, r) (:222), i.e. a seed keeps the blob that holds
the TRANSPOSED pixel.  The default build looks the seed's own pixel up.  On square images the reference's behaviour
is well defined; this test holds the switch to a numpy / scipy restatement of exactly that, in a process of its own (the switch is
read once per process)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy import ndimage

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import annonet_amd as aa
from conftest import random_params
from oracle.oracle import OracleNet
o = OracleNet(1, 3, 3, 0.25, 4)
p, r = random_params(o, 13)
net = aa.RuntimeNet(aa.net_config(1, 3, 3, 0.25, 4, aa.ANH_FP32))
net.set_params(p, r)
rng = np.random.default_rng(2)
side = 121
image = np.kron(rng.integers(0, 256, (side // 11, side // 11, 3)), np.ones((11, 11, 1))).astype(np.uint8)   # blocky: multi-pixel blobs
tp = aa.tiling.parameters(64, 64, o.required_input_dim(), o.required_input_dim())
plain, planes = aa.annonet_infer(net, image, tiling_parameters=tp, want_blended=True)
spread = np.sort(planes, axis=0)
det = [0.0, float(np.median(spread[2] - spread[0])), float(np.median(spread[2] - spread[0]))]
filtered = aa.annonet_infer(net, image, detection_levels=det, tiling_parameters=tp)
np.savez(sys.argv[2], plain=plain, planes=planes, filtered=filtered, det=np.array(det))
'''


def run(tmp_path, name, env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", SCRIPT, os.path.dirname(HERE), out], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def restate(plain, planes, det, transposed):
    h, w = plain.shape
    lab = plain.astype(np.int64)
    own = np.take_along_axis(planes, np.clip(lab, 0, 2)[None], axis=0)[0]
    seeds = (lab != 0) & ((own - planes[0]).astype(np.float64) > det[np.clip(lab, 0, 2)] - det[0])   # float difference, compared in double (annonet_infer.cpp:205-212)
    if transposed:
        t = np.zeros_like(seeds)
        rr, cc = np.nonzero(seeds)
        ok = (cc < h) & (rr < w)
        t[cc[ok], rr[ok]] = True
        seeds = t
    out = plain.copy()
    for v in np.unique(lab):
        if v == 0:
            continue
        blobs, n = ndimage.label(lab == v, structure=np.ones((3, 3)))
        keep = np.unique(blobs[seeds & (blobs > 0)])
        out[(blobs > 0) & ~np.isin(blobs, keep)] = 0
    return out


def test_reference_seed_order_switch(tmp_path):
    a = run(tmp_path, "default", {})
    b = run(tmp_path, "reference", {"ANH_DET_SEED_REFERENCE_ORDER": "1"})
    np.testing.assert_array_equal(a["plain"], b["plain"])
    np.testing.assert_array_equal(a["filtered"], restate(a["plain"], a["planes"], a["det"], transposed=False))
    np.testing.assert_array_equal(b["filtered"], restate(b["plain"], b["planes"], b["det"], transposed=True))
    assert (a["filtered"] != a["plain"]).any(), "the detection levels removed nothing: the test does not exercise the filter"
    assert (a["filtered"] != b["filtered"]).any(), "both lookups agree on this image: the test does not exercise the switch"
