"""Generates the committed golden vectors from the oracle (oracle/annonet_oracle.cpp).

The reference cannot be built or imported here (its arithmetic lives in absent submodules), so these
are vectors of the build's CPU restatement, frozen so that neither the oracle nor the HIP path can drift silently.
The reference's own known answers (set_weights, test/annonet_test.cpp:54-120) are stored alongside as data.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz, *.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import random_params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.oracle import OracleNet  # noqa: E402


def infer_case():
    cfg = dict(levels=2, in_ch=3, classes=3, scaler=0.25, min_filters=4)
    net = OracleNet(**cfg)
    p, r = random_params(net, 20260116)
    net.params[:], net.running[:] = p, r
    rng = np.random.default_rng(3)
    tile = rng.integers(0, 256, (1, 27, 31, 3), dtype=np.uint8)
    logits = net.forward(tile)
    image = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    ov = net.required_input_dim()
    gains = np.array([0.0, 0.25, -0.125])
    labels, blended = net.infer(image, gains=gains, max_tile=(75, 75), overlap=ov, want_blended=True)
    np.savez_compressed(os.path.join(HERE, "infer_fp32.npz"), cfg=json.dumps(cfg), params_seed=20260116, tile=tile, logits=logits, image=image,
                        gains=gains, max_tile=75, overlap=ov, labels=labels, blended=blended.astype(np.float32))


def train_case():
    cfg = dict(levels=1, in_ch=3, classes=3, scaler=0.25, min_filters=4)
    net = OracleNet(**cfg)
    p, r = random_params(net, 77)
    net.params[:], net.running[:] = p, r
    net.set_hyper(lr=0.05, wd=0.0005, mom=0.9, bn_window=100)
    rng = np.random.default_rng(5)
    d = net.recommended_input_dim(19)
    img = rng.integers(0, 256, (2, d, d, 3), dtype=np.uint8)
    lab = rng.integers(0, 3, (2, d, d)).astype(np.uint16)
    lab[rng.random((2, d, d)) < 0.1] = 65535
    w = np.stack([orc.set_weights(lab[i], 0.5, 0.5) for i in range(2)])
    loss = net.train_step(img, lab, w)
    np.savez_compressed(os.path.join(HERE, "train_step_fp32.npz"), cfg=json.dumps(cfg), params_seed=77, images=img, labels=lab, weights=w, loss=loss,
                        grads=net.grads.copy(), params_after=net.params.copy(), running_after=net.running.copy(), lr=0.05)


def host_cases():
    data = {
        # test/annonet_test.cpp:11-18,54-120 — the reference's known answers for set_weights
        "set_weights": {"labels": [[0, 65535, 1, 0, 0]],
                        "cases": [{"class_weight": 0.0, "image_weight": 0.0, "weights": [1.0, 0.0, 1.0, 1.0, 1.0], "total": 4.0, "tol": 0.0},
                                  {"class_weight": 1.0, "image_weight": 0.0, "weights": [0.666667, 0.0, 2.0, 0.666667, 0.666667], "total": 4.0, "tol": 1e-6},
                                  {"class_weight": 0.5, "image_weight": 0.0, "weights": [0.845299, 0.0, 0.845299 * 3 ** 0.5, 0.845299, 0.845299], "total": 4.0, "tol": 1e-6},
                                  {"class_weight": 0.0, "image_weight": 1.0, "weights": [1.25, 0.0, 1.25, 1.25, 1.25], "total": 5.0, "tol": 0.0}]},
        "tiles": [{"args": a, "tiles": orc.get_tiles(*a)} for a in
                  [(227, 227, 1024, 1024, 35, 35), (4096, 4096, 1024, 1024, 35, 35), (1025, 700, 1024, 1024, 35, 35), (300, 200, 120, 110, 15, 35)]],
        "dims": {str(levels): {"required": OracleNet(levels).required_input_dim(),
                               "recommended": {str(n): OracleNet(levels).recommended_input_dim(n) for n in (1, 35, 105, 227, 1024)}} for levels in range(4)},
    }
    with open(os.path.join(HERE, "host_logic.json"), "w") as f:
        json.dump(data, f, indent=1)


if __name__ == "__main__":
    infer_case()
    train_case()
    host_cases()
    print("golden vectors written to", HERE)
