"""The reference's own unit tests (test/annonet_test.cpp:54-130), re-hosted.

These are the only known answers the reference holds for the path; they pin
set_weights (annonet_train.h:20-83) and random_rect_containing_point (annonet_train.h:85-105).
Each case runs against the oracle here and, in test_host_logic.py, against the product library.
"""
import math

import numpy as np
import pytest

from oracle import oracle as orc

IGN = orc.IGNORE
LABELS = np.array([[0, IGN, 1, 0, 0]], dtype=np.uint16)  # test/annonet_test.cpp:11-18

# (class_weight, image_weight) -> expected weights, expected total, exact?
CASES = [
    ((0.0, 0.0), [1.0, 0.0, 1.0, 1.0, 1.0], 4.0, True),                                      # WeighsPixelsEquivalent
    ((1.0, 0.0), [0.666667, 0.0, 2.0, 0.666667, 0.666667], 4.0, False),                      # WeighsClassesEquivalent
    ((0.5, 0.0), [0.845299, 0.0, 0.845299 * math.sqrt(3), 0.845299, 0.845299], 4.0, False),  # WeighsEvenInBetween
    ((0.0, 1.0), [1.25, 0.0, 1.25, 1.25, 1.25], 5.0, True),                                  # WeighsImagesEquivalent
]


def check_set_weights(fn):
    for (cw, iw), want, total, exact in CASES:
        got = fn(LABELS, cw, iw)
        assert got.shape == LABELS.shape
        assert got[0, 1] == 0.0
        if exact:
            assert got[0].tolist() == want
            assert float(got.astype(np.float64).sum()) == total
        else:
            np.testing.assert_allclose(got[0], want, atol=1e-6, rtol=0)
            assert abs(float(got.astype(np.float64).sum()) - total) < 1e-6


def check_random_rect(fn):
    # GeneratesRandomRectContainingPoint: width/height exact, contains the point — for many draws
    rng = np.random.default_rng(0)
    for _ in range(200):
        dx, dy = (int(v) for v in rng.integers(0, 2**32, 2))
        w, h = (int(v) for v in rng.integers(1, 40, 2))
        l, t, r, b = fn(dx, dy, 50, 50, w, h)
        assert r - l + 1 == w and b - t + 1 == h
        assert l <= 50 <= r and t <= 50 <= b
    l, t, r, b = fn(12345, 67890, 50, 50, 10, 10)
    assert (r - l + 1, b - t + 1) == (10, 10)


def test_set_weights_known_answers_oracle():
    check_set_weights(orc.set_weights)


def test_random_rect_oracle():
    check_random_rect(orc.random_rect_containing_point)


def test_set_weights_all_ignored_and_sparse_labels():
    lab = np.full((3, 4), IGN, np.uint16)
    assert not orc.set_weights(lab, 0.5, 0.5).any()
    lab = np.array([[40, 40, 2, IGN]], np.uint16)  # label ids need not be dense (annonet_train.h:29-34)
    w = orc.set_weights(lab, 1.0, 0.0)
    assert w[0, 3] == 0 and abs(w.sum() - 3.0) < 1e-6 and w[0, 2] == pytest.approx(2 * w[0, 0])
