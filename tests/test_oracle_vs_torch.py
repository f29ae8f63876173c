"""The oracle's net arithmetic against an independent PyTorch-CPU (fp64) statement of the same spec.

"Parity unpinned" for this part of the path: the reference holds no test or fixture for the net
(SURVEY.md §8c), so the oracle is validated against torch's conv2d / conv_transpose2d / batch-norm /
weighted cross-entropy instead.
"""
import numpy as np
import pytest
import torch

import torch_ref
from conftest import random_params
from oracle.oracle import OracleNet, IGNORE

torch.set_num_threads(4)


def make_batch(rng, n, d, in_ch, classes, ignore_frac=0.05):
    img = rng.integers(0, 256, (n, d, d, in_ch), dtype=np.uint8)
    lab = rng.integers(0, classes, (n, d, d)).astype(np.uint16)
    lab[rng.random((n, d, d)) < ignore_frac] = IGNORE
    w = rng.uniform(0.2, 2.0, (n, d, d)).astype(np.float32)
    w[lab == IGNORE] = 0
    return img, lab, w


@pytest.mark.parametrize("levels,in_ch,scaler", [(0, 3, 0.25), (1, 3, 0.25), (2, 3, 0.25), (3, 1, 0.125), (2, 1, 0.5)])
def test_forward_inference_matches_torch(levels, in_ch, scaler):
    net = OracleNet(levels, in_ch, 3, scaler, 4)
    p, r = random_params(net, 7)
    net.params[:] = p
    net.running[:] = r
    d = net.recommended_input_dim(21)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (2, d, d, in_ch), dtype=np.uint8)
    got = net.forward(img)
    want, _, _ = torch_ref.forward(net.layers, torch.from_numpy(p).double(), r, img, training=False)
    want = want.numpy()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


def test_spec_shapes_and_dims():
    for levels in range(4):
        net = OracleNet(levels, 3, 3, 1.0, 1)
        assert len(net.layers) == 2 + 4 * levels
        q = 1 << levels
        for n in (1, 5, 35, 100, 227, 1024):
            d = net.recommended_input_dim(n)
            assert d >= n and (d - (q - 1)) % q == 0 and d >= 2 * q - 1
            if n >= 2 * q - 1:
                assert d - n < q
    assert OracleNet(2).recommended_input_dim(227) == 227
    assert OracleNet(2).required_input_dim() == 35
    assert OracleNet(0).required_input_dim() == 5
    assert OracleNet(1).required_input_dim() == 15


def test_receptive_field_is_tight():
    """A pixel further than required_input_dim()//2 from a change must not move; one inside does."""
    net = OracleNet(2, 3, 3, 0.25, 4)
    p, r = random_params(net, 3)
    net.params[:] = p
    net.running[:] = r
    rf = net.required_input_dim()
    d = net.recommended_input_dim(2 * rf + 9)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1, d, d, 3), dtype=np.uint8)
    base = net.forward(img)
    c = d // 2
    img2 = img.copy()
    img2[0, c, c] ^= 0xFF
    out = net.forward(img2)
    diff = np.abs(out - base).max(axis=(0, 1))
    ys, xs = np.nonzero(diff)
    assert ys.size > 0
    half = rf // 2
    assert max(abs(ys - c).max(), abs(xs - c).max()) <= half


@pytest.mark.parametrize("levels,in_ch", [(0, 3), (1, 1), (2, 3)])
def test_train_step_matches_torch(levels, in_ch):
    classes = 3
    net = OracleNet(levels, in_ch, classes, 0.25, 4)
    p, r = random_params(net, 11)
    net.params[:] = p
    net.running[:] = r
    net.set_hyper(lr=0.05, wd=0.0005, mom=0.9, bn_window=100)
    rng = np.random.default_rng(2)
    mom0 = rng.normal(0, 1e-3, net.n_params).astype(np.float32)
    net.momentum[:] = mom0
    d = net.recommended_input_dim(19)
    img, lab, w = make_batch(rng, 3, d, in_ch, classes)
    loss = net.train_step(img, lab, w)

    tp = torch.from_numpy(p).double().requires_grad_(True)
    logits, raws, stats = torch_ref.forward(net.layers, tp, r, img, training=True)
    tl = torch_ref.loss_fn(logits, lab, w, 3)
    tl.backward()
    assert abs(loss - tl.item()) < 1e-5 * max(1.0, abs(tl.item()))
    g = tp.grad.numpy()
    scale = np.abs(g).max()
    np.testing.assert_allclose(net.grads, g, rtol=2e-3, atol=2e-5 * scale)
    # SGD with momentum + weight decay on filters only
    wd_mask = np.zeros(net.n_params, dtype=np.float64)
    for L in net.layers:
        wd_mask[L.w_off:L.w_off + L.k * L.k * L.cin * L.cout] = 1.0
    v = 0.9 * mom0 - 0.0005 * 0.05 * wd_mask * p - 0.05 * g
    np.testing.assert_allclose(net.momentum, v, rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(net.params, p + v, rtol=1e-4, atol=1e-6)
    # running stats: first update has averaging factor 1
    for L, st in zip(net.layers, stats):
        if st is None:
            continue
        m, var = st
        P = 3 * raws[net.layers.index(L)].shape[2] * raws[net.layers.index(L)].shape[3]
        np.testing.assert_allclose(net.running[L.rs_off:L.rs_off + L.cout], m.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(net.running[L.rs_off + L.cout:L.rs_off + 2 * L.cout], var.numpy() * P / (P - 1), rtol=1e-4, atol=1e-6)


def test_loss_scale_uses_global_batch():
    net = OracleNet(0, 3, 3, 0.25, 4)
    p, r = random_params(net, 4)
    rng = np.random.default_rng(3)
    img, lab, w = make_batch(rng, 2, 9, 3, 3)
    net.params[:] = p
    l1 = net.train_step(img, lab, w, loss_scale_n=2, apply_update=False)
    g1 = net.grads.copy()
    net.params[:] = p
    l8 = net.train_step(img, lab, w, loss_scale_n=8, apply_update=False)
    assert abs(l1 / 4 - l8) < 1e-7
    np.testing.assert_allclose(net.grads, g1 / 4, rtol=1e-5, atol=1e-9)


def test_bad_label_and_bad_size_raise():
    net = OracleNet(2, 3, 3, 0.25, 4)
    img = np.zeros((1, 24, 24, 3), np.uint8)
    with pytest.raises(RuntimeError):
        net.forward(img)  # 24 is not 4m+3
    img = np.zeros((1, 23, 23, 3), np.uint8)
    lab = np.full((1, 23, 23), 7, np.uint16)
    with pytest.raises(RuntimeError):
        net.train_step(img, lab, np.ones((1, 23, 23), np.float32))
