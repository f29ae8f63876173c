"""Independent PyTorch-CPU statement of the net, built from a layer list.

Used only to cross-check the oracle ("PyTorch is an independent checker, not the
reference").  Works from the canonical parameter blob, whose filter layouts equal torch's
(Conv2d: [co][ci][kh][kw]; ConvTranspose2d: [ci][co][kh][kw]).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-4
IGNORE = 65535


def _slice(params, off, n):
    return params[off:off + n]


def forward(layers, params, running, images_u8, training, double=True):
    """Returns (logits NCHW, per-layer raw outputs, bn batch stats) ; params is a torch tensor (requires_grad ok)."""
    dt = torch.float64 if double else torch.float32
    x0 = torch.from_numpy(images_u8.astype(np.float32) / 256.0).to(dt).permute(0, 3, 1, 2)
    acts, raws, stats = [], [], []
    for L in layers:
        a = x0 if L.in_a < 0 else acts[L.in_a]
        if L.in_b >= 0:
            a = a + acts[L.in_b]
        nw = L.k * L.k * L.cin * L.cout
        w = _slice(params, L.w_off, nw)
        if L.type == 0:
            y = F.conv2d(a, w.view(L.cout, L.cin, L.k, L.k), stride=L.stride, padding=L.pad)
        else:
            y = F.conv_transpose2d(a, w.view(L.cin, L.cout, L.k, L.k), stride=L.stride, padding=L.pad)
        raws.append(y)
        if L.has_bn:
            g = _slice(params, L.g_off, L.cout).view(1, -1, 1, 1)
            b = _slice(params, L.beta_off, L.cout).view(1, -1, 1, 1)
            if training:
                m = y.mean(dim=(0, 2, 3), keepdim=True)
                v = y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
                stats.append((m.flatten().detach(), v.flatten().detach()))
            else:
                rm = torch.from_numpy(running[L.rs_off:L.rs_off + L.cout]).to(dt).view(1, -1, 1, 1)
                rv = torch.from_numpy(running[L.rs_off + L.cout:L.rs_off + 2 * L.cout]).to(dt).view(1, -1, 1, 1)
                m, v = rm, rv
                stats.append(None)
            z = (y - m) / torch.sqrt(v + BN_EPS) * g + b
            acts.append(torch.relu(z))
        else:
            stats.append(None)
            if L.has_bias:
                y = y + _slice(params, L.b_off, L.cout).view(1, -1, 1, 1)
            acts.append(y)
    return acts[-1], raws, stats


def loss_fn(logits, labels, weights, loss_scale_n):
    """loss_multiclass_log_per_pixel_weighted: sum_p w_p * -log softmax(z_p)[y_p] / (N*H*W); ignore=65535."""
    n, k, h, w = logits.shape
    lab = torch.from_numpy(labels.astype(np.int64))
    valid = lab != IGNORE
    lab = torch.where(valid, lab, torch.zeros_like(lab))
    logp = F.log_softmax(logits, dim=1)
    picked = logp.gather(1, lab.view(n, 1, h, w)).view(n, h, w)
    wt = torch.from_numpy(weights).to(logits.dtype) * valid.to(logits.dtype)
    return -(picked * wt).sum() / (loss_scale_n * h * w)
