import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def random_params(net, seed, scale=1.0):
    """Glorot-uniform filters, non-trivial bn gamma/beta and running stats: same blob for oracle and product."""
    rng = np.random.default_rng(seed)
    p = np.zeros(net.n_params, dtype=np.float32)
    r = np.zeros(net.n_running, dtype=np.float32)
    for L in net.layers:
        nw = L.k * L.k * L.cin * L.cout
        fan = L.k * L.k * (L.cin + L.cout)
        lim = scale * np.sqrt(6.0 / fan)
        p[L.w_off:L.w_off + nw] = rng.uniform(-lim, lim, nw)
        if L.has_bias:
            p[L.b_off:L.b_off + L.cout] = rng.uniform(-0.1, 0.1, L.cout)
        if L.has_bn:
            p[L.g_off:L.g_off + L.cout] = rng.uniform(0.5, 1.5, L.cout)
            p[L.beta_off:L.beta_off + L.cout] = rng.uniform(-0.3, 0.3, L.cout)
            r[L.rs_off:L.rs_off + L.cout] = rng.uniform(-0.2, 0.2, L.cout)
            r[L.rs_off + L.cout:L.rs_off + 2 * L.cout] = rng.uniform(0.02, 0.3, L.cout)
    return p, r


@pytest.fixture
def rp():
    return random_params
