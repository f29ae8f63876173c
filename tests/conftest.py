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


# Collection order (the driver runs the suite with -x): comparisons of the HIP path with the oracle / the golden fixtures run
# first, the end-to-end tool runs after them, bench.py's line contract last — a failure in a harness test can then never hide a
# parity test.  Files not named here keep their alphabetical place between the groups.
_ORDER_FIRST = ["test_golden", "test_gpu_ops", "test_gpu_ops_regime", "test_gpu_parity", "test_gpu_crops", "test_gpu_det_seed_order", "test_gpu_sharded_infer",
                "test_gpu_multidev", "test_gpu_trainer_state", "test_gpu_schedules", "test_gpu_trained_precision", "test_gpu_convergence",
                "test_gpu_errors", "test_dnn_envelope", "test_cpp_shim"]
_ORDER_LAST = ["test_gpu_infer_main", "test_gpu_train_main", "test_gpu_exit_order", "test_gpu_bench_contract"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _ORDER_FIRST:
            return (0, _ORDER_FIRST.index(name))
        if name in _ORDER_LAST:
            return (2, _ORDER_LAST.index(name))
        return (1, 0)
    items.sort(key=rank)   # stable: the order inside a file is kept


def free_port():
    """a free TCP port on 127.0.0.1 for a torch.distributed rendezvous (every test takes its own)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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
