"""The per-op oracle bars of test_gpu_ops.py in the regime bench.py runs.

The persistent MFMA kernels (conv3x3_ws, wgrad3x3_ws, the stem kernels) run 256 workgroups; at batch 32 x 227 x 227 a workgroup
walks 29 items (32->32 at side 227), 7.5 (64->64 at 113), 3.5 (128->128 at 56).  The shapes of test_gpu_ops.py are small enough for
the oracle to finish in milliseconds, which leaves <= 1 item per workgroup there: ring-buffer wrap-around, the XCD-band walk, the
producer-issued stores and sums of the tile finished two items ago and the tile-ahead prefetch of read-modify-write destinations
never reach their steady state.  Here the same assertions (tests/helpers/run_ops_regime.py calls the test functions of
test_gpu_ops.py) run in child processes whose grids are shrunk to 3 and 8 workgroups — every workgroup walks >= 4 items, grid 8
takes the band walk, grid 3 the strided one — and on full-width planes (n = 2, side 227) at the default grid and at 64 workgroups.
Reference call whose arithmetic this guards: TrainingNet::StartTraining, /root/reference/annonet_train_main.cpp:609.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def run_child(which, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "run_ops_regime.py"), which], env=e, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-6000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "failed cases: 0" in r.stdout, tail
    return r.stdout


@pytest.mark.parametrize("wgs", [3, 8])
def test_mfma_ops_keep_their_oracle_bars_when_every_workgroup_walks_many_items(wgs):
    out = run_child("small", {"ANH_WS_WGS": str(wgs), "ANH_WGRAD_WGS": str(wgs), "ANH_STEM_BLOCKS": str(wgs), "ANH_STEM_WGRAD_BLOCKS": str(wgs)})
    assert out.count("ok   ") >= 7 * 11 + 3


@pytest.mark.parametrize("wgs", [0, 64])
def test_mfma_ops_keep_their_oracle_bars_on_full_width_planes(wgs):
    env = {} if wgs == 0 else {"ANH_WS_WGS": str(wgs), "ANH_WGRAD_WGS": str(wgs)}
    out = run_child("full", env)
    assert out.count("ok   ") >= 3 * 11
