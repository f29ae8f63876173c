"""Every kernel-schedule switch of the library must train the same net.

The switches are environment variables read once per process, so each variant runs
tests/helpers/run_train_steps.py in its own process: 3 bf16 steps on a seeded batch of 6 tiles 99x99.
  * variants that only move work between streams or kernels WITHOUT changing any summation order are bit-identical to
    the default: one stream instead of two; the stem's filter gradient queued on the second stream instead of the main one;
    the conv filter slabs streamed through LDS with every patch instead of staying resident; the gradient under the fused
    head written to memory instead of recomputed by its bn backward pass; stream priorities; who issues the backward-data convs' stores;
  * variants that change a summation order (bn sums in accumulator tables folded by their consumers vs. per-workgroup
    partials with finalize kernels; bn statistics / bn backward sums in a conv epilogue vs. the separate kernels; the classic one-tile conv kernels; dy materialised for the stem) agree to bf16-training tolerance.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def run_variant(tmp_path, name, env):
    out = str(tmp_path / f"{name}.npz")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "run_train_steps.py"), out], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    return run_variant(tmp_path_factory.mktemp("sched"), "default", {})


@pytest.mark.parametrize("name,env", [
    ("one_stream", {"ANH_CONCURRENT_WGRAD": "0"}),
    ("head_gradient_materialised", {"ANH_HEAD_DA_VIRTUAL": "0"}),
    ("main_stream_above_filter_gradient_stream", {"ANH_STREAM_PRIORITY": "1"}),
    ("stem_filter_gradient_on_second_stream", {"ANH_STEM_WGRAD_MAIN": "0"}),
    ("conv_filters_streamed_with_every_patch", {"ANH_WS_WEIGHT_RESIDENT": "0"}),
    ("skip_gradient_written_to_both_sources", {"ANH_SKIP_GRAD_ONCE": "0"}),
    ("backward_data_stores_issued_by_the_consumer_waves", {"ANH_WS_PSTAT": "1"}),     # round 3's form; the default (7) lets the producer waves issue them
    ("backward_data_stores_by_the_producers_for_stride_1_and_down_only", {"ANH_WS_PSTAT": "8"}),
    ("filter_fragments_of_the_32_channel_conv_read_from_lds_every_item", {"ANH_WS_FILTER_REGS": "0"}),
])
def test_schedule_is_bit_identical(tmp_path, default_run, name, env):
    got = run_variant(tmp_path, name, env)
    np.testing.assert_array_equal(got["losses"], default_run["losses"])
    np.testing.assert_array_equal(got["params"], default_run["params"])
    np.testing.assert_array_equal(got["running"], default_run["running"])


@pytest.mark.parametrize("name,env", [
    ("bn_sums_as_partials_with_finalize_kernels", {"ANH_BN_TABLES": "0"}),
    ("conv_tiles_walked_with_the_grid_stride", {"ANH_WS_XCD_BANDS": "0"}),
    ("filter_gradient_tiles_walked_in_xcd_bands", {"ANH_WGRAD_XCD_BANDS": "1"}),
    ("filter_gradient_tiles_walked_with_the_grid_stride", {"ANH_WGRAD_XCD_BANDS": "0"}),
    ("separate_reduce_pass_for_the_layer_behind_the_64_channel_up_conv", {"ANH_WS_WIDE_PS": "0"}),
    ("fused_head_on_768_workgroups", {"ANH_HEAD_BLOCKS": "768"}),
    ("separate_bn_statistics", {"ANH_FUSE_BN_STATS": "0"}),
    ("separate_bn_backward_reduction", {"ANH_FUSE_BN_BWD_REDUCE": "0"}),
    ("stem_dy_materialised", {"ANH_FUSE_STEM_BN_APPLY": "0"}),
    ("classic_conv_kernels", {"ANH_CONV_WS": "0"}),
    ("single_role_filter_gradient_kernels", {"ANH_WGRAD_WS": "0"}),
])
def test_schedule_agrees_within_bf16_training_tolerance(tmp_path, default_run, name, env):
    got = run_variant(tmp_path, name, env)
    np.testing.assert_allclose(got["losses"], default_run["losses"], rtol=2e-3)
    dp = got["params"] - default_run["params"]
    ref = default_run["params"]
    assert np.linalg.norm(dp) <= 2e-3 * np.linalg.norm(ref), float(np.linalg.norm(dp) / np.linalg.norm(ref))
    run = default_run["running"]   # means (near zero) and variances: absolute bar relative to the largest statistic
    np.testing.assert_allclose(got["running"], run, rtol=2e-3, atol=2e-3 * np.abs(run).max())


def test_backward_replayed_as_a_captured_graph_is_bit_identical(tmp_path):
    """ANH_STEP_GRAPH=1: the launches behind the fused head (both streams, the dy hand-overs and the join as graph edges) captured
    once per (shape, buffers, input pointer) and replayed.  StartTraining alternates two staging sets, so a key is captured on its
    third use: ten steps replay each of the two graphs twice.  Same kernels, same arguments, same order: bit-identical."""
    steps = {"ANH_TEST_STEPS": "10"}
    eager = run_variant(tmp_path, "eager10", steps)
    graph = run_variant(tmp_path, "graph10", dict(steps, ANH_STEP_GRAPH="1"))
    assert tuple(eager["step_graph"]) == (0, 0)
    captures, launches = (int(v) for v in graph["step_graph"])
    assert captures == 2 and launches >= 4, (captures, launches)   # (a key changes once while the first steps size their scratch buffers)
    np.testing.assert_array_equal(graph["losses"], eager["losses"])
    np.testing.assert_array_equal(graph["params"], eager["params"])
    np.testing.assert_array_equal(graph["running"], eager["running"])
