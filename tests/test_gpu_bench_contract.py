"""bench.py's one-line contract (the driver parses it): metric / value / unit / step counts, the roofline object of the
dominant kernel measured live, the CPU baseline of the same run, and the fields that must NOT claim more than was measured.
These tests check the SHAPE of the line (fields, units, consistency between fields) — never a rate: a wall-clock bound in a
test is noise on a shared box, and under -x it would hide every test collected after it (tests/conftest.py runs this file last)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout       # exactly ONE json line on stdout
    return json.loads(lines[0])


def test_training_bench_line():
    d = run_bench("--gpus", "1", "--steps", "8", "--warmup", "2", "--prewarm-s", "0.5", "--infer-steps", "2")
    assert d["metric"].startswith("227x227 RGB tiles/sec fwd+bwd") and d["unit"] == "tiles/s"
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]     # value = tiles of all ranks / time
    assert d["value"] > 0 and d["prewarm_s"] == 0.5 and d["prewarm_steps"] >= 4
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    assert r["kernel"].split(":")[1].split("_")[0] in ("fwd", "dgrad", "wgrad")   # a §8d conv pass, not a bn / reduction kernel
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "tiles/s" and 1 <= c["cores"] <= 16 and c["value"] > 0 and c["sample"]
    assert c["one_thread"]["cores"] == 1 and c["one_thread"]["value"] > 0
    assert "pytorch_cpu" in c and ("value" in c["pytorch_cpu"] or "error" in c["pytorch_cpu"])
    # per-layer table: every 3x3 / 5x5 layer x {fwd, bwd-data, bwd-filter} (the stem has no bwd-data), each with its §8d bound
    rows = d["layers"]
    assert len(rows) == 9 * 3 - 1
    assert {x["pass"] for x in rows} == {"fwd", "bwd-data", "bwd-filter"}
    for x in rows:
        assert x["bound"] in ("hbm", "mfma") and 0 < x["frac"] < 1.0 and x["time_us"] >= x["floor_us"] > 0
    assert abs(sum(x["gflop"] for x in rows) - (564.6 - 3 * 0.32 - 7.91)) < 1.5      # 564.6 GFLOP per step incl. the 1x1 head (3 passes) and the stem's absent bwd-data
    assert any(o["entry"].startswith("bn_") for o in d["overhead_kernels"])
    assert d["ranks_seen"] == 1 and d["devices"] == [0]
    # round 4: the host-vector rate of the same K steps (the reference loop's StartTraining), the main stream's critical path, and the
    # matrix-pipe occupancy of every conv entry with its denominator stated — shapes only
    assert d["value_host_inputs"] > 0 and "PCIe-inclusive" in d["value_note"]
    cp = d["critical_path"]
    for k in ("forward", "head", "backward_data", "apply", "tail", "hidden_second_stream", "main_stream_kernels", "step", "gaps"):
        assert isinstance(cp[k], float), k
    assert cp["forward"] > 0 and cp["backward_data"] > 0 and cp["apply"] > 0 and cp["tail"] > 0 and cp["hidden_second_stream"] > 0
    assert abs(cp["step"] - 1e3 * d["ms_per_step"]) < 0.1 and abs(cp["main_stream_kernels"] + cp["gaps"] - cp["step"]) < 0.2
    assert "SQ_VALU_MFMA_BUSY_CYCLES" in d["mfma_busy_note"] and "SQ_BUSY_CYCLES" in d["mfma_busy_note"]
    for x in rows:
        assert x["mfma_busy_pct"] is None or 0 < x["mfma_busy_pct"] < 100
    assert "direct_oracle" in c and "im2col" in c["sample"]
    # the same line carries tiled inference (BASELINE.json configs[2]), measured by a child process after the training measurement
    i = d["infer"]
    assert "error" not in i, i
    assert i["unit"] == "Mpx/s" and i["n_gpus"] == 1 and i["value"] > 0 and i["value_labels_on_host"] > 0 and "4096x4096" in i["metric"]
    assert i["roofline"]["bytes_convention"].startswith("SURVEY 8d minimum") and i["cpu_baseline"]["unit"] == "Mpx/s"
    assert [o["metric"].split(",")[1].strip() for o in i["other_sizes"]] == ["16384x16384 image"]


def test_self_launch_refuses_more_gpus_than_visible():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr


def test_launched_as_a_rank_takes_the_rccl_path():
    """python -m torch.distributed.run --nproc-per-node 1 (what the self-launcher starts, at N = 1): one rank, RCCL all-reduce on the bucket"""
    from conftest import free_port
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--prewarm-s", "0.2", "--no-cpu-baseline", "--no-infer"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["ranks_seen"] == 1 and d["n_gpus"] == 1 and d["value"] > 0 and "infer" not in d


def test_inference_bench_line():
    d = run_bench("--mode", "infer", "--image-side", "2048", "--steps", "3", "--warmup", "1", "--prewarm-s", "0.2")
    assert d["unit"] == "Mpx/s" and d["n_gpus"] == 1 and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["config"]["exchanged_pixels"] == 0 and d["value"] > 0 and d["value_labels_on_host"] > 0 and d["steps"] == 3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1.0 and "fwd_L" in r["kernel"]
    assert r["algorithmic_bytes_per_launch"] <= r["design_bytes_per_launch"] * 1.001 and r["tiles_per_launch"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mpx/s" and c["value"] > 0 and c["one_thread"]["cores"] == 1


def test_fp32_inference_entries_name_the_matrix_kernels():
    d = run_bench("--mode", "infer", "--image-side", "1024", "--steps", "1", "--warmup", "1", "--prewarm-s", "0", "--precision", "fp32", "--no-cpu-baseline")
    assert d["roofline"]["kernel"].startswith("conv_mfma_f32:fwd_L")     # the fp32 parity mode runs on v_mfma_f32_32x32x2_f32, and says so


def test_two_rank_job_rehearsed_on_one_gpu():
    """The N > 1 paths of bench.py as a REAL two-process job — self-launch, rank bookkeeping, the two-part gradient exchange
    (dist.EarlyReduce), the agreed pre-warm, and in the inference child the overlap exchange and the label gather — with both ranks on
    device 0 and gloo as the transport (RCCL wants one GPU per rank; the one-GPU test box cannot run it at world size 2)."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--one-gpu", "--steps", "3", "--warmup", "1", "--prewarm-s", "0.3", "--no-cpu-baseline",
                  "--infer-steps", "2", "--infer-sides", "2048")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["devices"] == [0, 0] and d["config"]["global_batch"] == 64
    assert d["value"] > 0 and abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]      # whole-job tiles per second
    assert d["cpu_baseline"] is None and d["scaling"] == "weak" and d["prewarm_steps"] >= 4
    i = d["infer"]
    assert "error" not in i, i
    assert i["n_gpus"] == 2 and i["ranks_seen"] == 2 and i["config"]["parallelism"] == "tile-shard2" and i["config"]["exchanged_pixels"] > 0
    assert i["value"] > 0 and i["value_labels_on_host"] > 0
    # what makes ONE driver run at N = 8 diagnosable (round 4): the exchange step's device time, every rank's own step time, and host (B)
    x = d["exchange"]
    assert x["early_reduce"] is True and set(x["allreduce_us"]) == {"tail", "head"} and all(v > 0 for v in x["allreduce_us"].values())
    assert x["backend"] == "gloo" and x["rccl_version"] is None and x["bucket_bytes"] == (418435 + 1) * 4
    r = x["ms_per_step_per_rank"]
    assert len(r["all"]) == 2 and 0 < r["min"] <= r["max"] and r["max"] <= d["ms_per_step"] * 1.0001
    b = d["in_process"]
    assert "error" not in b, b
    assert b["replicas"] == 2 and b["devices"] == [0, 0] and b["global_batch"] == 64 and b["value"] > 0 and b["host_us_per_start_training"] > 0
    assert abs(b["value"] - 64 / (b["ms_per_step"] * 1e-3)) <= 1e-6 * b["value"]


def test_in_process_host_rehearsed_with_two_replicas_on_one_gpu():
    """Host (B) as its own line: ONE process, anh_set_devices([0, 0]) — the path /root/reference/annonet_train_main.cpp:583-614 would use
    unchanged — StartTraining with 64 host samples per step, persistent worker threads, the in-library exchange (rehearsal transport)."""
    d = run_bench("--in-process", "--devices", "0,0", "--steps", "9", "--warmup", "2", "--prewarm-s", "0.3")
    assert d["replicas"] == 2 and d["devices"] == [0, 0] and d["n_gpus"] == 1 and d["global_batch"] == 64 and d["unit"] == "tiles/s"
    assert d["value"] > 0 and abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 0 < d["host_busy_us_per_start_training"] <= d["host_us_per_start_training"] <= 1e3 * d["ms_per_step"] * 1.5
    assert d["worker_wakeups_per_step"] == 1.0                    # ONE wake-up of the persistent workers per StartTraining
    x = d["exchange"]
    assert x["early_reduce"] is True and x["sampled_steps"] >= 1 and x["allreduce_tail_us"] > 0 and x["allreduce_head_us"] > 0
    assert x["transport"].startswith("repeated-device") and x["bucket_bytes"] == (418435 + 1) * 4 and x["rccl_version"] > 0
