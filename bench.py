#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: 227x227 RGB tiles/s, training step (forward + loss + backward +
SGD update), batch 32 per GPU, bf16, random-init net (levels 2, width 1.0, K=3), synthetic tiles.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N rank processes itself, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --mode infer [--image-side 4096]         BASELINE.json configs[2] / [4]: tiled inference, Mpx/s

One process per GPU; for N > 1 the only exchange step of the path is the all-reduce (RCCL) of the flat gradient
bucket (training) / of the cross-rank overlap sums (inference).  Inputs are resident in HBM before the timed region.
Before the W warm-up steps the workload runs untimed for `prewarm_s` seconds (declared in the line) so that the clocks have
settled when a short --steps / --warmup run is timed.  The default (training) line also carries an `infer` object: tiled
inference over a 4096^2 image (BASELINE.json configs[2]; 16384^2 = configs[4] for N > 1), measured by a child process that
runs `bench.py --mode infer` after the training measurement has finished (a failure there is reported inside `infer`, it
cannot take the training line with it).
Prints ONE JSON line on rank 0, with
  roofline     — the §8d conv entry (layer x pass) with the largest time: algorithmic flops and minimum bytes per launch
                  / its mean launch time, measured with HIP events on the compute stream inside the
                 timed region; `traffic` = HBM-side bytes per launch from the committed rocprofv3 --pmc passes;
  layers       — every layer x {fwd, bwd-data, bwd-filter}: time (HIP events, untimed profile pass of the same step),
                 §8d flops and minimum bytes, bound = min(MFMA, AI x HBM), fraction of that bound;
  overhead_kernels — everything in the step that is not a §8d conv pass (bn finalize / apply, partial sums, SGD ...);
  cpu_baseline — the oracle (a CPU port of the path, see oracle/) timed on this host on a bounded sample, all threads and
                 one thread, plus a PyTorch-CPU (oneDNN) figure of the same net as a second, clearly labelled baseline;
  ranks_seen / devices — an all-reduce of ones and the device index of every rank.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises: keeps the trainer's two streams on separate hardware queues (annonet_amd/_lib.py)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE = 227
BATCH = 32
LEVELS, WIDTH, CLASSES = 2, 1.0, 3
# peaks from /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec), bf16 MFMA ~2.5 PFLOP/s dense (spec)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0
RIDGE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
PROFILE_STEPS = 4          # untimed all-kernel profile pass (feeds `layers`)
PREWARM_S = 3.0            # untimed, time-based pre-warm before the W warm-up steps (declared in the line as prewarm_s)


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one (the PMC passes are collected with tools/collect_profiles.sh)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r02_" + suffix)


# (tools/collect_profiles.sh points these at the counter passes it has just made on the same box)
TRAFFIC_JSON = os.environ.get("ANH_TRAFFIC_JSON") or _latest_profile("traffic.json")
INFER_TRAFFIC_JSON = os.environ.get("ANH_INFER_TRAFFIC_JSON") or _latest_profile("infer_traffic.json")
SQ_JSON = os.environ.get("ANH_SQ_JSON") or _latest_profile("sq_counters.json")


# ----------------------------------------------------------------------------------------------------------------------
# N > 1 from `python bench.py --gpus N` alone: start the rank processes as fresh children BEFORE anything touches the GPU
# ----------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """WORLD_SIZE unset and --gpus N > 1: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a
    child (one fresh process per GPU, rendezvous on 127.0.0.1), pass its output through and exit with its return code.
    This process never initialises HIP (counting devices does not)."""
    import torch
    have = torch.cuda.device_count()
    if args.one_gpu:
        have = args.gpus if have >= 1 else 0   # rehearsal: every rank drives device 0
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:        # rank 0's JSON line (and nothing else goes to the children's stdout)
        sys.stdout.write(line)
        sys.stdout.flush()
    sys.exit(proc.wait())


# ----------------------------------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------------------------------
def synthetic_batch(rank):
    import annonet_amd as aa
    rng_img = np.random.default_rng(1000 * rank + 0)
    rng_lab = np.random.default_rng(1000 * rank + 1)
    img = rng_img.integers(0, 256, (BATCH, TILE, TILE, 3), dtype=np.uint8)
    lab = rng_lab.integers(0, CLASSES, (BATCH, TILE, TILE)).astype(np.uint16)
    lab[rng_lab.random((BATCH, TILE, TILE)) < 0.05] = 65535
    w = np.stack([aa.set_weights(lab[i], 0.5, 0.5)["weight"] for i in range(BATCH)])  # annonet_train_main.cpp:292-293 defaults
    return img, lab, w


def layer_geometry(aa, cfg, side):
    """(h_in, h_out) of every layer for a square input of `side` (the spec's dimension rule: con (h+2p-k)/s+1, cont (h-1)s+k-2p)."""
    layers = aa.net_layers(cfg)
    dims = []
    for L in layers:
        h_in = side if L.in_a < 0 else dims[L.in_a][1]
        h_out = (h_in + 2 * L.pad - L.k) // L.stride + 1 if L.type == 0 else (h_in - 1) * L.stride + L.k - 2 * L.pad
        dims.append((h_in, h_out))
    return layers, dims


def section8d(layers, dims, li, n, first_layer_u8=True):
    """SURVEY.md §8d per pass of layer li: flops = 2 k^2 Cin Cout P, minimum bf16 traffic = 2 (Cin P_in + Cout P_out) + 2 k^2 Cin Cout bytes
    (P = pixels of the low-res side for stride 2; the stem reads a u8 image: 1 byte per value)."""
    L = layers[li]
    p_in, p_out = n * dims[li][0] ** 2, n * dims[li][1] ** 2
    flops = 2.0 * L.k * L.k * L.cin * L.cout * (p_out if L.type == 0 else p_in)
    in_bytes = L.cin * p_in * (1 if (L.in_a < 0 and first_layer_u8) else 2)
    min_bytes = in_bytes + 2.0 * L.cout * p_out + 2.0 * L.k * L.k * L.cin * L.cout
    return flops, min_bytes


def bound_of(flops, min_bytes, seconds):
    """min(MFMA, AI x HBM) roofline of one launch: the floor time is the larger of the two resource times."""
    t_mfma, t_hbm = flops / (PEAK_BF16_TFLOPS * 1e12), min_bytes / (PEAK_HBM_GBS * 1e9)
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    out = {"bound": bound, "floor_us": 1e6 * max(t_mfma, t_hbm), "frac": max(t_mfma, t_hbm) / seconds if seconds > 0 else 0.0}
    if bound == "mfma":
        out.update(achieved=flops / seconds / 1e12, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s")
    else:
        out.update(achieved=min_bytes / seconds / 1e9, peak=PEAK_HBM_GBS, unit="GB/s")
    return out


def useful_over_issued(L, dims_li, which):
    """Fraction of a conv launch's MFMAs that multiply pixels INSIDE the image: the kernels walk 32-pixel-wide tiles (8 rows for the
    stride-1 geometry and the stem, 4 rows of the low-resolution side for the stride-2 geometries and for filter gradients with 128
    tile channels), so a side that is not a multiple of the tile computes padding columns / rows (227 -> 256 x 232, 113 -> 128 x 120
    or 116, 56 -> 64 x 56).  The padding pixels' operands are zeros: issued, not useful."""
    h_in, h_out = dims_li
    low = min(h_in, h_out)                       # the side the tiles partition: output of con, input of cont, = both at stride 1
    if L.stride == 1:
        th = 4 if (which == "wgrad" and max(L.cin, L.cout) // 32 >= 4 and L.k == 3) else 8
    else:
        th = 4
    up = lambda v, m: (v + m - 1) // m * m
    return round(low * low / float(up(low, th) * up(low, 32)), 4)


ENTRY_RE = re.compile(r":(fwd|dgrad|wgrad)_L(\d+)_(\w+?)_(\d+)x(\d+)$")
PASS_NAME = {"fwd": "fwd", "dgrad": "bwd-data", "wgrad": "bwd-filter"}


def split_entries(prof, layers, dims, n, per_steps):
    """-> (layers array, overhead array) from the profiler entries of `per_steps` passes."""
    convs, other = [], []
    total_ms = sum(e["total_ms"] for e in prof) or 1.0
    for e in prof:
        if not e["launches"]:
            continue
        m = ENTRY_RE.search(e["name"])
        t_us = 1e3 * e["total_ms"] / e["launches"]
        if m and layers[int(m.group(2))].k > 1:
            li = int(m.group(2))
            flops, min_bytes = section8d(layers, dims, li, n)
            row = {"entry": e["name"], "layer": li, "kind": m.group(3), "cin": int(m.group(4)), "cout": int(m.group(5)), "side": dims[li][1],
                   "pass": PASS_NAME[m.group(1)], "time_us": round(t_us, 2), "gflop": round(flops / 1e9, 3), "min_mb": round(min_bytes / 1e6, 2),
                   "design_mb": round(e["bytes"] / e["launches"] / 1e6, 2), "flop_per_byte": round(flops / min_bytes, 1)}
            b = bound_of(flops, min_bytes, t_us * 1e-6)
            row.update(bound=b["bound"], floor_us=round(b["floor_us"], 2), frac=round(b["frac"], 4), achieved=round(b["achieved"], 1), unit=b["unit"])
            row["mfma_busy_pct"] = mfma_busy(e["name"])
            row["useful_over_issued"] = useful_over_issued(layers[li], dims[li], m.group(1))   # tile-quantisation share of the issued MFMAs
            convs.append(row)
        else:
            other.append({"entry": e["name"], "launches_per_step": e["launches"] / per_steps, "us_per_step": round(1e3 * e["total_ms"] / per_steps, 2),
                          "share_pct": round(100 * e["total_ms"] / total_ms, 1), "design_mb_per_step": round(e["bytes"] / per_steps / 1e6, 2)})
    convs.sort(key=lambda r: (r["layer"], r["pass"]))
    other.sort(key=lambda r: -r["us_per_step"])
    return convs, other


def pmc_traffic(entry, path=None):
    """HBM-side bytes per launch of a profiler entry, from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py)."""
    try:
        with open(path or TRAFFIC_JSON) as f:
            e = json.load(f)["entries"].get(entry)
        return e["traffic_bytes_per_launch"] if e else None
    except (OSError, ValueError, KeyError):
        return None


def mfma_busy(entry):
    """Matrix-pipe occupancy of a profiler entry from the committed SQ counter pass (tools/pmc_counters.py): SQ_VALU_MFMA_BUSY_CYCLES
    (cycles, summed over the 1024 SIMDs: 32 per v_mfma_f32_32x32x16_bf16) over the kernel's own cycles x 1024 SIMDs, the kernel's cycles
    being SQ_BUSY_CYCLES / 32 (the counter is summed over the 32 shader engines).  None without a counter pass for the entry."""
    try:
        with open(SQ_JSON) as f:
            e = json.load(f)["entries"].get(entry)
        if not e or not e.get("SQ_BUSY_CYCLES") or "SQ_VALU_MFMA_BUSY_CYCLES" not in e:
            return None
        return round(100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["SQ_BUSY_CYCLES"] / 32.0 * 1024.0), 1)
    except (OSError, ValueError, KeyError):
        return None


def critical_path(prof, per_steps, step_us):
    """us per step on the MAIN stream (the step's critical path), from the event pairs of the untimed profile pass: the second stream's
    filter gradients are listed as `hidden`.  `gaps` = step time - sum of the main-stream kernels: the cross-queue hand-over packets
    (one per filter gradient handed to the second stream), the join and launch boundaries."""
    acc = {"forward": 0.0, "head": 0.0, "backward_data": 0.0, "apply": 0.0, "tail": 0.0, "hidden_second_stream": 0.0}
    for e in prof:
        us = 1e3 * e["total_ms"] / per_steps
        n = e["name"]
        m = ENTRY_RE.search(n)
        if m and m.group(1) == "fwd":
            acc["forward"] += us
        elif m and m.group(1) == "dgrad":
            acc["backward_data"] += us
        elif m and m.group(1) == "wgrad":
            acc["tail" if int(m.group(2)) == 0 else "hidden_second_stream"] += us
        elif n.startswith("head_") or n in ("softmax_logloss",):
            acc["head"] += us
        elif n.startswith("bn_bwd_") or n.startswith("bn_"):
            acc["apply"] += us
        elif n in ("wgrad_reduce_partials_main", "sgd_momentum_wd"):
            acc["tail"] += us
        elif n == "wgrad_reduce_partials":
            acc["hidden_second_stream"] += us
        else:
            acc["tail"] += us
    main = sum(v for k, v in acc.items() if k != "hidden_second_stream")
    out = {k: round(v, 1) for k, v in acc.items()}
    out.update(main_stream_kernels=round(main, 1), step=round(step_us, 1), gaps=round(step_us - main, 1),
               note="kernel durations by HIP events in the untimed profile pass (every launch instrumented); step = the timed region's ms_per_step; forward = stem + 3x3 convs, head = fused 1x1 head + loss + its backward, apply = bn backward apply passes (8 launches), tail = stem filter gradient + its reduce + SGD update")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; bounded samples)
# ----------------------------------------------------------------------------------------------------------------------
def _omp_set_threads(n):
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


def torch_cpu_net(aa, cfg):
    """The same encoder-decoder in PyTorch (fp32, CPU, oneDNN convolutions) — a second, independent CPU baseline; random init."""
    import torch
    import torch.nn as nn
    layers = aa.net_layers(cfg)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.convs, self.bns = nn.ModuleList(), nn.ModuleList()
            for L in layers:
                conv = (nn.Conv2d if L.type == 0 else nn.ConvTranspose2d)(L.cin, L.cout, L.k, stride=L.stride, padding=L.pad, bias=bool(L.has_bias))
                self.convs.append(conv)
                self.bns.append(nn.BatchNorm2d(L.cout, eps=1e-4) if L.has_bn else nn.Identity())

        def forward(self, x):
            acts = []
            for i, L in enumerate(layers):
                a = x if L.in_a < 0 else acts[L.in_a]
                if L.in_b >= 0:
                    a = a + acts[L.in_b]
                y = self.bns[i](self.convs[i](a))
                acts.append(torch.relu(y) if L.has_bn else y)
            return acts[-1]
    return Net()


def cpu_baseline_train(aa, cfg):
    """The CPU restatement of the reference path on a bounded sample of the same workload, as SURVEY.md §8d specifies it: im2col +
    blocked SGEMM with OpenMP (dlib's CPU design, cpu_dlib.cpp + BLAS: /root/reference/annonet_train_cpu.vcxproj:93,113,230 — the
    oracle's conv_algo = 1, held to its direct loops by tests/test_oracle_gemm.py), all threads of this job's CPU share, then one
    thread; the direct-convolution parity oracle and PyTorch-CPU (oneDNN) on the same kind of sample as labelled second figures."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import random_params
    from oracle.oracle import OracleNet
    o = OracleNet(LEVELS, 3, CLASSES, WIDTH, 1)
    p, r = random_params(o, 2)
    o.params[:], o.running[:] = p, r
    o.set_conv_algorithm(1)
    img, lab, w = synthetic_batch(0)
    cores = int(os.environ["OMP_NUM_THREADS"])   # set by oracle/oracle.py: the CPUs this job may use, capped at 16 (a one-GPU box's share)
    o.train_step(img[:1], lab[:1], w[:1])  # warm-up (page-in, OpenMP pool)
    t0 = time.perf_counter()
    o.train_step(img[:4], lab[:4], w[:4], apply_update=False)  # estimate of the per-tile cost
    per_tile = (time.perf_counter() - t0) / 4
    n_all = len(img)                                                      # the WHOLE batch of 32 tiles per step: samples of 11-15 tiles read 3.4-7.6 tiles/s box to box
    t0 = time.perf_counter()
    o.train_step(img[:n_all], lab[:n_all], w[:n_all])                     # one whole step, untimed: sizes the sample (and pages the batch-sized buffers in)
    reps = int(max(1, min(6, round(12.0 / max(time.perf_counter() - t0, 1e-3)))))   # 10-30 s of CPU work in all
    t0 = time.perf_counter()
    for _ in range(reps):
        o.train_step(img[:n_all], lab[:n_all], w[:n_all])
    dt_all = time.perf_counter() - t0
    _omp_set_threads(1)
    n_one = int(max(1, min(n_all, round(6.0 / max(per_tile * cores, 1e-3)))))   # ~6 s on one thread
    t0 = time.perf_counter()
    o.train_step(img[:n_one], lab[:n_one], w[:n_one], apply_update=False)
    dt_one = time.perf_counter() - t0
    _omp_set_threads(cores)
    out = {"value": reps * n_all / dt_all, "unit": "tiles/s", "cores": cores, "kind": "port",
           "sample": f"{reps} training step(s) on {n_all} tiles of {TILE}x{TILE}x3 (same net, fp32 CPU restatement: im2col + blocked AVX2 SGEMM, OpenMP; {dt_all:.2f} s)",
           "one_thread": {"value": n_one / dt_one, "unit": "tiles/s", "cores": 1, "sample": f"fwd+bwd on {n_one} tile(s), {dt_one:.2f} s"}}
    try:   # the parity oracle itself (direct convolution: one k-ordered fmaf chain per output), a few tiles
        o.set_conv_algorithm(0)
        nd = int(max(2, min(n_all, 8)))
        t0 = time.perf_counter()
        o.train_step(img[:nd], lab[:nd], w[:nd], apply_update=False)
        dt = time.perf_counter() - t0
        out["direct_oracle"] = {"value": nd / dt, "unit": "tiles/s", "cores": cores, "kind": "the parity oracle's direct-convolution loops (not the baseline: the form the GPU parity mode is bit-exact against)",
                                "sample": f"fwd+bwd on {nd} tiles, {dt:.2f} s"}
    except Exception as e:
        out["direct_oracle"] = {"error": repr(e)}
    try:
        import torch
        torch.set_num_threads(cores)
        net = torch_cpu_net(aa, cfg)
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=0.0005)
        nb = min(n_all, 8)
        x = torch.from_numpy(img[:nb].astype(np.float32) / 256.0).permute(0, 3, 1, 2).contiguous()
        y = torch.from_numpy(np.where(lab[:nb] == 65535, -100, lab[:nb].astype(np.int64)))

        def step():
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.cross_entropy(net(x), y, ignore_index=-100)
            loss.backward()
            opt.step()
        step()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 6.0:
            step()
            reps += 1
        dt = time.perf_counter() - t0
        out["pytorch_cpu"] = {"value": reps * nb / dt, "unit": "tiles/s", "cores": cores, "kind": "independent PyTorch-CPU (oneDNN) fp32 statement of the same net, not the reference",
                              "sample": f"{reps} training steps on {nb} tiles, {dt:.2f} s"}
    except Exception as e:   # a baseline, never a reason to lose the bench line
        out["pytorch_cpu"] = {"error": repr(e)}
    return out


def cpu_baseline_infer(aa, cfg, ov):
    """The oracle's annonet_infer() on a bounded crop (all threads, then one thread), Mpx/s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import random_params
    from oracle.oracle import OracleNet
    o = OracleNet(LEVELS, 3, CLASSES, WIDTH, 1)
    p, r = random_params(o, 2)
    o.params[:], o.running[:] = p, r
    o.set_conv_algorithm(1)   # im2col + blocked SGEMM, as cpu_baseline_train
    cores = int(os.environ["OMP_NUM_THREADS"])
    rng = np.random.default_rng(3)
    side = 227
    o.infer(rng.integers(0, 256, (side, side, 3), dtype=np.uint8))
    t0 = time.perf_counter()
    o.infer(rng.integers(0, 256, (side, side, 3), dtype=np.uint8))
    per_px = (time.perf_counter() - t0) / (side * side)
    side_all = int(min(3000, max(300, (8.0 / per_px) ** 0.5)))     # ~8 s of CPU work
    image = rng.integers(0, 256, (side_all, side_all, 3), dtype=np.uint8)
    t0 = time.perf_counter()
    o.infer(image, max_tile=(1024, 1024), overlap=ov)
    dt_all = time.perf_counter() - t0
    _omp_set_threads(1)
    side_one = int(max(227, side_all / max(cores, 1) ** 0.5))
    t0 = time.perf_counter()
    o.infer(image[:side_one, :side_one], max_tile=(1024, 1024), overlap=ov)
    dt_one = time.perf_counter() - t0
    _omp_set_threads(cores)
    return {"value": side_all * side_all / dt_all / 1e6, "unit": "Mpx/s", "cores": cores, "kind": "port",
            "sample": f"annonet_infer() of a {side_all}x{side_all} crop of the same synthetic image (fp32 CPU restatement: im2col + blocked AVX2 SGEMM, OpenMP; {dt_all:.2f} s)",
            "one_thread": {"value": side_one * side_one / dt_one / 1e6, "unit": "Mpx/s", "cores": 1, "sample": f"{side_one}x{side_one} crop, {dt_one:.2f} s"}}


# ----------------------------------------------------------------------------------------------------------------------
def dist_facts(torch, dist, dev, local_rank, world, use_dist):
    """ranks_seen = all-reduce of ones; devices = every rank's device index (proves N distinct ranks drove N devices).
    A job whose ranks do not add up leaves with an error instead of reporting a number for fewer GPUs than it claims."""
    if not use_dist:
        return 1, [local_rank]
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    mine = torch.tensor([local_rank], device=dev, dtype=torch.int64)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    seen = int(ones.item())
    if seen != world:
        print(f"bench.py: {seen} rank(s) answered the all-reduce of ones, WORLD_SIZE is {world}", file=sys.stderr)
        sys.exit(3)
    return seen, [int(g.item()) for g in gathered]


def prewarm(run_once, sync, seconds, agree=None):
    """untimed: repeats the workload for `seconds` of wall time (clocks and caches settled before the warm-up steps).
    `agree(more)` (N > 1): every rank must run the same number of passes — they hold collectives — so the ranks continue
    as long as ANY of them wants to (an all-reduce MAX of the flag)."""
    if seconds <= 0:
        return 0
    t0, n = time.perf_counter(), 0
    while True:
        more = time.perf_counter() - t0 < seconds
        if agree is not None:
            more = agree(more)
        if not more:
            return n
        for _ in range(4):
            run_once()
        n += 4
        sync()


def make_agree(torch, dist, dev, use_dist):
    if not use_dist:
        return None

    def agree(more):
        flag = torch.tensor([1 if more else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return bool(flag.item())
    return agree


def infer_one_size(args, side, with_cpu_baseline, aa, aad, torch, dist, prec, rank, local_rank, world, use_dist):
    """One image size of the inference bench -> the result object (rank 0) or None."""
    dev = torch.device("cuda", local_rank)
    cfg = aa.net_config(LEVELS, 3, CLASSES, WIDTH, 1, prec)
    tr = aa.TrainingNet(LEVELS, 3, prec, seed=2)
    tr.SetNetWidth(WIDTH, 1); tr.SetClassCount(CLASSES); tr.Initialize()
    net = tr.GetRuntimeNet(prec)
    del tr
    net_stream = aad.handle_stream(net)   # torch's view of the net's own stream: the exchange step is enqueued there
    rng = np.random.default_rng(3)
    image = torch.from_numpy(rng.integers(0, 256, (side, side, 3), dtype=np.uint8)).to(dev)
    # two label maps (device + pinned host), used alternately: the copy of image i to the host overlaps the passes of image i + 1
    label_sets = [(torch.empty((side, side), dtype=torch.int16, device=dev), torch.empty((side, side), dtype=torch.int16).pin_memory()) for _ in range(2)]
    labels = label_sets[0][0]
    copy_stream = torch.cuda.Stream(device=dev)
    copied = [None, None]   # event: the host copy of that set's previous image has finished
    turn = [0]
    blended = torch.empty((CLASSES, side, side), dtype=torch.float32, device=dev)
    import ctypes as C
    ov = aa.lib().anh_required_input_dim(C.byref(cfg))
    tp = aa.tiling.parameters(1024, 1024, ov, ov)  # GPU defaults of the reference (annonet_infer_main.cpp:300-303,423-427)
    tiles = aa.tiling.get_tiles(side, side, tp)
    mine = aad.shard_tiles(tiles, rank, world)
    exchange = aad.OverlapExchange(tiles, world, side, side, dev)   # the pixels where tiles of different ranks overlap (none at world 1)
    gather = aad.LabelGather(tiles, world, rank, side, side, dev, CLASSES) if world > 1 else None   # ONE label map on rank 0 (resident leg)
    ranks_seen, devices = dist_facts(torch, dist, dev, local_rank, world, use_dist)
    # N > 1, labels on the host: ONE host map per label set in POSIX shared memory, every rank delivering the cells of its own tiles over
    # its own PCIe link (dist.HostLabelMap) — rank 0 copies only its share, nothing is gathered first
    host_maps = None
    if world > 1:
        host_maps = []
        for _ in range(2):
            names = [None]
            if rank == 0:
                hm = aad.HostLabelMap(tiles, world, rank, side, side)
                names = [hm.name]
            dist.broadcast_object_list(names, src=0)
            if rank != 0:
                hm = aad.HostLabelMap(tiles, world, rank, side, side, name=names[0])
            host_maps.append(hm)

    def run(to_host=False):
        # blend this rank's tiles -> all-reduce the plane sums of the cross-rank overlaps -> label this rank's rows
        k = turn[0] if to_host else 0
        labels, host_labels = label_sets[k]
        if to_host:
            turn[0] ^= 1
            if copied[k] is not None:   # this set's previous image must have left for the host before its map is overwritten
                net_stream.wait_event(copied[k])
        row0, row1 = aad.sharded_infer(net, image, labels, blended, tiles, rank, world, exchange, tiling_parameters=tp, stream=net_stream)
        if to_host and host_maps is not None:   # N > 1: every rank's own rows go straight to the shared host map
            copy_stream.wait_stream(net_stream)
            host_maps[k].deliver(labels.data_ptr(), copy_stream.cuda_stream)
            copied[k] = torch.cuda.Event()
            copied[k].record(copy_stream)
            return
        if gather is not None:   # N > 1: the result of annonet_infer() is one map — the ranks' shares are assembled on rank 0
            with torch.cuda.stream(net_stream):   # everything that touches `labels` stays on the net's stream
                whole = gather.run(labels)
                if whole is not None:
                    labels.copy_(whole)
                    row0, row1 = 0, side
        if to_host and row1 > row0:   # D2H on a copy stream, ordered after the net's stream; the net goes on with the other set
            copy_stream.wait_stream(net_stream)
            with torch.cuda.stream(copy_stream):
                host_labels[row0:row1].copy_(labels[row0:row1], non_blocking=True)
                copied[k] = torch.cuda.Event()
                copied[k].record(copy_stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, **kw):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            run(**kw)
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
        return elapsed

    prewarm_passes = prewarm(run, torch.cuda.synchronize, args.prewarm_s, make_agree(torch, dist, dev, use_dist))
    for _ in range(max(args.warmup, 1)):
        run()
    # untimed profile pass: every kernel class of one image
    net.synchronize()
    net.profile_enable(True)
    run()
    net.synchronize()
    prof_all = net.profile()
    if args.dump_launch_order and rank == 0:
        with open(args.dump_launch_order, "w") as f:
            json.dump({"order": net.profile_launch_order()}, f, indent=0)
    net.profile_reset()
    if os.environ.get("ANH_BENCH_VERBOSE") and rank == 0:   # every kernel class of one image, serialised by the event pairs
        for e in sorted(prof_all, key=lambda e: -e["total_ms"]):
            print(f"  {e['name']:52s} {e['total_ms']:8.3f} ms/image  {e['launches']:4d} launches", file=sys.stderr)
    win = aa.RuntimeNet.GetRecommendedInputDimension(LEVELS, tiles[0][0][2] - tiles[0][0][0] + 1)   # side of a tile's input window
    layers, dims = layer_geometry(aa, cfg, win)
    convs = [e for e in prof_all if ENTRY_RE.search(e["name"]) and layers[int(ENTRY_RE.search(e["name"]).group(2))].k > 1]
    dominant = max(convs, key=lambda e: e["total_ms"])["name"] if convs else ""
    # Both timed regions run under the SAME instrumentation (event pairs on the dominant conv entry only) and the same step count:
    # `value` with image and label map resident, `value_labels_on_host` with the label map copied to pinned host memory per pass.
    net.profile_set_filter(dominant)
    elapsed = timed(args.steps)
    net.synchronize()
    dom = [e for e in net.profile() if e["name"] == dominant]
    elapsed_host = timed(args.steps, to_host=True)
    net.synchronize()
    net.profile_enable(False)
    # bytes each rank copies to the host per image in the labels-on-host leg (N > 1: its own cells of the shared map; N = 1: the whole map)
    host_bytes = [side * side * 2]
    if host_maps is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, host_maps[0].bytes)
        host_bytes = gathered
        if rank == 0 and sum(gathered) != side * side * 2:
            raise SystemExit(f"host label map: the ranks' cells cover {sum(gathered)} bytes of {side * side * 2}")
        for hm in host_maps:
            hm.close()
    out = None
    if rank == 0:
        roof = None
        if dom and dom[0]["launches"]:
            e = dom[0]
            li = int(ENTRY_RE.search(dominant).group(2))
            avg_s = e["total_ms"] / 1e3 / e["launches"]
            # SURVEY.md §8d per launch: the library's launches carry 1..16 tiles, so the mean tile count per launch comes from the
            # flops the profiler summed (2 k^2 Cin Cout P per launch); bytes = the §8d MINIMUM (one input tensor + the output tensor +
            # the filters — a skip-add layer's second operand is design traffic, not algorithmic), the convention of the training line
            flops = e["flops"] / e["launches"]
            flops_1, bytes_1 = section8d(layers, dims, li, 1)
            w_bytes = 2.0 * layers[li].k ** 2 * layers[li].cin * layers[li].cout
            tiles_per_launch = flops / flops_1
            min_bytes = (bytes_1 - w_bytes) * tiles_per_launch + w_bytes
            roof = bound_of(flops, min_bytes, avg_s)
            roof.pop("floor_us")
            roof.update(traffic=pmc_traffic(dominant, INFER_TRAFFIC_JSON) if side == 4096 and world == 1 else None,
                        traffic_source=(f"not measured in this run: HBM bytes per launch of this entry from the committed rocprofv3 --pmc passes ({os.path.basename(INFER_TRAFFIC_JSON)}; tools/pmc_traffic.py)" if INFER_TRAFFIC_JSON and side == 4096 and world == 1 else None),
                        kernel=dominant, layer=li,
                        avg_launch_us=avg_s * 1e6, launches=e["launches"], tiles_per_launch=tiles_per_launch,
                        algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=min_bytes, design_bytes_per_launch=e["bytes"] / e["launches"],
                        arithmetic_intensity=flops / min_bytes, bytes_convention="SURVEY 8d minimum: one input + output + filters (skip operand not counted)")
        total_ms = sum(e["total_ms"] for e in prof_all) or 1.0
        share = sorted(((e["name"], round(100 * e["total_ms"] / total_ms, 1)) for e in prof_all), key=lambda x: -x[1])[:10]
        out = {"metric": f"Mpixels/s tiled inference, {side}x{side} image, 1024^2 tiles, overlap {ov}", "value": side * side * args.steps / elapsed / 1e6,
               "unit": "Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_s": args.prewarm_s if prewarm_passes else 0.0,
               "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": {"workload": f"annonet_infer(): {side}x{side} image, {len(tiles)} tiles ({len(mine)} on rank 0), window {win}, levels={LEVELS} width={WIDTH} K={CLASSES}",
                          "parallelism": f"tile-shard{world}", "exchanged_pixels": exchange.pixels(),
                          "host_bytes_per_rank": host_bytes},
               "value_labels_on_host": side * side * args.steps / elapsed_host / 1e6,
               "value_note": "value: image and label map resident in HBM (N > 1: ONE map assembled on rank 0, dist.LabelGather); value_labels_on_host: the same passes with the label map copied to pinned host memory inside the timed region (N > 1: every rank copies the cells of its own tiles into ONE shared-memory host map over its own PCIe link, dist.HostLabelMap: config.host_bytes_per_rank) (SURVEY 8d's unit; two label maps used alternately, so the copy of one image overlaps the passes of the next; every copy ends inside the region); same step count and instrumentation",
               "roofline": roof, "kernel_time_share_pct": share, "ranks_seen": ranks_seen, "devices": devices,
               "cpu_baseline": cpu_baseline_infer(aa, cfg, ov) if with_cpu_baseline else None}
    torch.cuda.synchronize()
    return out


def bench_infer(args, aa, aad, torch, dist, prec, rank, local_rank, world, use_dist):
    """BASELINE.json configs[2] / [4]: tiled sliding-window inference over one synthetic image, tiles sharded across ranks.
    Timed: tile cut + forward + blend, the exchange of the cross-rank overlap sums (one RCCL all-reduce of ~2 % of the
    planes; nothing at one GPU) and argmax; image and label map resident in HBM.  `value_labels_on_host` times the same
    passes with the label map copied to (pinned) host memory inside the timed region (SURVEY.md §8d's unit).
    --image-side a,b,...: the first size is the line, the others ride in `other_sizes` (no CPU baseline)."""
    sides = [int(x) for x in str(args.image_side).split(",")]
    out = None
    for i, side in enumerate(sides):
        res = infer_one_size(args, side, i == 0 and not args.no_cpu_baseline and world == 1, aa, aad, torch, dist, prec, rank, local_rank, world, use_dist)
        if rank == 0:
            if i == 0:
                out = res
            else:
                out.setdefault("other_sizes", []).append({k: res[k] for k in ("metric", "value", "value_labels_on_host", "ms_per_step", "steps", "config", "roofline")})
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def infer_in_child(args, world):
    """The inference workload for the training line's `infer` object: `bench.py --mode infer` as a CHILD process (for N > 1 it
    starts its own ranks), after this process has finished its own measurement.  Whatever happens there — an error, a hang cut at
    the timeout — stays inside the returned object."""
    sides = args.infer_sides or ("4096,16384" if world == 1 else "16384")
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "infer", "--gpus", str(world), "--image-side", sides, "--steps", str(args.infer_steps),
           "--warmup", "2", "--precision", args.precision, "--prewarm-s", str(min(args.prewarm_s, 1.0))]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.backend != "nccl":
        cmd += ["--backend", args.backend]
    if args.one_gpu:
        cmd.append("--one-gpu")
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
            "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_") and not k.startswith("TORCH_NCCL_")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.infer_timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"inference child exceeded {args.infer_timeout} s", "command": " ".join(cmd[1:])}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return {"error": f"inference child returned {r.returncode}", "stderr_tail": r.stderr[-800:], "command": " ".join(cmd[1:])}
    d = json.loads(lines[0])
    d["measured_by"] = "child process: " + " ".join(["bench.py"] + cmd[2:])
    return d


def bench_in_process(args):
    """Host (B): ONE process drives several GPUs behind the C ABI — anh_set_devices (NetPimpl::SetDevices), then the reference's own loop
    unchanged: StartTraining with host vectors (/root/reference/annonet_train_main.cpp:583-614), the library splitting the mini-batch
    along N, RCCL inside the library, persistent worker threads per replica.  Global batch 32 per device; PCIe-inclusive by nature of
    the interface.  `--devices 0,0` is the rehearsal on one GPU (repeated-device backend: same splits, a fixed-order sum instead of RCCL)."""
    import ctypes as C
    import annonet_amd as aa
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    R = len(devices)
    aa.set_devices(devices)
    prec = aa.ANH_BF16 if args.precision == "bf16" else aa.ANH_FP32
    t = aa.TrainingNet(LEVELS, 3, prec, seed=2)
    t.SetNetWidth(WIDTH, 1)
    t.SetClassCount(CLASSES)
    t.Initialize()
    t.SetLearningRate(0.1)
    imgs, wls = [], []
    for r in range(R):   # replica r gets the shard rank r of the one-process-per-GPU job would hold
        img, lab, _ = synthetic_batch(r)
        imgs += [np.ascontiguousarray(img[i]) for i in range(BATCH)]
        wls += [aa.set_weights(lab[i], 0.5, 0.5) for i in range(BATCH)]
    n = BATCH * R
    ip = (C.c_void_p * n)(*[a.ctypes.data for a in imgs])
    lp = (C.c_void_p * n)(*[a.ctypes.data for a in wls])

    def step():
        aa._lib.check(t.L.anh_trainer_step(t.h, ip, lp, n, TILE, TILE))
    prewarm_steps = prewarm(step, t.synchronize, args.prewarm_s)
    for _ in range(max(args.warmup, 1)):
        step()
    t.synchronize()
    t.reset_exchange_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t.synchronize()
    elapsed = time.perf_counter() - t0
    st = t.exchange_stats()
    out = {"metric": "227x227 RGB tiles/sec fwd+bwd @ batch 32 per GPU, one process driving all GPUs (anh_set_devices), host vectors every step",
           "value": n * args.steps / elapsed, "unit": "tiles/s", "n_gpus": len(set(devices)), "replicas": R, "devices": devices, "steps": args.steps, "warmup": args.warmup,
           "prewarm_s": args.prewarm_s if prewarm_steps else 0.0, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
           "dtype": args.precision, "data": "synthetic", "global_batch": n,
           "host_us_per_start_training": round(st["host_us_mean"], 1),
           "host_busy_us_per_start_training": round(st["host_us_mean"] - st["host_wait_us_mean"], 1),
           "host_note": "host_us: wall time the calling thread spends inside one StartTraining; host_busy_us: the same minus the time it is blocked on the GPU (a staging set is reused every second step: the call waits until the upload of step k-2 has left it) = what the host itself needs (packing 32 x replicas samples into pinned staging, uploads, the step's launches on every replica by persistent worker threads, the collectives, the update): the one-thread reference loop cannot step faster than this",
           "worker_wakeups_per_step": round(st["worker_calls"] / max(st["steps"], 1), 2),
           "exchange": {"allreduce_tail_us": round(st["allreduce_tail_us_mean"], 1), "allreduce_head_us": round(st["allreduce_head_us_mean"], 1), "sampled_steps": st["samples"],
                        "early_reduce": bool(st["early_reduce"]), "transport": {1: "rccl", 2: "peer copies between distinct devices (RCCL communicators unavailable or ANH_COLLECTIVE_TRANSPORT=peer)"}.get(st["uses_rccl"], "repeated-device rehearsal (fixed-order sum kernel + copies)"),
                        "rccl_version": st["rccl_version"], "bucket_bytes": st["bucket_bytes"],
                        "note": "device time of the all-reduce's two parts on replica 0 (event pairs on every 8th step): tail = bucket[first:] on a side stream while backward still runs, head = bucket[:first] on the main stream"},
           "final_loss": t.get_last_loss()}
    print(json.dumps(out), flush=True)


def in_process_child(args, world):
    """`bench.py --in-process --gpus N` as a child, after this job's own measurement (the GPUs are free again): host (B)'s figure beside host (A)'s."""
    cmd = [sys.executable, os.path.abspath(__file__), "--in-process", "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--precision", args.precision, "--prewarm-s", str(min(args.prewarm_s, 1.0))]
    if args.one_gpu:
        cmd += ["--devices", ",".join(["0"] * world)]
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_") and not k.startswith("TORCH_NCCL_")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.infer_timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"in-process child exceeded {args.infer_timeout} s", "command": " ".join(cmd[1:])}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return {"error": f"in-process child returned {r.returncode}", "stderr_tail": r.stderr[-800:], "command": " ".join(cmd[1:])}
    d = json.loads(lines[0])
    d["measured_by"] = "child process: " + " ".join(["bench.py"] + cmd[2:])
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "infer"], help="train = BASELINE.json's metric (default); infer = tiled inference over a 4096x4096 image")
    ap.add_argument("--image-side", default="4096", help="infer mode: image side, or a comma list (the first is the line, the others ride in other_sizes)")
    ap.add_argument("--prewarm-s", type=float, default=PREWARM_S, help="untimed time-based pre-warm before the warm-up steps (declared in the line)")
    ap.add_argument("--no-infer", action="store_true", help="train mode: leave the `infer` object out (profile collection under rocprofv3)")
    ap.add_argument("--infer-steps", type=int, default=10)
    ap.add_argument("--infer-sides", default=None, help="image sides of the `infer` object (default: 4096,16384 at one GPU = BASELINE.json configs[2], 16384 = configs[4] otherwise)")
    ap.add_argument("--infer-timeout", type=float, default=300.0)
    ap.add_argument("--collective-timeout", type=float, default=180.0, help="seconds after which a collective that a rank never joined fails the job")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend: nccl = RCCL over xGMI (the job); gloo = rehearsal transport")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal of the N > 1 paths on a one-GPU box: every rank drives device 0 (needs --backend gloo: RCCL wants one GPU per rank)")
    ap.add_argument("--no-in-process", action="store_true", help="N > 1: leave the `in_process` object (host B measured by a child process) out")
    ap.add_argument("--in-process", action="store_true", help="host (B): ONE process drives --gpus N devices (or --devices) through anh_set_devices and StartTraining with host vectors; prints its own line")
    ap.add_argument("--devices", default=None, help="--in-process: explicit device list, e.g. 0,0 = two replicas rehearsed on one GPU")
    ap.add_argument("--dump-launch-order", default=None, help="write the step's launch order (profiler entry per kernel-class launch) as JSON: input of tools/pmc_traffic.py")
    args = ap.parse_args()
    if args.mode == "infer" and args.steps == 200:
        args.steps, args.warmup = 20, 3

    if args.in_process:
        return bench_in_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # does not return

    import torch
    import annonet_amd as aa
    from annonet_amd import dist as aad

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.one_gpu:
        if args.backend != "gloo":
            raise SystemExit("--one-gpu is a rehearsal on one device: it needs --backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    aa._lib.check(aa.lib().anh_set_device(local_rank))
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run: take the RCCL path even at world 1
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # a rank that never arrives, or dies, must end the job with an error — not leave the others waiting: every collective has a deadline
        kw = {"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.collective_timeout), **kw)

    prec = aa.ANH_BF16 if args.precision == "bf16" else aa.ANH_FP32
    if args.mode == "infer":
        return bench_infer(args, aa, aad, torch, dist, prec, rank, local_rank, world, use_dist)
    cfg = aa.net_config(LEVELS, 3, CLASSES, WIDTH, 1, prec)
    t = aa.TrainingNet(LEVELS, 3, prec, seed=2)
    t.SetNetWidth(WIDTH, 1)
    t.SetClassCount(CLASSES)
    t.Initialize()
    t.SetLearningRate(0.1)
    t_stream = aad.handle_stream(t)       # torch's view of the trainer's own stream: the all-reduce is enqueued there
    bucket = aad.grad_bucket_tensor(t)

    img, lab, w = synthetic_batch(rank)
    dev = torch.device("cuda", local_rank)
    d_img = torch.from_numpy(img).to(dev)
    d_lab = torch.from_numpy(lab.view(np.int16)).to(dev)
    d_w = torch.from_numpy(w).to(dev)
    ranks_seen, devices = dist_facts(torch, dist, dev, local_rank, world, use_dist)

    # N > 1: the tail of the bucket (all layers but the first two) is all-reduced on a side stream while backward still runs
    # (two collectives cost ~20 us more than one: only worth it where the all-reduce moves data, i.e. N > 1; ANH_EARLY_REDUCE=0/1 forces)
    er = os.environ.get("ANH_EARLY_REDUCE")
    early = aad.EarlyReduce(t, bucket) if use_dist and (er == "1" or (er is None and world > 1)) else None

    def step():
        aad.data_parallel_step(t, bucket, d_img.data_ptr(), d_lab.data_ptr(), d_w.data_ptr(), BATCH, TILE, TILE, world, force_collective=use_dist, stream=t_stream, early=early)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: time-based pre-warm, the W warm-up steps, then an all-kernel profile pass (per-layer table, and which §8d conv entry
    # is the dominant one)
    prewarm_steps = prewarm(step, t.synchronize, args.prewarm_s, make_agree(torch, dist, dev, use_dist))
    for _ in range(max(args.warmup, 1)):
        step()
    t.synchronize()
    t.profile_enable(True)
    for _ in range(PROFILE_STEPS):
        step()
    t.synchronize()
    prof_all = t.profile()
    if args.dump_launch_order and rank == 0:
        with open(args.dump_launch_order, "w") as f:
            json.dump({"order": t.profile_launch_order()}, f, indent=0)
    t.profile_reset()
    layers, dims = layer_geometry(aa, cfg, TILE)
    layer_rows, overhead = split_entries(prof_all, layers, dims, BATCH, PROFILE_STEPS)
    dominant = max(layer_rows, key=lambda r: r["time_us"])["entry"] if layer_rows else ""
    t.profile_set_filter(dominant)
    # An event pair costs ~6 us of stream time per launch, so the timed region times the dominant entry on a sample of
    # its steps (every 4th), not on all of them: the average launch duration is estimated from >= steps/4 launches.
    sample_every = 4 if args.steps >= 8 else 1
    t.profile_set_sampling(sample_every)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    t.synchronize()
    dom = [e for e in t.profile() if e["name"] == dominant]
    loss = t.get_last_loss()
    # N > 1 (or launched as a rank): what ONE driver run must carry to be diagnosable — every rank's own step time, and the device time of
    # the exchange step measured by event pairs around each collective on four extra, untimed steps
    exchange = None
    if use_dist:
        mine = torch.tensor([1e3 * own_elapsed / args.steps], device=dev, dtype=torch.float64)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        ev = []
        for _ in range(4):
            aad.data_parallel_step(t, bucket, d_img.data_ptr(), d_lab.data_ptr(), d_w.data_ptr(), BATCH, TILE, TILE, world, force_collective=True, stream=t_stream, early=early, events=ev)
        t.synchronize()
        torch.cuda.synchronize()
        parts = {}
        for name, a, b in ev:
            parts.setdefault(name, []).append(1e3 * a.elapsed_time(b))
        exchange = {"allreduce_us": {k: round(sum(v) / len(v), 1) for k, v in parts.items()}, "sampled_steps": 4,
                    "early_reduce": bool(early is not None and early.split), "backend": args.backend,
                    "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if args.backend == "nccl" else None,
                    "bucket_bytes": int(bucket.numel()) * 4,
                    "ms_per_step_per_rank": {"min": round(min(float(x.item()) for x in per_rank), 4), "max": round(max(float(x.item()) for x in per_rank), 4),
                                             "all": [round(float(x.item()), 4) for x in per_rank]},
                    "note": "allreduce_us: device time between event pairs around each collective of a step (tail = bucket[first:] on the side stream, head = bucket[:first], or whole = one all-reduce), mean of 4 untimed steps on rank 0; ms_per_step_per_rank: every rank's own wall time over the timed region / K"}
    # The same K steps with the mini-batch handed over as HOST vectors every step, the way the reference's loop drives StartTraining
    # (/root/reference/annonet_train_main.cpp:585-609: 32 images + 32 weighted-label images per call; anh_trainer_step packs them into
    # pinned staging, uploads and runs the step behind the upload) — same step count, same instrumentation (event pairs on the dominant
    # entry, sampled).  PCIe-inclusive: reported beside `value`, never as it.  One process per GPU only (the N > 1 job feeds resident shards).
    elapsed_host = None
    if world == 1 and not use_dist:
        import ctypes as C
        wl = [aa.set_weights(lab[i], 0.5, 0.5) for i in range(BATCH)]
        imgs = [np.ascontiguousarray(img[i]) for i in range(BATCH)]
        ip = (C.c_void_p * BATCH)(*[a.ctypes.data for a in imgs])
        lp = (C.c_void_p * BATCH)(*[a.ctypes.data for a in wl])

        def host_step():
            aa._lib.check(t.L.anh_trainer_step(t.h, ip, lp, BATCH, TILE, TILE))
        for _ in range(max(args.warmup, 3)):
            host_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            host_step()
        fence()
        elapsed_host = time.perf_counter() - t0
    t.profile_enable(False)

    if rank == 0:
        roof = None
        if dom and dom[0]["launches"]:
            e = dom[0]
            row = [r for r in layer_rows if r["entry"] == dominant][0]
            avg_s = e["total_ms"] / 1e3 / e["launches"]
            flops, min_bytes = row["gflop"] * 1e9, row["min_mb"] * 1e6
            roof = bound_of(flops, min_bytes, avg_s)
            roof.pop("floor_us")
            roof.update(traffic=pmc_traffic(dominant), traffic_source=f"not measured in this run: HBM bytes per launch of this entry from the committed rocprofv3 --pmc passes ({os.path.basename(TRAFFIC_JSON)}; tools/pmc_traffic.py, FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes)" if TRAFFIC_JSON else None,
                        kernel=dominant, layer=row["layer"], **{"pass": row["pass"]},
                        avg_launch_us=avg_s * 1e6, launches=e["launches"], sampled_every_n_steps=sample_every,
                        algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=min_bytes, arithmetic_intensity=flops / min_bytes)
        step_gflop = 3 * sum(section8d(layers, dims, li, BATCH)[0] for li in range(len(layers))) / 1e9
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": "227x227 RGB tiles/sec fwd+bwd @ batch 32", "value": BATCH * world * args.steps / elapsed, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_s": args.prewarm_s if prewarm_steps else 0.0, "prewarm_steps": prewarm_steps, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"training step (fwd+loss+bwd+SGD), batch {BATCH}x3x{TILE}x{TILE} per GPU, encoder-decoder levels={LEVELS} width={WIDTH} K={CLASSES}, random init",
                       "global_batch": BATCH * world, "parallelism": f"dp{world}"},
            "value_host_inputs": None if elapsed_host is None else BATCH * args.steps / elapsed_host,
            "value_note": "value: the mini-batch resident in HBM when the timed region starts; value_host_inputs: the same K steps through anh_trainer_step with 32 host images + 32 host weighted-label images handed over every step (reference loop annonet_train_main.cpp:585-609; PCIe-inclusive, same step count and instrumentation)",
            "roofline": roof,
            "critical_path": critical_path(prof_all, PROFILE_STEPS, 1e3 * ms),
            "exchange": exchange,
            "mfma_busy_note": "layers[].mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs) from the committed counter pass " + os.path.basename(SQ_JSON) + " (kernels serialised by the profiler); cycles at the clock the kernel held, not the 2.4 GHz of the 2.5 PFLOP/s peak",
            "step_roofline": {"gflop_per_step": round(step_gflop, 1), "achieved_tflops": round(step_gflop / ms, 1), "frac_of_bf16_mfma_peak": round(step_gflop / ms / PEAK_BF16_TFLOPS, 4),
                              "sum_of_layer_floors_us": round(sum(r["floor_us"] for r in layer_rows), 1)},
            "layers": layer_rows,
            "layers_measured": f"HIP events on the launching stream, untimed profile pass of {PROFILE_STEPS} steps right before the timed region",
            "overhead_kernels": overhead,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline_train(aa, cfg),   # rank 0 at N=1 only
            "ranks_seen": ranks_seen, "devices": devices,
            "final_loss": loss,
        }
        if os.environ.get("ANH_BENCH_VERBOSE"):
            for e in sorted(prof_all, key=lambda e: -e["total_ms"]):
                per = e["total_ms"] / PROFILE_STEPS
                print(f"  {e['name']:52s} {per:8.3f} ms/step  {e['launches'] // PROFILE_STEPS:3d} launches  "
                      f"{e['flops'] / PROFILE_STEPS / per / 1e9 if per else 0:9.1f} TFLOP/s  {e['bytes'] / PROFILE_STEPS / per / 1e6 if per else 0:9.1f} GB/s", file=sys.stderr)
    # everything of this process's own measurement is done: release the GPU side, then (rank 0) let a child measure tiled inference
    del early, bucket, t_stream, t
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROFILER_") or k.startswith("ROCP_") for k in os.environ)
        if not args.no_infer and not under_profiler:
            out["infer"] = infer_in_child(args, world)
        if world > 1 and not under_profiler and not args.no_in_process:
            out["in_process"] = in_process_child(args, world)   # host (B): the same N GPUs from ONE process (anh_set_devices)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
