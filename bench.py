#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config: 227x227 RGB tiles/s, training step (forward + loss + backward +
SGD update), batch 32 per GPU, bf16, random-init net (levels 2, width 1.0, K=3), synthetic tiles.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; for N > 1 the only exchange step of the path is the all-reduce (RCCL) of the flat gradient
bucket.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0, with
  roofline     — the dominant kernel class: algorithmic flops (or bytes) per launch / its mean launch time, measured
                 with HIP events on the compute stream inside the timed region;
  cpu_baseline — the oracle (a CPU port of the path, see oracle/) timed on this host on a bounded sample.
"""
import argparse
import json
import re
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises: keeps the trainer's two streams on separate hardware queues (annonet_amd/_lib.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE = 227
BATCH = 32
LEVELS, WIDTH, CLASSES = 2, 1.0, 3
# peaks from /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec), bf16 MFMA ~2.5 PFLOP/s dense (spec)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0
RIDGE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)


def pmc_traffic(entry):
    """HBM-side bytes per launch of a profiler entry, from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            entries = json.load(f)["entries"]
            e = entries.get(entry) or entries.get(re.sub(r"_\d+x\d+$", "", entry))  # conv/wgrad entries carry a channel-shape suffix
        return e["traffic_bytes_per_launch"] if e else None
    except (OSError, ValueError, KeyError):
        return None


def synthetic_batch(rank):
    import annonet_amd as aa
    rng_img = np.random.default_rng(1000 * rank + 0)
    rng_lab = np.random.default_rng(1000 * rank + 1)
    img = rng_img.integers(0, 256, (BATCH, TILE, TILE, 3), dtype=np.uint8)
    lab = rng_lab.integers(0, CLASSES, (BATCH, TILE, TILE)).astype(np.uint16)
    lab[rng_lab.random((BATCH, TILE, TILE)) < 0.05] = 65535
    w = np.stack([aa.set_weights(lab[i], 0.5, 0.5)["weight"] for i in range(BATCH)])  # annonet_train_main.cpp:292-293 defaults
    return img, lab, w


def cpu_baseline(sample_tiles=4):
    """The oracle's training step on `sample_tiles` tiles of the same workload, all host cores (OpenMP)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import random_params
    from oracle.oracle import OracleNet
    o = OracleNet(LEVELS, 3, CLASSES, WIDTH, 1)
    p, r = random_params(o, 2)
    o.params[:], o.running[:] = p, r
    img, lab, w = synthetic_batch(0)
    o.train_step(img[:1], lab[:1], w[:1])  # warm-up (page-in, OpenMP pool)
    t0 = time.perf_counter()
    o.train_step(img[:2], lab[:2], w[:2], apply_update=False)  # estimate of the per-tile cost
    per_tile = (time.perf_counter() - t0) / 2
    sample_tiles = int(min(len(img), max(sample_tiles, round(20.0 / max(per_tile, 1e-3)))))  # ~10-20 s of CPU work, at most the whole batch
    img, lab, w = img[:sample_tiles], lab[:sample_tiles], w[:sample_tiles]
    t0 = time.perf_counter()
    o.train_step(img, lab, w)
    dt = time.perf_counter() - t0
    cores = int(os.environ["OMP_NUM_THREADS"])   # set by oracle/oracle.py: the CPUs this job may use, capped at 16 (a one-GPU box's share)
    return {"value": sample_tiles / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"1 training step on {sample_tiles} tiles of {TILE}x{TILE}x3 (same net, fp32 CPU oracle, {dt:.2f} s)"}


def bench_infer(args, aa, aad, torch, dist, prec, rank, local_rank, world, use_dist):
    """BASELINE.json configs[2] / [4]: tiled sliding-window inference over one synthetic image, tiles sharded across ranks.
    Timed: tile cut + forward + blend, the exchange of the cross-rank overlap sums (one RCCL all-reduce of ~2 % of the
    planes; nothing at one GPU) and argmax; image and label map resident in HBM."""
    side = args.image_side
    dev = torch.device("cuda", local_rank)
    cfg = aa.net_config(LEVELS, 3, CLASSES, WIDTH, 1, prec)
    tr = aa.TrainingNet(LEVELS, 3, prec, seed=2)
    tr.SetNetWidth(WIDTH, 1); tr.SetClassCount(CLASSES); tr.Initialize()
    net = tr.GetRuntimeNet(prec)
    del tr
    net_stream = aad.handle_stream(net)   # torch's view of the net's own stream: the exchange step is enqueued there
    rng = np.random.default_rng(3)
    image = torch.from_numpy(rng.integers(0, 256, (side, side, 3), dtype=np.uint8)).to(dev)
    labels = torch.empty((side, side), dtype=torch.int16, device=dev)
    blended = torch.empty((CLASSES, side, side), dtype=torch.float32, device=dev)
    import ctypes as C
    ov = aa.lib().anh_required_input_dim(C.byref(cfg))
    tp = aa.tiling.parameters(1024, 1024, ov, ov)  # GPU defaults of the reference (annonet_infer_main.cpp:300-303,423-427)
    tiles = aa.tiling.get_tiles(side, side, tp)
    mine = aad.shard_tiles(tiles, rank, world)
    exchange = aad.OverlapExchange(tiles, world, side, side, dev)   # the pixels where tiles of different ranks overlap (none at world 1)

    def run():
        # blend this rank's tiles -> all-reduce the plane sums of the cross-rank overlaps -> label this rank's rows
        aad.sharded_infer(net, image, labels, blended, tiles, rank, world, exchange, tiling_parameters=tp, stream=net_stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    if rank == 0:
        out = {"metric": f"Mpixels/s tiled inference, {side}x{side} image, 1024^2 tiles, overlap {ov}", "value": side * side * args.steps / elapsed / 1e6,
               "unit": "Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": {"workload": f"annonet_infer(): {len(tiles)} tiles ({len(mine)} on rank 0), levels={LEVELS} width={WIDTH} K={CLASSES}", "parallelism": f"tile-shard{world}", "exchanged_pixels": exchange.pixels()}}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "infer"], help="train = BASELINE.json's metric (default); infer = tiled inference over a 4096x4096 image")
    ap.add_argument("--image-side", type=int, default=4096)
    args = ap.parse_args()

    import torch
    import annonet_amd as aa
    from annonet_amd import dist as aad

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    aa._lib.check(aa.lib().anh_set_device(local_rank))
    import torch.distributed as dist
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run: take the RCCL path even at world 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    prec = aa.ANH_BF16 if args.precision == "bf16" else aa.ANH_FP32
    if args.mode == "infer":
        return bench_infer(args, aa, aad, torch, dist, prec, rank, local_rank, world, use_dist)
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

    def step():
        aad.data_parallel_step(t, bucket, d_img.data_ptr(), d_lab.data_ptr(), d_w.data_ptr(), BATCH, TILE, TILE, world, force_collective=use_dist, stream=t_stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: warm-up, then a short all-kernel profile to find the dominant kernel class
    for _ in range(max(args.warmup, 1)):
        step()
    t.synchronize()
    t.profile_enable(True)
    for _ in range(2):
        step()
    t.synchronize()
    prof_all = t.profile()
    t.profile_reset()
    dominant = max(prof_all, key=lambda e: e["total_ms"])["name"] if prof_all else ""
    t.profile_set_filter(dominant)
    # An event pair costs ~6 us of stream time per launch, so the timed region times the dominant kernel on a sample of
    # its steps (every 4th), not on all of them: the average launch duration is estimated from >= steps/4 * launches.
    sample_every = 4 if args.steps >= 8 else 1
    t.profile_set_sampling(sample_every)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    t.synchronize()
    dom = [e for e in t.profile() if e["name"] == dominant]
    loss = t.get_last_loss()
    t.profile_enable(False)

    if rank == 0:
        roof = None
        if dom and dom[0]["launches"]:
            e = dom[0]
            avg_s = e["total_ms"] / 1e3 / e["launches"]
            flops, byts = e["flops"] / e["launches"], e["bytes"] / e["launches"]
            ai = flops / byts if byts else 0.0
            if flops > 0 and ai >= RIDGE:  # above the ridge point (peak flops / peak bytes): the matrix cores bound it
                roof = {"bound": "mfma", "achieved": flops / avg_s / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s"}
            else:                          # below it: HBM traffic bounds it
                roof = {"bound": "hbm", "achieved": byts / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s"}
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["traffic"] = pmc_traffic(dominant)
            roof["algorithmic_bytes_per_launch"] = byts
            roof["kernel"] = dominant
            roof["avg_launch_us"] = avg_s * 1e6
            roof["launches"] = e["launches"]
            roof["sampled_every_n_steps"] = sample_every
            roof["arithmetic_intensity"] = ai
        total_ms = sum(e["total_ms"] for e in prof_all) or 1.0
        breakdown = sorted(((e["name"], round(100 * e["total_ms"] / total_ms, 1)) for e in prof_all), key=lambda x: -x[1])[:8]
        if os.environ.get("ANH_BENCH_VERBOSE"):
            for e in sorted(prof_all, key=lambda e: -e["total_ms"]):
                per = e["total_ms"] / 2.0
                print(f"  {e['name']:44s} {per:8.3f} ms/step  {e['launches'] // 2:3d} launches  "
                      f"{e['flops'] / 2 / per / 1e9 if per else 0:9.1f} TFLOP/s  {e['bytes'] / 2 / per / 1e6 if per else 0:9.1f} GB/s", file=sys.stderr)
        out = {
            "metric": "227x227 RGB tiles/sec fwd+bwd @ batch 32", "value": BATCH * world * args.steps / elapsed, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"training step (fwd+loss+bwd+SGD), batch {BATCH}x3x{TILE}x{TILE} per GPU, encoder-decoder levels={LEVELS} width={WIDTH} K={CLASSES}, random init",
                       "global_batch": BATCH * world, "parallelism": f"dp{world}"},
            "roofline": roof,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(),   # rank 0 at N=1 only
            "kernel_time_share_pct": breakdown,
            "final_loss": loss,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
