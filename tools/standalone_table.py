#!/usr/bin/env python3
"""Stand-alone rate of every layer x pass against the bf16 MFMA peak and the HBM peak, from a bench line measured on the ONE-stream
schedule (ANH_CONCURRENT_WGRAD=0: no kernel shares the chip with another, so an entry's HIP-event time is its stand-alone time).
usage: python tools/standalone_table.py <bench_one_stream.json>"""
import json
import sys

b = json.load(open(sys.argv[1]))
print("stand-alone (one-stream schedule) times of every layer x pass; MFMA column = algorithmic flops / time / 2500 TFLOP/s, HBM column = SURVEY 8d minimum bytes / time / 8000 GB/s")
print(f"{'entry':50s} {'us':>7s} {'GFLOP':>7s} {'TFLOP/s':>8s} {'of MFMA peak':>12s} {'min MB':>7s} {'GB/s':>7s} {'of HBM peak':>11s}  bound")
for l in b["layers"]:
    t = l["time_us"] * 1e-6
    tf, gb = l["gflop"] / 1e3 / t, l["min_mb"] / 1e3 / t
    print(f"{l['entry']:50s} {l['time_us']:7.1f} {l['gflop']:7.2f} {tf:8.0f} {tf / 2500:12.3f} {l['min_mb']:7.1f} {gb:7.0f} {gb / 8000:11.3f}  {l['bound']}")
print(f"step (one stream): {b['ms_per_step']:.3f} ms")
