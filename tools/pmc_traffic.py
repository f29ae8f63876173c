#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Collected as /opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots) prescribes:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py ...      (own run)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ... -- python3 bench.py ...      (own run)
Both counters are in KiB.  gfx950 correction from the guide: FETCH_SIZE tallies 128-byte requests of wide (16 B/lane)
streaming reads at 64 B, i.e. reports half the bytes -> doubled here; WRITE_SIZE is exact for 16-B-per-lane streaming
stores.  Infinity-Cache hits are counted by these memory-side counters (they are "traffic leaving L2", an upper
bound on HBM traffic).  The JSON is keyed by bench.py's profiler entry names; each entry aggregates the kernel
instantiations that the entry launches (regexes below) and gives the AVERAGE bytes per launch.
"""
import csv
import json
import re
import sys
from collections import defaultdict

ENTRY_KERNELS = {
    "wgrad_mfma_bf16:wgrad_con3x3s1": r"wgrad3x3_(ws|mfma)_kernel<\d, \d, 0, 1(, false)?>",
    "wgrad_mfma_bf16:wgrad_con3x3s2": r"wgrad3x3_(ws|mfma)_kernel<\d, \d, 0, 2(, false)?>",
    "wgrad_mfma_bf16:wgrad_cont3x3s2": r"wgrad3x3_(ws|mfma)_kernel<\d, 0, [12], 2(, true)?>",
    "wgrad_mfma_bf16:wgrad_stem": r"wgrad_stem_mfma_kernel",
    "wgrad_reduce_partials": r"reduce_partials_kernel",
    "bn_bwd_reduce": r"bn_bwd_reduce",
    "bn_bwd_apply": r"bn_bwd_apply",
    "bn_bwd_finalize": r"bn_bwd_finalize",
    "bn_forward_stats": r"bn_stats",
    "bn_forward_finalize": r"bn_finalize_kernel",
    "conv_mfma_bf16:fwd_con3x3s1": r"conv3x3_ws_kernel<.*GeoS1, \d, [12]>",
    "conv_mfma_bf16:dgrad_con3x3s1": r"conv3x3_ws_kernel<.*GeoS1, \d, 0>",
    "conv_mfma_bf16:fwd_con3x3s2": r"conv3x3_ws_kernel<.*GeoDown, \d, [12]>",
    "conv_mfma_bf16:dgrad_cont3x3s2": r"conv3x3_ws_kernel<.*GeoDown, \d, 0>",
    "conv_mfma_bf16:fwd_cont3x3s2": r"conv3x3_ws_kernel<.*GeoUp, \d, [12]>",
    "conv_mfma_bf16:dgrad_con3x3s2": r"conv3x3_ws_kernel<.*GeoUp, \d, 0>",
    "conv_mfma_bf16:fwd_stem": r"(?<!wgrad_)stem_mfma_kernel",
    "head_fused_fwd_loss_bwd": r"head_train_kernel",
    "sgd_momentum_wd": r"sgd_kernel",
}


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            tot[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for entry, rx in ENTRY_KERNELS.items():
        pat = re.compile(rx)
        f = sum(v for k, v in ft.items() if pat.search(k))
        fn = sum(v for k, v in fc.items() if pat.search(k))
        w = sum(v for k, v in wt.items() if pat.search(k))
        wn = sum(v for k, v in wc.items() if pat.search(k))
        if not fn or not wn:
            continue
        fetch_b = 2.0 * 1024.0 * f / fn   # KiB -> bytes, x2: the gfx950 FETCH_SIZE correction
        write_b = 1024.0 * w / wn
        res[entry] = {"fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
                      "traffic_bytes_per_launch": fetch_b + write_b, "launches_sampled": int(min(fn, wn))}
    with open(out, "w") as f:
        json.dump({"unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, memory side of L2)", "entries": res}, f, indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]):
        print(f"{k:40s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB  per launch  ({v['launches_sampled']} launches)")


if __name__ == "__main__":
    main()
