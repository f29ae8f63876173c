#!/usr/bin/env python3
"""Turns gpurun_out/prof (tools/collect_profiles.sh) into the committed, judged set under profiles/ with the round's prefix:
kernel stats, per-entry HBM traffic (tools/pmc_traffic.py), SQ counters per entry (tools/pmc_counters.py), per-launch timelines
(tools/timeline.py), the bench lines.     usage: python tools/publish_profiles.py r02"""
import glob
import gzip
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def one(pattern):
    m = glob.glob(os.path.join(SRC, pattern))
    assert len(m) == 1, (pattern, m)
    return m[0]


def run(*cmd, out):
    with open(out, "w") as f:
        subprocess.check_call([sys.executable, *cmd], stdout=f)


for name in ("bench.json", "bench_under_rocprof.json", "bench_infer_4096_bf16.json", "bench_infer_under_rocprof.json", "bench_infer_4096_fp32.json", "bench_infer_16384_bf16.json",
             "bench_one_stream.json", "launch_order.json", "infer_launch_order.json"):
    shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{tag}_{name}"))
shutil.copy(one("kt/*/*_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "standalone_table.txt"), os.path.join(DST, f"{tag}_standalone_table.txt"))
shutil.copy(one("ikt/*/*_kernel_stats.csv"), os.path.join(DST, f"{tag}_infer_kernel_stats.csv"))
run(os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(SRC, "infer_launch_order.json"), one("ifetch/*/*_counter_collection.csv"), one("iwrite/*/*_counter_collection.csv"),
    os.path.join(DST, f"{tag}_infer_traffic.json"), out=os.path.join(DST, f"{tag}_infer_traffic.txt"))
order = os.path.join(SRC, "launch_order.json")
fetch, write, sq = one("fetch/*/*_counter_collection.csv"), one("write/*/*_counter_collection.csv"), one("sq/*/*_counter_collection.csv")
run(os.path.join(ROOT, "tools", "pmc_traffic.py"), order, fetch, write, os.path.join(DST, f"{tag}_traffic.json"), out=os.path.join(DST, f"{tag}_traffic.txt"))
run(os.path.join(ROOT, "tools", "pmc_counters.py"), order, sq, os.path.join(DST, f"{tag}_sq_counters.json"), out=os.path.join(DST, f"{tag}_sq_counters.txt"))
run(os.path.join(ROOT, "tools", "timeline.py"), one("kt/*/*_kernel_trace.csv"), out=os.path.join(DST, f"{tag}_timeline_two_streams.txt"))
run(os.path.join(ROOT, "tools", "timeline.py"), one("kt1/*/*_kernel_trace.csv"), out=os.path.join(DST, f"{tag}_timeline_one_stream.txt"))
for src, name in ((fetch, "pmc_FETCH_SIZE"), (write, "pmc_WRITE_SIZE")):      # the raw per-dispatch counter rows, compressed
    with open(src, "rb") as f, gzip.open(os.path.join(DST, f"{tag}_{name}.csv.gz"), "wb") as g:
        g.write(f.read())
print("published:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(DST, tag + "_*"))))
