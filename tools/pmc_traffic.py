#!/usr/bin/env python3
"""HBM-side traffic per launch, PER bench.py PROFILER ENTRY (layer x pass), from two rocprofv3 counter passes
(FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

usage: python tools/pmc_traffic.py <launch_order.json> <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Collected as /opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots) prescribes:
  python3 bench.py --dump-launch-order launch_order.json ...                                 (any run: the order is static)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ... -- python3 bench.py ...      (own run)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ... -- python3 bench.py ...      (own run)
Both counters are in KiB.  gfx950 correction from the guide: FETCH_SIZE tallies 128-byte requests of wide (16 B/lane)
streaming reads at 64 B, i.e. reports half the bytes -> doubled here; WRITE_SIZE is exact for 16-B-per-lane streaming
stores.  Infinity-Cache hits are counted by these memory-side counters (they are "traffic leaving L2", an upper
bound on HBM traffic).

Mapping dispatches to entries.  Several layers run the SAME kernel instantiation, so a kernel name does not identify a layer.
The library records the entry of every launch of a step in host enqueue order (anh_profile_launch_order; bench.py
--dump-launch-order); rocprofv3's Dispatch_Id follows the same order.  Every entry launches a fixed kernel sequence (KERNELS
below); the dispatch stream is cut into steps by matching that sequence, and a dispatch whose kernel does not fit its entry, or a
profiled kernel of this library that fits no step, is an ERROR (the script fails rather than leaving holes in the json).
"""
import csv
import json
import re
import sys
from collections import defaultdict

# entry name (regex) -> the kernels one launch of that entry dispatches, in order (regexes on the demangled kernel name)
KERNELS = [
    (r"^conv_mfma_bf16:fwd_L\d+_stem_", [r"(?<!wgrad_)stem_mfma_kernel"]),
    (r"^conv_mfma_bf16:(fwd|dgrad)_L\d+_", [r"conv3x3_ws_kernel|conv3x3s1_mfma_kernel|conv3x3_down_mfma_kernel|conv3x3_up_mfma_kernel"]),
    (r"^wgrad_mfma_bf16:wgrad_L\d+_stem_", [r"wgrad_stem_mfma_kernel"]),
    (r"^wgrad_mfma_bf16:wgrad_L\d+_", [r"wgrad3x3_(ws|mfma)_kernel"]),
    (r"^conv_mfma_f32:", [r"conv_f32_mfma"]),
    (r"^conv_generic_(bf16|f32):", [r"conv_generic|stem_forward_kernel|head_forward"]),
    (r"^wgrad_generic_(bf16|f32):", [r"wgrad_generic|wgrad_stem"]),
    (r"^wgrad_reduce_partials(_main)?$", [r"reduce_partials_kernel"]),
    (r"^bn_forward_stats$", [r"bn_stats"]),
    (r"^bn_forward_finalize$", [r"bn_finalize_kernel"]),
    (r"^bn_bwd_reduce$", [r"bn_bwd_reduce"]),
    (r"^bn_bwd_finalize$", [r"bn_bwd_finalize_kernel"]),
    (r"^bn_bwd_apply$", [r"bn_bwd_apply"]),
    (r"^head_fused_fwd_loss_bwd$", [r"head_train_kernel", r"head_finalize_kernel"]),
    (r"^softmax_logloss$", [r"loss_kernel", r"loss_finalize"]),
    (r"^sgd_momentum_wd$", [r"sgd_kernel"]),
    (r"^bn_tables_zero$", []),                       # a runtime fill, not a kernel of this library
    (r"^bn_fold_all$", [r"bn_fold_all_kernel"]),
    (r"^head_blend_fused$", [r"head_blend_kernel"]),
    (r"^blend_accumulate$", [r"(?<!head_)blend(_batch)?_kernel"]),
    (r"^argmax_gain$", [r"argmax_kernel"]),
]
OURS = re.compile(r"anh::|_ZN3anh")          # kernels of this library (everything else in the trace: runtime copies / fills)


def kernels_of(entry):
    for rx, ks in KERNELS:
        if re.search(rx, entry):
            return [re.compile(k) for k in ks]
    raise SystemExit(f"pmc_traffic: no kernel pattern for profiler entry {entry!r} — add it to KERNELS")


def read_dispatches(path, counter):
    """-> list of (dispatch id, kernel name, counter value) sorted by dispatch id"""
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return [(i,) + rows[i] for i in sorted(rows)]


def per_entry(dispatches, order, what):
    """cuts the dispatch stream into steps by the launch order; -> {entry: [value per launch ...]}, steps matched"""
    flat = [(e, k) for e in order for k in kernels_of(e)]     # (entry, kernel regex) per dispatch of one step
    ours = [d for d in dispatches if OURS.search(d[1])]
    out = defaultdict(list)
    i = steps = 0
    unmatched = []
    while i < len(ours):
        if i + len(flat) <= len(ours) and all(rx.search(ours[i + j][1]) for j, (_, rx) in enumerate(flat)):
            per = defaultdict(float)
            for j, (e, _) in enumerate(flat):
                per[(e, j)] = ours[i + j][2]
            # one value per LAUNCH of an entry (an entry that appears several times per step, e.g. bn_bwd_finalize, gives several values)
            pos = 0
            for e in order:
                n = len(kernels_of(e))
                out[e].append(sum(ours[i + pos + t][2] for t in range(n)))
                pos += n
            i += len(flat)
            steps += 1
        else:
            unmatched.append(ours[i][1])
            i += 1
    if steps < 2:
        raise SystemExit(f"pmc_traffic: only {steps} complete steps of the launch order found in the {what} trace — the order does not fit the trace")
    # kernels outside complete steps: the first / last partial step of the run and set-up launches (layout refresh) are expected; anything
    # that is a large share of the trace is not
    if len(unmatched) > max(3 * len(flat), len(ours) // 10):
        raise SystemExit(f"pmc_traffic: {len(unmatched)} of {len(ours)} profiled kernels fit no step of the launch order ({what}); first: {unmatched[0][:120]}")
    return out, steps


def main():
    order_json, fetch_csv, write_csv, out = sys.argv[1:5]
    order = json.load(open(order_json))["order"]
    fetch, fs = per_entry(read_dispatches(fetch_csv, "FETCH_SIZE"), order, "FETCH_SIZE")
    write, ws = per_entry(read_dispatches(write_csv, "WRITE_SIZE"), order, "WRITE_SIZE")
    res = {}
    launches_per_step = defaultdict(int)
    for e in order:
        launches_per_step[e] += 1
    for e in launches_per_step:
        f, w = fetch[e], write[e]
        fetch_b = 2.0 * 1024.0 * sum(f) / len(f)   # KiB -> bytes, x2: the gfx950 FETCH_SIZE correction
        write_b = 1024.0 * sum(w) / len(w)
        res[e] = {"fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
                  "launches_per_step": launches_per_step[e], "launches_sampled": min(len(f), len(w))}
    total = sum(v["traffic_bytes_per_launch"] * v["launches_per_step"] for v in res.values())
    with open(out, "w") as f:
        json.dump({"unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, memory side of L2)", "steps_matched": {"fetch": fs, "write": ws},
                   "total_bytes_per_step": total, "entries": res}, f, indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches_per_step"]):
        print(f"{k:48s} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB  per launch x {v['launches_per_step']}  ({v['launches_sampled']} sampled)")
    print(f"total {total / 1e9:.3f} GB per step ({fs} / {ws} steps matched)")


if __name__ == "__main__":
    main()
