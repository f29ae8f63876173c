#!/usr/bin/env python3
"""Any rocprofv3 --pmc counters per bench.py profiler entry (layer x pass): the dispatch -> entry mapping of tools/pmc_traffic.py
applied to every counter in one counter_collection.csv.

usage: python tools/pmc_counters.py <launch_order.json> <counter_collection.csv> [out.json]
Prints one line per entry with the per-launch average of every counter, plus ratios that read directly:
  valu%  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES   stall% = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  mfma%  = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) when both are present (quad-cycle vs cycle units: MI355X_MICROARCH.md)
"""
import csv
import json
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import per_entry  # noqa: E402


def main():
    order = json.load(open(sys.argv[1]))["order"]
    by_counter = defaultdict(dict)
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            by_counter[r["Counter_Name"]][int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    res = defaultdict(dict)
    for c, rows in by_counter.items():
        disp = [(i,) + rows[i] for i in sorted(rows)]
        vals, steps = per_entry(disp, order, c)
        for e, v in vals.items():
            res[e][c] = sum(v) / len(v)
    names = sorted(by_counter)
    print("entry".ljust(48), " ".join(n[-18:].rjust(18) for n in names))
    for e in dict.fromkeys(order):
        row = res[e]
        extra = ""
        wc = row.get("SQ_WAVE_CYCLES")
        if wc:
            for label, key in (("valu", "SQ_ACTIVE_INST_VALU"), ("wait", "SQ_WAIT_ANY"), ("stall", "SQ_WAIT_INST_ANY"), ("lds", "SQ_ACTIVE_INST_LDS"), ("any", "SQ_ACTIVE_INST_ANY")):
                if key in row:
                    extra += f" {label} {100 * row[key] / wc:5.1f}%"
        print(e.ljust(48), " ".join(f"{row.get(n, float('nan')):18.4g}" for n in names), extra)
    if len(sys.argv) > 3:
        json.dump({"counters": names, "entries": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
