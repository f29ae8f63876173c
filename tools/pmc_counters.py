#!/usr/bin/env python3
"""Any rocprofv3 --pmc counters per bench.py profiler entry (layer x pass): the dispatch -> entry mapping of tools/pmc_traffic.py
applied to every counter in one counter_collection.csv.

usage: python tools/pmc_counters.py <launch_order.json> <counter_collection.csv> [out.json]
Prints one line per entry with the per-launch average of every counter, plus ratios that read directly:
  valu%  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES   stall% = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  mfma%  = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024): the matrix-pipe occupancy of the kernel.  Numerator: cycles, summed
           over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16: MI355X_MICROARCH.md; checked here: the 128->128 forward issues
           903,168 x 1.143 (tile quantisation) MFMAs = 3.30e7 cycles, the counter reads 3.303e7).  Denominator: the kernel's own cycles
           x 1024 SIMDs, the kernel's cycles being SQ_BUSY_CYCLES / 32 (summed over the 32 shader engines; checked against the kernel's
           duration: 3.0e6 / 32 = 94.5 k cycles = 47 us at 2.0 GHz).  At the clock the kernel HELD under the profiler (kernels serialised),
           not the 2.4 GHz the 2.5 PFLOP/s peak assumes: x (held clock / 2.4 GHz) x (useful / issued MFMAs) gives the fraction of peak.
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
        if row.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            row["mfma_busy_pct"] = 100.0 * row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
            extra += f" mfma {row['mfma_busy_pct']:5.1f}%"
        print(e.ljust(48), " ".join(f"{row.get(n, float('nan')):18.4g}" for n in names), extra)
    if len(sys.argv) > 3:
        json.dump({"counters": names, "entries": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
