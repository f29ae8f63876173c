#!/usr/bin/env python3
"""The per-layer table of DESIGN.md §7.1 from the committed bench line and traffic file.   usage: python tools/design_table.py r02"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
b = json.load(open(os.path.join(root, "profiles", f"{tag}_bench.json")))
t = json.load(open(os.path.join(root, "profiles", f"{tag}_traffic.json")))["entries"]
names = {"stem": "stem", "con3x3s2": "con3x3s2", "con3x3s1": "con3x3s1", "cont3x3s2": "cont3x3s2"}
print("| layer | pass | µs | GFLOP | min MB | flop/B | bound | floor µs | frac | PMC traffic MB |")
print("|---|---|---|---|---|---|---|---|---|---|")
for l in b["layers"]:
    tr = t.get(l["entry"], {}).get("traffic_bytes_per_launch")
    print(f"| L{l['layer']} {names.get(l['kind'], l['kind'])} {l['cin']}→{l['cout']} @{l['side']} | {l['pass']} | {l['time_us']:.1f} | {l['gflop']:.2f} | {l['min_mb']:.1f} | "
          f"{l['flop_per_byte']:.0f} | {l['bound']} | {l['floor_us']:.1f} | {l['frac']:.2f} | {tr / 1e6:.0f} |" if tr else f"| L{l['layer']} | {l['pass']} | {l['time_us']:.1f} | | | | | | {l['frac']:.2f} | |")
