#!/usr/bin/env python3
"""Same-box A/B of two builds of libannonet_hip.so on the training bench: alternates the builds, prints ms/step of every run and the
per-entry times side by side.   usage: python tools/ab_bench.py <libA.so> <libB.so> [rounds] [extra bench.py args ...]"""
import json
import os
import subprocess
import sys

a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
extra = sys.argv[4:]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
res = {a: [], b: []}
for r in range(rounds):
    for lib in (a, b):
        env = dict(os.environ, ANH_LIBRARY=lib)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-infer", "--prewarm-s", "1"] + extra, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            sys.exit(out.stderr[-3000:])
        res[lib].append(json.loads(line[-1]))
        print(("A" if lib == a else "B"), f"{res[lib][-1]['ms_per_step']:.4f} ms/step", flush=True)
la, lb = res[a][-1], res[b][-1]
tb = {l["entry"]: l["time_us"] for l in lb["layers"]}
for l in la["layers"]:
    print(f"{l['entry']:48s} A {l['time_us']:8.1f}  B {tb.get(l['entry'], float('nan')):8.1f}")
oa, ob = ({o["entry"]: o["us_per_step"] for o in d["overhead_kernels"]} for d in (la, lb))
for k in oa:
    print(f"{k:48s} A {oa[k]:8.1f}  B {ob.get(k, float('nan')):8.1f}")
print("A best %.4f  B best %.4f ms/step" % (min(x["ms_per_step"] for x in res[a]), min(x["ms_per_step"] for x in res[b])))
