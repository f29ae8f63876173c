import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"])
for l in d["layers"]:
    if l["pass"] == "fwd":
        print("  ", l["entry"].ljust(44), l["time_us"])
for o in d["overhead_kernels"]:
    if "forward" in o["entry"] or "stats" in o["entry"]:
        print("  ", o["entry"], o["us_per_step"])
