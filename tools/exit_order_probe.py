#!/usr/bin/env python3
"""Exit-time crash probe (round 2: `bench.py --mode infer` printed its line and died with SIGSEGV; the caller-side `del` ordering hid it).
Runs the old flow of the inference bench at a small size in child processes, one per VARIANT of what is alive / dropped in which
order when the function returns, each under `-X faulthandler`, and prints return code + the fault's Python stack.

    python tools/exit_order_probe.py            -> table of variants (parent never touches the GPU)
    python tools/exit_order_probe.py <variant>  -> one child
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = ["natural", "natural_nocpu", "no_pinned", "no_waitstream", "net_first", "net_first_nosync", "handle_last", "workaround"]


def flow(variant):
    import numpy as np
    import torch
    import annonet_amd as aa
    from annonet_amd import dist as aad
    import bench
    side = 2048
    dev = torch.device("cuda", 0)
    prec = aa.ANH_BF16
    cfg = aa.net_config(2, 3, 3, 1.0, 1, prec)
    tr = aa.TrainingNet(2, 3, prec, seed=2)
    tr.SetNetWidth(1.0, 1); tr.SetClassCount(3); tr.Initialize()
    net = tr.GetRuntimeNet(prec)
    del tr
    net_stream = torch.cuda.ExternalStream(net.stream_ptr(), device=dev)   # the bare wrapper (dist.handle_stream would keep `net` alive): the LIBRARY must cope
    rng = np.random.default_rng(3)
    image = torch.from_numpy(rng.integers(0, 256, (side, side, 3), dtype=np.uint8)).to(dev)
    labels = torch.empty((side, side), dtype=torch.int16, device=dev)
    host_labels = torch.empty((side, side), dtype=torch.int16).pin_memory() if variant != "no_pinned" else None
    blended = torch.empty((3, side, side), dtype=torch.float32, device=dev)
    import ctypes as C
    ov = aa.lib().anh_required_input_dim(C.byref(cfg))
    tp = aa.tiling.parameters(1024, 1024, ov, ov)
    tiles = aa.tiling.get_tiles(side, side, tp)
    exchange = aad.OverlapExchange(tiles, 1, side, side, dev)

    def run(to_host=False):
        row0, row1 = aad.sharded_infer(net, image, labels, blended, tiles, 0, 1, exchange, tiling_parameters=tp, stream=net_stream)
        if to_host and host_labels is not None:
            cur = torch.cuda.current_stream()
            if variant != "no_waitstream":
                cur.wait_stream(net_stream)
            else:
                net.synchronize()
            host_labels[row0:row1].copy_(labels[row0:row1], non_blocking=True)
            if variant != "no_waitstream":
                net_stream.wait_stream(cur)

    for _ in range(2):
        run()
    net.synchronize()
    net.profile_enable(True)
    run()
    net.synchronize()
    prof_all = net.profile()
    net.profile_reset()
    net.profile_set_filter("fwd_L8")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    net.synchronize()
    net.profile_enable(False)
    run(to_host=True)
    torch.cuda.synchronize()
    if variant not in ("natural_nocpu",):
        args = type("A", (), {"no_cpu_baseline": False})()
        bench.cpu_baseline_infer(aa, cfg, ov)
    print("line printed", flush=True)
    if variant == "workaround":
        torch.cuda.synchronize()
        del host_labels, exchange, net_stream
    elif variant == "net_first":
        torch.cuda.synchronize()
        del net
    elif variant == "net_first_nosync":
        del net
    elif variant == "handle_last":
        del host_labels, exchange, net_stream, image, labels, blended
        torch.cuda.synchronize()
    # everything else dies when the frame goes


def main():
    if len(sys.argv) > 1:
        import faulthandler
        faulthandler.enable()
        flow(sys.argv[1])
        print("flow returned", flush=True)
        return
    for v in VARIANTS:
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), v], capture_output=True, text=True, timeout=600, cwd=ROOT)
        err = [l for l in r.stderr.splitlines() if "amdgpu.ids" not in l]
        print(f"=== {v}: rc {r.returncode}; stdout: {r.stdout.strip().splitlines()[-2:]}")
        for l in err[-25:]:
            print("   ", l)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
