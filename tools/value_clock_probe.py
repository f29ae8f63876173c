#!/usr/bin/env python3
"""Does the step's time depend on the VALUES it computes on?  The same trainer, the same batch, the same launches: a phase of steps at learning
rate 0.1 (the bench's: the weights move), a phase at learning rate 1e-30 (every store still happens, the weights stay where the first phase left
them), then a fresh trainer stepped at learning rate 1e-30 from its initial weights.  Prints ms per step of each phase and, if rocm-smi answers, the
shader clock sampled during it.   usage (GPU box): python tools/value_clock_probe.py [steps per phase = 1500]"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import annonet_amd as aa  # noqa: E402
from annonet_amd import dist as aad  # noqa: E402
import bench  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500


def sclk_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5)
            for line in r.stdout.splitlines():
                if line.startswith("card0"):
                    for field in line.split(","):
                        f = field.strip().strip("()").lower()
                        if f.endswith("mhz"):
                            out.append(field.strip())
                            break
        except Exception:
            pass
        time.sleep(0.2)


def make():
    t = aa.TrainingNet(bench.LEVELS, 3, aa.ANH_BF16, seed=2)
    t.SetNetWidth(bench.WIDTH, 1); t.SetClassCount(bench.CLASSES); t.Initialize()
    return t


def phase(t, lr, d, name):
    t.SetLearningRate(lr)
    stream = aad.handle_stream(t)
    bucket = aad.grad_bucket_tensor(t)
    def step():
        aad.data_parallel_step(t, bucket, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), bench.BATCH, bench.TILE, bench.TILE, 1, force_collective=False, stream=stream, early=None)
    for _ in range(50): step()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sclk_sampler, args=(stop, samples)); th.start()
    t0 = time.perf_counter()
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / STEPS
    stop.set(); th.join()
    print(f"{name:58s} {ms:.4f} ms/step   loss {t.get_last_loss():.4f}   sclk samples {samples[:3]} ... {samples[-3:]}", flush=True)


def main():
    img, lab, w = bench.synthetic_batch(0)
    dev = torch.device("cuda", 0)
    d = (torch.from_numpy(img).to(dev), torch.from_numpy(lab.view(np.int16)).to(dev), torch.from_numpy(w).to(dev))
    a = make()
    phase(a, 1e-30, d, "fresh trainer, learning rate 1e-30 (initial weights)")
    phase(a, 0.1, d, "same trainer, learning rate 0.1 (weights move)")
    phase(a, 1e-30, d, "same trainer, learning rate 1e-30 (trained weights, fixed)")
    phase(a, 0.1, d, "same trainer, learning rate 0.1 again")
    b = make()
    phase(b, 1e-30, d, "second fresh trainer, learning rate 1e-30 (initial weights)")


if __name__ == "__main__":
    main()
