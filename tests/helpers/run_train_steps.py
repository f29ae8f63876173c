"""Runs a few bf16 training steps on a fixed seeded batch and saves (loss history, parameters, running statistics).
Used by tests/test_gpu_schedules.py: each kernel-schedule switch of the library is an environment variable read once per
process, so every variant runs in its own process and the results are compared afterwards.

usage: python tests/helpers/run_train_steps.py <out.npz>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import annonet_amd as aa  # noqa: E402


def main():
    out = sys.argv[1]
    rng = np.random.default_rng(123)
    n, d = 6, 99   # 99 = 4*24 + 3: valid for two levels; ragged against every tile size
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=11)
    t.SetNetWidth(1.0, 1)
    t.SetClassCount(3)
    t.Initialize()
    t.SetLearningRate(0.05)
    img = rng.integers(0, 256, (n, d, d, 3), dtype=np.uint8)
    lab = rng.integers(0, 3, (n, d, d)).astype(np.uint16)
    lab[rng.random((n, d, d)) < 0.1] = aa.LABEL_IGNORE
    wl = [aa.set_weights(l, 0.5, 0.5) for l in lab]
    losses = []
    for _ in range(int(os.environ.get("ANH_TEST_STEPS", "3"))):
        t.StartTraining(list(img), wl)
        t.synchronize()
        losses.append(t.get_last_loss())
    p, r = t.get_params()
    np.savez(out, losses=np.array(losses), params=p, running=r, step_graph=np.array(t.step_graph_stats()))


if __name__ == "__main__":
    main()
