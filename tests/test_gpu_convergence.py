"""End-to-end sanity of the bf16 training path with every fused epilogue active: the net must LEARN.

150 SGD steps on blocky synthetic tiles whose label is the brightest colour channel of each 9x9 block; the loss has to fall
well below its start and the snapshot RuntimeNet has to label its last training batch.  (Parity proper is in
test_gpu_parity.py / test_gpu_ops.py; this guards against errors that keep every single step plausible — e.g. batch-norm
statistics or their gradients slightly off — but stop learning.)
"""
import numpy as np
import pytest

import annonet_amd as aa

pytestmark = pytest.mark.gpu


def test_bf16_training_learns_a_colour_rule():
    rng = np.random.default_rng(0)
    n, d = 8, 99
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=3)
    t.SetNetWidth(1.0, 1)
    t.SetClassCount(3)
    t.Initialize()
    t.SetLearningRate(0.05)
    losses = []
    for step in range(150):
        base = rng.integers(0, 256, (n, d // 9 + 1, d // 9 + 1, 3))
        img = np.kron(base, np.ones((1, 9, 9, 1)))[:, :d, :d, :].astype(np.uint8)
        lab = img.argmax(-1).astype(np.uint16)
        t.StartTraining(list(img), [aa.set_weights(l, 0.5, 0.5) for l in lab])
        if step % 10 == 9:
            t.synchronize()
            losses.append(t.get_last_loss())
    assert losses[-1] < 0.5 * losses[0], losses
    out = t.GetRuntimeNet(aa.ANH_BF16).Forward(img[0])
    assert (out.argmax(0) == lab[0]).mean() > 0.85
