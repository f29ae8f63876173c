"""bf16 versus fp32 label maps of a TRAINED net.  On random-init nets the logits of the classes lie within bf16 rounding of each
other almost everywhere, which is why tests/test_gpu_parity.py can only bound the bf16 / fp32 disagreement loosely.  north_star
asks for label-map parity; the meaningful statement is on a net that has learnt something: after ~300 steps on a separable
synthetic task the class margins dwarf bf16 rounding and the two precisions must label held-out images identically except for a
fraction << 1 % of pixels, all of them at small top-2 margins.  (ANH_FP32 remains the bit-exact parity mode; INTEGRATION.md states
why the C++ shim defaults to bf16.)"""
import numpy as np
import pytest

import annonet_amd as aa

pytestmark = pytest.mark.gpu

D = 67   # 4 * 16 + 3: valid for two levels


def scene(rng, h, w):
    """three classes separable by brightness and texture: background noise, mid-gray blobs, bright striped blobs"""
    lab = np.zeros((h, w), np.uint16)
    for _ in range(rng.integers(3, 8)):
        y, x = rng.integers(0, h - 8), rng.integers(0, w - 8)
        lab[y:y + rng.integers(8, 30), x:x + rng.integers(8, 30)] = rng.integers(1, 3)
    img = rng.integers(0, 110, (h, w, 3)).astype(np.int64)     # heavy pixel noise: single pixels are ambiguous, neighbourhoods are not
    img[lab == 1] += 45
    img[lab == 2] += 95
    img[(lab == 2) & ((np.arange(w)[None, :] // 2) % 2 == 0)] -= 35
    return np.clip(img, 0, 255).astype(np.uint8), lab


def test_bf16_and_fp32_label_maps_agree_on_a_trained_net():
    rng = np.random.default_rng(0)
    t = aa.TrainingNet(2, 3, aa.ANH_BF16, seed=1)
    t.SetNetWidth(0.5, 8); t.SetClassCount(3); t.Initialize()
    t.SetLearningRate(0.05)
    losses = []
    for step in range(300):
        imgs, wls = [], []
        for _ in range(8):
            img, lab = scene(rng, D, D)
            imgs.append(img); wls.append(aa.set_weights(lab, 0.5, 0.5))
        t.StartTraining(imgs, wls)
        if step % 50 == 0 or step == 299:
            losses.append(t.get_last_loss())
    assert losses[-1] < 0.25 * losses[0], losses          # it learnt
    fp32, bf16 = t.GetRuntimeNet(aa.ANH_FP32), t.GetRuntimeNet(aa.ANH_BF16)
    ov = t.GetRequiredInputDimension()
    tp = aa.tiling.parameters(160, 160, ov, ov)
    total = mismatch = wrong = 0
    worst_margin = 0.0
    for _ in range(6):                                   # held-out images, tiled (several tiles each)
        img, lab = scene(rng, 300, 420)
        a, planes = aa.annonet_infer(fp32, img, tiling_parameters=tp, want_blended=True)
        b = aa.annonet_infer(bf16, img, tiling_parameters=tp)
        differ = a != b
        total += a.size; mismatch += int(differ.sum()); wrong += int((a != lab).sum())
        if differ.any():
            srt = np.sort(planes, axis=0)
            worst_margin = max(worst_margin, float((srt[-1] - srt[-2])[differ].max() / (planes.max() - planes.min())))
    assert wrong / total < 0.2                            # the fp32 labels are mostly right: the margins are real
    assert mismatch / total < 2e-3, (mismatch, total)     # << 1 %: bf16 flips only near-ties
    assert worst_margin < 0.05, worst_margin             # ... and only where the top-2 margin is a few percent of the logit range
    print(f"bf16 vs fp32 label mismatch on a trained net: {mismatch}/{total} = {mismatch / total:.2e}; fp32 error vs truth {wrong / total:.3f}; worst flipped margin {worst_margin:.3f} of the logit range")
