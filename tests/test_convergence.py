"""Does training in the 16-bit modes converge like training at the reference's precision?  (trainer.py:89-141 + :158-196.)

tools/convergence.py trains the BASELINE cfg #1-size networks from one initialisation on identical batches: three exact-f32
runs that differ only in their label noise (the fp32 run-to-run band) and one bf16 / f16 run.  Gates, in half-widths of that
band (floored at +-2 % of the value): every smoothed loss from iteration 100 on within 6 (measured on the MI355X: <= 2.4 with
the shipped kernels, profiles/r03_convergence.txt; <= 5.4 across the kernel versions of round 3 -- GAN training is chaotic, any
change of summation order moves the trajectory by a band or two), the first 50 iterations -- where the three f32 runs have not
spread yet and the band is a hair -- within 12 (measured 4.9), final PSNR / SSIM of the generator on a held-out batch within
4.5 (measured 0.6); everything finite.  What the gate excludes is the failure round 3 found and fixed: fp16 with the former
static loss scale of 2^14 ended 300 iterations with a content loss 5-10x the fp32 band's (12.7 half-widths) and 6 dB PSNR."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LATE, EARLY, QUALITY = 6.0, 12.0, 4.5      # half-widths of the fp32 band: iterations >= 100 / the first checkpoints / PSNR, SSIM


def _tool():
    spec = importlib.util.spec_from_file_location("convergence", os.path.join(ROOT, "tools", "convergence.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_16bit_training_tracks_fp32_training(pkg):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    conv = _tool()
    lines = []
    results, rows, worst = conv.main(300, modes=("bf16", "f16"), log=lines.append)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "convergence.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    for r in results["f32"] + [results["bf16"], results["f16"]]:
        assert r["finite"]
    # the f32 runs themselves learn: pre-training lowers the pixel loss, the discriminator separates real from fake
    pre = results["f32"][0]["curves"]["pretrain_loss"]
    assert sum(pre[-10:]) < 0.5 * sum(pre[:10]), (pre[:3], pre[-3:])
    bad = []
    for mode, k, t, lo, hi, v, dist in rows:
        gate = QUALITY if k in ("psnr", "ssim") else (LATE if t >= 100 and k != "pretrain_loss" else EARLY)
        if not dist < gate:
            bad.append((mode, k, t, round(lo, 5), round(hi, 5), round(v, 5), round(dist, 2)))
    assert not bad, bad
