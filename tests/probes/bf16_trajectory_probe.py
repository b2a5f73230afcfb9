"""How much of the 16-bit convergence result is trajectory noise?  Runs tools/convergence.run() for ONE 16-bit mode with several
label-noise seeds (and whatever FSR_* switches / FSR_HIP_LIB the environment sets) and prints the late content loss, PSNR, SSIM.
python tests/probes/bf16_trajectory_probe.py [mode] [seed ...]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import convergence as C  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
seeds = [int(v) for v in sys.argv[2:]] or [0, 1]
pkg = importlib.import_module("fast-srgan_amd")
pkg._lib.lib()


def reduce4(x):
    return torch.nn.functional.interpolate(x, size=(96, 96), mode="bicubic", antialias=True, align_corners=False)


hr_all = C.synthetic_dataset(16, 384, seed=7)
lr_all = reduce4(hr_all)
hr_eval = C.synthetic_dataset(8, 384, seed=8)
lr_eval = reduce4(hr_eval)
tag = "%s %s" % (os.path.basename(os.environ.get("FSR_HIP_LIB", "shipped")), " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("FSR_WGRAD")))
for s in seeds:
    r = C.run(pkg, mode, 300, s, hr_all, lr_all, hr_eval, lr_eval)
    c = r["curves"]["content_loss"]
    print("%-40s %s seed %d: content_loss @150/200/250/300 = %.5f %.5f %.5f %.5f  psnr %.2f ssim %.4f" % (
        tag, mode, s, C.smooth_at(c, 150, 25), C.smooth_at(c, 200, 25), C.smooth_at(c, 250, 25), C.smooth_at(c, 300, 25), r["psnr"], r["ssim"]), flush=True)
