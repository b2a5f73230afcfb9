"""Diagnostic: fp16 training of tools/convergence.py's scenario at several INITIAL loss scales (dynamic scaler on)."""
import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import convergence as C
pkg = importlib.import_module("fast-srgan_amd")
red = lambda x: torch.nn.functional.interpolate(x, size=(96, 96), mode="bicubic", antialias=True, align_corners=False)
hr_all = C.synthetic_dataset(16, 384, 7); lr_all = red(hr_all); hr_eval = C.synthetic_dataset(8, 384, 8); lr_eval = red(hr_eval)
for mode, scale in (("f16", "16384"), ("f16", "1048576"), ("f16", "67108864"), ("f16", "256")):
    os.environ["CONV_LOSS_SCALE"] = scale
    r = C.run(pkg, mode, 300, 0, hr_all, lr_all, hr_eval, lr_eval)
    c = r["curves"]
    print(mode, "scale", scale, "content@100/200/300 %.5f %.5f %.5f" % tuple(C.smooth_at(c["content_loss"], t, 25) for t in (100, 200, 300)),
          "adv@300 %.4f" % C.smooth_at(c["adv_loss"], 300, 25), "psnr %.2f ssim %.4f" % (r["psnr"], r["ssim"]), "scale/skipped", r["loss_scale"], flush=True)
