"""TEST INFRASTRUCTURE: repeats the GPU half of the small golden checks (test_modules.py) and prints the worst errors."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
from backend import select, relerr
from conftest import load_npz, sd_from
pkg = importlib.import_module("fast-srgan_amd")
ns = types.SimpleNamespace
dev = select("hip")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, make, stem in (("G", lambda: pkg.Generator(ns(n_filters=16, n_layers=2), compute_dtype="f32"), "g"),
                         ("D", lambda: pkg.Discriminator(ns(n_filters=16, n_layers=7), compute_dtype="f32"), "d")):
    z = load_npz(stem + "_small.npz")
    M = make()
    M.load_state_dict(sd_from(z, "sd."))
    M.to(dev)
    ref = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
    wy = wdx = wg = 0.0
    who = None
    for it in range(N):
        for p in M.parameters():
            p.grad = None
        x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
        y = M(x)
        (y * torch.from_numpy(z["r"]).to(dev)).sum().backward()
        wy = max(wy, relerr(y, torch.from_numpy(z["y"])))
        wdx = max(wdx, relerr(x.grad, torch.from_numpy(z["dx"])))
        for k, p in M.named_parameters():
            e = relerr(p.grad, ref[k])
            if e > wg:
                wg, who = e, k
    print("%s small golden, %d runs: worst y %.2e dx %.2e param grad %.2e (%s)" % (name, N, wy, wdx, wg, who), flush=True)
