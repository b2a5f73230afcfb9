"""Generator inference at 180x320 (BASELINE configs[1]), model only, hipGraph replay, by batch size: does a smaller batch --
whose activations (118 MB per image after the last up-sampling stage) stay in the 256 MB Infinity Cache between the kernels --
run faster per image than batch 32?"""
import importlib, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
model_mod = importlib.import_module("fast-srgan_amd.model")
dev = "cuda:0"
torch.manual_seed(0)
G = pkg.Generator(types.SimpleNamespace(n_filters=64, n_layers=8), compute_dtype="bf16").to(dev).eval()
for h, w in ((180, 320), (90, 160)):
    for b in (1, 2, 4, 8, 16, 32):
        x = torch.rand(b, 3, h, w, device=dev) * 2 - 1
        gg = model_mod.GraphedGenerator(G, x)
        for _ in range(3):
            gg(x)
        torch.cuda.synchronize()
        n = max(4, 64 // b)
        t0 = time.perf_counter()
        for _ in range(n):
            gg(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%dx%d batch %2d: %7.3f ms per batch  %8.1f FPS" % (h, w, b, dt * 1e3, b / dt), flush=True)
        del gg
