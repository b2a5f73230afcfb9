"""Generator inference at 180x320, batch 32 (BASELINE configs[1]), eager launches: run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of the inference metric."""
import importlib, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
dev = "cuda:0"
torch.manual_seed(0)
G = pkg.Generator(types.SimpleNamespace(n_filters=64, n_layers=8), compute_dtype=os.environ.get("INF_DTYPE", "bf16")).to(dev).eval()
x = torch.rand(32, 3, 180, 320, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(2):
        G(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        G(x)
    torch.cuda.synchronize()
print("eager: %.2f ms per batch of 32 (%.0f FPS)" % ((time.perf_counter() - t0) / n * 1e3, 32 * n / (time.perf_counter() - t0)))
