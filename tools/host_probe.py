"""Host-side staging costs on the GPU box: CPU writes/reads of pinned memory, pageable vs pinned copies, uint8 vs float head."""
import importlib, os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
dev = "cuda:0"
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
h, w = 180, 320
rng = np.random.default_rng(0)
f = rng.integers(0, 256, size=(8, h, w, 3), dtype=np.uint8)
pin = torch.empty((8, h, w, 3), dtype=torch.uint8).pin_memory()
pag = torch.empty((8, h, w, 3), dtype=torch.uint8)
pin_np = pin.numpy()
print("fill pinned via torch copy_  ms", t(lambda: pin.copy_(torch.from_numpy(f))))
print("fill pinned via np.copyto    ms", t(lambda: np.copyto(pin_np, f)))
print("fill pageable via np.copyto  ms", t(lambda: np.copyto(pag.numpy(), f)))
d = torch.empty((8, h, w, 3), dtype=torch.uint8, device=dev)
print("h2d from pinned nb           ms", t(lambda: d.copy_(pin, non_blocking=True)))
print("h2d from pageable            ms", t(lambda: d.copy_(pag)))
print("h2d from numpy direct        ms", t(lambda: d.copy_(torch.from_numpy(f))))
big = torch.empty((8, 4 * h, 4 * w, 3), dtype=torch.uint8, device=dev)
pout = torch.empty((8, 4 * h, 4 * w, 3), dtype=torch.uint8).pin_memory()
gout = torch.empty((8, 4 * h, 4 * w, 3), dtype=torch.uint8)
print("d2h to pinned nb (22MB)      ms", t(lambda: pout.copy_(big, non_blocking=True)))
print("d2h to pageable (22MB)       ms", t(lambda: gout.copy_(big)))
print("read pinned -> numpy copy    ms", t(lambda: pout.numpy().copy()))
print("read pageable -> numpy copy  ms", t(lambda: gout.numpy().copy()))
print("torch threads", torch.get_num_threads())
G = pkg.Generator(types.SimpleNamespace(n_filters=64, n_layers=8)).to(dev).eval()
for bsz in (1, 8, 32):
    xu = torch.randint(0, 256, (bsz, h, w, 3), dtype=torch.uint8, device=dev)
    xf = torch.rand(bsz, 3, h, w, device=dev) * 2 - 1
    with torch.no_grad():
        G.forward_u8(xu); G(xf); torch.cuda.synchronize()
        gu, gf = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gu): yu = G.forward_u8(xu)
        with torch.cuda.graph(gf): yf = G(xf)
    print("batch", bsz, "u8 graph ms", round(t(gu.replay), 3), "float graph ms", round(t(gf.replay), 3), flush=True)
