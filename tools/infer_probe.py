"""Where does the time of InferencePipeline go?  (stage timings on the host, one shape)"""
import importlib, os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
dev = "cuda:0"
G = pkg.Generator(types.SimpleNamespace(n_filters=64, n_layers=8)).to(dev).eval()
rng = np.random.default_rng(0)
h, w = 180, 320
for bsz in (1, 8):
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(64)]
    pipe = pkg.InferencePipeline(G, dev, batch=bsz, depth=2)
    for _ in pipe.run(frames[:2 * bsz]):
        pass
    sl = pipe._slot(h, w, 0)
    T = {}
    def tick(name, t0):
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    reps = 20
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(bsz):
            sl.host_in[i].copy_(torch.from_numpy(frames[i]))
        tick("fill", t0)
        t0 = time.perf_counter(); sl.x.copy_(sl.host_in, non_blocking=True); tick("h2d", t0)
        t0 = time.perf_counter(); sl.graph.replay(); tick("graph", t0)
        t0 = time.perf_counter(); sl.host_out.copy_(sl.y, non_blocking=True); tick("d2h", t0)
        t0 = time.perf_counter(); out = [sl.host_out[i].numpy().copy() for i in range(bsz)]; tick("copyout", t0)
        t0 = time.perf_counter(); ev = torch.cuda.Event(); ev.record(); ev.synchronize(); tick("event", t0)
    print("batch", bsz, {k: round(v / reps * 1e3, 3) for k, v in T.items()}, "ms per batch", flush=True)
    t0 = time.perf_counter(); n = 0
    for _ in range(4):
        for y in pipe.run(frames):
            n += 1
    print("batch", bsz, "pipeline fps", round(n / (time.perf_counter() - t0), 1), flush=True)
