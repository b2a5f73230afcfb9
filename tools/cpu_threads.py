"""CPU-baseline thread sweep (round-5 verdict item 9): the oracle's full GAN iteration (oracle/srgan_cpu.train_step, fp32, torch CPU /
oneDNN) at batch 4, 96 -> 384, with 16 / 32 / 64 / 128 / 256 threads on the GPU box's host -- bench.py's cpu_baseline uses the
fastest count (BASELINE.md section 3 asks for os.cpu_count(); the sweep shows what that costs).

    python tools/cpu_threads.py [threads,threads,...] > profiles/r06_cpu_threads.txt"""
import importlib
import os
import sys
import time
import types
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import srgan_cpu as O  # noqa: E402


def main():
    pkg = importlib.import_module("fast-srgan_amd")
    ns = types.SimpleNamespace
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    counts = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 32, 64, 128, 256]
    counts = sorted({min(c, avail) for c in counts})
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G = pkg.Generator(ns(n_filters=64, n_layers=8))
        D = pkg.Discriminator(ns(n_filters=64, n_layers=7))
    g_sd = {k: v.detach().clone() for k, v in G.state_dict().items()}
    d_sd = {k: v.detach().clone() for k, v in D.state_dict().items()}
    v_sd = O.vgg_standin_state_dict(1234, 1)
    b = 4
    lr, hr = torch.rand(b, 3, 96, 96) * 2 - 1, torch.rand(b, 3, 384, 384) * 2 - 1
    noise = [torch.rand(b, 1, 24, 24) for _ in range(3)]
    print("host: os.cpu_count() = %s, usable = %d, torch %s" % (os.cpu_count(), avail, torch.__version__))
    print("workload: oracle.train_step, batch %d, 96 -> 384, fp32; 1 warm-up + 2 timed iterations per thread count" % b)
    best = None
    for c in counts:
        torch.set_num_threads(c)
        O.train_step(g_sd, d_sd, v_sd, lr, hr, noise, {}, {})
        t0 = time.perf_counter()
        for _ in range(2):
            O.train_step(g_sd, d_sd, v_sd, lr, hr, noise, {}, {})
        dt = (time.perf_counter() - t0) / 2
        print("threads %4d   %.3f s per iteration   %.4f images/s" % (c, dt, b / dt), flush=True)
        if best is None or b / dt > best[1]:
            best = (c, b / dt)
    print("fastest: %d threads, %.4f images/s" % best)


if __name__ == "__main__":
    main()
