"""Shader clock while the weight-gradient kernel runs back to back (~2 s per variant library given in FSR_HIP_LIB):
separates 'the DMA traffic costs clock' from 'the DMA traffic costs issue slots'.  python tools/wgrad_clock.py [cin cout hw batch]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ClockSampler  # noqa: E402

ops = importlib.import_module("fast-srgan_amd.ops")
cin, cout, hw, n = (int(v) for v in (sys.argv[1:5] + ["256", "256", "96", "32"][len(sys.argv) - 1:]))
cd = ops.Compute("bf16")
dev = torch.device("cuda:0")
x = torch.randn(n, hw, hw, cin, device=dev).to(cd.torch_dtype)
dy = torch.randn(n, hw, hw, cout, device=dev).to(cd.torch_dtype)
fn = lambda: ops.conv3x3_wgrad_raw(cd, x, dy, cout, cin, 1)  # noqa: E731
fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        fn()
g.replay()
torch.cuda.synchronize()
with ClockSampler(0) as cs:
    t0 = time.time()
    reps = 0
    while time.time() - t0 < 2.0:
        g.replay()
        reps += 1
    torch.cuda.synchronize()
    dt = time.time() - t0
s = cs.samples
us = dt / (reps * 20) * 1e6
gf = 2.0 * n * hw * hw * cin * cout * 9 / 1e9
print("%-12s %7.1f us  %7.1f TFLOP/s  sclk mean %.0f min %.0f max %.0f MHz (%d samples)" % (
    os.path.basename(os.environ.get("FSR_HIP_LIB", "shipped")), us, gf / us * 1e-3 * 1e3 / 1e3 * 1e0 if False else gf / (us * 1e-6) / 1e3,
    sum(s) / max(len(s), 1), min(s or [0]), max(s or [0]), len(s)))
