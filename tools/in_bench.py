"""Per-shape timing of the InstanceNorm kernels (forward apply; backward reduce + apply) against the bytes they move.
Usage on the GPU box: python tools/in_bench.py [--batch 32]"""
import argparse
import importlib
import os
import sys

import faulthandler

import torch

faulthandler.enable()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
L = importlib.import_module("fast-srgan_amd._lib")
ops = importlib.import_module("fast-srgan_amd.ops")
from conv_bench import timeit  # noqa: E402

SHAPES = [("G 64ch @96 (+res)", 96, 64, True), ("D 64ch @192", 192, 64, False), ("D 128ch @192", 192, 128, False),
          ("D 128ch @96", 96, 128, False), ("D 256ch @96", 96, 256, False), ("D 256ch @48", 48, 256, False),
          ("D 512ch @48", 48, 512, False), ("D 512ch @24", 24, 512, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    cd = ops.Compute("bf16")
    dev = torch.device("cuda:0")
    n = args.batch
    print("%-20s %8s | %8s %7s | %8s %7s" % ("layer", "MB", "fwd us", "TB/s", "bwd us", "TB/s"), flush=True)
    for name, hw, c, has_res in SHAPES:
        x = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
        res = torch.randn_like(x) if has_res else None
        xf = x.float()
        stats = torch.stack([xf.sum((1, 2)), (xf * xf).sum((1, 2))], -1).contiguous()
        prelu = torch.tensor([0.25], device=dev)
        mb = x.numel() * 2 / 1e6

        def fwd():
            return ops.instnorm_act(x, stats, res, prelu, cd, act=L.ACT_PRELU)

        t_f = timeit(fwd) * 1e3
        g = torch.randn_like(x)
        lib = L.lib()
        sums = torch.zeros((n, c, 2), device=dev)
        dprelu = torch.zeros((1,), device=dev)
        scr = torch.empty(lib.fsr_instnorm_act_bwd_scratch(n, hw * hw, c) // 4 + 1, device=dev)
        dx = torch.empty_like(x)
        P = ops._p

        def bwd():
            st = ops._stream()
            L.check(lib.fsr_instnorm_act_bwd_reduce(cd.code, P(g), P(x), P(stats), L.ACT_PRELU, 0.0, P(prelu), P(sums), P(dprelu), P(scr),
                                                    n, hw * hw, c, st), "reduce")
            L.check(lib.fsr_instnorm_act_bwd_apply(cd.code, P(g), P(x), P(stats), P(sums), L.ACT_PRELU, 0.0, P(prelu), P(dx), n, hw * hw,
                                                   c, st), "apply")

        t_b = timeit(bwd) * 1e3
        fb = mb * (3 if has_res else 2)
        bb = mb * 5          # reduce reads g, x; apply reads g, x and writes dx
        print("%-20s %8.1f | %8.1f %7.2f | %8.1f %7.2f" % (name, mb, t_f, fb / t_f, t_b, bb / t_b), flush=True)


if __name__ == "__main__":
    main()
