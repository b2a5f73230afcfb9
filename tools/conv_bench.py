"""Per-layer timing of the 3x3 convolution kernels (forward, data gradient, weight gradient) at the shapes of
the bench workload (batch 32, 96->384), next to torch/MIOpen's conv on the same tensors as a same-hardware
yardstick.  Usage on the GPU box: python tools/conv_bench.py [--batch 32] [--only fwd,dgrad,wgrad] [--miopen]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("fast-srgan_amd._lib")
ops = importlib.import_module("fast-srgan_amd.ops")

# (name, cin, cout, H, W (input), stride, pixel_shuffle)
SHAPES = [
    ("G stem 64->64 @96", 64, 64, 96, 96, 1, False),
    ("G up0 64->256 @96", 64, 256, 96, 96, 1, True),
    ("G up1 64->256 @192", 64, 256, 192, 192, 1, True),
    ("G head 64->3 @384", 64, 3, 384, 384, 1, False),
    ("D/VGG first 3->64 @384", 3, 64, 384, 384, 1, False),
    ("D s2 64->64 @384", 64, 64, 384, 384, 2, False),
    ("D 64->128 @192", 64, 128, 192, 192, 1, False),
    ("D s2 128->128 @192", 128, 128, 192, 192, 2, False),
    ("D 128->256 @96", 128, 256, 96, 96, 1, False),
    ("D s2 256->256 @96", 256, 256, 96, 96, 2, False),
    ("D 256->512 @48", 256, 512, 48, 48, 1, False),
    ("D s2 512->512 @48", 512, 512, 48, 48, 2, False),
    ("VGG 64->64 @384", 64, 64, 384, 384, 1, False),
    ("VGG 128->128 @192", 128, 128, 192, 192, 1, False),
    ("VGG 256->256 @96", 256, 256, 96, 96, 1, False),
    ("VGG 512->512 @48", 512, 512, 48, 48, 1, False),
    ("VGG 512->512 @24", 512, 512, 24, 24, 1, False),
]


def timeit(fn, iters=10, eager=False):
    """Kernel time per call: `iters` calls captured into one hipGraph (no host launch gaps, no allocator calls between
    the kernels) and replayed; falls back to eager launches if the capture fails."""
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        if eager or os.environ.get("FSR_BENCH_EAGER") == "1":    # PMC passes / library calls that cannot be captured
            raise RuntimeError("eager requested")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s.record()
        for _ in range(3):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / (3 * iters)
    except Exception:  # noqa: BLE001
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--miopen", action="store_true")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--filter", default="")
    ap.add_argument("--mask", action="store_true", help="data gradients with the fused activation-gradient mask (the saved forward input)")
    args = ap.parse_args()
    cd = ops.Compute(args.dtype)
    dev = torch.device("cuda:0")
    n = args.batch
    only = args.only.split(",")
    print("%-26s %9s | %8s %7s | %8s %7s | %8s %7s | %s" % ("layer", "GFLOP", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s", "miopen fwd/dgrad/wgrad TF/s"))
    for name, cin, cout, h, w, stride, ps in SHAPES:
        if args.filter and args.filter not in name:
            continue
        cin_pad = cd.pad(cin)
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        x = ops.to_storage(cd, torch.randn(n, h, w, cin_pad, device=dev))
        wt = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        gflop = 2.0 * n * oh * ow * cout * cin * 9 / 1e9
        wpk = ops.packed_filter(cd, wt, L.PACK_FWD_PS if ps else L.PACK_FWD, cin_pad)
        res = []
        cout_pad = cd.pad(cout)
        dy = ops.to_storage(cd, torch.randn((n, 2 * oh, 2 * ow, cout // 4) if ps else (n, oh, ow, cout_pad), device=dev))
        if "fwd" in only:
            t = timeit(lambda: ops.conv3x3_raw(cd, x, wpk, cout, stride=stride, pixel_shuffle=ps, out_f32=(cout == 3),
                                               want_stats=(not ps and cout != 3 and "VGG" not in name and "first" not in name)))
            res += [t * 1e3, gflop / t]
        else:
            res += [0, 0]
        if "dgrad" in only:
            wpd = ops.packed_filter(cd, wt, L.PACK_DGRAD_PS if ps else L.PACK_DGRAD, cout_pad)
            mask = x if (args.mask and cin > 3) else None      # the fused activation gradient: mask = the saved forward input
            t = timeit(lambda: ops.conv3x3_raw(cd, dy, wpd, cin_pad if cin > 3 else 3, mode=L.CONV_DGRAD, out_hw=(h, w),
                                               stride=stride, in_pixel_shuffled=ps, out_f32=(cin == 3), dact_mask=mask, dact_slope=0.2))
            res += [t * 1e3, gflop / t]
        else:
            res += [0, 0]
        if "wgrad" in only:
            t = timeit(lambda: ops.conv3x3_wgrad_raw(cd, x, dy, cout, cin, stride, dy_pixel_shuffled=ps))
            res += [t * 1e3, gflop / t]
        else:
            res += [0, 0]
        extra = ""
        if args.miopen and cin > 3 and cout > 3:
            xt = torch.randn(n, cin, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wtt = wt.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = torch.nn.functional.conv2d(xt, wtt, None, stride, 1)
            g = torch.randn_like(y)
            tf = timeit(lambda: torch.nn.functional.conv2d(xt, wtt, None, stride, 1), 20, eager=True)
            td = timeit(lambda: torch.autograd.grad(y, xt, g, retain_graph=True), 20, eager=True)
            tw = timeit(lambda: torch.autograd.grad(y, wtt, g, retain_graph=True), 20, eager=True)
            extra = "%.0f / %.0f / %.0f" % (gflop / tf, gflop / td, gflop / tw)
        print("%-26s %9.1f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f | %s" % (name, gflop, *res, extra), flush=True)


if __name__ == "__main__":
    main()
