"""Timing of the stride-2 data gradient of the discriminator's first strided block (64->64, 384->192)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fast_srgan_amd as pkg
from fast_srgan_amd import ops, _lib as L

dev = torch.device("cuda:0")
cd = ops.Compute("bf16")
for n in (32, 64):
    for c in (64, 128):
        h = 384 if c == 64 else 192
        dz = torch.randn(n, h // 2, h // 2, c, device=dev).to(torch.bfloat16)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        mask = torch.randn(n, h, h, c, device=dev).to(torch.bfloat16)
        wpk = ops.packed_filter(cd, w, L.PACK_DGRAD, c)
        for use_mask in (False, True):
            ts = []
            for it in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dx, _, _ = ops.conv3x3_raw(cd, dz, wpk, c, mode=L.CONV_DGRAD, out_hw=(h, h), stride=2,
                                           dact_mask=mask if use_mask else None, dact_slope=0.2)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            gb = (dz.numel() + dx.numel() * (2 if use_mask else 1)) * 2 / 1e9
            print("n=%d c=%d mask=%s: %.1f us  (%.2f GB -> %.2f TB/s)" % (n, c, use_mask, min(ts) * 1e3, gb, gb / min(ts) / 1e-3 / 1e3), flush=True)
        del dz, mask, dx
