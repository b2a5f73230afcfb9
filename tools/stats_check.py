import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fast_srgan_amd as pkg
from fast_srgan_amd import ops, _lib as L
dev = torch.device("cuda:0")
cd = ops.Compute("bf16")
for (n, hw) in ((32, 96), (32, 384)):
    x = torch.randn(n, hw, hw, 64, device=dev).to(torch.bfloat16)
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    wpk = ops.packed_filter(cd, w, L.PACK_FWD, 64)
    for st in (False, True):
        ts = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y, _, s = ops.conv3x3_raw(cd, x, wpk, 64, want_stats=st)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("n=%d %dx%d stats=%s: %.1f us" % (n, hw, hw, st, min(ts) * 1e3), flush=True)
