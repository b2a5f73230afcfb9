"""InstanceNorm backward (reduce + apply) on a discriminator-sized tensor: whole batch vs image chunks that fit the
256 MB Infinity Cache (the apply pass re-reads what the reduce pass just read)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fast_srgan_amd as pkg
from fast_srgan_amd import ops, _lib as L

dev = torch.device("cuda:0")
cd = ops.Compute("bf16")
lib = L.lib()
for (n, hw, c) in ((64, 192 * 192, 128), (64, 192 * 192, 64), (64, 96 * 96, 256), (32, 96 * 96, 64)):
    x = torch.randn(n, hw, c, device=dev).to(torch.bfloat16)
    g = torch.randn(n, hw, c, device=dev).to(torch.bfloat16)
    dx = torch.empty_like(x)
    stats = torch.rand(n, c, 2, device=dev) * hw
    stats[..., 1] += stats[..., 0] ** 2 / hw
    st = torch.cuda.current_stream().cuda_stream
    per_img = hw * c * 2
    for chunk in (n, 16, 8, 4, 2):
        if chunk > n:
            continue
        ts = []
        for it in range(5):
            sums = torch.zeros(n, c, 2, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i0 in range(0, n, chunk):
                m = min(chunk, n - i0)
                gp, xp, dp = g.data_ptr() + i0 * per_img, x.data_ptr() + i0 * per_img, dx.data_ptr() + i0 * per_img
                sp, qp = stats.data_ptr() + i0 * c * 8, sums.data_ptr() + i0 * c * 8
                L.check(lib.fsr_instnorm_act_bwd_reduce(cd.code, gp, xp, sp, L.ACT_LEAKY, 0.01, None, qp, None, ops._workspace(lib.fsr_instnorm_act_bwd_scratch(m, hw, c), dev).data_ptr(), m, hw, c, st))
                L.check(lib.fsr_instnorm_act_bwd_apply(cd.code, gp, xp, sp, qp, L.ACT_LEAKY, 0.01, None, dp, m, hw, c, st))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("n=%d hw=%d c=%d (%.0f MB/tensor) chunk %2d: %.1f us" % (n, hw, c, n * per_img / 1e6, chunk, min(ts) * 1e3), flush=True)
