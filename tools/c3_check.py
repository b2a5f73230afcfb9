"""GPU check + timing of the first-layer kernels against the padded-tensor path (no oracle needed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import fast_srgan_amd as pkg
from fast_srgan_amd import ops, _lib as L

dev = torch.device("cuda:0")


def run(cdn, cout, act, n, h, w, nhwc, use_c3):
    cd = ops.Compute(cdn)
    torch.manual_seed(11)
    img = torch.rand(n, 3, h, w) * 2 - 1
    if nhwc:
        img = img.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    wt = (torch.randn(cout, 3, 3, 3) * 0.2).to(cd.torch_dtype).float()
    b = torch.randn(cout) * 0.1
    cfg = ops.ConvCfg(cd, act=act, slope=0.0, image_in=True, in_scale=(0.9, 1.1, 1.3), in_shift=(0.1, -0.2, 0.3))
    xi = img.to(dev).requires_grad_(True)
    wd = wt.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    ops.USE_C3_KERNELS = use_c3
    y, _ = ops.conv3x3(xi, wd, bd, None, cfg)
    g = torch.randn(n, h, w, cout).to(cd.torch_dtype)
    y.backward(g.to(dev))
    ops.USE_C3_KERNELS = True
    # float64 reference
    xr = img.double().requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    xn = xr * torch.tensor((0.9, 1.1, 1.3)).double().view(1, 3, 1, 1) + torch.tensor((0.1, -0.2, 0.3)).double().view(1, 3, 1, 1)
    xn = xn + (xn.detach().float().to(cd.torch_dtype).double() - xn.detach())
    z = F.conv2d(xn, wr, b.double(), 1, 1)
    yr = F.relu(z) if act == L.ACT_RELU else z
    yr.backward(g.double().permute(0, 3, 1, 2))
    return y.float().cpu(), wd.grad.cpu(), yr.float(), wr.grad.float()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


if "--check" in sys.argv:
    for cdn in ("f32", "bf16"):
        for cout, act, nhwc in ((128, L.ACT_RELU, True), (128, L.ACT_RELU, False), (64, L.ACT_NONE, True)):
            y1, dw1, yr, dwr = run(cdn, cout, act, 6, 163, 210, nhwc, True)
            y0, dw0, _, _ = run(cdn, cout, act, 6, 163, 210, nhwc, False)
            print(cdn, cout, act, nhwc, "y c3/pad vs f64: %.2e %.2e" % (rel(y1.permute(0, 3, 1, 2), yr), rel(y0.permute(0, 3, 1, 2), yr)),
                  "dw c3/pad vs f64: %.2e %.2e" % (rel(dw1, dwr), rel(dw0, dwr)), flush=True)

if "--time" in sys.argv:
    cd = ops.Compute("bf16")
    for n in (32, 64):
        img = (torch.rand(n, 3, 384, 384, device=dev) * 2 - 1)
        wt = torch.randn(64, 3, 3, 3, device=dev) * 0.2
        b = torch.zeros(64, device=dev)
        cfg = ops.ConvCfg(cd, act=L.ACT_LEAKY, slope=0.2, image_in=True)
        for use in (True, False):
            ops.USE_C3_KERNELS = use
            xi = img.clone().requires_grad_(True)
            wd = wt.clone().requires_grad_(True)
            ts = []
            for it in range(6):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                y, _ = ops.conv3x3(xi, wd, b, None, cfg)
                e[1].record()
                y.backward(y)
                e[2].record()
                torch.cuda.synchronize()
                ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
            print("n=%d c3=%s fwd %.3f ms  bwd(act+dgrad+wgrad) %.3f ms" % (n, use, min(t[0] for t in ts), min(t[1] for t in ts)), flush=True)
    ops.USE_C3_KERNELS = True
