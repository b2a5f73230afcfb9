"""Which branch of the generator step carries the bf16 gradient-norm discrepancy?  HIP bf16 vs fp32 oracle, dL/d(sr) of the
content branch (VGG + SmoothL1) and of the adversarial branch (D + BCE) separately, cfg1 size."""
import importlib, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import srgan_cpu as O
pkg = importlib.import_module("fast-srgan_amd")
ops = importlib.import_module("fast-srgan_amd.ops")
ns = types.SimpleNamespace
dev = "cuda:0"
torch.manual_seed(6)
B = 2
sr = (torch.rand(B, 3, 384, 384) * 2 - 1) * 0.8
hr = torch.rand(B, 3, 384, 384) * 2 - 1
v_sd = O.vgg_standin_state_dict(1234, 1)
def l2(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def cos(a, b): return float((a.double() * b.double()).sum() / a.double().norm() / b.double().norm())
# oracle, content branch
for q, name in ((None, "fp32"), (O.Q_BF16, "bf16q")):
    x = sr.clone().requires_grad_(True)
    loss = O.smooth_l1(O.vgg_forward(v_sd, x, q), O.vgg_forward(v_sd, hr, q).detach())
    loss.backward()
    if q is None: ref_loss, ref_dx = float(loss), x.grad.clone()
    else: q_loss, q_dx = float(loss), x.grad.clone()
print("oracle content loss fp32 %.6g bf16q %.6g ; dx bf16q vs fp32: l2 %.4f cos %.5f normratio %.4f" % (ref_loss, q_loss, l2(q_dx, ref_dx), cos(q_dx, ref_dx), float(q_dx.norm() / ref_dx.norm())))
for cdn in ("f32", "bf16"):
    V = pkg.VGG19(compute_dtype=cdn, seed=1234).to(dev)
    ops.zero_pool_reset(torch.device(dev))
    x = sr.to(dev).requires_grad_(True)
    with torch.no_grad():
        tgt = V.features_nhwc(hr.to(dev))
    loss = ops.smooth_l1(V.features_nhwc(x), tgt)
    loss.backward()
    torch.cuda.synchronize()
    g = x.grad.cpu()
    print("HIP %s content loss %.6g ; dx vs fp32 oracle: l2 %.4f cos %.5f normratio %.4f ; vs bf16q: l2 %.4f cos %.5f normratio %.4f" % (
        cdn, float(loss), l2(g, ref_dx), cos(g, ref_dx), float(g.norm() / ref_dx.norm()), l2(g, q_dx), cos(g, q_dx), float(g.norm() / q_dx.norm())))
    ops.zero_pool_end(torch.device(dev))
