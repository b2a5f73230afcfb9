import importlib, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("fast-srgan_amd._lib")
ops = importlib.import_module("fast-srgan_amd.ops")
pkg = importlib.import_module("fast-srgan_amd")
from oracle import srgan_cpu as O
dev = torch.device("cuda:0")
torch.manual_seed(0)
print("== act_bwd dprelu / dbias")
for cdn in ("f32", "bf16"):
    cd = ops.Compute(cdn)
    for (n, h, w, c, ps) in [(2, 24, 40, 64, 0), (2, 48, 80, 64, 1), (1, 10, 14, 32, 1), (4, 96, 96, 64, 0)]:
        g = torch.randn(n, h, w, c).to(cd.torch_dtype)
        sv = torch.randn(n, h, w, c).to(cd.torch_dtype)
        a = torch.tensor([-0.25])
        ref_dp = (g.float() * sv.float().clamp(max=0)).sum()
        dzr = g.float() * torch.where(sv.float() > 0, torch.ones(()), a)
        for rep in range(3):
            dz = torch.empty_like(g, device=dev)
            dbias = torch.zeros(c * (4 if ps else 1), device=dev)
            dp = torch.zeros(1, device=dev)
            L.check(L.lib().fsr_act_bwd(cd.code, g.to(dev).data_ptr(), sv.to(dev).data_ptr(), L.ACT_PRELU, 0.0, a.to(dev).data_ptr(),
                                        dz.data_ptr(), dbias.data_ptr(), dp.data_ptr(), n, h, w, c, ps, ops._stream()))
            torch.cuda.synchronize()
            print(cdn, (n, h, w, c, ps), "dprelu", float(dp), "ref", float(ref_dp), "dz err", float((dz.float().cpu() - dzr).abs().max()),
                  "dbias sum", float(dbias.sum()), "ref", float(dzr.sum()))
print("== per-layer gradient errors, full-width G(2 blocks)+D")
ns = types.SimpleNamespace
for cdn in ("f32", "bf16"):
    torch.manual_seed(3)
    G = pkg.Generator(ns(n_filters=64, n_layers=2), compute_dtype=cdn)
    D = pkg.Discriminator(ns(n_filters=64, n_layers=7), compute_dtype=cdn)
    gsd = {k: v.clone() for k, v in G.state_dict().items()}
    dsd = {k: v.clone() for k, v in D.state_dict().items()}
    G.to(dev), D.to(dev)
    x = torch.rand(2, 3, 24, 40) * 2 - 1
    sr = G(x.to(dev))
    logits = D(sr)
    r = torch.randn(logits.shape)
    (logits * r.to(dev)).sum().backward()
    gp = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    dp = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    sr_ref = O.generator_forward(gp, x)
    lg_ref = O.discriminator_forward(dp, sr_ref)
    grads = torch.autograd.grad((lg_ref * r).sum(), list(gp.values()) + list(dp.values()))
    ref = dict(zip([("g", k) for k in gp] + [("d", k) for k in dp], grads))
    rel = lambda a, b: float((a.detach().float().cpu() - b).abs().max() / b.abs().max())
    print(cdn, "sr", rel(sr, sr_ref), "logits", rel(logits, lg_ref))
    for k, p in list(D.named_parameters())[::-1]:
        print(cdn, "d", k, "%.4f" % rel(p.grad, ref[("d", k)]))
    for k, p in list(G.named_parameters())[::-1]:
        print(cdn, "g", k, "%.4f" % rel(p.grad, ref[("g", k)]))
