"""TEST INFRASTRUCTURE: repeats the GPU half of test_full_size_modules_vs_oracle_gpu[f32] against ONE oracle evaluation and
prints the spread of the errors (InstanceNorm statistics meet through float atomics: run-to-run differences of 1e-7
flip LeakyReLU signs now and then)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
from backend import select, relerr, relerr2
from oracle import srgan_cpu as O
pkg = importlib.import_module("fast-srgan_amd")
ns = types.SimpleNamespace
dev = select("hip")
torch.manual_seed(3)
G = pkg.Generator(ns(n_filters=64, n_layers=2), compute_dtype="f32")
D = pkg.Discriminator(ns(n_filters=64, n_layers=7), compute_dtype="f32")
gsd = {k: v.clone() for k, v in G.state_dict().items()}
dsd = {k: v.clone() for k, v in D.state_dict().items()}
G.to(dev), D.to(dev)
x = torch.rand(2, 3, 24, 40) * 2 - 1
r = torch.randn(2, 1, 6, 10)
gp = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
dp = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
sr_ref = O.generator_forward(gp, x)
lg_ref = O.discriminator_forward(dp, sr_ref)
r = torch.randn(lg_ref.shape)
grads = torch.autograd.grad((lg_ref * r).sum(), list(gp.values()) + list(dp.values()))
ref = dict(zip([("g", k) for k in gp] + [("d", k) for k in dp], grads))
worst = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for m in (G, D):
        for p in m.parameters():
            p.grad = None
    sr = G(x.to(dev))
    logits = D(sr)
    (logits * r.to(dev)).sum().backward()
    e_out, e_lg = relerr(sr, sr_ref), relerr(logits, lg_ref)
    e_g, who = 0.0, None
    for tag, mod in (("g", G), ("d", D)):
        for k, p in mod.named_parameters():
            e = relerr2(p.grad, ref[(tag, k)])
            if e > e_g:
                e_g, who = e, (tag, k)
    worst.append((e_out, e_lg, e_g, who))
    print("run %2d: out %.2e logits %.2e worst grad %.2e %s" % (it, e_out, e_lg, e_g, who), flush=True)
print("max over runs: out %.2e logits %.2e grad %.2e" % (max(w[0] for w in worst), max(w[1] for w in worst), max(w[2] for w in worst)))
