"""Shows that the fp32 ORACLE gradients of G+D move by ~1% (max-norm) under a 1e-6 relative weight perturbation:
activation-sign flips make element-wise gradient comparisons ill-conditioned (see tests/backend.py relerr2)."""
import sys, types, torch, importlib
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))  # repo root
from oracle import srgan_cpu as O
pkg = importlib.import_module("fast-srgan_amd")
ns=types.SimpleNamespace
torch.manual_seed(3)
G = pkg.Generator(ns(n_filters=64, n_layers=2), compute_dtype="f32")
D = pkg.Discriminator(ns(n_filters=64, n_layers=7), compute_dtype="f32")
gsd = {k: v.clone() for k, v in G.state_dict().items()}
dsd = {k: v.clone() for k, v in D.state_dict().items()}
x = torch.rand(2, 3, 24, 40) * 2 - 1
r = torch.randn(2,1,6,10)
def grads(x, eps=0.0):
    gp = {k: (v*(1+eps*torch.randn_like(v))).clone().requires_grad_(True) for k, v in gsd.items()}
    dp = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    lg = O.discriminator_forward(dp, O.generator_forward(gp, x))
    g = torch.autograd.grad((lg*r).sum(), list(gp.values())+list(dp.values()))
    return dict(zip([("g",k) for k in gp]+[("d",k) for k in dp], g))
a = grads(x); b = grads(x, 1e-6)
for k in a:
    print(k, "%.5f" % float((a[k]-b[k]).abs().max()/a[k].abs().max()))
