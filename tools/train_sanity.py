"""Runs N eager + graphed training iterations on one fixed synthetic batch and checks that every loss stays finite and
the generator output changes (a cheap guard against NaNs / stale filter images / lost gradients on the GPU)."""
import importlib, os, sys, types, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
ns = types.SimpleNamespace
B = 8
cfg = ns(experiment=ns(name="sanity", seed=1), generator=ns(n_filters=64, n_layers=8), discriminator=ns(n_filters=64, n_layers=7),
         training=ns(compiled=False, device="cuda:0", log_iter=10 ** 9, checkpoint_iter=10 ** 9, generator_lr=1e-4,
                     discriminator_lr=1e-4, batch_size=B, compute_dtype="bf16"))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    T = pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype="bf16", seed=1234))   # kaiming-normal stand-in (no ImageNet weights offline)
torch.manual_seed(0)
hr = torch.rand(B, 3, 384, 384, device="cuda:0") * 2 - 1
lr = torch.nn.functional.avg_pool2d(hr, 4)
with torch.no_grad():
    y0 = T.generator(lr).float().clone()
hist = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    out = T.train_step(lr, hr)
    vals = {k: float(v) for k, v in out.items()}
    assert all(v == v and abs(v) < 1e4 for v in vals.values()), (it, vals)
    hist.append(vals)
T.capture_train_step(lr, hr)
for it in range(40):
    out = T.graphed_train_step(lr, hr)
vals = {k: float(v) for k, v in out.items()}
assert all(v == v and abs(v) < 1e4 for v in vals.values()), vals
with torch.no_grad():
    y1 = T.generator(lr).float()
l1_0 = float((y0 - hr).abs().mean())
l1_1 = float((y1 - hr).abs().mean())
print("first", {k: round(v, 4) for k, v in hist[0].items()})
print("last eager", {k: round(v, 4) for k, v in hist[-1].items()})
print("last graphed", {k: round(v, 4) for k, v in vals.items()})
print("mean |G(lr) - hr|: %.4f -> %.4f (the GAN phase has no pixel loss; informational)" % (l1_0, l1_1))
assert vals["content_loss"] < hist[0]["content_loss"], "the perceptual loss did not decrease"
assert float((y1 - y0).abs().max()) > 1e-3, "the generator did not change"
print("train sanity ok")
