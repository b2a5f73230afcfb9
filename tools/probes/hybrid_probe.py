import os, sys, types, time, warnings, importlib
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch
torch.set_num_threads(16)
pkg = importlib.import_module("fast-srgan_amd")
from oracle import srgan_cpu as O
ns = types.SimpleNamespace
dev = "cuda:0"
def cfg(B, cdt, **kw):
    return ns(experiment=ns(name="h", seed=1234), generator=ns(n_filters=64, n_layers=8), discriminator=ns(n_filters=64, n_layers=7),
              training=ns(compiled=False, device=dev, log_iter=1, checkpoint_iter=10**9, generator_lr=1e-4, discriminator_lr=1e-4, batch_size=B, compute_dtype=cdt, **kw))
# parity at cfg1
torch.manual_seed(6)
res = {}
for name, cdt, vdt, kw in (("x3", "x3", "x3", {}), ("x3+f16vgg", "x3", "f16", dict(loss_scale=1048576.0)), ("x3+bf16vgg", "x3", "bf16", {})):
    torch.manual_seed(6)
    T = pkg.Trainer(cfg(4, cdt, **kw), perceptual_network=pkg.VGG19(compute_dtype=vdt, seed=1234))
    if name == "x3":
        g0 = {k: v.detach().cpu().clone() for k, v in T.generator.state_dict().items()}
        d0 = {k: v.detach().cpu().clone() for k, v in T.discriminator.state_dict().items()}
        v_sd = O.vgg_standin_state_dict(1234, 1)
        lr, hr = torch.rand(4, 3, 96, 96) * 2 - 1, torch.rand(4, 3, 384, 384) * 2 - 1
        noise = [torch.rand(4, 1, 24, 24) for _ in range(3)]
        ref = {}
        want = O.train_step({k: v.clone() for k, v in g0.items()}, {k: v.clone() for k, v in d0.items()}, v_sd, lr, hr, noise, {}, {}, grads_out=ref)
    else:
        T.generator.load_state_dict(g0); T.discriminator.load_state_dict(d0)
    got = T.train_step(lr.to(dev), hr.to(dev), [n.to(dev) for n in noise])
    torch.cuda.synchronize()
    inv = 1.0 / T.loss_scale
    errs = {k: abs(float(got[k]) - float(want[k])) / abs(float(want[k])) for k in want}
    num = den = 0.0
    for k, p in T.generator.named_parameters():
        r = ref["g." + k].double(); g = p.grad.detach().double().cpu() * inv
        num += float((g - r).norm()) ** 2; den += float(r.norm()) ** 2
    print(name, {k: "%.2e" % v for k, v in errs.items()}, "G network grad rel-L2 vs fp32 oracle %.4f" % ((num / den) ** 0.5), flush=True)
    del T
# speed at b32
for name, cdt, vdt, kw in (("x3", "x3", "x3", {}), ("x3+f16vgg", "x3", "f16", dict(loss_scale=1048576.0)), ("f16", "f16", "f16", {})):
    torch.manual_seed(1)
    T = pkg.Trainer(cfg(32, cdt, **kw), perceptual_network=pkg.VGG19(compute_dtype=vdt, seed=1234))
    lr, hr = torch.rand(32, 3, 96, 96, device=dev) * 2 - 1, torch.rand(32, 3, 384, 384, device=dev) * 2 - 1
    T.capture_train_step(lr, hr)
    for _ in range(3): T.graphed_train_step(lr, hr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): T.graphed_train_step(lr, hr)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s: %.2f ms/step %.1f images/s" % (name, dt * 1e3, 32 / dt), flush=True)
    del T; torch.cuda.empty_cache()
