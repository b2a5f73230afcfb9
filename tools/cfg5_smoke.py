"""BASELINE configs[4] shape check: 12 residual blocks, 8x upscale (three pixel-shuffle stages), 128 -> 1024 crops,
full GAN iteration in bf16 (the config names fp16 MFMA; this framework's 16-bit mode is bf16).  One GPU, small batch."""
import importlib, os, sys, time, types, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
ns = types.SimpleNamespace
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = ns(experiment=ns(name="cfg5", seed=1234), generator=ns(n_filters=64, n_layers=12, n_upsample=3),
         discriminator=ns(n_filters=64, n_layers=7),
         training=ns(compiled=False, device="cuda:0", log_iter=10 ** 9, checkpoint_iter=10 ** 9, generator_lr=1e-4,
                     discriminator_lr=1e-4, batch_size=B, compute_dtype="bf16"))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    T = pkg.Trainer(cfg)
lr = torch.rand(B, 3, 128, 128, device="cuda:0") * 2 - 1
hr = torch.rand(B, 3, 1024, 1024, device="cuda:0") * 2 - 1
for _ in range(2):
    out = T.train_step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    out = T.train_step(lr, hr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print({k: round(float(v), 5) for k, v in out.items()})
print("cfg5 batch %d: %.1f ms/iteration, %.2f images/s, peak memory %.1f GB" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2 ** 30))
