"""Generates the golden fixtures of tests/golden/ by running the REFERENCE's own modules.

Run in the authoring container only (needs /root/reference; the GPU box has none):
    python tests/golden/make_golden.py
The reference is imported, never copied.  Third-party packages that are absent offline are
stubbed in sys.modules *only as far as the import statements need*:
  torchvision.models.vgg   -> vgg19() returns the structural features stack with seeded
                              kaiming-normal weights (oracle.srgan_cpu.vgg_standin_state_dict)
  torchvision.transforms.v2-> Resize = F.interpolate(bicubic, antialias=True) (the torch kernel the
                              real class forwards to for float tensors)
  torch.utils.tensorboard  -> SummaryWriter that records add_scalar calls
  torchmetrics.image       -> no-op metric objects
  hydra / omegaconf        -> not needed (configs are SimpleNamespaces)
Fixtures are small .npz files; tests/test_oracle.py replays them against oracle/srgan_cpu.py and
tests/test_parity_gpu.py against the HIP path.
"""
import os
import random
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch
import torch.nn.functional as F

from oracle import srgan_cpu as O

VGG_WIDTH_DIV = 4  # golden VGG stand-in: 16..128 channels instead of 64..512 (fixture size)


def install_stubs(vgg_seed=1234, vgg_width_div=VGG_WIDTH_DIV):
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvv = types.ModuleType("torchvision.models.vgg")

    class VGG19_Weights:
        IMAGENET1K_V1 = "IMAGENET1K_V1"

    def vgg19(weights=None):
        sd = O.vgg_standin_state_dict(vgg_seed, vgg_width_div)
        layers, cin, k = [], 3, 0
        for v in O.VGG_CFG + [512, "M"]:  # torchvision cfg "E": 16 convs; [:34] drops the last
            if v == "M":
                layers.append(torch.nn.MaxPool2d(2, 2))
                continue
            cout = v // vgg_width_div
            conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
            if k < len(O.VGG_CONV_IDX):
                with torch.no_grad():
                    conv.weight.copy_(sd[f"vgg.{O.VGG_CONV_IDX[k]}.weight"])
                    conv.bias.zero_()
            layers += [conv, torch.nn.ReLU(inplace=True)]
            cin = cout
            k += 1
        return types.SimpleNamespace(features=torch.nn.Sequential(*layers))

    tvv.VGG19_Weights, tvv.vgg19 = VGG19_Weights, vgg19
    tvt = types.ModuleType("torchvision.transforms")
    tv2 = types.ModuleType("torchvision.transforms.v2")

    class InterpolationMode:
        BICUBIC = "bicubic"

    class Resize:
        def __init__(self, size, antialias=True, interpolation="bicubic"):
            self.size, self.antialias, self.mode = size, antialias, interpolation

        def __call__(self, x):
            return F.interpolate(x[None], self.size, mode=self.mode, antialias=self.antialias, align_corners=False)[0]

    tv2.Resize, tv2.InterpolationMode = Resize, InterpolationMode
    tvt.v2 = tv2
    tv.models, tv.transforms, tvm.vgg = tvm, tvt, tvv

    tb = types.ModuleType("torch.utils.tensorboard")
    tbw = types.ModuleType("torch.utils.tensorboard.writer")

    class SummaryWriter:
        log = []

        def __init__(self, *a, **k):
            pass

        def add_scalar(self, tag, value, global_step=None):
            SummaryWriter.log.append((tag, float(value), global_step))

        def add_images(self, *a, **k):
            pass

        def flush(self):
            pass

    tbw.SummaryWriter = SummaryWriter
    tm = types.ModuleType("torchmetrics")
    tmi = types.ModuleType("torchmetrics.image")

    class _Metric:
        def __init__(self, *a, **k):
            pass

        def to(self, *_):
            return self

        def reset(self):
            pass

        def update(self, *a):
            pass

        def compute(self):
            return torch.zeros(1)

    tmi.PeakSignalNoiseRatio = tmi.StructuralSimilarityIndexMeasure = _Metric
    for name, mod in {"torchvision": tv, "torchvision.models": tvm, "torchvision.models.vgg": tvv,
                      "torchvision.transforms": tvt, "torchvision.transforms.v2": tv2,
                      "torch.utils.tensorboard": tb, "torch.utils.tensorboard.writer": tbw,
                      "torchmetrics": tm, "torchmetrics.image": tmi}.items():
        sys.modules[name] = mod
    return SummaryWriter


def ns(**k):
    return types.SimpleNamespace(**k)


def sd_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (authoring container only)")
    writer_cls = install_stubs()
    sys.path.insert(0, REF)
    import dataloader as ref_data
    import model as ref_model
    import trainer as ref_trainer

    # ---- 1. real-weights KAT for Generator.forward (models/model.pt)
    w = torch.load(os.path.join(REF, "models", "model.pt"), map_location="cpu")
    w = {k.replace("_orig_mod.", ""): v for k, v in w.items()}          # inference.py:31-33
    G = ref_model.Generator(ns(n_filters=64, n_layers=8))
    G.load_state_dict(w)
    G.eval()
    torch.manual_seed(0)
    x = torch.rand(4, 3, 96, 96) * 2 - 1
    with torch.no_grad():
        y = G(x)
        xs = x[:2, :, :24, :32].contiguous()
        ys = G(xs)
    np.savez(os.path.join(HERE, "g_model_pt.npz"), x_small=xs.numpy(), y_small=ys.numpy(),
             y_sum=np.float64(y.double().sum().item()), y_mean=np.float64(y.double().mean().item()),
             y_absmean=np.float64(y.double().abs().mean().item()), y_row0=y[0, 0, 0, :4].numpy(),
             y_tail=y[3, 2, 383, 380:].numpy(), y_strided=y[:, :, ::16, ::16].contiguous().numpy(),
             **sd_np(w, "sd."))

    # ---- 2. small Generator: forward + all gradients
    torch.manual_seed(11)
    Gs = ref_model.Generator(ns(n_filters=16, n_layers=2))
    with torch.no_grad():  # non-trivial PReLU slopes, one of them negative like the shipped checkpoint
        Gs.neck[1].weight.fill_(0.2)
        Gs.stem[0].relu1.weight.fill_(-0.28)
        Gs.upsampling[1].relu.weight.fill_(0.1)
    x = (torch.rand(2, 3, 12, 20) * 2 - 1).requires_grad_(True)
    r = torch.randn(2, 3, 48, 80)
    y = Gs(x)
    (y * r).sum().backward()
    np.savez(os.path.join(HERE, "g_small.npz"), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), **sd_np(Gs.state_dict(), "sd."),
             **{"grad." + k: p.grad.numpy() for k, p in Gs.named_parameters()})

    # ---- 3. small Discriminator: forward + gradients (parameters and input)
    torch.manual_seed(12)
    Ds = ref_model.Discriminator(ns(n_filters=16, n_layers=7))
    x = (torch.rand(2, 3, 64, 48) * 2 - 1).requires_grad_(True)
    y = Ds(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    np.savez(os.path.join(HERE, "d_small.npz"), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), **sd_np(Ds.state_dict(), "sd."),
             **{"grad." + k: p.grad.numpy() for k, p in Ds.named_parameters()})

    # ---- 4. VGG19 wrapper (model.py:5-23) on the width/8 stand-in: forward + input gradient
    V = ref_model.VGG19()
    x = (torch.rand(2, 3, 32, 48) * 2 - 1).requires_grad_(True)
    y = V(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    np.savez(os.path.join(HERE, "vgg_small.npz"), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), width_div=VGG_WIDTH_DIV, seed=1234,
             w0_sum=np.float64(V.vgg[0].weight.double().sum().item()))

    # ---- 4b. tiny variants of 2-4 for the CPU suite (the host emulator runs them in seconds)
    torch.manual_seed(13)
    Gt = ref_model.Generator(ns(n_filters=16, n_layers=1))
    with torch.no_grad():
        Gt.stem[0].relu1.weight.fill_(-0.28)
    x = (torch.rand(1, 3, 5, 6) * 2 - 1).requires_grad_(True)
    y = Gt(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    np.savez(os.path.join(HERE, "g_tiny.npz"), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), **sd_np(Gt.state_dict(), "sd."),
             **{"grad." + k: p.grad.numpy() for k, p in Gt.named_parameters()})
    torch.manual_seed(14)
    Dt = ref_model.Discriminator(ns(n_filters=16, n_layers=7))
    x = (torch.rand(1, 3, 32, 16) * 2 - 1).requires_grad_(True)
    y = Dt(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    np.savez(os.path.join(HERE, "d_tiny.npz"), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), **sd_np(Dt.state_dict(), "sd."),
             **{"grad." + k: p.grad.numpy() for k, p in Dt.named_parameters()})
    x = (torch.rand(1, 3, 16, 16) * 2 - 1).requires_grad_(True)
    y = V(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    np.savez(os.path.join(HERE, "vgg_tiny.npz"), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), width_div=VGG_WIDTH_DIV, seed=1234,
             w0_sum=np.float64(V.vgg[0].weight.double().sum().item()))

    # ---- 5. two iterations of the reference's own Trainer.train loop (trainer.py:158-233)
    cfg = ns(experiment=ns(name="golden", seed=1234),
             generator=ns(n_filters=16, n_layers=1), discriminator=ns(n_filters=16, n_layers=7),
             training=ns(compiled=False, device="cpu", log_iter=1, checkpoint_iter=10 ** 9,
                         generator_lr=1e-4, discriminator_lr=1e-4, batch_size=2))
    cwd = os.getcwd()
    os.chdir("/tmp")
    torch.manual_seed(21)
    T = ref_trainer.Trainer(cfg)
    g0 = {k: v.clone() for k, v in T.generator.state_dict().items()}
    d0 = {k: v.clone() for k, v in T.discriminator.state_dict().items()}
    batches = [((torch.rand(2, 3, 8, 8) * 2 - 1), (torch.rand(2, 3, 32, 32) * 2 - 1)) for _ in range(2)]
    noise = []
    real_rand_like = torch.rand_like

    def recording_rand_like(t, *a, **k):
        n = real_rand_like(t, *a, **k)
        noise.append(n.clone())
        return n

    torch.rand_like = recording_rand_like
    writer_cls.log.clear()
    try:
        T.train(batches, [])           # val loader empty: metrics stub does nothing
    finally:
        torch.rand_like = real_rand_like
        os.chdir(cwd)
    losses = np.array([v for (t, v, _) in writer_cls.log if t.startswith("Loss/")], dtype=np.float64).reshape(2, 4)
    out = {"losses": losses, "vgg_width_div": VGG_WIDTH_DIV, "vgg_seed": 1234}
    for i, (l, h) in enumerate(batches):
        out[f"lr{i}"], out[f"hr{i}"] = l.numpy(), h.numpy()
    for i, n in enumerate(noise):
        out[f"noise{i}"] = n.numpy()
    out.update(sd_np(g0, "g0."))
    out.update(sd_np(d0, "d0."))
    out.update(sd_np(T.generator.state_dict(), "g2."))
    out.update(sd_np(T.discriminator.state_dict(), "d2."))
    np.savez(os.path.join(HERE, "train_steps.npz"), **out)

    # ---- 6. NumpyImagesDataset.__getitem__ (dataloader.py:24-38) with the Resize stub
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(3, 75, 90), dtype=np.uint8)
    path = "/tmp/_golden_img.npy"
    np.save(path, img)
    ds = ref_data.NumpyImagesDataset([path], lr_image_size=12, scale_factor=4)
    random.seed(77)
    items = [ds[0] for _ in range(3)]
    np.savez(os.path.join(HERE, "dataset.npz"), image=img, seed=77,
             **{f"lr{i}": it[0].numpy() for i, it in enumerate(items)},
             **{f"hr{i}": it[1].numpy() for i, it in enumerate(items)})
    os.remove(path)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
