"""TEST INFRASTRUCTURE: CPU fp32 restatement of the Fast-SRGAN hot path (the parity oracle).

Every function cites the reference lines it follows (paths relative to /root/reference).  All
tensors are NCHW float32 on the CPU, exactly as the reference sees them; parameters are passed as
a state_dict with the reference's key names so that one dict drives both the oracle and the HIP
modules under test.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512]  # the 15 convs of [:34]
# torchvision vgg19.features indices of the convolutions kept by features[:34] (model.py:8)
VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32]
VGG_POOL_AFTER = {2, 7, 16, 25}  # a MaxPool2d(2,2) follows the ReLU of these convs
VGG_MEAN = (0.485, 0.456, 0.406)  # model.py:13
VGG_STD = (0.229, 0.224, 0.225)   # model.py:17


# ---------------------------------------------------------------------------------- bf16 storage model
# The benched mode of the HIP path keeps every activation (and activation gradient) tensor in bf16 between kernels and
# feeds bf16 operands to the matrix cores; accumulation, statistics, parameters and losses stay fp32.  `Q_BF16` restates the
# reference with exactly those roundings (and nothing else changed), so that the bf16 kernels can be held to TIGHT bounds:
# against the plain fp32 oracle a bf16 run differs by ~1e-2 per activation, which flips a percent of the ReLU / LeakyReLU /
# max-pool decisions per layer and moves whole-network gradients by tens of per cent (measured) --
# a property of bf16 arithmetic on this network, not of any kernel.  q = None everywhere is the reference itself.
class _StoreBF16(torch.autograd.Function):
    """A tensor the kernels store in bf16: the value is rounded in the forward pass, its gradient in the backward pass."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class Q_BF16:
    """store(x): activation kept in bf16 (gradient too); operand(w): a float tensor rounded on its way into the MFMA (packed
    filters, the image read by the first layers) -- value only, its gradient stays fp32."""

    @staticmethod
    def store(x):
        return _StoreBF16.apply(x)

    @staticmethod
    def operand(w):
        return w + (w.detach().bfloat16().float() - w.detach())


def _st(q, x):
    return x if q is None else q.store(x)


def _op(q, w):
    return w if q is None else q.operand(w)


def instance_norm_stored(u, q, eps=1e-5):
    """InstanceNorm2d as the kernels evaluate it in the bf16 mode: statistics from the fp32 accumulators `u` of the
    producing convolution (its epilogue), normalisation of the STORED (rounded) tensor."""
    if q is None:
        return instance_norm(u, eps)
    mean = u.mean(dim=(2, 3), keepdim=True)
    var = ((u - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    return (q.store(u) - mean) / torch.sqrt(var + eps)


def instance_norm(x, eps=1e-5):
    """torch.nn.InstanceNorm2d defaults (model.py:55,65,94,132): biased variance over H*W per
    (n, c), eps 1e-5, no affine, no running statistics."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def prelu(x, a):
    """torch.nn.PReLU() with one scalar weight (model.py:37,56,77)."""
    return torch.clamp(x, min=0) + a.reshape(1, 1, 1, 1) * torch.clamp(x, max=0)


def pixel_shuffle2(x):
    """torch.nn.PixelShuffle(2) (model.py:36): out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w]."""
    n, c4, h, w = x.shape
    c = c4 // 4
    return x.reshape(n, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, c, 2 * h, 2 * w)


def generator_forward(sd, x, q=None):
    """Generator.forward (model.py:112-117); n_layers / n_upsample inferred from the state_dict.  q: None (the reference)
    or Q_BF16 (the reference with the bf16 mode's storage roundings)."""
    r = _st(q, prelu(F.conv2d(_op(q, x), _op(q, sd["neck.0.weight"]), sd["neck.0.bias"], padding=1), sd["neck.1.weight"]))  # :75-78
    y = r
    i = 0
    while f"stem.{i}.conv1.weight" in sd:  # ResidualBlock.forward, model.py:67-69
        t = _st(q, prelu(instance_norm_stored(F.conv2d(y, _op(q, sd[f"stem.{i}.conv1.weight"]), None, padding=1), q), sd[f"stem.{i}.relu1.weight"]))
        y = _st(q, instance_norm_stored(F.conv2d(t, _op(q, sd[f"stem.{i}.conv2.weight"]), None, padding=1), q) + y)
        i += 1
    y = _st(q, instance_norm_stored(F.conv2d(y, _op(q, sd["bottleneck.0.weight"]), None, padding=1), q) + r)  # :86-95, :115
    j = 0
    while f"upsampling.{j}.conv.weight" in sd:  # UpSamplingBlock.forward, model.py:39-40
        y = F.conv2d(y, _op(q, sd[f"upsampling.{j}.conv.weight"]), sd[f"upsampling.{j}.conv.bias"], padding=1)
        y = _st(q, prelu(pixel_shuffle2(y), sd[f"upsampling.{j}.relu.weight"]))
        j += 1
    return torch.tanh(F.conv2d(y, _op(q, sd["head.0.weight"]), sd["head.0.bias"], padding=1))  # :102-110


D_STRIDES = (2, 1, 2, 1, 2, 1, 2)  # model.py:148-183


def discriminator_forward(sd, x, q=None):
    """Discriminator.forward (model.py:139-193): neck conv + LeakyReLU(0.2); 7 SimpleBlocks
    (conv no-bias, InstanceNorm, LeakyReLU(default 0.01), model.py:120-136); 1x1 conv."""
    y = _st(q, F.leaky_relu(F.conv2d(_op(q, x), _op(q, sd["neck.0.weight"]), sd["neck.0.bias"], padding=1), 0.2))
    for i, s in enumerate(D_STRIDES):
        u = F.conv2d(y, _op(q, sd[f"stem.{i}.conv.weight"]), None, stride=s, padding=1)
        y = _st(q, F.leaky_relu(instance_norm_stored(u, q), 0.01))
    return F.conv2d(y, sd["stem.7.weight"], sd["stem.7.bias"])


def vgg_forward(sd, x, q=None):
    """VGG19.forward (model.py:20-23) over vgg19.features[:34] (model.py:8): 15x[conv3x3+ReLU],
    MaxPool2d(2) after convs 2, 4, 8, 12 (torchvision cfg 'E'), ending at the ReLU after conv5_3...
    (index 33)."""
    mean = sd["mean"] if "mean" in sd else torch.tensor(VGG_MEAN).view(1, 3, 1, 1)
    std = sd["std"] if "std" in sd else torch.tensor(VGG_STD).view(1, 3, 1, 1)
    if q is None:
        y = (x + 1.0) / 2.0
        y = (y - mean) / std
    else:   # the first-layer kernel folds both lines into one fused multiply-add per channel and rounds the result
        y = _op(q, x * (0.5 / std) + (0.5 - mean) / std)
    for idx in VGG_CONV_IDX:
        y = _st(q, F.relu(F.conv2d(y, _op(q, sd[f"vgg.{idx}.weight"]), sd[f"vgg.{idx}.bias"], padding=1)))
        if idx in VGG_POOL_AFTER:
            y = F.max_pool2d(y, 2, 2)
    return y


def vgg_standin_state_dict(seed=1234, width_div=1):
    """Structural stand-in for torchvision's vgg19(weights=IMAGENET1K_V1): the ImageNet weights
    cannot be downloaded offline, so weights are drawn the way torchvision initialises vgg19 when
    weights=None -- kaiming_normal_(mode='fan_out', nonlinearity='relu'), bias 0 -- from a seeded
    CPU generator.  width_div > 1 shrinks every layer (for small golden fixtures)."""
    g = torch.Generator().manual_seed(seed)
    sd = {"mean": torch.tensor(VGG_MEAN).view(1, 3, 1, 1), "std": torch.tensor(VGG_STD).view(1, 3, 1, 1)}
    cin, k = 3, 0
    for v in VGG_CFG:
        if v == "M":
            continue
        cout = v // width_div
        stdv = math.sqrt(2.0 / (cout * 9))
        sd[f"vgg.{VGG_CONV_IDX[k]}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * stdv
        sd[f"vgg.{VGG_CONV_IDX[k]}.bias"] = torch.zeros(cout)
        cin = cout
        k += 1
    return sd


def bce_with_logits(x, t):
    """torch.nn.BCEWithLogitsLoss() (trainer.py:41): mean(max(x,0) - x*t + log1p(exp(-|x|)))."""
    return (torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-x.abs()))).mean()


def smooth_l1(a, b):
    """torch.nn.SmoothL1Loss() (trainer.py:43): beta = 1, mean reduction."""
    d = (a - b).abs()
    return torch.where(d < 1.0, 0.5 * d * d, d - 0.5).mean()


def adamw_step(params, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, wd=1e-2):
    """torch.optim.AdamW defaults as built at trainer.py:33-38 (decoupled weight decay 0.01 on
    every parameter, PReLU weights and biases included)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    for k in params:
        if grads.get(k) is None:
            continue
        g = grads[k]
        m = state.setdefault(("m", k), torch.zeros_like(params[k]))
        v = state.setdefault(("v", k), torch.zeros_like(params[k]))
        params[k].mul_(1 - lr * wd)
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m, denom, value=-lr / bc1)


def train_step(g_sd, d_sd, v_sd, lr_images, hr_images, noise, g_state, d_state, g_lr=1e-4, d_lr=1e-4, grads_out=None, q=None,
               chunk=None):
    """One iteration of Trainer.train's loop body, trainer.py:171-196, line for line.  `noise` is the
    three torch.rand_like draws of :175, :176, :187 (injected: device RNG streams differ).
    g_sd / d_sd are updated in place; returns the four logged losses (:199-218).  grads_out (optional dict) receives the
    gradients of the two backward passes under "d.<key>" / "g.<key>".  q = Q_BF16: the same iteration with the bf16
    mode's storage roundings.
    chunk (optional, must divide the batch): the batch is walked in slices of `chunk` samples and the gradients are
    accumulated -- the SAME iteration, because every layer of the three networks is per-sample (InstanceNorm, no BatchNorm) and
    every loss is a batch mean, so loss = sum_c (|c| / B) loss_c and likewise its gradient.  It bounds the autograd memory of
    the batch-32 parity test to that of a batch of `chunk` (tests/test_oracle.py checks chunked == whole)."""
    B = lr_images.shape[0]
    chunk = B if chunk is None else int(chunk)
    if chunk <= 0 or B % chunk != 0:
        raise ValueError("chunk must divide the batch")
    slices = [slice(i, i + chunk) for i in range(0, B, chunk)]
    wgt = float(chunk) / float(B)

    def accumulate(total, part):
        return [p.detach().clone() for p in part] if total is None else [t.add_(p.detach()) for t, p in zip(total, part)]

    # ---- discriminator step, :171-181
    dp = {k: v.detach().clone().requires_grad_(True) for k, v in d_sd.items()}
    grads, loss_real, loss_fake = None, 0.0, 0.0
    for sl in slices:
        y_real = discriminator_forward(dp, hr_images[sl], q)                   # :172
        with torch.no_grad():
            sr = generator_forward(g_sd, lr_images[sl], q)                     # :173 (.detach())
        y_fake = discriminator_forward(dp, sr, q)                              # :174
        real_labels = 0.3 * noise[0][sl] + 0.8                                 # :175
        fake_labels = 0.3 * noise[1][sl]                                       # :176
        lr_c = bce_with_logits(y_real, real_labels)                            # :177
        lf_c = bce_with_logits(y_fake, fake_labels)                            # :178
        d_loss = 0.5 * lr_c + 0.5 * lf_c                                       # :179
        grads = accumulate(grads, torch.autograd.grad(wgt * d_loss, list(dp.values())))   # :180
        loss_real = loss_real + wgt * lr_c.detach()
        loss_fake = loss_fake + wgt * lf_c.detach()
    if grads_out is not None:
        grads_out.update({"d." + k: g.detach().clone() for k, g in zip(dp.keys(), grads)})
    adamw_step(d_sd, dict(zip(dp.keys(), grads)), d_state, lr=d_lr)            # :181
    # ---- generator step, :184-196
    gp = {k: v.detach().clone().requires_grad_(True) for k, v in g_sd.items()}
    grads, adv_loss, content_loss = None, 0.0, 0.0
    for sl in slices:
        sr = generator_forward(gp, lr_images[sl], q)                           # :185
        y_fake = discriminator_forward(d_sd, sr, q)                            # :186 (updated D)
        real_labels = 0.3 * noise[2][sl] + 0.7                                 # :187
        adv_c = 1e-1 * bce_with_logits(y_fake, real_labels)                    # :188
        fake_features = vgg_forward(v_sd, sr, q)                               # :190
        real_features = vgg_forward(v_sd, hr_images[sl], q)                    # :191
        content_c = smooth_l1(fake_features, real_features)                    # :192
        g_loss = 0.5 * adv_c + 0.5 * content_c                                 # :194
        grads = accumulate(grads, torch.autograd.grad(wgt * g_loss, list(gp.values())))   # :195
        adv_loss = adv_loss + wgt * adv_c.detach()
        content_loss = content_loss + wgt * content_c.detach()
    if grads_out is not None:
        grads_out.update({"g." + k: g.detach().clone() for k, g in zip(gp.keys(), grads)})
    adamw_step(g_sd, dict(zip(gp.keys(), grads)), g_state, lr=g_lr)            # :196
    return {"loss_real": loss_real, "loss_fake": loss_fake, "adv_loss": adv_loss, "content_loss": content_loss}


def pretrain_step(g_sd, lr_images, hr_images, g_state, g_lr=1e-4):
    """Trainer.pretrain loop body, trainer.py:107-111: SmoothL1 on pixels, AdamW."""
    gp = {k: v.detach().clone().requires_grad_(True) for k, v in g_sd.items()}
    loss = smooth_l1(generator_forward(gp, lr_images), hr_images)
    grads = torch.autograd.grad(loss, list(gp.values()))
    adamw_step(g_sd, dict(zip(gp.keys(), grads)), g_state, lr=g_lr)
    return loss.detach()


# ------------------------------------------------------------------ data path
def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def aa_bicubic_weights(in_size, out_size):
    """Separable antialiased-bicubic taps of torch's upsample_bicubic2d_aa kernel (the kernel
    torchvision.transforms.v2.Resize(antialias=True, BICUBIC) forwards to for float tensors,
    dataloader.py:15-19,34): scale = in/out, support = 2*scale, center = scale*(i+0.5),
    xmin = max(int(center-support+0.5), 0), xsize = min(int(center+support+0.5), in) - xmin,
    w_j = cubic((j + xmin - center + 0.5)/scale), normalised.  Returns (xmin[out], W[out][kmax])."""
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    kmax = int(math.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, dtype=np.int32)
    sizes = np.zeros(out_size, dtype=np.int32)
    W = np.zeros((out_size, kmax), dtype=np.float32)
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), in_size) - xmin
        ws = np.array([_cubic((j + xmin - center + 0.5) * invscale) for j in range(xsize)], dtype=np.float32)
        tot = np.float32(ws.sum(dtype=np.float32))
        W[i, :xsize] = ws / tot
        xmins[i] = xmin
        sizes[i] = xsize
    return xmins, sizes, W


def resize_bicubic_aa(img, out_h, out_w):
    """(C,H,W) float32 -> (C,out_h,out_w): horizontal pass then vertical pass, as the torch CPU
    kernel orders them (separable; no clamp, no rounding)."""
    c, h, w = img.shape
    xm, xs, wx = aa_bicubic_weights(w, out_w)
    ym, ys, wy = aa_bicubic_weights(h, out_h)
    tmp = np.zeros((c, h, out_w), dtype=np.float32)
    for i in range(out_w):
        tmp[:, :, i] = (img[:, :, xm[i]:xm[i] + xs[i]] * wx[i, :xs[i]]).sum(axis=2, dtype=np.float32)
    out = np.zeros((c, out_h, out_w), dtype=np.float32)
    for i in range(out_h):
        out[:, i, :] = (tmp[:, ym[i]:ym[i] + ys[i], :] * wy[i, :ys[i], None]).sum(axis=1, dtype=np.float32)
    return out


def dataset_item(image_u8, lr_image_size, scale_factor, rng=random):
    """NumpyImagesDataset.__getitem__ (dataloader.py:24-38) for one uint8 CHW array: two inclusive
    random.randint draws (crop_h then crop_w), float32 crop, antialiased bicubic down-scale of the
    UNSCALED 0..255 crop, then both mapped to [-1,1] by x/127.5 - 1."""
    hr_size = lr_image_size * scale_factor
    _, h, w = image_u8.shape
    crop_h, crop_w = rng.randint(0, h - hr_size), rng.randint(0, w - hr_size)
    hr = image_u8[:, crop_h:crop_h + hr_size, crop_w:crop_w + hr_size].astype(np.float32)
    lr = resize_bicubic_aa(hr, lr_image_size, lr_image_size)
    return torch.from_numpy(lr / np.float32(127.5) - np.float32(1.0)), torch.from_numpy(hr / np.float32(127.5) - np.float32(1.0)), (crop_h, crop_w)


def postprocess_u8(sr):
    """inference.py:53-56: (y+1)/2 -> NHWC -> *255 -> astype(uint8) (C truncation toward zero,
    no rounding, no clamp; tanh keeps y inside (-1,1) so the product stays inside [0,255))."""
    y = (sr + 1.0) / 2.0
    y = y.permute(0, 2, 3, 1).squeeze()
    return (y * 255).numpy().astype(np.uint8)


# ---------------------------------------------------------------------------------- validation metrics (trainer.py:46-69)
# torchmetrics (pinned 1.4.0 in the reference's Pipfile) is NOT installed here and cannot be fetched: the two metrics are
# restated from its published algorithm (functional/image/ssim.py::_ssim_update, functional/image/psnr.py) -- parity
# UNPINNED at the torchmetrics boundary (no golden vector of the real package exists in the reference or here).
def ssim_per_image(preds, target, data_range=1.0, sigma=1.5, k1=0.01, k2=0.03):
    """StructuralSimilarityIndexMeasure(data_range=1.0, reduction="none") on (N,C,H,W) batches in [0,1] (trainer.py:46-48):
    gaussian window of size int(3.5 sigma + .5) * 2 + 1 = 11, reflect padding by 5, five depthwise convolutions, clamped
    variances, crop by 5, mean over (C, H-10, W-10) per image."""
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    ks = int(3.5 * sigma + 0.5) * 2 + 1
    pad = (ks - 1) // 2
    dist = torch.arange((1 - ks) / 2, (1 + ks) / 2, 1, dtype=preds.dtype)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    ch = preds.shape[1]
    kernel = torch.matmul(g.t(), g).expand(ch, 1, ks, ks)
    p = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    outs = F.conv2d(torch.cat((p, t, p * p, t * t, p * t)), kernel, groups=ch).split(preds.shape[0])
    mu_pp, mu_tt, mu_pt = outs[0].pow(2), outs[1].pow(2), outs[0] * outs[1]
    s_pp = torch.clamp(outs[2] - mu_pp, min=0.0)
    s_tt = torch.clamp(outs[3] - mu_tt, min=0.0)
    s_pt = outs[4] - mu_pt
    full = ((2 * mu_pt + c1) * (2 * s_pt + c2)) / ((mu_pp + mu_tt + c1) * (s_pp + s_tt + c2))
    return full[..., pad:-pad, pad:-pad].reshape(preds.shape[0], -1).mean(-1)


def psnr_global(batches, data_range=1.0):
    """PeakSignalNoiseRatio(data_range=1.0, reduction="none"), dim=None (trainer.py:49-51): the squared error and the
    element count accumulate over every update; one global 10 log10(range^2 / mse).  `batches`: [(preds, target), ...]."""
    sse = sum(float(((p.double() - t.double()) ** 2).sum()) for p, t in batches)
    n = sum(t.numel() for _, t in batches)
    return 10.0 * math.log10(data_range ** 2 / (sse / n))
