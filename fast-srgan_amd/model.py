"""Generator / Discriminator / VGG19 with the reference's nn.Module surface on MI355X kernels.

Drop-in for /root/reference/model.py: same class names, constructor arguments, submodule names and
therefore the same state_dict keys and shapes (the shipped models/model.pt loads unchanged), same
forward signatures ((N,3,H,W) float32 in; (N,3,4H,4W) / (N,1,H/16,W/16) float32 out).  The torch.nn
submodules are kept purely as PARAMETER CONTAINERS (and so that default initialisation draws the
same random numbers in the same order as the reference); their forward methods are never called.
All math runs through fast-srgan_amd.ops on NHWC activations in the module's compute dtype.

Extra constructor keyword (not in the reference): compute_dtype = "f16" (default since round 5; fp16 MFMA with
f32 accumulation, f32 parameters/statistics), "bf16" (bf16 MFMA), "x3" (split-bf16 operands, three bf16
MFMAs per product: the fast mode inside the reference's 1e-3 fp32 tolerance on outputs and losses), "f32" (exact-f32 MFMA),
or "x3v" (round 6: the x3 mode for the TRAINED networks -- Generator, Discriminator -- and fp16 for the FROZEN perceptual
network, whose only product is the content loss: that loss stays within 2e-4 of the fp32 reference's, the gradients are as far
from float64 as pure x3's, and the iteration is 31 % faster; `split_compute_dtype`).
"""
import os
import warnings

import torch

from . import _lib as L
from . import ops

_DEFAULT_DTYPE = os.environ.get("FSR_COMPUTE_DTYPE", "f16")


def split_compute_dtype(name):
    """(dtype of the trained networks, dtype of the frozen perceptual network) of a compute-mode name.  Every mode but "x3v" uses
    one dtype for both; "x3v" = x3 for Generator / Discriminator, fp16 for VGG19 (its backward then runs under the Trainer's
    dynamic loss scale, like the fp16 mode's)."""
    if name == "x3v":
        return "x3", "f16"
    return name, name


class UpSamplingBlock(torch.nn.Module):
    """model.py:26-40: conv 3x3 C->4C (bias) -> PixelShuffle(2) -> PReLU, one fused kernel here."""

    def __init__(self, config):
        super().__init__()
        self.conv = torch.nn.Conv2d(config.n_filters, config.n_filters * 4, kernel_size=3, padding=1)
        self.phase_shift = torch.nn.PixelShuffle(upscale_factor=2)
        self.relu = torch.nn.PReLU()


class ResidualBlock(torch.nn.Module):
    """model.py:43-69."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = torch.nn.InstanceNorm2d(out_channels)
        self.relu1 = torch.nn.PReLU()
        self.conv2 = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = torch.nn.InstanceNorm2d(out_channels)


class Generator(torch.nn.Module):
    """model.py:72-117.  `config.n_upsample` (default 2, i.e. 4x) is an extension for BASELINE cfg #5."""

    def __init__(self, config, compute_dtype=None):
        super().__init__()
        nf = config.n_filters
        self.compute = ops.Compute(split_compute_dtype(compute_dtype or _DEFAULT_DTYPE)[0])
        if nf % self.compute.cpad:
            raise ValueError("n_filters=%d must be a multiple of %d in %s mode" % (nf, self.compute.cpad, self.compute.name))
        self.neck = torch.nn.Sequential(torch.nn.Conv2d(3, nf, kernel_size=3, padding=1), torch.nn.PReLU())
        self.stem = torch.nn.Sequential(*[ResidualBlock(nf, nf) for _ in range(config.n_layers)])
        self.bottleneck = torch.nn.Sequential(
            torch.nn.Conv2d(nf, nf, kernel_size=3, padding=1, bias=False), torch.nn.InstanceNorm2d(nf))
        self.upsampling = torch.nn.Sequential(*[UpSamplingBlock(config) for _ in range(getattr(config, "n_upsample", 2))])
        self.head = torch.nn.Sequential(torch.nn.Conv2d(nf, 3, kernel_size=3, padding=1), torch.nn.Tanh())
        cd = self.compute
        self._cfg_neck = ops.ConvCfg(cd, act=L.ACT_PRELU, image_in=True)
        self._cfg_in = ops.ConvCfg(cd, stats=True)
        # first convolution of a residual block: also hands out aliases of its input for the skip connection(s), whose
        # gradients its data-gradient launch adds in the epilogue (no autograd accumulation kernels on the block inputs)
        self._cfg_in_skip = ops.ConvCfg(cd, stats=True, n_alias=1)
        self._cfg_in_skip2 = ops.ConvCfg(cd, stats=True, n_alias=2)
        self._cfg_up = ops.ConvCfg(cd, act=L.ACT_PRELU, pixel_shuffle=True)
        self._cfg_head = ops.ConvCfg(cd, tanh_head=True)
        self._cfg_head_u8 = ops.ConvCfg(cd, tanh_head=True, u8_head=True)

    def forward_u8(self, frames):
        """Inference on raw frames (inference.py:47-57 without the host round trips): (N,H,W,3) uint8 in, (N,4H,4W,3) uint8
        out.  The [-1,1] mapping (:48) is one small kernel in front of the neck; the head's epilogue applies :53-56
        ((y+1)/2 * 255, truncating cast) and stores bytes -- a 720p frame leaves the device as 2.8 MB instead of 11 MB of
        floats."""
        with torch.no_grad():
            return self._forward(ops.u8_to_image(frames), self._cfg_head_u8)

    def forward(self, x):
        return self._forward(x, self._cfg_head)

    def _forward(self, x, cfg_head):
        cd = self.compute
        r, _ = ops.conv3x3(x, self.neck[0].weight, self.neck[0].bias, self.neck[1].weight, self._cfg_neck)  # :113
        y = r
        training = torch.is_grad_enabled()
        for k, blk in enumerate(self.stem):                                                             # :114
            if training:
                # block 0's input r also feeds the long skip of :115: two aliases
                t, st, *skip = ops.conv3x3(y, blk.conv1.weight, None, None, self._cfg_in_skip2 if k == 0 else self._cfg_in_skip)
            else:
                t, st = ops.conv3x3(y, blk.conv1.weight, None, None, self._cfg_in)
                skip = [y, y]
            if k == 0:
                r = skip[1]
            t = ops.instnorm_act(t, st, None, blk.relu1.weight, cd, L.ACT_PRELU)
            u, st = ops.conv3x3(t, blk.conv2.weight, None, None, self._cfg_in)
            y = ops.instnorm_act(u, st, skip[0], None, cd)
        u, st = ops.conv3x3(y, self.bottleneck[0].weight, None, None, self._cfg_in)                      # :115
        y = ops.instnorm_act(u, st, r, None, cd)
        for up in self.upsampling:                                                                      # :116
            y, _ = ops.conv3x3(y, up.conv.weight, up.conv.bias, up.relu.weight, self._cfg_up)
        out, _ = ops.conv3x3(y, self.head[0].weight, self.head[0].bias, None, cfg_head)                # :117
        return out


class GraphedGenerator:
    """Inference replay of a Generator at one fixed input shape as a single hipGraph launch.

    At batch 1 a 720p frame is ~45 kernels of 10-30 us each, so eager inference is bound by host launch overhead;
    one graph launch per frame removes it.  `gg = GraphedGenerator(model, example)`; `y = gg(x)` copies x into the
    static input, replays, and returns the static output tensor (clone it if it must outlive the next call)."""

    def __init__(self, model, example, warmup=2):
        self.model = model.eval()
        self.x = example.detach().clone()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.model(self.x)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.y = self.model(self.x)

    def __call__(self, x):
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.y


class SimpleBlock(torch.nn.Module):
    """model.py:120-136 (LeakyReLU with the default slope 0.01)."""

    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.conv = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, stride=stride, bias=False)
        self.bn = torch.nn.InstanceNorm2d(out_channels)
        self.act = torch.nn.LeakyReLU()
        self.stride = stride


class Discriminator(torch.nn.Module):
    """model.py:139-193.  config.n_layers is accepted and ignored, as in the reference."""

    def __init__(self, config, compute_dtype=None):
        super().__init__()
        self.config = config
        nf = config.n_filters
        self.compute = ops.Compute(split_compute_dtype(compute_dtype or _DEFAULT_DTYPE)[0])
        if nf % self.compute.cpad:
            raise ValueError("n_filters=%d must be a multiple of %d in %s mode" % (nf, self.compute.cpad, self.compute.name))
        self.neck = torch.nn.Sequential(torch.nn.Conv2d(3, nf, kernel_size=3, padding=1), torch.nn.LeakyReLU(negative_slope=0.2))
        self.stem = torch.nn.Sequential(
            SimpleBlock(nf, nf, 2), SimpleBlock(nf, nf * 2, 1), SimpleBlock(nf * 2, nf * 2, 2),
            SimpleBlock(nf * 2, nf * 4, 1), SimpleBlock(nf * 4, nf * 4, 2), SimpleBlock(nf * 4, nf * 8, 1),
            SimpleBlock(nf * 8, nf * 8, 2),
            torch.nn.Conv2d(nf * 8, 1, kernel_size=1, padding=0, stride=1))
        cd = self.compute
        # the neck's LeakyReLU(0.2) backward is applied by stem.0's data-gradient epilogue (mask = the neck output it
        # saved as its input); the neck then only needs the column sums of that gradient for its bias
        self._cfg_neck = ops.ConvCfg(cd, act=L.ACT_LEAKY, slope=0.2, image_in=True, act_bwd_by_consumer=True, emit_signs=True)
        self._cfg_s = {1: ops.ConvCfg(cd, stride=1, stats=True), 2: ops.ConvCfg(cd, stride=2, stats=True)}
        self._cfg_s0 = ops.ConvCfg(cd, stride=2, stats=True, input_act_bwd=0.2)

    def forward(self, x):
        cd = self.compute
        y, _ = ops.conv3x3(x, self.neck[0].weight, self.neck[0].bias, None, self._cfg_neck)
        for i, blk in enumerate(list(self.stem)[:7]):
            u, st = ops.conv3x3(y, blk.conv.weight, None, None, self._cfg_s0 if i == 0 else self._cfg_s[blk.stride])
            y = ops.instnorm_act(u, st, None, None, cd, L.ACT_LEAKY, 0.01)
        return ops.conv1x1_to_logits(y, self.stem[7].weight, self.stem[7].bias, cd)


_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


class VGG19(torch.nn.Module):
    """model.py:5-23: ImageNet normalisation + torchvision vgg19.features[:34], frozen.

    torchvision is not a dependency: the feature stack is rebuilt with the same module indices (state_dict
    keys vgg.{0,2,5,...,32}.{weight,bias} + buffers mean/std).  Pretrained weights: pass `weights=` a path
    to torchvision's vgg19 checkpoint (keys features.N.*) or set FSR_VGG19_WEIGHTS; when torchvision is
    installed and has the ImageNet file cached, it is used as the reference does (model.py:8).  WITHOUT ImageNet
    weights the constructor RAISES: a perceptual loss against random features trains a silently different model.
    The random stand-in (torchvision's weights=None initialisation, kaiming-normal fan_out) is an explicit opt-in
    for tests and benchmarks: `seed=` or `allow_random=True`.
    """

    def __init__(self, weights=None, compute_dtype=None, width_div=1, seed=None, allow_random=False):
        super().__init__()
        self.compute = ops.Compute(split_compute_dtype(compute_dtype or _DEFAULT_DTYPE)[1])
        layers, cin = [], 3
        for v in _VGG_CFG:
            if v == "M":
                layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [torch.nn.Conv2d(cin, v // width_div, kernel_size=3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v // width_div
        self.vgg = torch.nn.Sequential(*layers)[:34]
        path = weights or os.environ.get("FSR_VGG19_WEIGHTS")
        if path:
            sd = torch.load(path, map_location="cpu")
            self.vgg.load_state_dict({k[len("features."):]: v for k, v in sd.items()
                                      if k.startswith("features.") and int(k.split(".")[1]) < 34})
        elif seed is not None or allow_random:
            if seed is None:
                warnings.warn("VGG19: random kaiming-normal stand-in instead of the ImageNet weights (allow_random=True)")
            gen = torch.Generator().manual_seed(1234 if seed is None else seed)
            for m in self.vgg:
                if isinstance(m, torch.nn.Conv2d):
                    with torch.no_grad():
                        std = (2.0 / (m.out_channels * 9)) ** 0.5
                        m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * std)
                        m.bias.zero_()
        else:
            try:        # what the reference does (model.py:8); needs torchvision and its cached / downloadable checkpoint
                from torchvision.models.vgg import VGG19_Weights, vgg19
                tv = vgg19(weights=VGG19_Weights.IMAGENET1K_V1).features[:34]
                self.vgg.load_state_dict(tv.state_dict())
            except Exception as exc:
                raise L.FsrError(
                    "VGG19: no ImageNet weights. Pass weights=<torchvision vgg19 checkpoint>, set FSR_VGG19_WEIGHTS or the "
                    "config key training.vgg19_weights; the reference uses vgg19(IMAGENET1K_V1) (model.py:8) and a random "
                    "feature network would silently train a different model. For tests / benchmarks opt in to the random "
                    "stand-in with VGG19(seed=...) or training.allow_random_vgg=true. (torchvision: %s)" % (exc,)) from exc
        for param in self.vgg.parameters():
            param.requires_grad = False
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406], requires_grad=False).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225], requires_grad=False).view(1, 3, 1, 1))
        mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        cd = self.compute
        # (x+1)/2 then (x-mean)/std  ==  x * 1/(2 std) + (0.5-mean)/std, fused into the first conv's input load
        scale, shift = tuple(0.5 / s for s in std), tuple((0.5 - m) / s for m, s in zip(mean, std))
        # Backward plan (the stack is frozen, so no bias gradients exist): every conv's ReLU backward is applied
        # by its CONSUMER -- the next conv's data-gradient epilogue (mask = its own saved input) or the pool's
        # backward -- instead of a separate elementwise pass; only the last conv handles its own.
        mods = list(self.vgg)
        convs = [i for i, m in enumerate(mods) if isinstance(m, torch.nn.Conv2d)]
        self._plan = []
        for k, i in enumerate(convs):
            prev_is_relu = k > 0 and isinstance(mods[i - 1], torch.nn.ReLU)
            nxt = mods[i + 2] if i + 2 < len(mods) else None
            self._plan.append((i, ops.ConvCfg(cd, act=L.ACT_RELU, image_in=(k == 0), in_scale=scale if k == 0 else (1.0, 1.0, 1.0),
                                              in_shift=shift if k == 0 else (0.0, 0.0, 0.0),
                                              input_act_bwd=0.0 if prev_is_relu else None,
                                              act_bwd_by_consumer=nxt is not None),
                               isinstance(nxt, torch.nn.MaxPool2d)))

    @property
    def _plan_pooled(self):
        plan = getattr(self, "_plan_pooled_cache", None)
        if plan is None:
            plan = {k: ops.ConvCfg(self.compute, act=L.ACT_RELU, pool_after=True) for k, (_, _, pool) in enumerate(self._plan) if pool}
            self._plan_pooled_cache = plan
        return plan

    def features_nhwc(self, x):
        y = x
        # no gradient wanted (the target features of trainer.py:191, inference): conv + ReLU + MaxPool2d as ONE kernel in the
        # 16-bit modes -- the full-resolution tensor, which only the pool's backward would read, is never written
        fuse_pool = (not torch.is_grad_enabled()) and self.compute.name != "f32"
        for k, (i, cfg, pool_after) in enumerate(self._plan):
            m = self.vgg[i]
            if pool_after and fuse_pool and not cfg.image_in and y.shape[1] % 2 == 0 and y.shape[2] % 2 == 0:
                y, _ = ops.conv3x3(y, m.weight, m.bias, None, self._plan_pooled[k])
                continue
            y, _ = ops.conv3x3(y, m.weight, m.bias, None, cfg)
            if pool_after:
                y = ops.maxpool2(y, self.compute, relu_mask=True)
        return y

    def forward(self, x):
        """(N,3,H,W) in [-1,1] -> (N,512,H/16,W/16) feature VALUES (model.py:20-23).  In the x3 mode the kernels' output is a
        float32 container of bf16 hi / lo planes: it is decoded here (and the cotangent re-encoded on the way back), so
        callers of V(x) never see the container; the trainer's loss kernels take features_nhwc() directly."""
        return ops.values(self.compute, self.features_nhwc(x)).permute(0, 3, 1, 2)
