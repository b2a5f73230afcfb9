"""Parity at the configurations bench.py times (BASELINE.json configs[0..2]): the full-width VGG19 stack, one whole GAN
iteration at cfg #1 size (batch 4, 96 -> 384, 64 filters / 8 blocks, full-width VGG) and generator inference at 90x160
and 180x320 -- HIP path vs the CPU oracle, in the exact-f32 MFMA mode (north-star tolerance 1e-3) and in the benched
bf16 mode (gates ~2x the errors measured on the MI355X; every value is logged through backend.report)."""
import types
import warnings

import pytest
import torch

from backend import relerr, relerr2, report, select
from conftest import load_npz, sd_from
from oracle import srgan_cpu as O

pytestmark = pytest.mark.gpu


def ns(**k):
    return types.SimpleNamespace(**k)


# bf16 gates (~2x measured, gpurun_out/parity_errors.log)
VGG_BF16_OUT, VGG_BF16_DX = 4e-2, 8e-2
STEP_BF16_LOSS, STEP_BF16_GRAD, STEP_BF16_SCALAR = 3e-2, 0.3, 0.5
INF_BF16_MEAN, INF_BF16_MAX = 0.02, 0.5


@pytest.mark.parametrize("cdn", ["f32", "bf16"])
def test_full_width_vgg19_forward_and_input_gradient(pkg, cdn):
    """VGG19.forward (model.py:5-23) at its real width (64 .. 512 channels, 15 convolutions, 4 pools): features and the
    gradient with respect to the image, the only gradient the frozen network produces (trainer.py:190-195)."""
    dev = select("hip")
    torch.manual_seed(5)
    V = pkg.VGG19(compute_dtype=cdn, seed=1234).to(dev)
    v_sd = O.vgg_standin_state_dict(1234, 1)
    for k, v in V.state_dict().items():
        assert torch.equal(v.cpu(), v_sd[k]), k
    x = torch.rand(2, 3, 64, 96) * 2 - 1
    xd = x.to(dev).requires_grad_(True)
    y = V(xd)
    xr = x.clone().requires_grad_(True)
    yr = O.vgg_forward(v_sd, xr)
    assert y.shape == yr.shape == (2, 512, 4, 6)
    r = torch.randn(yr.shape)
    (y.float() * r.to(dev)).sum().backward()
    (yr * r).sum().backward()
    t_out, t_dx = (1e-3, 1e-3) if cdn == "f32" else (VGG_BF16_OUT, VGG_BF16_DX)
    assert report("vgg_full.%s.features" % cdn, relerr(y, yr)) < t_out
    assert report("vgg_full.%s.dx_l2" % cdn, relerr2(xd.grad, xr.grad)) < t_dx
    if cdn == "f32":
        assert relerr(xd.grad, xr.grad) < 1e-2      # max-norm as well in the parity mode


@pytest.mark.parametrize("cdn", ["f32", "bf16"])
def test_train_step_at_baseline_cfg1_size(pkg, cdn):
    """One full iteration (trainer.py:171-196) at BASELINE configs[0]: batch 4, 96x96 -> 384x384, 64 filters / 8 blocks,
    full-width VGG stand-in, injected label noise: the four losses and the gradients of both backward passes."""
    dev = select("hip")
    torch.manual_seed(6)
    cfg = ns(experiment=ns(name="cfg1", seed=1234), generator=ns(n_filters=64, n_layers=8),
             discriminator=ns(n_filters=64, n_layers=7),
             training=ns(compiled=False, device=str(dev), log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4,
                         discriminator_lr=1e-4, batch_size=4, compute_dtype=cdn))
    T = pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype=cdn, seed=1234))
    g_sd = {k: v.detach().cpu().clone() for k, v in T.generator.state_dict().items()}
    d_sd = {k: v.detach().cpu().clone() for k, v in T.discriminator.state_dict().items()}
    v_sd = O.vgg_standin_state_dict(1234, 1)
    lr, hr = torch.rand(4, 3, 96, 96) * 2 - 1, torch.rand(4, 3, 384, 384) * 2 - 1
    noise = [torch.rand(4, 1, 24, 24) for _ in range(3)]
    got = T.train_step(lr.to(dev), hr.to(dev), [n.to(dev) for n in noise])
    torch.cuda.synchronize()
    ref_grads = {}
    want = O.train_step(g_sd, d_sd, v_sd, lr, hr, noise, {}, {}, grads_out=ref_grads)
    t_loss, t_grad, t_scalar = (1e-3, 1e-2, 1e-2) if cdn == "f32" else (STEP_BF16_LOSS, STEP_BF16_GRAD, STEP_BF16_SCALAR)
    for k in want:
        e = report("cfg1.%s.%s" % (cdn, k), abs(float(got[k]) - float(want[k])) / abs(float(want[k])))
        assert e < t_loss, (k, float(got[k]), float(want[k]))
    # gradients left in the arenas: the discriminator's from the D step (:180), the generator's from the G step (:195)
    for tag, mod in (("d", T.discriminator), ("g", T.generator)):
        for k, p in mod.named_parameters():
            e = report("cfg1.%s.grad.%s.%s" % (cdn, tag, k), relerr2(p.grad, ref_grads[tag + "." + k]))
            assert e < (t_scalar if p.numel() < 1000 else t_grad), (tag, k, e)


@pytest.mark.parametrize("hw", [(90, 160), (180, 320)])
def test_generator_inference_shapes_vs_oracle(pkg, hw):
    """BASELINE configs[1]: the shipped generator at 90x160 -> 360x640 and 180x320 -> 720x1280 (ragged 90-row tiles at
    full width) -- fp32 mode against the oracle at 1e-3, bf16 against the same reference at ~2x its measured error."""
    dev = select("hip")
    z = load_npz("g_model_pt.npz")
    sd = sd_from(z, "sd.")
    torch.manual_seed(7)
    x = torch.rand(1, 3, *hw) * 2 - 1
    want = O.generator_forward(sd, x)
    for cdn in ("f32", "bf16"):
        G = pkg.Generator(ns(n_filters=64, n_layers=8), compute_dtype=cdn)
        G.load_state_dict(sd)
        G.to(dev).eval()
        with torch.no_grad():
            y = G(x.to(dev)).cpu()
        assert y.shape == (1, 3, 4 * hw[0], 4 * hw[1])
        if cdn == "f32":
            assert report("infer.%dx%d.f32" % hw, relerr(y, want)) < 1e-3
        else:
            assert report("infer.%dx%d.bf16.mean_abs" % hw, float((y - want).abs().mean())) < INF_BF16_MEAN
            assert report("infer.%dx%d.bf16.max_abs" % hw, float((y - want).abs().max())) < INF_BF16_MAX
