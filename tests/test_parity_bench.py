"""Parity at the configurations bench.py times (BASELINE.json configs[0..2]): the full-width VGG19 stack, one whole GAN
iteration at cfg #1 size (batch 4, 96 -> 384, 64 filters / 8 blocks, full-width VGG) and generator inference at 90x160
and 180x320 -- HIP path vs the CPU oracle, in the exact-f32 MFMA mode (north-star tolerance 1e-3) and in the benched
bf16 mode (gates ~2x the errors measured on the MI355X; every value is logged through backend.report)."""
import types
import warnings

import pytest
import torch

from backend import check_grads, relerr, relerr2, report, select
from conftest import load_npz, sd_from
from oracle import srgan_cpu as O

pytestmark = pytest.mark.gpu


def ns(**k):
    return types.SimpleNamespace(**k)


# Gates (~2x the errors measured on the MI355X, gpurun_out/parity_errors.log).  Two facts shape them (DESIGN.md section 5,
# tests/probes/conditioning_probe.py):
#  * whole-network GRADIENTS of this model are ill-conditioned: the fp32 oracle itself, run in float64, moves the generator's
#    gradients of this very iteration by 9.4 % (relative L2), the discriminator's by 0.5 %, the VGG image gradient by 0.6 % --
#    ReLU / LeakyReLU(0.01) / max-pool decisions flip under 1e-7 perturbations.  The f32-mode gates are 2x THAT, losses and
#    forward outputs keep the north-star 1e-3;
#  * the bf16 mode is held to the oracle with the bf16 storage roundings (O.Q_BF16); its distance to the plain fp32 oracle is
#    reported and loosely bounded.
F32_VGG_DX, F32_D_GRAD, F32_G_GRAD, F32_G_COS = 2e-2, 2e-2, 0.2, 0.98
VGGQ_OUT, VGGQ_DX, VGG_BF16_OUT, VGG_BF16_DX = 1.2e-2, 0.45, 2e-2, 0.7
#    Measured at cfg #1 (bf16 kernels vs the bf16-storage oracle): the four losses 3e-5, 3e-5, 1.7e-4, 3e-6; gradient tensors
#    0.15-0.5 relative L2 with norm ratios 0.94-1.01 and cosines 0.995 (D) / 0.90 (G): the forward pass is reproduced to 1e-3,
#    the gradients are as far apart as two bf16 evaluations of this network that differ in fp32 summation order are (one bf16
#    ulp in 0.02 % of the activations after the second discriminator block has become a difference in 67 % of them after the
#    seventh, with 0.1 % LeakyReLU(0.01) sign flips per layer -- tests/probes/conditioning_probe.py).
#    Round 3 (verdict: "a 100 % L2 gate is not a test"): the gates below are 1.4-1.6x the values measured on the MI355X with the
#    round-3 kernels (gpurun_out/parity_errors.log, copied to profiles/r03_parity_errors.log) -- per network, with the norm ratio
#    of every tensor and the PReLU slopes bounded as well:
#      vs the bf16-storage oracle: D tensors <= 0.479, G tensors <= 0.492, cosines 0.9954 (D) / 0.905 (G), slopes <= 0.076;
#      vs the plain fp32 oracle:   D tensors <= 0.370, G tensors <= 0.522, cosine 0.992, norm ratios 0.97-1.05; PReLU slopes <= 0.07 of
#                                  the generator's largest slope gradient (<= 0.0013 of the largest 1-element gradient of both networks).
#    What holds the 16-bit mode to the reference beyond single-iteration gradients is tests/test_convergence.py (300 iterations
#    of f32 against bf16 / f16 training from one initialisation).
STEPQ_LOSS, STEPQ_D_GRAD, STEPQ_G_GRAD, STEPQ_COS_D, STEPQ_COS_G, STEPQ_SLOPE = 1e-3, 0.7, 0.7, 0.99, 0.85, 0.15
STEP_BF16_LOSS, STEP_BF16_D_GRAD, STEP_BF16_G_GRAD, STEP_BF16_COS, STEP_BF16_SLOPE, STEP_BF16_SLOPE_ALL = 4e-3, 0.55, 0.75, 0.984, 0.15, 0.01
STEP_NORM = (0.88, 1.14)
INF_BF16_MEAN, INF_BF16_MAX = 6e-3, 7e-2


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
    r = torch.randn(2, 512, 4, 6)
    (y.float() * r.to(dev)).sum().backward()

    def oracle(q):
        xr = x.clone().requires_grad_(True)
        yr = O.vgg_forward(v_sd, xr, q)
        (yr * r).sum().backward()
        return yr.detach(), xr.grad

    assert y.shape == (2, 512, 4, 6)
    if cdn == "f32":
        yr, dxr = oracle(None)
        assert report("vgg_full.f32.features", relerr(y, yr)) < 1e-3
        assert report("vgg_full.f32.dx_l2", relerr2(xd.grad, dxr)) < F32_VGG_DX
        return
    yr, dxr = oracle(O.Q_BF16)
    assert report("vgg_full.bf16q.features", relerr(y, yr)) < VGGQ_OUT
    assert report("vgg_full.bf16q.dx_l2", relerr2(xd.grad, dxr)) < VGGQ_DX
    yr, dxr = oracle(None)
    assert report("vgg_full.bf16.features", relerr(y, yr)) < VGG_BF16_OUT
    assert report("vgg_full.bf16.dx_l2", relerr2(xd.grad, dxr)) < VGG_BF16_DX


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
    g0 = {k: v.detach().cpu().clone() for k, v in T.generator.state_dict().items()}
    d0 = {k: v.detach().cpu().clone() for k, v in T.discriminator.state_dict().items()}
    v_sd = O.vgg_standin_state_dict(1234, 1)
    lr, hr = torch.rand(4, 3, 96, 96) * 2 - 1, torch.rand(4, 3, 384, 384) * 2 - 1
    noise = [torch.rand(4, 1, 24, 24) for _ in range(3)]
    got = T.train_step(lr.to(dev), hr.to(dev), [n.to(dev) for n in noise])
    torch.cuda.synchronize()
    # gradients left in the arenas: the discriminator's from the D step (:180), the generator's from the G step (:195)
    named_d = [("d." + k, p.grad) for k, p in T.discriminator.named_parameters()]
    named_g = [("g." + k, p.grad) for k, p in T.generator.named_parameters()]

    def oracle(q):
        ref = {}
        want = O.train_step({k: v.clone() for k, v in g0.items()}, {k: v.clone() for k, v in d0.items()}, v_sd, lr, hr, noise,
                            {}, {}, grads_out=ref, q=q)
        return want, ref

    def losses(tag, want, tol):
        for k in want:
            e = report("cfg1.%s.%s" % (tag, k), abs(float(got[k]) - float(want[k])) / abs(float(want[k])))
            assert e < tol, (k, float(got[k]), float(want[k]))

    if cdn == "f32":
        want, ref = oracle(None)
        losses("f32", want, 1e-3)
        bad = check_grads("cfg1.f32.grad", named_d, ref, t_tensor=F32_D_GRAD, t_cos=0.9995)
        bad += check_grads("cfg1.f32.grad", named_g, ref, t_tensor=F32_G_GRAD, t_cos=F32_G_COS)
        assert not bad, bad
        return
    want, ref = oracle(O.Q_BF16)
    losses("bf16q", want, STEPQ_LOSS)
    bad = check_grads("cfg1.bf16q.grad", named_d, ref, t_tensor=STEPQ_D_GRAD, t_slope=STEPQ_SLOPE, t_cos=STEPQ_COS_D, t_norm=STEP_NORM)
    bad += check_grads("cfg1.bf16q.grad", named_g, ref, t_tensor=STEPQ_G_GRAD, t_slope=STEPQ_SLOPE, t_cos=STEPQ_COS_G, t_norm=STEP_NORM)
    assert not bad, bad
    want, ref = oracle(None)
    losses("bf16", want, STEP_BF16_LOSS)
    bad = check_grads("cfg1.bf16.grad", named_d, ref, t_tensor=STEP_BF16_D_GRAD, t_slope=STEP_BF16_SLOPE, t_norm=STEP_NORM)
    bad += check_grads("cfg1.bf16.grad", named_g, ref, t_tensor=STEP_BF16_G_GRAD, t_slope=STEP_BF16_SLOPE, t_norm=STEP_NORM)
    bad += check_grads("cfg1.bf16.grad.all", named_d + named_g, ref, t_tensor=STEP_BF16_G_GRAD, t_slope=STEP_BF16_SLOPE_ALL, t_cos=STEP_BF16_COS)
    assert not bad, bad


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


def test_generator_cfg5_full_size_vs_oracle(pkg):
    """BASELINE configs[4]: 12 residual blocks, three pixel-shuffle stages, 128x128 -> 1024x1024 at FULL size: fp32 mode
    against the oracle (1e-3), the 16-bit MFMA mode against the same reference (bounded like the inference shapes)."""
    dev = select("hip")
    torch.manual_seed(8)
    G32 = pkg.Generator(ns(n_filters=64, n_layers=12, n_upsample=3), compute_dtype="f32")
    sd = {k: v.clone() for k, v in G32.state_dict().items()}
    x = torch.rand(1, 3, 128, 128) * 2 - 1
    want = O.generator_forward(sd, x)
    assert want.shape == (1, 3, 1024, 1024)
    with torch.no_grad():
        y32 = G32.to(dev).eval()(x.to(dev)).cpu()
    assert report("cfg5.f32.sr", relerr(y32, want)) < 1e-3
    for cdn, t_mean, t_max in (("bf16", 1e-3, 6e-3), ("f16", 2e-4, 1.5e-3)):      # ~2x measured (fp16: the dtype configs[4] names)
        G16 = pkg.Generator(ns(n_filters=64, n_layers=12, n_upsample=3), compute_dtype=cdn)
        G16.load_state_dict(sd)
        with torch.no_grad():
            y16 = G16.to(dev).eval()(x.to(dev)).cpu()
        assert report("cfg5.%s.mean_abs" % cdn, float((y16 - want).abs().mean())) < t_mean
        assert report("cfg5.%s.max_abs" % cdn, float((y16 - want).abs().max())) < t_max
