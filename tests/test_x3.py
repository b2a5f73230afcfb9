"""The x3 compute mode (FSR_X3: split-bf16 storage, three bf16 MFMAs per product, f32 accumulate) -- the fast mode that is held
to the FP32 gates: every kernel against the plain PyTorch fp32 op it replaces, the modules and one training iteration against
the fp32 oracle, all through the C ABI on both backends (tests/backend.py).  Tolerances are the f32 mode's 1e-3 on outputs and
losses (north_star) and a few 1e-5 at operator level (bf16 pairs carry 16 mantissa bits, the dropped lo x lo term is 2^-16)."""
import importlib
import types
import warnings

import pytest
import torch
import torch.nn.functional as F

from backend import BACKENDS, L, check_grads, ops, relerr, report, select
from oracle import srgan_cpu as O

OP_TOL = 5e-5       # operator level, max-norm relative (measured 4e-6 .. 8e-6)


def ns(**k):
    return types.SimpleNamespace(**k)


@pytest.fixture(params=BACKENDS)
def dev(request):
    return select(request.param)


@pytest.fixture()
def cd():
    return ops.Compute("x3")


def _nhwc(x, cd, dev):
    return ops.to_storage(cd, x.permute(0, 2, 3, 1).contiguous()).to(dev)


def _nchw(y, cd):
    return ops.from_storage(cd, y.cpu()).permute(0, 3, 1, 2)


def leaf(t):
    return t.detach().clone().requires_grad_(True)


def test_x3_storage_layout_and_roundtrip(cd):
    """Per pixel and 32-channel group: 64 bytes of bf16(v), then 64 bytes of bf16(v - hi); decode(encode(v)) is v to 2^-16."""
    torch.manual_seed(0)
    v = torch.randn(2, 3, 5, 64) * torch.logspace(-6, 3, 64)
    t = ops.x3_encode(v)
    assert t.dtype == torch.float32 and t.shape == v.shape and t.data_ptr() % 128 == 0
    raw = t.view(torch.bfloat16).view(2, 3, 5, 2, 2, 32)          # [group][hi | lo][32]
    hi = v.to(torch.bfloat16)
    assert torch.equal(raw[..., 0, :].reshape(v.shape), hi)
    assert torch.equal(raw[..., 1, :].reshape(v.shape), (v - hi.float()).to(torch.bfloat16))
    back = ops.x3_decode(t)
    assert float(((back - v).abs() / v.abs()).max()) < 2.0 ** -16
    with pytest.raises(ValueError):
        ops.x3_encode(torch.zeros(1, 1, 1, 48))


@pytest.mark.parametrize("stride,cin,cout,ps", [(1, 32, 64, False), (1, 64, 64, False), (2, 64, 64, False), (1, 32, 128, True), (1, 64, 3, False),
                                                (1, 64, 128, False), (1, 64, 256, True), (2, 64, 128, False), (1, 128, 128, False), (2, 128, 128, False)])
def test_x3_conv_fwd_dgrad_wgrad(dev, cd, stride, cin, cout, ps):
    torch.manual_seed(1)
    big = dev.type == "cuda"
    n, h, w = (3, 37, 45) if big else (1, 7, 19)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) * 0.1
    bias = torch.randn(cout) * 0.1
    xd = _nhwc(x, cd, dev)
    wpk = ops.packed_filter(cd, wt.to(dev), L.PACK_FWD_PS if ps else L.PACK_FWD, cin)
    y, _, stats = ops.conv3x3_raw(cd, xd, wpk, cout, stride=stride, bias=bias.to(dev), pixel_shuffle=ps, want_stats=(not ps and cout % 16 == 0),
                                  out_f32=(cout == 3))
    ref = F.conv2d(x, wt, bias, stride, 1)
    refo = F.pixel_shuffle(ref, 2) if ps else ref
    yo = y.float().cpu().permute(0, 3, 1, 2) if cout == 3 else _nchw(y, cd)
    assert report("x3.conv.fwd", relerr(yo, refo)) < OP_TOL
    if stats is not None:
        s = stats.cpu()
        assert relerr(s[..., 0], ref.sum((2, 3))) < 1e-4
        assert relerr(s[..., 1], (ref * ref).sum((2, 3))) < 1e-4
    g = torch.randn_like(refo)
    xr, wr = leaf(x), leaf(wt)
    yr = F.conv2d(xr, wr, None, stride, 1)
    yr = F.pixel_shuffle(yr, 2) if ps else yr
    yr.backward(g)
    cpad_out = cd.pad(cout)
    gd = torch.zeros(n, g.shape[2], g.shape[3], (cpad_out // 4) if ps else cpad_out)
    gd[..., :g.shape[1]] = g.permute(0, 2, 3, 1)
    gd = ops.to_storage(cd, gd).to(dev)
    wpk_d = ops.packed_filter(cd, wt.to(dev), L.PACK_DGRAD_PS if ps else L.PACK_DGRAD, cpad_out)
    dx, _, _ = ops.conv3x3_raw(cd, gd, wpk_d, cin, mode=L.CONV_DGRAD, out_hw=(h, w), stride=stride, in_pixel_shuffled=ps)
    assert report("x3.conv.dgrad", relerr(_nchw(dx, cd), xr.grad)) < OP_TOL
    if stride == 2 and cin % 64 == 0:      # conv_s2d3's x3 form with the fused LeakyReLU(0.2) mask (the saved forward input)
        assert L.lib().fsr_last_kernel().decode() == "conv_s2d3_kernel<x3>"
        mask = torch.randn(n, cin, h, w)
        dxm, _, _ = ops.conv3x3_raw(cd, gd, wpk_d, cin, mode=L.CONV_DGRAD, out_hw=(h, w), stride=2, dact_mask=_nhwc(mask, cd, dev), dact_slope=0.2)
        want = xr.grad * torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, 0.2))
        assert report("x3.conv.dgrad_masked", relerr(_nchw(dxm, cd), want)) < OP_TOL
    dw = ops.conv3x3_wgrad_raw(cd, xd, gd, cout, cin, stride, dy_pixel_shuffled=ps)
    assert report("x3.conv.wgrad", relerr(dw, wr.grad)) < OP_TOL


@pytest.mark.parametrize("cus", [1, 3])
def test_x3_thin_kernel_head_and_image_gradient(dev, cd, cus, monkeypatch):
    """conv64_thin_kernel<x3> (round 6): the 64 -> 3 ends of the networks on x3 storage -- the generator's head with its tanh
    (model.py:102-110; float and uint8 output) and a 64 -> 3 data gradient with the per-channel scale of the VGG normalisation --
    streamed as two channel-group passes per tile (hi hi + lo hi + hi lo on the 16-bit kernel's four fragment reads), on maps
    that are not multiples of the 16 x 16 tile, one and several tile ranges per workgroup."""
    monkeypatch.setenv("FSR_PERSIST_CUS", str(cus))
    torch.manual_seed(4)
    big = dev.type == "cuda"
    n, h, w = (3, 37, 45) if big else (2, 17, 19)
    x = torch.randn(n, 64, h, w)
    wt = torch.randn(3, 64, 3, 3) * 0.1
    bias = torch.randn(3) * 0.1
    xd = _nhwc(x, cd, dev)
    wpk = ops.packed_filter(cd, wt.to(dev), L.PACK_FWD, 64)
    y, _, _ = ops.conv3x3_raw(cd, xd, wpk, 3, bias=bias.to(dev), act=L.ACT_TANH, out_f32=True)
    assert L.lib().fsr_last_kernel().decode() == "conv64_thin_kernel<x3>", L.lib().fsr_last_kernel()
    ref = torch.tanh(F.conv2d(x, wt, bias, 1, 1))
    assert y.dtype == torch.float32 and y.shape == (n, h, w, 3)
    assert report("x3.thin.head", relerr(y.cpu().permute(0, 3, 1, 2), ref)) < OP_TOL
    yu, _, _ = ops.conv3x3_raw(cd, xd, wpk, 3, bias=bias.to(dev), act=L.ACT_TANH, out_u8=True)
    assert L.lib().fsr_last_kernel().decode() == "conv64_thin_kernel<x3>"
    want_u8 = (((y.cpu() + 1.0) / 2.0) * 255.0).to(torch.uint8)          # inference.py:53-56 on the float output of the same kernel
    assert yu.dtype == torch.uint8 and torch.equal(yu.cpu(), want_u8)
    # the data gradient of a 3 -> 64 first layer: 64 gradient channels in, 3 float channels out, scaled per channel
    w1 = torch.randn(64, 3, 3, 3) * 0.1
    g = torch.randn(n, 64, h, w)
    xr = leaf(torch.randn(n, 3, h, w))
    F.conv2d(xr, w1, None, 1, 1).backward(g)
    scale = torch.tensor([2.0, 0.5, 1.5])
    wpk_d = ops.packed_filter(cd, w1.to(dev), L.PACK_DGRAD, 64)
    dx, _, _ = ops.conv3x3_raw(cd, _nhwc(g, cd, dev), wpk_d, 3, mode=L.CONV_DGRAD, out_hw=(h, w), out_f32=True, oscale=scale.to(dev))
    assert L.lib().fsr_last_kernel().decode() == "conv64_thin_kernel<x3>", L.lib().fsr_last_kernel()
    assert report("x3.thin.image_gradient", relerr(dx.cpu().permute(0, 3, 1, 2), xr.grad * scale.view(1, 3, 1, 1))) < OP_TOL


def test_x3_conv_epilogues_autograd(dev, cd):
    """Conv3x3Fn with the fused epilogues in x3 storage: bias + PReLU + PixelShuffle (pre-activation copy, act_bwd with its
    bias / slope reductions), LeakyReLU applied by the consumer's data-gradient mask, the residual skip added in the
    data-gradient epilogue, InstanceNorm forward / backward -- against torch autograd on the same values."""
    torch.manual_seed(2)
    n, c, h, w = (2, 64, 18, 22) if dev.type == "cuda" else (1, 32, 6, 9)
    x = torch.randn(n, c, h, w)
    w1, b1, a1 = torch.randn(4 * c, c, 3, 3) * 0.05, torch.randn(4 * c) * 0.1, torch.tensor([0.25])
    w2 = torch.randn(c, c, 3, 3) * 0.05
    xd = leaf(_nhwc(x, cd, dev))
    p = [leaf(t.to(dev)) for t in (w1, b1, a1, w2)]
    up, _ = ops.conv3x3(xd, p[0], p[1], p[2], ops.ConvCfg(cd, act=L.ACT_PRELU, pixel_shuffle=True))
    u, st, skip = ops.conv3x3(up, p[3], None, None, ops.ConvCfg(cd, stats=True, n_alias=1))
    y = ops.instnorm_act(u, st, skip, None, cd, L.ACT_LEAKY, 0.2)
    r = torch.randn(n, c, 2 * h, 2 * w)
    # the seed gradient is an x3 tensor too
    y.backward(_nhwc(r, cd, dev))
    xr = leaf(x)
    q = [leaf(t) for t in (w1, b1, a1, w2)]
    upr = F.prelu(F.pixel_shuffle(F.conv2d(xr, q[0], q[1], 1, 1), 2), q[2])
    ur = F.conv2d(upr, q[3], None, 1, 1)
    yr = F.leaky_relu(F.instance_norm(ur), 0.2) + upr
    yr.backward(r)
    assert report("x3.epi.y", relerr(_nchw(y.detach(), cd), yr)) < 1e-4
    assert report("x3.epi.dx", relerr(_nchw(xd.grad, cd), xr.grad)) < 2e-4
    for name, a, b in zip(("w1", "b1", "a1", "w2"), p, q):
        # (the PReLU slope gradient is ONE cancelling sum over the layer: its error is relative to the sum of magnitudes)
        assert report("x3.epi.d" + name, relerr(a.grad, b.grad)) < (2e-3 if name == "a1" else 2e-4), name


def test_x3_pool_conv1x1_smoothl1_add(dev, cd):
    torch.manual_seed(3)
    n, c, h, w = (2, 64, 12, 20) if dev.type == "cuda" else (1, 32, 4, 6)
    x, t = torch.randn(n, c, h, w), torch.randn(n, c, h // 2, w // 2)
    w1, b1 = torch.randn(1, c, 1, 1) * 0.1, torch.randn(1)
    # (an x3 tensor must have ONE consumer in autograd -- torch would accumulate two gradient containers with float adds;
    # the modules route their skip connections through Conv3x3Fn's aliases for exactly that reason -- so: two graphs)
    for relu_mask in (False, True):
        for head in ("l1", "c1"):
            xd = leaf(_nhwc(x, cd, dev))
            wd, bd = leaf(w1.to(dev)), leaf(b1.to(dev))
            pooled = ops.maxpool2(xd, cd, relu_mask=relu_mask)
            xr, wr, br = leaf(x), leaf(w1), leaf(b1)
            pr = F.max_pool2d(xr, 2)
            if head == "l1":
                loss, lref = ops.smooth_l1(pooled, _nhwc(t, cd, dev), cd=cd), F.smooth_l1_loss(pr, t)
            else:
                loss, lref = ops.conv1x1_to_logits(pooled, wd, bd, cd).sum(), F.conv2d(pr, wr, br).sum()
            loss.backward()
            lref.backward()
            # relu_mask: the pool's backward also applies the backward of the ReLU that produced x (dx = 0 where x <= 0)
            gref = xr.grad * (x > 0) if relu_mask else xr.grad
            assert report("x3.pool.loss", abs(float(loss.detach()) - float(lref.detach())) / abs(float(lref.detach()))) < 1e-5
            assert report("x3.pool.dx", relerr(_nchw(xd.grad, cd), gref)) < OP_TOL
            if head == "c1":
                assert relerr(wd.grad, wr.grad) < OP_TOL and relerr(bd.grad, br.grad) < OP_TOL
    a, b = torch.randn(n, c, h, w), torch.randn(n, c, h, w) * 1e-3
    s = ops.add(cd, _nhwc(a, cd, dev), _nhwc(b, cd, dev))
    assert relerr(_nchw(s, cd), a + b) < 3e-5


def test_x3_generator_discriminator_vs_oracle(dev):
    """G (32 filters, 1 block) -> D, forward and every gradient against the fp32 oracle at the F32 mode's gates."""
    pkg = importlib.import_module("fast-srgan_amd")
    torch.manual_seed(5)
    big = dev.type == "cuda"
    nf = 64 if big else 32
    G = pkg.Generator(ns(n_filters=nf, n_layers=2 if big else 1), compute_dtype="x3")
    D = pkg.Discriminator(ns(n_filters=nf, n_layers=7), compute_dtype="x3")
    gsd = {k: v.clone() for k, v in G.state_dict().items()}
    dsd = {k: v.clone() for k, v in D.state_dict().items()}
    G.to(dev), D.to(dev)
    x = torch.rand(2, 3, 24, 40) * 2 - 1 if big else torch.rand(1, 3, 8, 12) * 2 - 1
    sr = G(x.to(dev))
    assert sr.dtype == torch.float32
    logits = D(sr)
    r = torch.randn(logits.shape)
    (logits * r.to(dev)).sum().backward()
    gp = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    dp = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    sr_ref = O.generator_forward(gp, x)
    lg_ref = O.discriminator_forward(dp, sr_ref)
    grads = torch.autograd.grad((lg_ref * r).sum(), list(gp.values()) + list(dp.values()))
    ref = dict(zip(["g." + k for k in gp] + ["d." + k for k in dp], grads))
    assert report("modules.x3.sr.%s" % dev.type, relerr(sr, sr_ref)) < 1e-3
    assert report("modules.x3.logits.%s" % dev.type, relerr(logits, lg_ref)) < 1e-3
    named = [("g." + k, p.grad) for k, p in G.named_parameters()] + [("d." + k, p.grad) for k, p in D.named_parameters()]
    # gradients: 2x what the MI355X measures at 64 filters (tensors 2.0 %, slopes 3.6 %, cosine 0.99982 -- the f32 mode's gates
    # are 1 % / 1 % / 0.9999: x3 products carry 2^-17 instead of 2^-24, and LeakyReLU(0.01) decisions near zero flip)
    # (tensors of fewer than 64 elements -- the head's 3 biases -- are cancelling sums over a whole layer: 10 %)
    bad = check_grads("modules.x3.grad.%s" % dev.type, [(k, g) for k, g in named if g.numel() >= 64 or g.numel() == 1], ref,
                      t_tensor=4e-2, t_slope=8e-2, t_cos=0.9995)
    bad += check_grads("modules.x3.grad_small.%s" % dev.type, [(k, g) for k, g in named if 1 < g.numel() < 64], ref, t_tensor=0.1)
    assert not bad, bad


@pytest.mark.parametrize("mode", ["x3", "x3v"])
def test_x3_train_step_vs_oracle(dev, mode):
    """One whole iteration (trainer.py:171-196) in x3 mode against the fp32 oracle: four losses to 1e-3 (measured ~1e-5),
    one AdamW update of both networks in the mean.  mode x3v: the same with the frozen perceptual network in fp16 (x3 Generator /
    Discriminator, the dynamic loss scale of the fp16 mode) -- same gates; the content loss is the one quantity fp16 touches."""
    pkg = importlib.import_module("fast-srgan_amd")
    if mode == "x3" and dev.type != "cuda":
        pytest.skip("emulator: the x3v case runs the same x3 Generator / Discriminator kernels (35 s saved); x3 VGG kernels: the operator tests")
    torch.manual_seed(9)
    big = dev.type == "cuda"
    nf, wd, nl = (64, 2, 2) if big else (32, 2, 1)
    cfg = ns(experiment=ns(name="t", seed=1234), generator=ns(n_filters=nf, n_layers=nl), discriminator=ns(n_filters=nf, n_layers=7),
             training=ns(compiled=False, device=str(dev), log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4, discriminator_lr=1e-4,
                         batch_size=2, compute_dtype=mode))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        V = pkg.VGG19(compute_dtype=mode, width_div=wd, seed=1234)
        T = pkg.Trainer(cfg, perceptual_network=V)
    assert T.generator.compute.name == "x3" and T.discriminator.compute.name == "x3"
    assert V.compute.name == ("f16" if mode == "x3v" else "x3") and (T.loss_scale_state() is not None) == (mode == "x3v")
    g_sd = {k: v.detach().cpu().clone() for k, v in T.generator.state_dict().items()}
    d_sd = {k: v.detach().cpu().clone() for k, v in T.discriminator.state_dict().items()}
    g0, d0 = {k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}
    v_sd = O.vgg_standin_state_dict(1234, wd)
    b, s = (2, 16) if big else (1, 8)
    lr, hr = torch.rand(b, 3, s, s) * 2 - 1, torch.rand(b, 3, 4 * s, 4 * s) * 2 - 1
    noise = [torch.rand(b, 1, s // 4, s // 4) for _ in range(3)]
    got = T.train_step(lr.to(dev), hr.to(dev), [t.to(dev) for t in noise])
    want = O.train_step(g_sd, d_sd, v_sd, lr, hr, noise, {}, {})
    for k in want:
        e = report("step.%s.%s.%s" % (mode, dev.type, k), abs(float(got[k]) - float(want[k])) / abs(float(want[k])))
        assert e <= 1e-3, (k, float(got[k]), float(want[k]))
    for sd_ref, sd0, mod in ((g_sd, g0, T.generator), (d_sd, d0, T.discriminator)):
        for k, p in mod.state_dict().items():
            upd = (sd_ref[k] - sd0[k]).abs().mean()
            # (Adam normalises every element's update to ~lr: an element whose tiny gradient changes sign under 2^-17 products moves
            # by 2 lr -- measured up to 0.10 of the mean update on a bias vector; the f32 mode's bound is 0.1)
            assert (p.cpu() - sd_ref[k]).abs().mean() <= 0.2 * upd + 1e-12, k
    if mode == "x3v":
        assert T.loss_scale_state()[1] == 0          # nothing overflowed: no skipped update


def test_x3v_overflow_in_the_fp16_perceptual_branch_skips_the_iteration(dev):
    """x3v under a loss scale that fp16 cannot carry (2^100: the content-loss seed alone overflows): the perceptual network's backward
    produces inf / NaN, they reach the x3 generator's gradient arena, and the device-side check skips the generator's update of the
    iteration, leaves its parameters, moments and step counter untouched and halves the scale -- the next iterations do the same
    until the scale fits.  Nothing is poisoned for good.  (Until round 6 the NaNs never arrived: a data-gradient epilogue spelled
    its activation max(v, 0) + slope * min(v, 0), which maps NaN to 0 -- fsr_common.h: act_slope.)"""
    pkg = importlib.import_module("fast-srgan_amd")
    torch.manual_seed(12)
    cfg = ns(experiment=ns(name="t", seed=1234), generator=ns(n_filters=32, n_layers=1), discriminator=ns(n_filters=32, n_layers=7),
             training=ns(compiled=False, device=str(dev), log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4, discriminator_lr=1e-4,
                         batch_size=1, compute_dtype="x3v", loss_scale=2.0 ** 100))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        T = pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype="x3v", width_div=2, seed=1234))
    assert T.dynamic_loss_scale and T.loss_scale_state() == (2.0 ** 100, 0)
    g0, d0 = T.optim_generator.flat_param.clone(), T.optim_discriminator.flat_param.clone()
    lr, hr = torch.rand(1, 3, 8, 8) * 2 - 1, torch.rand(1, 3, 32, 32) * 2 - 1
    noise = [torch.rand(1, 1, 2, 2) for _ in range(3)]
    out = T.train_step(lr.to(dev), hr.to(dev), [t.to(dev) for t in noise])
    assert all(torch.isfinite(v).all() for v in out.values())          # the LOSSES are forward quantities: finite
    scale, skipped = T.loss_scale_state()
    assert skipped == 1 and scale == 2.0 ** 99
    # the generator -- the network the perceptual gradient flows into -- did not move: parameters, moments, step counter
    assert torch.equal(T.optim_generator.flat_param, g0)
    assert float(T.optim_generator.step_dev) == 0.0 and float(T.optim_generator.exp_avg.abs().sum()) == 0.0
    # (the discriminator stepped earlier in the iteration, trainer.py:180, on its own finite x3 gradients -- correctly: the scale is
    # divided out on the device; the shared flag only couples the other way, a D overflow also skipping G)
    assert float(T.optim_discriminator.step_dev) == 1.0 and torch.isfinite(T.optim_discriminator.flat_param).all()
    assert not torch.equal(T.optim_discriminator.flat_param, d0)
    # x3v does not grow the scale (only an overflow moves it): 2^99 stays after clean iterations would have been counted
    assert T.loss_scale_growth == 1.0


@pytest.mark.parametrize("case", ["s1_relu_pool", "s1_stats", "s1_narrow_addend", "s1_narrow_stats", "s1_mask", "s2_stats", "s2_bias_leaky_mask", "rows12",
                                  "s1_ps_prelu_preact", "s2_narrow_stats"])
def test_x3_conv_tall3_forms(dev, cd, case, monkeypatch):
    """conv_tall3.hip in x3 form (three virtual chunks per channel group: (x_hi, w_hi), (x_lo, w_hi), (x_hi, w_lo)) with few
    workgroups, so that every workgroup walks several tiles and the DMA stream runs on across tile and channel-block boundaries:
    ReLU + fused max-pool, InstanceNorm statistics (bit-reproducible), the 64-channel block with a skip-gradient addend, a
    sign mask (read from the hi parts), the stride-2 form (four parity planes) and 12-row tiles; filters through FilterSpec,
    i.e. in the x3 stage-contiguous pack the kernel asks for."""
    monkeypatch.setenv("FSR_PERSIST_CUS", "3" if dev.type == "cuda" else "1")
    torch.manual_seed(12)
    big = dev.type == "cuda"
    stride = 2 if case.startswith("s2") else 1
    cin, cout = {"s1_narrow_addend": (128, 64), "s1_narrow_stats": (64, 64), "s2_narrow_stats": (64, 64), "s2_bias_leaky_mask": (128, 256)}.get(case, (128, 128))
    if not big:
        cin = 64 if case != "s1_narrow_addend" else 64      # (emulated lanes are slow: 64 logical = 128 physical channels)
    n = 3 if big else 1
    h, w = ((37, 45) if stride == 1 else (41, 38)) if big else ((18, 20) if stride == 1 else (19, 22))
    if case == "rows12":
        h, w = 24, (40 if big else 17)
    if case == "s1_relu_pool":
        h, w = (36, 44) if big else (18, 20)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) * 0.05
    xd = _nhwc(x, cd, dev)
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    spec = ops.FilterSpec(wt.to(dev), L.PACK_FWD, cin)
    kw = {}
    bias = None
    ref = F.conv2d(x, wt, None, stride, 1)
    if case == "s1_relu_pool":
        bias = torch.randn(cout) * 0.1
        kw = dict(bias=bias.to(dev), act=L.ACT_RELU, pool2=True)
        ref = F.max_pool2d(F.relu(F.conv2d(x, wt, bias, 1, 1)), 2)
    elif case in ("s1_stats", "s1_narrow_stats", "s2_stats", "s2_narrow_stats", "rows12"):
        kw = dict(want_stats=True)
    elif case == "s1_narrow_addend":
        add = torch.randn(n, cout, oh, ow)
        kw = dict(dact_mask=_nhwc(add, cd, dev), dact_add=True)
        ref = ref + add
    elif case == "s1_mask":
        mask = torch.randn(n, cout, oh, ow)
        kw = dict(dact_mask=_nhwc(mask, cd, dev), dact_slope=0.0, act=L.ACT_NONE)
        ref = ref * (mask > 0)
    elif case == "s1_ps_prelu_preact":      # the generator's up-sampling convolution: bias + PixelShuffle(2) + PReLU, pre-activation kept
        bias = torch.randn(cout) * 0.1
        slope_t = torch.tensor([-0.3])      # (a trained PReLU weight may have any sign)
        spec = ops.FilterSpec(wt.to(dev), L.PACK_FWD_PS, cin)
        kw = dict(bias=bias.to(dev), act=L.ACT_PRELU, prelu=slope_t.to(dev), pixel_shuffle=True, want_preact=True)
        pre_ref = F.pixel_shuffle(F.conv2d(x, wt, bias, 1, 1), 2)
        ref = F.prelu(pre_ref, slope_t)
    elif case == "s2_bias_leaky_mask":
        bias = torch.randn(cout) * 0.1
        mask = torch.randn(n, cout, oh, ow)
        kw = dict(bias=bias.to(dev), act=L.ACT_LEAKY, slope=0.2, dact_mask=_nhwc(mask, cd, dev), dact_slope=0.5)
        pre = F.conv2d(x, wt, bias, 2, 1)
        ref = F.leaky_relu(pre * torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, 0.5)), 0.2)
    y, pre, stats = ops.conv3x3_raw(cd, xd, spec, cout, stride=stride, **kw)
    name = L.lib().fsr_last_kernel().decode()
    assert name.startswith("conv_tall3_kernel<x3"), name
    if case == "s1_ps_prelu_preact":
        assert relerr(_nchw(pre, cd), pre_ref) < OP_TOL
    assert ("s2>" in name) == (stride == 2) and ("stats" in name) == ("want_stats" in kw), name
    assert report("x3.tall3.%s" % case, relerr(_nchw(y, cd), ref)) < OP_TOL
    if stats is not None:
        st = stats.cpu()
        assert relerr(st[..., 0], ref.sum((2, 3))) < 1e-4 and relerr(st[..., 1], (ref * ref).sum((2, 3))) < 1e-4
        _, _, stats2 = ops.conv3x3_raw(cd, xd, spec, cout, stride=stride, **kw)
        assert torch.equal(stats2.cpu(), st)                    # order-fixed partial slots: bit-reproducible


@pytest.mark.parametrize("act", ["leaky", "prelu"])
def test_x3_first_layer_kernels(dev, cd, act):
    """Conv2d(3 -> 64) straight from the float image (conv_c3.hip) on x3 storage: exact-f32 arithmetic, 16 + 16-byte hi / lo stores
    of the 64-channel block, the pre-activation copy of a PReLU layer, weight / bias / slope gradients and the image gradient."""
    torch.manual_seed(4)
    n, h, w = (2, 37, 45) if dev.type == "cuda" else (1, 9, 20)
    img = torch.rand(n, 3, h, w) * 2 - 1
    wt, b, a1 = torch.randn(64, 3, 3, 3) * 0.2, torch.randn(64) * 0.1, torch.tensor([0.25])
    xd = img.to(dev).requires_grad_(True)
    p = [leaf(t.to(dev)) for t in (wt, b, a1)]
    cfg = ops.ConvCfg(cd, act=L.ACT_PRELU, image_in=True) if act == "prelu" else ops.ConvCfg(cd, act=L.ACT_LEAKY, slope=0.2, image_in=True)
    y, _ = ops.conv3x3(xd, p[0], p[1], p[2] if act == "prelu" else None, cfg)
    r = torch.randn(n, 64, h, w)
    y.backward(_nhwc(r, cd, dev))
    xr = leaf(img)
    q = [leaf(t) for t in (wt, b, a1)]
    pre = F.conv2d(xr, q[0], q[1], 1, 1)
    yr = F.prelu(pre, q[2]) if act == "prelu" else F.leaky_relu(pre, 0.2)
    yr.backward(r)
    assert report("x3.c3.%s.y" % act, relerr(_nchw(y.detach(), cd), yr)) < 2e-5      # f32 arithmetic, then the 16-bit pair: 2^-17
    assert report("x3.c3.%s.dx" % act, relerr(xd.grad, xr.grad)) < OP_TOL
    assert relerr(p[0].grad, q[0].grad) < OP_TOL and relerr(p[1].grad, q[1].grad) < OP_TOL
    if act == "prelu":
        assert relerr(p[2].grad, q[2].grad) < 2e-3


def test_x3_argument_checks(dev, cd):
    """Refusals of the x3 entry points: nothing is enqueued, fsr_last_error says why -- channel counts that are not whole hi / lo
    groups of 32, tensors whose base is not 128-byte aligned (the split addressing works on 128-byte blocks), filter packs of the
    first-layer kernels in the wrong element type are impossible by construction (float image: checked by size)."""
    import ctypes
    lib = L.lib()
    x = ops.to_storage(cd, torch.randn(1, 4, 6, 64)).to(dev)
    out = ops._empty((1, 4, 6, 64), torch.float32, dev)
    st = torch.zeros(1, 64, 2, dtype=torch.float32, device=dev)
    # an unaligned view: 16 channels (64 bytes) into a pixel
    flat = ops._empty((1 * 4 * 6 * 64 + 64,), torch.float32, dev)
    mis = flat[16:16 + 4 * 6 * 64].view(1, 4, 6, 64)
    assert mis.data_ptr() % 128 != 0
    rc = lib.fsr_instnorm_act_fwd(L.FSR_X3, mis.data_ptr(), st.data_ptr(), None, L.ACT_NONE, 0.0, None, out.data_ptr(), 1, 24, 64, ops._stream())
    assert rc == -2 and b"128-byte aligned" in lib.fsr_last_error()
    rc = lib.fsr_add(L.FSR_X3, x.data_ptr(), mis.data_ptr(), out.data_ptr(), x.numel(), ops._stream())
    assert rc == -2 and b"128-byte aligned" in lib.fsr_last_error()
    # 48 channels: not whole groups of 32
    rc = lib.fsr_maxpool2_fwd(L.FSR_X3, x.data_ptr(), out.data_ptr(), None, 1, 4, 6, 48, ops._stream())
    assert rc == -2 and b"multiple of 32" in lib.fsr_last_error()
    d = L.ConvDesc(L.FSR_X3, L.CONV_FWD, 1, 4, 6, 48, 4, 6, 64, 1, L.ACT_NONE, 0.0, 0, 0, 0, 0, 0, 0)
    rc = lib.fsr_conv3x3(ctypes.byref(d), x.data_ptr(), x.data_ptr(), None, None, None, None, 0.0, out.data_ptr(), None, None, None, ops._stream())
    assert rc == -2 and b"multiple of 32" in lib.fsr_last_error()
    d = L.ConvDesc(L.FSR_X3, L.CONV_FWD, 1, 4, 6, 64, 4, 6, 64, 1, L.ACT_NONE, 0.0, 0, 0, 0, 0, 0, 0)
    rc = lib.fsr_conv3x3(ctypes.byref(d), mis.data_ptr(), x.data_ptr(), None, None, None, None, 0.0, out.data_ptr(), None, None, None, ops._stream())
    assert rc == -2 and b"128-byte aligned" in lib.fsr_last_error()
    wd = L.WgradDesc(L.FSR_X3, 1, 4, 6, 48, 48, 4, 6, 64, 64, 1, 0)
    assert lib.fsr_conv3x3_wgrad_workspace(ctypes.byref(wd)) == 0
    # the grouped weight gradient has no x3 form: refused, the caller launches the layers one by one (ops._wgrad_defer)
    wd = L.WgradDesc(L.FSR_X3, 1, 4, 6, 64, 64, 4, 6, 64, 64, 1, 0)
    arr = (ctypes.c_void_p * 1)(x.data_ptr())
    ws = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
    rc = lib.fsr_conv3x3_wgrad_grouped(ctypes.byref(wd), 1, arr, arr, arr, ws.data_ptr(), ops._stream())
    assert rc == -2
    # smooth L1 on x3 containers needs the caller to say so (a float32 container looks like float data)
    with pytest.raises(L.FsrError):
        ops.smooth_l1(x.to(torch.bfloat16), x.to(torch.bfloat16), cd=cd)
