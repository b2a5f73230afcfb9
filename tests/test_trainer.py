"""Step-level parity (trainer.py:171-196) and the host logic around it."""
import importlib
import os
import random
import types
import warnings

import numpy as np
import pytest
import torch

from backend import BACKENDS, L, ops, relerr, report, select
from conftest import load_npz, sd_from
from oracle import srgan_cpu as O


def ns(**k):
    return types.SimpleNamespace(**k)


def _cfg(device, cdt, nf=16, n_layers=1):
    return ns(experiment=ns(name="t", seed=1234), generator=ns(n_filters=nf, n_layers=n_layers),
              discriminator=ns(n_filters=nf, n_layers=7),
              training=ns(compiled=False, device=device, log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4,
                          discriminator_lr=1e-4, batch_size=2, compute_dtype=cdt))


def _trainer(pkg, dev, cdt, z=None, nf=16, n_layers=1, width_div=4):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        V = pkg.VGG19(compute_dtype=cdt, width_div=width_div, seed=1234)
        T = pkg.Trainer(_cfg(str(dev), cdt, nf, n_layers), perceptual_network=V)
    if z is not None:
        T.generator.load_state_dict(sd_from(z, "g0."))
        T.discriminator.load_state_dict(sd_from(z, "d0."))
    return T


@pytest.mark.parametrize("backend", BACKENDS)
def test_two_train_steps_match_reference_trainer_f32(pkg, backend):
    """Two iterations of the REFERENCE's Trainer.train (golden, recorded label noise) vs Trainer.train_step on the
    HIP path in exact-f32 MFMA mode: the four logged losses within 1e-3 relative, parameter updates in the mean.
    (Also runs on the host emulator: the same kernels, lane for lane, in the CPU suite.)"""
    dev = select(backend)
    z = load_npz("train_steps.npz")
    T = _trainer(pkg, dev, "f32", z, width_div=int(z["vgg_width_div"]))
    for it in range(2):
        noise = [torch.from_numpy(z[f"noise{3 * it + j}"]).to(dev) for j in range(3)]
        out = T.train_step(torch.from_numpy(z[f"lr{it}"]).to(dev), torch.from_numpy(z[f"hr{it}"]).to(dev), noise)
        got = np.array([float(out[k]) for k in ("loss_real", "loss_fake", "adv_loss", "content_loss")])
        assert np.allclose(got, z["losses"][it], rtol=1e-3), (it, got, z["losses"][it])
    for pre, mod in (("g", T.generator), ("d", T.discriminator)):
        for k, p in mod.state_dict().items():
            p0, p2 = torch.from_numpy(z[f"{pre}0.{k}"]), torch.from_numpy(z[f"{pre}2.{k}"])
            upd = (p2 - p0).abs().mean()
            lim = 0.3 if p.numel() == 1 else 0.12      # a lone scalar has no mean to average Adam's noise gain over
            assert (p.cpu() - p2).abs().mean() <= lim * upd, (pre, k)   # see tests/test_oracle.py on Adam's noise gain


@pytest.mark.gpu
@pytest.mark.parametrize("cdt", ["f32", "bf16", "f16"])
def test_full_width_train_step_vs_oracle(pkg, cdt):
    """64-filter G (2 blocks) and D, width/4 VGG stand-in, 16->64 crops: losses and one AdamW update vs the oracle."""
    dev = select("hip")
    torch.manual_seed(9)
    wd = 4 if cdt == "f32" else 2          # bf16 kernels want channel counts that are multiples of 32
    T = _trainer(pkg, dev, cdt, nf=64, n_layers=2, width_div=wd)
    g_sd = {k: v.detach().cpu().clone() for k, v in T.generator.state_dict().items()}
    d_sd = {k: v.detach().cpu().clone() for k, v in T.discriminator.state_dict().items()}
    g0, d0 = {k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}
    v_sd = O.vgg_standin_state_dict(1234, wd)
    lr, hr = torch.rand(2, 3, 16, 16) * 2 - 1, torch.rand(2, 3, 64, 64) * 2 - 1
    noise = [torch.rand(2, 1, 4, 4) for _ in range(3)]
    got = T.train_step(lr.to(dev), hr.to(dev), [n.to(dev) for n in noise])
    want = O.train_step(g_sd, d_sd, v_sd, lr, hr, noise, {}, {})
    tl = {"f32": 1e-3, "bf16": 5e-3, "f16": 1e-3}[cdt]      # 16-bit modes: ~2x the measured error (gpurun_out/parity_errors.log)
    for k in want:
        e = report("step16.%s.%s" % (cdt, k), abs(float(got[k]) - float(want[k])) / abs(float(want[k])))
        assert e <= tl, (k, float(got[k]), float(want[k]))
    if cdt == "f32":   # one AdamW step each: compare the updates in the mean (Adam amplifies tiny-gradient noise)
        for sd_ref, sd0, mod in ((g_sd, g0, T.generator), (d_sd, d0, T.discriminator)):
            for k, p in mod.state_dict().items():
                upd = (sd_ref[k] - sd0[k]).abs().mean()
                assert (p.cpu() - sd_ref[k]).abs().mean() <= 0.1 * upd + 1e-12, k


@pytest.mark.parametrize("backend", BACKENDS)
def test_pretrain_step_vs_oracle(pkg, backend):
    dev = select(backend)
    torch.manual_seed(10)
    T = _trainer(pkg, dev, "f32")
    g_sd = {k: v.detach().cpu().clone() for k, v in T.generator.state_dict().items()}
    lr, hr = torch.rand(2, 3, 8, 12) * 2 - 1, torch.rand(2, 3, 32, 48) * 2 - 1
    got = T.pretrain_step(lr.to(dev), hr.to(dev))
    want = O.pretrain_step(g_sd, lr, hr, {})
    assert abs(float(got) - float(want)) <= 1e-4 * abs(float(want))
    for k, p in T.generator.state_dict().items():
        assert (p.cpu() - g_sd[k]).abs().mean() <= 0.1 * 1e-4 + 1e-9, k   # both moved by ~lr in the same direction


# ------------------------------------------------------------------ host logic (CPU suite)
def test_config_loader_keys_and_overrides(pkg, tmp_path):
    cfg = pkg.load_config(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "config.yaml"),
                          ["training.batch_size=8", "generator.n_layers=12", "experiment.name=x", "training.generator_lr=2e-4"])
    ref_keys = {"experiment": {"name", "seed"}, "data": {"image_dir", "numpy_dir", "lr_image_size", "scale_factor"},
                "generator": {"n_filters", "n_layers"}, "discriminator": {"n_filters", "n_layers"},
                "training": {"compiled", "pretrain_iterations", "iterations", "device", "log_iter", "checkpoint_iter",
                             "batch_size", "num_workers", "generator_lr", "discriminator_lr"}}
    for group, keys in ref_keys.items():                      # configs/config.yaml:1-25 of the reference
        assert keys <= set(vars(getattr(cfg, group))), group
    assert cfg.training.batch_size == 8 and cfg.generator.n_layers == 12 and cfg.experiment.name == "x"
    assert cfg.training.generator_lr == 2e-4 and cfg.training.discriminator_lr == 1e-4
    with pytest.raises(ValueError):
        pkg.load_config(None, ["nonsense"])


def test_reference_defaults_and_hydra_run_directory(pkg, tmp_path, monkeypatch):
    """configs/config.yaml carries the reference's values (configs/config.yaml:7,22), and train.main reproduces what
    `@hydra.main(version_base="1.1")` does around the reference's main (train.py:46): outputs/<date>/<time>, .hydra/."""
    import datetime
    import yaml
    cfgmod = importlib.import_module("fast-srgan_amd.config")
    root = os.path.dirname(os.path.dirname(__file__))
    cfg = pkg.load_config(os.path.join(root, "configs", "config.yaml"))
    assert cfg.data.lr_image_size == 24 and cfg.training.batch_size == 24 and cfg.data.scale_factor == 4
    assert cfg.training.pretrain_iterations == 100 and cfg.generator.n_layers == 8 and cfg.discriminator.n_layers == 7
    now = datetime.datetime(2024, 12, 18, 7, 8, 9)
    assert cfgmod.hydra_run_settings([], now) == (True, os.path.join("outputs", "2024-12-18", "07-08-09"))
    assert cfgmod.hydra_run_settings(["hydra.job.chdir=false", "a.b=1"], now)[0] is False
    assert cfgmod.hydra_run_settings(["hydra.run.dir=x/y"], now) == (True, "x/y")
    monkeypatch.chdir(tmp_path)
    ov = ["data.numpy_dir=np", "hydra.run.dir=x/y", "training.batch_size=3"]
    cfg = pkg.load_config(None, ov)                       # hydra.* keys are not config keys
    assert not hasattr(cfg, "hydra") and cfg.training.batch_size == 3
    with pytest.warns(UserWarning, match="hydra.verbose"):      # other hydra.* keys: ignored with a warning (a reference command line keeps working)
        assert pkg.load_config(None, ["hydra.verbose=true", "training.batch_size=5"]).training.batch_size == 5
    got = cfgmod.enter_run_dir(cfg, ov)
    assert got == str(tmp_path / "x" / "y") and os.getcwd() == got
    assert cfg.data.numpy_dir == str(tmp_path / "np")     # relative data paths survive the chdir
    assert yaml.safe_load(open(".hydra/overrides.yaml")) == ["data.numpy_dir=np", "training.batch_size=3"]
    assert yaml.safe_load(open(".hydra/config.yaml"))["training"]["batch_size"] == 3
    monkeypatch.chdir(tmp_path)
    assert cfgmod.enter_run_dir(pkg.load_config(None, []), ["hydra.job.chdir=false"]) is None and os.getcwd() == str(tmp_path)


def test_resize_taps_match_oracle(pkg):
    data = __import__("importlib").import_module("fast-srgan_amd.dataloader")
    for n_in, n_out in ((384, 96), (48, 12), (100, 25), (64, 8)):
        xmin, xsize, w, kmax = data.aa_bicubic_taps(n_in, n_out)
        oxm, oxs, ow = O.aa_bicubic_weights(n_in, n_out)
        assert np.array_equal(xmin, oxm) and np.array_equal(xsize, oxs) and np.array_equal(w, ow) and kmax == ow.shape[1]


def test_device_crop_pipeline_matches_reference_dataset(pkg, tmp_path):
    """NumpyImagesDataset on the (emulated) device kernels vs the golden samples the REFERENCE dataset produced."""
    dev = select("emu")
    z = load_npz("dataset.npz")
    path = str(tmp_path / "img.npy")
    np.save(path, z["image"])
    ds = pkg.NumpyImagesDataset([path], lr_image_size=12, scale_factor=4, device=dev)
    random.seed(int(z["seed"]))
    for i in range(3):
        lr, hr = ds[0]
        assert torch.equal(hr, torch.from_numpy(z[f"hr{i}"]))
        assert (lr - torch.from_numpy(z[f"lr{i}"])).abs().max() < 1e-5
    loader = pkg.DeviceBatchLoader(ds, batch_size=2, iterations=2, seed=3)
    a = [tuple(t.clone() for t in b) for b in loader]
    b = [tuple(t.clone() for t in b) for b in pkg.DeviceBatchLoader(ds, batch_size=2, iterations=2, seed=3)]
    assert len(a) == 2 and a[0][0].shape == (2, 3, 12, 12) and a[0][1].shape == (2, 3, 48, 48)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))


def test_arena_adamw_matches_torch(pkg):
    """ArenaAdamW (flat arena + fsr_adamw_step) vs torch.optim.AdamW over several parameters and steps."""
    dev = select("emu")
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 4, 3, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(1))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ropt = pkg.ArenaAdamW(ps, lr=1e-3), torch.optim.AdamW(ref, lr=1e-3)
    assert ps[0].data_ptr() == opt.flat_param.data_ptr() and ps[0].grad.data_ptr() == opt.flat_grad.data_ptr()
    for step in range(3):
        opt.zero_grad()
        ropt.zero_grad()
        for p, r in zip(ps, ref):
            g = torch.randn_like(p)
            p.grad.add_(g)
            r.grad = g.clone()
        opt.step()
        ropt.step()
    for p, r in zip(ps, ref):
        assert (p.detach() - r.detach()).abs().max() < 1e-6
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][0]["exp_avg"].shape == (5, 4, 3, 3)


def test_dynamic_loss_scale_skips_overflowed_steps(pkg):
    """fp16-mode loss scaling (one overflow must not poison the moments for good): with a device-side scale
    state, a step whose gradient arena holds an inf / NaN changes NOTHING and halves the scale; clean steps divide by the
    scale on the device and match torch.optim.AdamW on the unscaled gradients; enough clean iterations double the scale."""
    import ctypes
    dev = select("emu")
    L = __import__("importlib").import_module("fast-srgan_amd._lib")
    ops = __import__("importlib").import_module("fast-srgan_amd.ops")
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(6, 4, 3, 3)), torch.nn.Parameter(torch.randn(9))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ropt = pkg.ArenaAdamW(ps, lr=1e-3), torch.optim.AdamW(ref, lr=1e-3)
    state = torch.tensor([1024.0, 0.0, 0.0, 0.0])
    opt.scale_state = state

    def iteration(poison):
        opt.zero_grad()
        ropt.zero_grad()
        for p, r in zip(ps, ref):
            g = torch.randn_like(p)
            p.grad.add_(g * float(state[0]))        # what a backward pass seeded with the scale leaves in the arena
            r.grad = g.clone()
        if poison:
            ps[0].grad.view(-1)[5] = poison
        before = [t.clone() for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_dev)]
        opt.step()
        L.check(L.lib().fsr_loss_scale_update(ops._p(state), 3.0, 2.0, 0.5, None), "fsr_loss_scale_update")
        return before

    iteration(None)
    ropt.step()
    for p, r in zip(ps, ref):
        assert (p.detach() - r.detach()).abs().max() < 1e-6
    assert state.tolist() == [1024.0, 1.0, 0.0, 0.0]
    for poison in (float("inf"), float("nan")):
        before = iteration(poison)
        for a, b in zip(before, (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_dev)):
            assert torch.equal(a, b)                # nothing moved, the step counter included
    assert state.tolist() == [256.0, 0.0, 0.0, 2.0]
    for _ in range(3):                              # growth_interval = 3 clean iterations -> the scale doubles
        iteration(None)
        ropt.step()
    assert state.tolist() == [512.0, 0.0, 0.0, 2.0]
    for p, r in zip(ps, ref):
        assert (p.detach() - r.detach()).abs().max() < 1e-6


def test_sequential_loader_without_a_full_batch_warns(pkg):
    """train.py:81-91's validation loader drops the last partial batch; a data set smaller than one batch then yields nothing.
    The reference skips its metrics silently; here the loader says so."""
    class Tiny:
        def __len__(self):
            return 3
    with pytest.warns(UserWarning, match="no full batch"):
        loader = pkg.DeviceBatchLoader(Tiny(), 4, sequential=True)
    assert len(loader) == 0 and list(loader) == []


def test_trainer_rejects_bad_loss_scale(pkg):
    select("emu")
    import types
    ns = types.SimpleNamespace
    for bad in (0.0, -4.0, float("nan")):
        cfg = ns(experiment=ns(name="t", seed=1), generator=ns(n_filters=32, n_layers=1), discriminator=ns(n_filters=32, n_layers=7),
                 training=ns(compiled=False, device="cpu", log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4, discriminator_lr=1e-4,
                             batch_size=1, compute_dtype="f16", loss_scale=bad))
        with pytest.raises(ValueError, match="loss_scale"):
            pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype="f16", width_div=2, seed=1))
    for bad in (0, 0.5, float("nan")):            # validated on the host, not at the first device step
        cfg.training.loss_scale, cfg.training.loss_scale_growth_interval = 1024.0, bad
        with pytest.raises(ValueError, match="growth_interval"):
            pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype="f16", width_div=2, seed=1))


def test_loss_scale_state_survives_a_checkpoint(pkg):
    """the device-side {scale, clean count, flag, skipped} of the dynamic fp16 loss scale rides in the
    optimizer checkpoint as an extra key (torch's own load_state_dict ignores it) and is restored on resume -- with the
    non-finite flag down."""
    select("emu")
    ps = [torch.nn.Parameter(torch.randn(4, 4, 3, 3))]
    opt = pkg.ArenaAdamW(ps, lr=1e-3)
    assert "fsr_loss_scale_state" not in opt.state_dict()          # static scale: the reference's format, nothing added
    opt.scale_state = torch.tensor([4096.0, 17.0, 1.0, 3.0])
    sd = opt.state_dict()
    assert sd["fsr_loss_scale_state"].tolist() == [4096.0, 17.0, 1.0, 3.0]
    ref = torch.optim.AdamW([torch.nn.Parameter(torch.randn(4, 4, 3, 3))], lr=1e-3)
    ref.load_state_dict({k: v for k, v in sd.items()})             # a reference-format loader takes the file as it is
    opt2 = pkg.ArenaAdamW([torch.nn.Parameter(torch.randn(4, 4, 3, 3))], lr=1e-3)
    opt2.scale_state = torch.tensor([1048576.0, 0.0, 0.0, 0.0])
    opt2.load_state_dict(sd)
    assert opt2.scale_state.tolist() == [4096.0, 17.0, 0.0, 3.0]
    opt3 = pkg.ArenaAdamW([torch.nn.Parameter(torch.randn(4, 4, 3, 3))], lr=1e-3)   # a static-scale run ignores the key
    opt3.load_state_dict(sd)
    assert opt3.scale_state is None


@pytest.mark.gpu
def test_device_crop_pipeline_full_size_gpu(pkg, tmp_path):
    """NumpyImagesDataset at the bench geometry (96 -> 384 crops of 2K-wide images) on the HIP kernels vs the oracle's
    restatement of dataloader.py:24-38, same python RNG stream."""
    dev = select("hip")
    rng = np.random.default_rng(3)
    paths = []
    for i in range(2):
        p = str(tmp_path / f"img{i}.npy")
        np.save(p, rng.integers(0, 256, size=(3, 500 + 40 * i, 700), dtype=np.uint8))
        paths.append(p)
    ds = pkg.NumpyImagesDataset(paths, lr_image_size=96, scale_factor=4, device=dev)
    for idx in (0, 1, 0):
        random.seed(100 + idx)
        lr, hr = ds[idx]
        random.seed(100 + idx)
        lr_ref, hr_ref, _ = O.dataset_item(np.load(paths[idx]), 96, 4)
        assert lr.shape == (3, 96, 96) and hr.shape == (3, 384, 384)
        assert torch.equal(hr.cpu(), hr_ref)
        assert (lr.cpu() - lr_ref).abs().max() < 1e-5
    lr_b, hr_b = next(iter(pkg.DeviceBatchLoader(ds, batch_size=8, iterations=1, seed=5)))
    assert lr_b.shape == (8, 3, 96, 96) and hr_b.shape == (8, 3, 384, 384) and lr_b.is_cuda


@pytest.mark.gpu
def test_graph_replay_of_iteration_and_inference(pkg):
    """hipGraph paths: a captured training iteration keeps stepping the optimizers (device-side step counters,
    fresh label noise per replay), and GraphedGenerator reproduces eager inference BIT FOR BIT (same kernels, and the
    InstanceNorm statistics are order-fixed sums)."""
    dev = select("hip")
    torch.manual_seed(11)
    T = _trainer(pkg, dev, "bf16", nf=32, n_layers=1, width_div=2)
    lr, hr = (torch.rand(2, 3, 16, 16) * 2 - 1).to(dev), (torch.rand(2, 3, 64, 64) * 2 - 1).to(dev)
    T.capture_train_step(lr, hr, warmup=2)                # 2 warm-up iterations + 1 captured
    base = float(T.optim_generator.step_dev)
    p0 = T.optim_generator.flat_param.clone()
    outs = []
    for _ in range(3):
        out = T.graphed_train_step(lr, hr)
        outs.append(float(out["loss_fake"]))
    torch.cuda.synchronize()
    assert float(T.optim_generator.step_dev) == base + 3 and float(T.optim_discriminator.step_dev) == base + 3
    assert torch.isfinite(T.optim_generator.flat_param).all() and not torch.equal(T.optim_generator.flat_param, p0)
    assert len(set(outs)) == 3                             # new noise (and new weights) every replay
    G = T.generator.eval()
    x = (torch.rand(1, 3, 20, 24) * 2 - 1).to(dev)
    # eager code after replays must see the CURRENT weights (its packed-filter cache was invalidated by the replay)
    G2 = pkg.Generator(ns(n_filters=32, n_layers=1), compute_dtype="bf16")
    G2.load_state_dict({k: v.detach().cpu().clone() for k, v in G.state_dict().items()})
    with torch.no_grad():
        assert (G(x) - G2.to(dev).eval()(x)).abs().max() < 2e-2
    with torch.no_grad():
        want = G(x).clone()
    gg = pkg.GraphedGenerator(G, x)
    assert torch.equal(gg(x), want)
    x2 = (torch.rand(1, 3, 20, 24) * 2 - 1).to(dev)
    with torch.no_grad():
        assert torch.equal(gg(x2), G(x2))


@pytest.mark.gpu
@pytest.mark.parametrize("cdt", ["f32", "x3", "x3v", "f16", "bf16"])
def test_graphed_replays_equal_eager_steps(pkg, cdt):
    """N replays of the captured iteration == N eager train_steps, bit for bit (injected label noise, new batch every
    step), in every mode the bench times.  A replay that read filters packed at capture time, or statistics summed in another
    order, fails this; so does -- x3 -- one stray torch addition of two float32 CONTAINERS inside the captured graph, and -- f16 --
    a loss-scale decision taken on the host at capture time instead of on the device at replay time."""
    dev = select("hip")

    def batches(seed, count):
        g = torch.Generator().manual_seed(seed)
        return [((torch.rand(2, 3, 16, 16, generator=g) * 2 - 1).to(dev), (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).to(dev),
                 [torch.rand(2, 1, 4, 4, generator=g).to(dev) for _ in range(3)]) for _ in range(count)]

    data = batches(5, 6)
    torch.manual_seed(21)
    Te = _trainer(pkg, dev, cdt, nf=32, n_layers=1, width_div=4 if cdt == "f32" else 2)      # (16-bit and x3 tensors: multiples of 32 channels)
    torch.manual_seed(21)
    Tg = _trainer(pkg, dev, cdt, nf=32, n_layers=1, width_div=4 if cdt == "f32" else 2)
    for a, b in zip(Te.optim_generator.flat_param, Tg.optim_generator.flat_param):
        assert float(a) == float(b)
        break
    # eager: the capture's two warm-up iterations see batch 0 twice, then batches 1..5
    eager_losses = []
    for lr, hr, nz in [data[0], data[0]] + data[1:]:
        eager_losses.append({k: float(v) for k, v in Te.train_step(lr, hr, nz).items()})
    Tg.capture_train_step(data[0][0], data[0][1], warmup=2, noise=data[0][2])
    graph_losses = []
    for lr, hr, nz in data[1:]:
        graph_losses.append({k: float(v) for k, v in Tg.graphed_train_step(lr, hr, nz).items()})
    torch.cuda.synchronize()
    assert graph_losses == eager_losses[2:]
    for oe, og in ((Te.optim_generator, Tg.optim_generator), (Te.optim_discriminator, Tg.optim_discriminator)):
        assert torch.equal(oe.flat_param, og.flat_param) and torch.equal(oe.exp_avg_sq, og.exp_avg_sq)
        assert float(oe.step_dev) == float(og.step_dev) == 7.0
    if cdt in ("f16", "x3v"):
        assert Te.loss_scale_state() == Tg.loss_scale_state() and Tg.loss_scale_state()[1] == 0


@pytest.mark.gpu
def test_rccl_single_process_group_phase_graphs(pkg, tmp_path):
    """The data-parallel path on ONE GPU: a 1-rank RCCL process group (FSR_FORCE_DIST=1) runs the gradient all-reduces for
    real and the iteration is captured as three phase graphs around them; results equal the plain single-process run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "dist_check.py")
    outs = []
    for force in ("0", "1"):
        out = str(tmp_path / ("params_%s.pt" % force))
        env = dict(os.environ, FSR_FORCE_DIST=force, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29517")
        r = subprocess.run([sys.executable, script, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(out))
    assert outs[1]["segments"] == 3 and outs[0]["segments"] == 1 and outs[1]["backend"] == "nccl"
    assert torch.equal(outs[0]["g"], outs[1]["g"]) and torch.equal(outs[0]["d"], outs[1]["d"])
    assert outs[0]["losses"] == outs[1]["losses"]
