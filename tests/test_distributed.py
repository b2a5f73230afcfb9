"""Data-parallel path on a fake 2-rank cluster: gloo backend, CPU tensors, kernels on the host emulator."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    from backend import select
    select("emu")
    pkg = importlib.import_module("fast-srgan_amd")
    D = importlib.import_module("fast-srgan_amd.distributed")
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.rank() == rank
    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT parameters ...
    params = [torch.nn.Parameter(torch.randn(6, 5, 3, 3)), torch.nn.Parameter(torch.randn(11))]
    opt = pkg.ArenaAdamW(params, lr=1e-2)
    D.broadcast_parameters(opt)                        # ... and agree after the broadcast
    sync = D.GradSync(opt)
    assert abs(opt.grad_scale - 1.0 / world) < 1e-12
    start = opt.flat_param.clone()
    for step in range(2):
        opt.zero_grad()
        g = torch.Generator().manual_seed(1000 * step + rank)
        for p in params:
            p.grad.add_(torch.randn(p.shape, generator=g))
        sync.start()
        sync.wait()
        opt.step()
    torch.save({"start": start, "end": opt.flat_param.clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_average_equals_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(res[0]["start"], res[1]["start"])         # broadcast made the replicas identical
    assert torch.equal(res[0]["end"], res[1]["end"])             # and they stay identical after the steps
    # single-process reference: AdamW on the MEAN of the two ranks' gradients
    torch.manual_seed(100)
    ref = [torch.nn.Parameter(torch.randn(6, 5, 3, 3)), torch.nn.Parameter(torch.randn(11))]
    opt = torch.optim.AdamW(ref, lr=1e-2)
    for step in range(2):
        gens = [torch.Generator().manual_seed(1000 * step + r) for r in range(world)]
        for p in ref:
            p.grad = sum(torch.randn(p.shape, generator=g) for g in gens) / world
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in ref])
    assert (res[0]["end"] - flat).abs().max() < 1e-6


# ------------------------------------------------------------------ the whole iteration, batch-sharded over 2 ranks
def _tiny_trainer(pkg, dev, mode="f32"):
    """mode "x3v" (the bench headline): x3 Generator / Discriminator, fp16 perceptual network, the device-side dynamic loss scale --
    16-bit and x3 tensors carry multiples of 32 channels."""
    import types
    import warnings
    ns = types.SimpleNamespace
    nf, wd = (16, 4) if mode == "f32" else (32, 2)
    cfg = ns(experiment=ns(name="ddp", seed=1234), generator=ns(n_filters=nf, n_layers=1),
             discriminator=ns(n_filters=nf, n_layers=7),
             training=ns(compiled=False, device=str(dev), log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4,
                         discriminator_lr=1e-4, batch_size=1, compute_dtype=mode))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype=mode, width_div=wd, seed=1234))


def _ddp_data():
    g = torch.Generator().manual_seed(42)
    lr = torch.rand(2, 3, 8, 8, generator=g) * 2 - 1
    hr = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    noise = [torch.rand(2, 1, 2, 2, generator=g) for _ in range(3)]
    return lr, hr, noise


def _step_worker(rank, world, port, out_dir, mode="f32"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    from backend import select
    dev = select("emu")
    pkg = importlib.import_module("fast-srgan_amd")
    D = importlib.import_module("fast-srgan_amd.distributed")
    D.init_from_env(backend="gloo")
    torch.manual_seed(7 + rank)                       # different initialisation per rank; Trainer broadcasts rank 0's
    T = _tiny_trainer(pkg, dev, mode)
    init = {"g": T.optim_generator.flat_param.clone(), "d": T.optim_discriminator.flat_param.clone()}
    lr, hr, noise = _ddp_data()
    T.train_step(lr[rank:rank + 1], hr[rank:rank + 1], [n[rank:rank + 1] for n in noise])
    torch.save({"init": init, "g_grad": T.optim_generator.flat_grad / world, "d_grad": T.optim_discriminator.flat_grad / world,
                "g": T.optim_generator.flat_param.clone(), "d": T.optim_discriminator.flat_param.clone(),
                "scale": T.loss_scale_state()},
               os.path.join(out_dir, f"step_rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("mode", ["f32", "x3v"])
def test_two_rank_iteration_equals_single_process_at_global_batch(tmp_path, mode):
    """trainer.py:171-196 sharded by batch over 2 ranks (gloo, kernels on the host emulator) vs ONE process at the
    global batch: per-sample InstanceNorm + mean-reduced losses make the averaged gradients identical.
    x3v -- the mode bench.py's `value` is timed in --: the gradient arenas carry the loss scale S through the exchange (AdamW
    divides by it on the device, after the non-finite check every rank takes on the SUMMED arena: the ranks skip alike)."""
    world, port = 2, _free_port()
    mp.spawn(_step_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f"step_rank{r}.pt")) for r in range(world)]
    for k in ("g", "d"):
        assert torch.equal(res[0]["init"][k], res[1]["init"][k])
        assert torch.equal(res[0][k], res[1][k])                      # replicas stay identical after the iteration
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    from backend import relerr2, select
    dev = select("emu")
    pkg = importlib.import_module("fast-srgan_amd")
    T = _tiny_trainer(pkg, dev, mode)
    if mode == "x3v":      # both ranks kept the initial scale and skipped nothing; so does the single process
        assert res[0]["scale"] == res[1]["scale"] == T.loss_scale_state() and res[0]["scale"][1] == 0
    with torch.no_grad():
        T.optim_generator.flat_param.copy_(res[0]["init"]["g"])
        T.optim_discriminator.flat_param.copy_(res[0]["init"]["d"])
    lr, hr, noise = _ddp_data()
    T.train_step(lr, hr, noise)
    assert relerr2(res[0]["g_grad"], T.optim_generator.flat_grad) < 1e-3
    assert relerr2(res[0]["d_grad"], T.optim_discriminator.flat_grad) < 1e-3


# ------------------------------------------------------------------ four ranks, different data everywhere, several iterations
def _four_rank_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    from backend import select
    dev = select("emu")
    pkg = importlib.import_module("fast-srgan_amd")
    D = importlib.import_module("fast-srgan_amd.distributed")
    D.init_from_env(backend="gloo")
    assert D.all_ranks_ok(True) and not D.all_ranks_ok(rank != 2)      # one dissenting rank turns every rank's answer
    torch.manual_seed(50 + rank)
    T = _tiny_trainer(pkg, dev)
    g = torch.Generator().manual_seed(900 + 17 * rank)                 # every rank draws its OWN crops and label noise
    for _ in range(3):
        lr = torch.rand(1, 3, 8, 8, generator=g) * 2 - 1
        hr = torch.rand(1, 3, 32, 32, generator=g) * 2 - 1
        noise = [torch.rand(1, 1, 2, 2, generator=g) for _ in range(3)]
        T.train_step(lr, hr, noise)
    torch.save({"g": T.optim_generator.flat_param.clone(), "d": T.optim_discriminator.flat_param.clone(),
                "gm": T.optim_generator.exp_avg.clone(), "dm": T.optim_discriminator.exp_avg.clone(),
                "gstep": T.optim_generator.step_dev.clone()}, os.path.join(out_dir, f"four_rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_four_ranks_with_unequal_data_stay_bit_identical(tmp_path):
    """World size 4 (gloo, host emulator), per-rank data and label-noise seeds, three graph-less iterations of
    trainer.py:171-196 with the exchange started after each phase and awaited before its optimizer step: the replicas'
    parameters AND Adam moments are bit-identical afterwards, and they moved."""
    world, port = 4, _free_port()
    mp.spawn(_four_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f"four_rank{r}.pt")) for r in range(world)]
    for r in range(1, world):
        for k in res[0]:
            assert torch.equal(res[0][k], res[r][k]), (r, k)
    assert float(res[0]["gstep"]) == 3.0 and float(res[0]["gm"].abs().max()) > 0 and float(res[0]["dm"].abs().max()) > 0
