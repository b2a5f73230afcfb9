"""Data-parallel path on a fake 2-rank cluster: gloo backend, CPU tensors, kernels on the host emulator."""
import os
import socket
import sys

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
