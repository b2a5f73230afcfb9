"""One process, optionally inside a 1-rank RCCL process group (FSR_FORCE_DIST=1): capture the GAN iteration (one hipGraph, or
three phase graphs around the two gradient all-reduces), replay it on four batches and save the parameters.
tests/test_trainer.py compares the two modes bit for bit.  Usage: python tools/dist_check.py OUT.pt"""
import importlib
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("fast-srgan_amd")
D = importlib.import_module("fast-srgan_amd.distributed")


def ns(**k):
    return types.SimpleNamespace(**k)


def main():
    rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    torch.manual_seed(31)
    cfg = ns(experiment=ns(name="dist", seed=1234), generator=ns(n_filters=32, n_layers=1), discriminator=ns(n_filters=32, n_layers=7),
             training=ns(compiled=False, device=dev, log_iter=1, checkpoint_iter=10 ** 9, generator_lr=1e-4, discriminator_lr=1e-4,
                         batch_size=2, compute_dtype="bf16"))
    T = pkg.Trainer(cfg, perceptual_network=pkg.VGG19(compute_dtype="bf16", width_div=2, seed=1234))
    g = torch.Generator().manual_seed(7)
    data = [((torch.rand(2, 3, 16, 16, generator=g) * 2 - 1).to(dev), (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).to(dev),
             [torch.rand(2, 1, 4, 4, generator=g).to(dev) for _ in range(3)]) for _ in range(5)]
    T.capture_train_step(data[0][0], data[0][1], warmup=2, noise=data[0][2])
    losses = []
    for lr, hr, nz in data[1:]:
        losses.append({k: float(v) for k, v in T.graphed_train_step(lr, hr, nz).items()})
    torch.cuda.synchronize()
    backend = torch.distributed.get_backend() if D.is_distributed() else "none"
    torch.save({"g": T.optim_generator.flat_param.cpu(), "d": T.optim_discriminator.flat_param.cpu(), "losses": losses,
                "segments": len(T._graphs), "backend": backend}, sys.argv[1])
    if D.is_distributed():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
