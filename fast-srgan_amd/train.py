"""`python train.py [a.b=c ...]`: drop-in for /root/reference/train.py (hydra-style overrides, same keys).

One process per GPU: launch with `python -m torch.distributed.run --nproc-per-node 8 train.py ...` for
batch-sharded data parallelism (training.batch_size is the PER-GPU batch).  The PNG -> uint8 CHW .npy cache
(train.py:22-37) is kept as is (one-off, PIL-bound); batches are then cut on the device (dataloader.py).
"""
import os
import random
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import distributed as D
from .config import enter_run_dir, hydra_run_settings, load_config
from .dataloader import DeviceBatchLoader, NumpyImagesDataset
from .trainer import Trainer


def seed(seed_value):
    torch.manual_seed(seed_value)
    np.random.seed(seed_value)
    random.seed(seed_value)


def write_images_to_numpy_arrays(image_list, output_dir, threads=16):
    """train.py:22-37 of the reference (PIL decode -> RGB -> uint8 CHW -> np.save on a 16-thread pool), done natively:
    fsr_png_to_npy (csrc/ingest.cpp) inflates, unfilters and converts on `threads` worker threads without the GIL and writes the
    same (3, H, W) uint8 .npy files.  PNG kinds that decoder does not take (interlaced, 16-bit, grey below 8 bits: status -4) --
    and only those -- go through PIL as in the reference; any other failure is an error."""
    import ctypes

    from . import _lib as L
    os.makedirs(output_dir, exist_ok=True)
    image_list = list(image_list)
    outs = [os.path.join(output_dir, os.path.basename(p).replace(".png", "")) + ".npy" for p in image_list]
    n = len(image_list)
    if n == 0:
        return
    arr = ctypes.c_char_p * n
    status = (ctypes.c_int * n)()
    left = L.lib().fsr_png_to_npy(arr(*[os.fsencode(p) for p in image_list]), arr(*[os.fsencode(p) for p in outs]), n, int(threads), status)
    if left < 0:
        L.check(left, "fsr_png_to_npy")
    todo = [(p, o) for p, o, st in zip(image_list, outs, status) if st != 0]
    bad = [(p, st) for p, st in zip(image_list, status) if st not in (0, -4)]
    if bad:
        raise L.FsrError("fsr_png_to_npy: %d file(s) could not be converted, e.g. %s (status %d)" % (len(bad), bad[0][0], bad[0][1]))
    if todo:
        from PIL import Image

        def _write(image_path, numpy_path):
            image = np.array(Image.open(image_path).convert("RGB")).astype(np.uint8)
            np.save(numpy_path, np.transpose(image, (2, 0, 1)))

        with ThreadPoolExecutor(max_workers=threads) as executor:
            list(executor.map(lambda t: _write(*t), todo))


def main(argv=None, config_dir="configs"):
    argv = sys.argv[1:] if argv is None else list(argv)
    config = load_config(os.path.join(config_dir, "config.yaml"), argv)
    rank, world, local_rank = D.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    # hydra 1.1 (train.py:46) runs `main` inside outputs/<date>/<time>: rank 0 names the directory, every rank enters it
    run_dir = [hydra_run_settings(argv)[1]]
    if world > 1:
        torch.distributed.broadcast_object_list(run_dir, src=0)
    if rank == 0:
        enter_run_dir(config, argv, run_dir[0])
    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        # (a multi-node job need not share a working directory: a rank that does not see rank 0's directory makes its own)
        if hydra_run_settings(argv)[0]:
            os.makedirs(os.path.abspath(run_dir[0]), exist_ok=True)
        enter_run_dir(config, argv, run_dir[0], create=False)
    if rank == 0 and not os.path.exists(config.data.numpy_dir):
        write_images_to_numpy_arrays([os.path.join(config.data.image_dir, x) for x in os.listdir(config.data.image_dir)
                                      if x.endswith(".png")], config.data.numpy_dir)
    if world > 1:
        torch.distributed.barrier()
    seed(config.experiment.seed + rank)
    numpy_files = sorted(os.path.join(config.data.numpy_dir, x) for x in os.listdir(config.data.numpy_dir) if x.endswith(".npy"))
    dataset = NumpyImagesDataset(numpy_files, config.data.lr_image_size, config.data.scale_factor, device=config.training.device)
    bs = config.training.batch_size
    val = DeviceBatchLoader(dataset, bs, seed=config.experiment.seed + 7919 * (rank + 1), sequential=True)   # train.py:81-91
    pre = DeviceBatchLoader(dataset, bs, config.training.pretrain_iterations, seed=config.experiment.seed + rank)
    trn = DeviceBatchLoader(dataset, bs, config.training.iterations, seed=config.experiment.seed + 104729 * (rank + 1))
    trainer = Trainer(config)
    trainer.pretrain(pre, val)
    trainer.train(trn, val)


if __name__ == "__main__":
    main()
