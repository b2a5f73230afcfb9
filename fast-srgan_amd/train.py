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


def write_images_to_numpy_arrays(image_list, output_dir):
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)

    def _write(image_path, numpy_path):
        image = np.array(Image.open(image_path).convert("RGB")).astype(np.uint8)
        np.save(numpy_path, np.transpose(image, (2, 0, 1)))

    with ThreadPoolExecutor(max_workers=16) as executor:
        for image_path in image_list:
            executor.submit(_write, image_path, os.path.join(output_dir, os.path.basename(image_path).replace(".png", "")))


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
