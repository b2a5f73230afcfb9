"""HR/LR crop pipeline on the device: drop-in for /root/reference/dataloader.py.

`NumpyImagesDataset(numpy_paths, lr_image_size, scale_factor)` keeps the reference's constructor, `__len__`
and `__getitem__ -> (lr (3,l,l), hr (3,l*s,l*s))` float32 in [-1,1] (dataloader.py:11-38) -- but the uint8
CHW arrays are uploaded once and stay resident in HBM, and the crop / float conversion / antialiased bicubic
down-scale / [-1,1] mapping run in two HIP kernels (csrc/data.hip) instead of 16 DataLoader worker processes.
`DeviceBatchLoader` draws whole batches that way (what train.py's RandomSampler(replacement=True) +
DataLoader pair does, train.py:69-113) and is what the Trainer iterates.
"""
import math
import random

import numpy as np
import torch

from . import _lib as L
from .ops import _p, _stream


def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def aa_bicubic_taps(in_size, out_size):
    """Tap table of the antialiased bicubic down-scale (what v2.Resize(BICUBIC, antialias=True) computes for a
    float tensor, dataloader.py:15-19): support = 2*scale, center = scale*(i+.5), xmin = max(int(center-support+.5),0),
    xsize = min(int(center+support+.5), in) - xmin, w_j = cubic((j+xmin-center+.5)/scale), normalised in float32."""
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    kmax = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int32)
    xsize = np.zeros(out_size, dtype=np.int32)
    w = np.zeros((out_size, kmax), dtype=np.float32)
    for i in range(out_size):
        center = scale * (i + 0.5)
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        ws = np.array([_cubic((j + lo - center + 0.5) * inv) for j in range(n)], dtype=np.float32)
        w[i, :n] = ws / np.float32(ws.sum(dtype=np.float32))
        xmin[i], xsize[i] = lo, n
    return xmin, xsize, w, kmax


class NumpyImagesDataset(torch.utils.data.Dataset):
    def __init__(self, numpy_paths, lr_image_size, scale_factor, device="cuda"):
        self.numpy_paths = list(numpy_paths)
        self.lr_image_size = lr_image_size
        self.hr_image_size = lr_image_size * scale_factor
        self.scale_factor = scale_factor
        self.device = torch.device(device)
        self._images = [None] * len(self.numpy_paths)
        xmin, xsize, w, self._kmax = aa_bicubic_taps(self.hr_image_size, self.lr_image_size)
        self._xmin = torch.from_numpy(xmin).to(self.device)
        self._xsize = torch.from_numpy(xsize).to(self.device)
        self._w = torch.from_numpy(w).to(self.device)

    def __len__(self):
        return len(self.numpy_paths)

    def _image(self, idx):
        img = self._images[idx]
        if img is None:
            arr = np.load(self.numpy_paths[idx], mmap_mode="r")       # uint8 (3,H,W), train.py:29-31
            if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[0] != 3:
                raise ValueError("%s: expected a uint8 (3,H,W) array" % self.numpy_paths[idx])
            img = torch.from_numpy(np.array(arr)).to(self.device)
            self._images[idx] = img
        return img

    def draw_crop(self, idx, rng=random):
        """dataloader.py:27-29: two inclusive randint draws, crop_h first."""
        _, h, w = self._image(idx).shape
        return rng.randint(0, h - self.hr_image_size), rng.randint(0, w - self.hr_image_size)

    def batch(self, indices, crops):
        """(lr, hr) float32 NCHW device batches for image `indices` cropped at `crops` [(y, x), ...].
        The per-sample descriptors (image pointer, height, width, crop origin) travel as ONE pinned, non-blocking
        host-to-device copy; nothing here waits for the device."""
        n = len(indices)
        imgs = [self._image(i) for i in indices]
        # [n int64 pointers][4 x n int32: heights, widths, crop_y, crop_x]  (viewed as int64 words for one copy)
        desc = np.empty(3 * n, dtype=np.int64)
        desc[:n] = [im.data_ptr() for im in imgs]
        desc[n:].view(np.int32)[:] = ([im.shape[1] for im in imgs] + [im.shape[2] for im in imgs]
                                      + [c[0] for c in crops] + [c[1] for c in crops])
        host = torch.from_numpy(desc)
        if self.device.type == "cuda":
            host = host.pin_memory()
        dev_desc = host.to(self.device, non_blocking=True)
        meta = dev_desc[n:].view(torch.int32).view(4, n)
        hr, lr = self.hr_image_size, self.lr_image_size
        hr_out = torch.empty((n, 3, hr, hr), dtype=torch.float32, device=self.device)
        lr_out = torch.empty((n, 3, lr, lr), dtype=torch.float32, device=self.device)
        tmp = torch.empty((n, 3, hr, lr), dtype=torch.float32, device=self.device)
        L.check(L.lib().fsr_crop_resize(_p(dev_desc), _p(meta[0]), _p(meta[1]), _p(meta[2]), _p(meta[3]), n, hr,
                                        self.scale_factor, _p(self._w), _p(self._xmin), _p(self._xsize), self._kmax,
                                        _p(hr_out), _p(lr_out), _p(tmp), _stream()), "fsr_crop_resize")
        return lr_out, hr_out

    def __getitem__(self, idx):
        lr, hr = self.batch([idx], [self.draw_crop(idx)])
        return lr[0], hr[0]


class DeviceBatchLoader:
    """Iterable of device batches.

    Default (training, train.py:69-80,92-113): `iterations` batches whose image indices are drawn with replacement from a
    seeded torch.Generator (RandomSampler(replacement=True)), crops from a seeded `random.Random` (train.py:40-43 seeds
    python's RNG per worker).  sequential=True (validation, train.py:81-91: shuffle=False, drop_last=True): ONE pass over
    the data set in order, len(dataset) // batch_size batches, fresh random crops.  Under data parallelism each rank
    passes seed + rank."""

    def __init__(self, dataset, batch_size, iterations=None, seed=1234, sequential=False):
        self.dataset, self.batch_size, self.sequential = dataset, batch_size, sequential
        self.iterations = len(dataset) // batch_size if sequential else iterations
        if self.iterations is None:
            raise ValueError("DeviceBatchLoader needs `iterations` unless sequential=True")
        if sequential and self.iterations == 0:
            # the reference's DataLoader(drop_last=True) behaves the same way (train.py:81-91), but silently: SSIM / PSNR are
            # then never logged and no fixed validation images exist
            import warnings
            warnings.warn("DeviceBatchLoader(sequential=True): %d images give no full batch of %d -- validation metrics and the "
                          "fixed validation images will be skipped (drop_last semantics of train.py:81-91)" % (len(dataset), batch_size))
        self.gen = torch.Generator().manual_seed(seed)
        self.rng = random.Random(seed)

    def __len__(self):
        return self.iterations

    def __iter__(self):
        for it in range(self.iterations):
            if self.sequential:
                idx = list(range(it * self.batch_size, (it + 1) * self.batch_size))
            else:
                idx = torch.randint(len(self.dataset), (self.batch_size,), generator=self.gen).tolist()
            crops = [self.dataset.draw_crop(i, self.rng) for i in idx]
            yield self.dataset.batch(idx, crops)
