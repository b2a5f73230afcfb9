"""`python inference.py --image_dir D --output_dir O`: drop-in for /root/reference/inference.py.

Reads configs/config.yaml and models/model.pt relative to the working directory (inference.py:26-27), strips
the `_orig_mod.` prefix torch.compile left in the shipped checkpoint (:31-33), accepts .png/.jpg/jpeg in any
case (:37-45), writes each result under the same basename (:57) and converts with the reference's
TRUNCATING uint8 cast (:53-56) -- in the head kernel's epilogue.  Frames travel as bytes in both directions and are
batched and pipelined (InferencePipeline); decoding / encoding runs in worker threads.
"""
import os
from argparse import ArgumentParser

import numpy as np
import torch

from .config import load_config
from .model import Generator

parser = ArgumentParser("Real Time Image Super Resolution")
parser.add_argument("--image_dir", default=None, required=True, type=str)
parser.add_argument("--output_dir", default=None, required=True, type=str)
parser.add_argument("--compute_dtype", default=None, choices=["bf16", "f16", "x3", "x3v", "f32"], help="extension: kernel precision")
parser.add_argument("--batch", default=8, type=int, help="extension: frames per device batch (same-shape frames are batched)")


def load_generator(config, model_path, device="cuda", compute_dtype=None):
    model = Generator(config.generator, compute_dtype=compute_dtype or getattr(config.training, "compute_dtype", "f16"))
    weights = torch.load(model_path, map_location="cpu")
    model.load_state_dict({k.replace("_orig_mod.", ""): v for k, v in weights.items()})
    return model.to(device).eval()


@torch.no_grad()
def super_resolve(model, lr_u8_hwc, device="cuda"):
    """uint8 (H,W,3) -> uint8 (4H,4W,3), inference.py:48-56, one frame: bytes up, bytes down; the [-1,1] mapping and the
    (y+1)/2*255 truncating cast run on the device (Generator.forward_u8)."""
    frame = torch.from_numpy(np.ascontiguousarray(lr_u8_hwc)).unsqueeze(0).to(device)
    return model.forward_u8(frame)[0].cpu().numpy()


class InferencePipeline:
    """Batched, pipelined super-resolution of a stream of frames (replaces the per-image loop of
    inference.py:47-57).

      host thread   : packs `batch` uint8 frames into a pinned staging buffer
      compute stream: H2D of the bytes (0.17 MB per 180x320 frame) -> ONE hipGraph replay per batch (u8 -> [-1,1],
                      generator, head epilogue storing the uint8 frame)
      copy stream   : D2H of the finished uint8 frames (2.8 MB per 720p frame, 4x less than floats) into pinned memory,
                      overlapped with the next batch's compute (`depth` staging slots, each with its own graph)
    Frames are bucketed by shape; one set of graphs per (H, W), built lazily and kept for the most recently used shapes
    only.  `run` yields results in input order."""

    def __init__(self, model, device="cuda", batch=8, depth=2, use_graph=True, copy=True, max_shapes=4):
        """copy=False: `run` yields VIEWS of the pinned result buffers (valid until `depth` more batches have been
        submitted) instead of private arrays -- for consumers that encode / display a frame right away.
        max_shapes: plans (pinned staging, device buffers, captured graphs with their private pools) are kept for the
        `max_shapes` most recently used frame shapes only; a directory of differently sized images (which the reference's
        per-image loop handles, inference.py:47-57) would otherwise pin a few GB per distinct shape for good."""
        self.model, self.device, self.batch, self.depth, self.use_graph = model.eval(), torch.device(device), batch, depth, use_graph
        self.copy = copy
        self.max_shapes = max(1, int(max_shapes))
        self._plans = {}            # (h, w) -> list of `depth` slots, built lazily (a one-batch bucket only ever builds slot 0)
        self._lru = []              # shapes, least recently used first
        self._copy_stream = torch.cuda.Stream(device=self.device)

    class _Slot:
        pass

    def _touch(self, key):
        if key in self._lru:
            self._lru.remove(key)
        self._lru.append(key)
        while len(self._lru) > self.max_shapes:
            old = self._lru.pop(0)
            plan = self._plans.pop(old, None)
            if plan is not None:
                for sl in plan:
                    if sl is not None and sl.pending is not None:
                        sl.copied.synchronize()
                del plan            # graphs (and their private memory pools), device and pinned buffers go with the slots
                torch.cuda.synchronize(self.device)
                torch.cuda.empty_cache()

    def _slot(self, h, w, j):
        """Slot j of the plan for (h, w), created on first use."""
        key = (h, w)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = [None] * self.depth
        self._touch(key)
        if plan[j] is not None:
            return plan[j]
        sl = self._Slot()
        sl.host_in = torch.empty((self.batch, h, w, 3), dtype=torch.uint8).pin_memory()
        sl.x = torch.zeros((self.batch, h, w, 3), dtype=torch.uint8, device=self.device)
        with torch.no_grad():
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                y = self.model.forward_u8(sl.x)            # warm-up: creates every lazily allocated buffer
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            sl.graph = None
            if self.use_graph:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        y = self.model.forward_u8(sl.x)
                    sl.graph = g
                except Exception as exc:  # noqa: BLE001 -- eager launches are always available
                    print("InferencePipeline: hipGraph capture failed (%s: %s); eager launches" % (type(exc).__name__, exc))
                    torch.cuda.synchronize()
        sl.y = y
        sl.host_out = torch.empty(tuple(y.shape), dtype=torch.uint8).pin_memory()
        # plain numpy views for the host-side packing / unpacking: one memcpy per frame on this thread (torch's
        # multi-threaded copy_ pays a thread-pool wake-up per call, milliseconds when the pool has gone to sleep)
        sl.host_in_np, sl.host_out_np = sl.host_in.numpy(), sl.host_out.numpy()
        sl.done = torch.cuda.Event()
        sl.copied = torch.cuda.Event()
        sl.pending = None
        plan[j] = sl
        return sl

    @torch.no_grad()
    def _eager(self, frames):
        """A handful of frames of one shape (fewer than a batch): one eager launch at their TRUE count -- no staging plan, no
        graph capture, no padding of the batch with repeated frames."""
        x = torch.from_numpy(np.stack([np.ascontiguousarray(f) for f in frames])).to(self.device)
        y = self.model.forward_u8(x).cpu().numpy()
        return [y[i] for i in range(len(frames))]

    @torch.no_grad()
    def _submit(self, sl, frames):
        """Enqueue one batch on slot `sl` (its previous results have been collected): nothing here waits for the device."""
        n = len(frames)
        for i, f in enumerate(frames):
            np.copyto(sl.host_in_np[i], f)
        assert n == self.batch       # (ragged tails run eagerly: InferencePipeline.run)
        main = torch.cuda.current_stream()
        sl.x.copy_(sl.host_in, non_blocking=True)
        if sl.graph is not None:
            sl.graph.replay()
        else:
            sl.y = self.model.forward_u8(sl.x)
        sl.done.record(main)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(sl.done)
            sl.host_out.copy_(sl.y, non_blocking=True)
            sl.copied.record(self._copy_stream)
        # (a slot is resubmitted only after _collect has waited for `copied`: its y is never overwritten early)
        sl.pending = n

    def _collect(self, sl):
        sl.copied.synchronize()
        out = [sl.host_out_np[i].copy() if self.copy else sl.host_out_np[i] for i in range(sl.pending)]
        sl.pending = None
        return out

    def run(self, frames):
        """frames: iterable of uint8 (H,W,3) arrays, all of ONE shape per call (use `run_mixed` otherwise).  Yields uint8
        (4H,4W,3) arrays in order.  Full batches go through the pipelined slots; a ragged tail runs eagerly at its true size."""
        it = iter(frames)
        hw, k, inflight = None, 0, []
        while True:
            chunk = []
            for f in it:
                chunk.append(f)
                if len(chunk) == self.batch:
                    break
            if not chunk:
                break
            if hw is None:
                hw = (chunk[0].shape[0], chunk[0].shape[1])
            if len(chunk) < self.batch:           # the tail (or a bucket smaller than one batch)
                for sl in inflight:
                    yield from self._collect(sl)
                inflight = []
                yield from self._eager(chunk)
                break
            sl = self._slot(hw[0], hw[1], k % self.depth)
            if sl.pending is not None:
                inflight.remove(sl)
                yield from self._collect(sl)
            self._submit(sl, chunk)
            inflight.append(sl)
            k += 1
        for sl in inflight:
            yield from self._collect(sl)

    def run_mixed(self, frames):
        """Frames of any shapes: bucketed by (H, W), results returned as a list in input order."""
        frames = list(frames)
        out = [None] * len(frames)
        buckets = {}
        for i, f in enumerate(frames):
            buckets.setdefault((f.shape[0], f.shape[1]), []).append(i)
        for idx in buckets.values():
            for i, y in zip(idx, self.run(frames[j] for j in idx)):
                out[i] = y
        return out


def main(argv=None):
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image
    args = parser.parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("fast-srgan_amd runs on an MI355X only: no GPU is visible")
    device = "cuda"
    print(f"Using device: {device}")
    config = load_config("configs/config.yaml")
    model = load_generator(config, "models/model.pt", device, args.compute_dtype)
    image_paths = sorted(x for x in os.listdir(args.image_dir)
                         if x.lower().endswith(".png") or x.lower().endswith(".jpg") or x.lower().endswith("jpeg"))
    print(f"Found {len(image_paths)} to super resolve, starting...")
    pipe = InferencePipeline(model, device, batch=args.batch)

    def load(name):
        return np.array(Image.open(os.path.join(args.image_dir, name)).convert("RGB"))

    def save(name, arr):
        Image.fromarray(arr).save(os.path.join(args.output_dir, os.path.basename(name)))

    # decode and encode run in worker threads (PIL releases the GIL), the device pipeline in this one
    with ThreadPoolExecutor(max_workers=8) as pool:
        window = 16 * args.batch
        for start in range(0, len(image_paths), window):
            names = image_paths[start:start + window]
            frames = list(pool.map(load, names))
            saves = [pool.submit(save, n, y) for n, y in zip(names, pipe.run_mixed(frames))]
            for s_ in saves:
                s_.result()


if __name__ == "__main__":
    main()
