"""End-to-end drop-in entry points on the GPU: `train.py a.b=c` and `inference.py --image_dir --output_dir`."""
import importlib
import os
import types
import warnings

import numpy as np
import pytest
import torch

from backend import select
from conftest import load_npz, sd_from
from oracle import srgan_cpu as O


@pytest.mark.gpu
def test_train_entry_point_runs_and_checkpoints(tmp_path, monkeypatch):
    select("hip")
    train = importlib.import_module("fast-srgan_amd.train")
    trainer_mod = importlib.import_module("fast-srgan_amd.trainer")
    rng = np.random.default_rng(0)
    npdir = tmp_path / "np"
    npdir.mkdir()
    for i in range(3):
        np.save(npdir / f"img{i}.npy", rng.integers(0, 256, size=(3, 90 + 7 * i, 120), dtype=np.uint8))
    (tmp_path / "configs").mkdir()
    monkeypatch.chdir(tmp_path)
    real_vgg = trainer_mod.VGG19
    monkeypatch.setattr(trainer_mod, "VGG19", lambda **kw: real_vgg(width_div=8 if kw.get("compute_dtype") == "f32" else 2, seed=1, **kw))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        train.main([f"data.numpy_dir={npdir}", "data.lr_image_size=16", "data.scale_factor=4", "generator.n_filters=32",
                    "generator.n_layers=1", "discriminator.n_filters=32", "training.batch_size=2",
                    "training.pretrain_iterations=2", "training.iterations=3", "training.log_iter=1",
                    "training.checkpoint_iter=3", "experiment.name=cli", "hydra.run.dir=run1"])
    # hydra 1.1 semantics (train.py:46): main ran inside the run directory, so `runs/` lies there, next to .hydra/
    assert os.path.exists(tmp_path / "run1" / ".hydra" / "config.yaml") and os.path.exists(tmp_path / "run1" / ".hydra" / "overrides.yaml")
    tmp_path = tmp_path / "run1"
    for f in ("generator_epoch_3.pt", "discriminator_epoch_3.pt", "generator_optim_epoch_3.pt", "discriminator_optim_epoch_3.pt"):
        assert os.path.exists(tmp_path / "runs" / "cli" / f), f            # trainer.py:143-156 file names
    assert os.path.exists(tmp_path / "runs" / "pretrain_generator.pt")
    sd = torch.load(tmp_path / "runs" / "cli" / "generator_epoch_3.pt", map_location="cpu")
    assert set(sd) >= {"neck.0.weight", "stem.0.conv1.weight", "upsampling.1.relu.weight", "head.0.bias"}
    assert all(torch.isfinite(v).all() for v in sd.values())


@pytest.mark.gpu
def test_inference_entry_point_matches_oracle(tmp_path, monkeypatch):
    from PIL import Image
    select("hip")
    inference = importlib.import_module("fast-srgan_amd.inference")
    z = load_npz("g_model_pt.npz")
    sd = sd_from(z, "sd.")
    (tmp_path / "models").mkdir()
    (tmp_path / "configs").mkdir()
    (tmp_path / "in").mkdir()
    torch.save({"_orig_mod." + k: v for k, v in sd.items()}, tmp_path / "models" / "model.pt")   # as shipped (inference.py:31-33)
    (tmp_path / "configs" / "config.yaml").write_text("generator:\n  n_filters: 64\n  n_layers: 8\ntraining:\n  compute_dtype: f32\n")
    rng = np.random.default_rng(1)
    imgs = {"a.png": rng.integers(0, 256, size=(20, 28, 3), dtype=np.uint8), "B.JPEG": rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8)}
    for name, arr in imgs.items():
        Image.fromarray(arr).save(tmp_path / "in" / name, quality=100)
    (tmp_path / "in" / "notes.txt").write_text("ignored")
    monkeypatch.chdir(tmp_path)
    inference.main(["--image_dir", "in", "--output_dir", "out"])
    assert sorted(os.listdir(tmp_path / "out")) == ["B.JPEG", "a.png"]
    lr = np.array(Image.open(tmp_path / "in" / "a.png").convert("RGB"))
    got = np.array(Image.open(tmp_path / "out" / "a.png"))
    x = (torch.from_numpy(lr) / 127.5 - 1.0).permute(2, 0, 1).unsqueeze(0)
    want = O.postprocess_u8(O.generator_forward(sd, x))                  # inference.py:53-56 truncating cast
    assert got.shape == (80, 112, 3)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02                    # +-1 only where y*255 sits on an integer boundary


@pytest.mark.gpu
def test_inference_pipeline_batched_frames(tmp_path):
    """InferencePipeline: a stream of mixed-shape frames, bucketed, batched (ragged tail), replayed as
    hipGraphs with overlapped uint8 D2H -- every frame byte-identical to the single-frame path, in input order."""
    dev = select("hip")
    pkg = importlib.import_module("fast-srgan_amd")
    z = load_npz("g_model_pt.npz")
    sd = sd_from(z, "sd.")
    G = pkg.Generator(types.SimpleNamespace(n_filters=64, n_layers=8), compute_dtype="bf16")
    G.load_state_dict(sd)
    G.to(dev).eval()
    rng = np.random.default_rng(5)
    shapes = [(24, 40)] * 7 + [(30, 30)] * 3 + [(24, 40)] * 2
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in shapes]
    pipe = pkg.InferencePipeline(G, dev, batch=4, depth=2, max_shapes=1)
    outs = pipe.run_mixed(frames)
    assert len(outs) == len(frames)
    # plans are built lazily and only for full batches: the 3-frame bucket and the 1-frame tail ran eagerly at their true size
    assert list(pipe._plans) == [(24, 40)] and all(sl is not None for sl in pipe._plans[(24, 40)])
    # ... and are evicted least-recently-used first (a folder of differently sized images must not pin
    # graph pools per shape for good)
    more = [rng.integers(0, 256, size=(40, 24, 3), dtype=np.uint8) for _ in range(4)]
    outs2 = pipe.run_mixed(more)
    assert list(pipe._plans) == [(40, 24)] and pipe._plans[(40, 24)][1] is None      # one batch: only slot 0 was ever built
    frames, outs = frames + more, outs + outs2
    inference = importlib.import_module("fast-srgan_amd.inference")
    for f, y in zip(frames, outs):
        assert y.dtype == np.uint8 and y.shape == (4 * f.shape[0], 4 * f.shape[1], 3)
        one = inference.super_resolve(G, f, dev)
        assert np.array_equal(y, one)
    # against the oracle on one frame (fp32 reference; bf16 kernels: a few grey levels)
    x = (torch.from_numpy(frames[0]) / 127.5 - 1.0).permute(2, 0, 1).unsqueeze(0)
    want = O.postprocess_u8(O.generator_forward(sd, x))
    assert np.abs(outs[0].astype(int) - want.astype(int)).mean() < 3.0


@pytest.mark.gpu
def test_bench_self_spawns_ranks_and_runs_the_rccl_path():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (here N = 1, forced);
    with a process group (FSR_FORCE_DIST=1: a 1-rank RCCL world) the step runs as three phase graphs around two real
    all-reduces and rank 0 prints the contract's JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FSR_BENCH_FORCE_SPAWN="1", FSR_FORCE_DIST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline", "--no-inference", "--no-f32", "--no-x3", "--no-bf16", "--no-sustained"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "images/s" and d["value"] > 0
    assert d["config"]["collectives"] == "rccl world 1" and d["config"]["launch"].startswith("3 phase hipGraphs")
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


def test_native_png_ingest_matches_pil(tmp_path):
    """train.write_images_to_numpy_arrays (reference train.py:22-37) through fsr_png_to_npy: RGB, RGBA, grey, grey + alpha and palette
    (8, 4, 2, 1 bits) PNGs written by PIL with several compression / filter settings decode to EXACTLY np.array(Image.open(p).convert("RGB")),
    land as (3, H, W) uint8 .npy files np.load reads, and the kinds the native decoder leaves out (interlaced, 16-bit) go through
    PIL, per file.  Host code only: runs against the emulation build too (the entry point does no device work)."""
    import numpy as np
    from PIL import Image
    from backend import select
    select("emu")
    train = importlib.import_module("fast-srgan_amd.train")
    rng = np.random.default_rng(5)
    src = tmp_path / "png"
    src.mkdir()
    h, w = 37, 53
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgb[5:20, 7:30] = rgb[5, 7]                     # flat areas: the Sub / Up / Paeth filters get picked by the encoder
    grad = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(src / "rgb.png")
    Image.fromarray(rgb, "RGB").save(src / "rgb_opt.png", optimize=True, compress_level=9)
    Image.fromarray(np.dstack([rgb, grad]), "RGBA").save(src / "rgba.png")
    Image.fromarray(grad, "L").save(src / "grey.png")
    Image.fromarray(np.dstack([grad, 255 - grad]), "LA").save(src / "grey_alpha.png")
    Image.fromarray(np.stack([grad] * 3, axis=2), "RGB").save(src / "smooth.png", compress_level=1)
    pal = Image.fromarray(rgb, "RGB").quantize(200)
    pal.save(src / "pal8.png")
    Image.fromarray(rgb, "RGB").quantize(16).save(src / "pal4.png", bits=4)
    Image.fromarray(rgb, "RGB").quantize(4).save(src / "pal2.png", bits=2)
    Image.fromarray(rgb, "RGB").quantize(2).save(src / "pal1.png", bits=1)
    Image.fromarray((rng.integers(0, 65536, (h, w))).astype(np.uint16)).save(src / "grey16.png")          # left to PIL
    names = sorted(os.listdir(src))
    out = tmp_path / "npy"
    train.write_images_to_numpy_arrays([str(src / n) for n in names], str(out), threads=3)
    for n in names:
        want = np.transpose(np.array(Image.open(src / n).convert("RGB")).astype(np.uint8), (2, 0, 1))
        got = np.load(out / n.replace(".png", ".npy"))
        assert got.dtype == np.uint8 and got.shape == want.shape, n
        assert np.array_equal(got, want), n
    # the per-file status: the 16-bit file is the only one the native decoder declined
    import ctypes
    L = importlib.import_module("fast-srgan_amd._lib")
    hh, ww = ctypes.c_int(0), ctypes.c_int(0)
    assert L.lib().fsr_png_decode_chw(os.fsencode(str(src / "grey16.png")), None, 0, ctypes.byref(hh), ctypes.byref(ww)) == -4
    assert L.lib().fsr_png_decode_chw(os.fsencode(str(src / "rgb.png")), None, 0, ctypes.byref(hh), ctypes.byref(ww)) == 0 and (hh.value, ww.value) == (h, w)
    assert L.lib().fsr_png_decode_chw(os.fsencode(str(out / "rgb.npy")), None, 0, ctypes.byref(hh), ctypes.byref(ww)) == -1
    # a hostile header (round-4 advisor): IHDR claims 2^31 - 1 x 2^31 - 1 pixels over a few bytes of IDAT -- status -2, no
    # allocation of the claimed raster, no exception across the C ABI (the process used to abort in std::length_error)
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)

    for dims in ((0x7fffffff, 0x7fffffff), (60000, 60000)):
        evil = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", dims[0], dims[1], 8, 2, 0, 0, 0)) \
            + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
        (src / "evil.png").write_bytes(evil)
        assert L.lib().fsr_png_decode_chw(os.fsencode(str(src / "evil.png")), None, 0, ctypes.byref(hh), ctypes.byref(ww)) == -2
    st = (ctypes.c_int * 1)()
    pa, na = (ctypes.c_char_p * 1)(os.fsencode(str(src / "evil.png"))), (ctypes.c_char_p * 1)(os.fsencode(str(out / "evil.npy")))
    assert L.lib().fsr_png_to_npy(pa, na, 1, 2, st) == 1 and st[0] == -2
