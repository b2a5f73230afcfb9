"""Test helper: selects where the kernels under test run.

  "emu" : the UNMODIFIED kernel sources compiled for the host by tests/emu (thread-per-lane emulation of
          waves, MFMA, LDS) -- index arithmetic and numerics can be checked without a GPU (CPU suite);
  "hip" : libfsr_hip.so on cuda:0 -- the product path (`-m gpu`).
"""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

L = importlib.import_module("fast-srgan_amd._lib")
ops = importlib.import_module("fast-srgan_amd.ops")

_emu_lib = None


def select(kind):
    """Returns the torch device for `kind` after routing the binding to the right library."""
    global _emu_lib
    if kind == "emu":
        if _emu_lib is None:
            from build_emu import build_emu
            _emu_lib = build_emu()
        L._install_for_testing(_emu_lib)
        return torch.device("cpu")
    L._install_for_testing(None)
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L.lib()  # raises when libfsr_hip.so is missing: the GPU tests must exercise the HIP path
    return torch.device("cuda:0")


BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]


def tol(cd_name, f32=2e-4, bf16=3e-2):
    """f16 has 3 more mantissa bits than bf16: it passes the bf16 bounds with room to spare (an eighth is asserted)."""
    return f32 if cd_name == "f32" else (bf16 if cd_name == "bf16" else max(bf16 / 8, 2 * f32))


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def relerr2(a, b):
    """Relative L2 error.  Gradients of (Leaky/P)ReLU networks are discontinuous in the activations' signs: a 1e-6
    perturbation of the weights already moves single elements of the fp32 ORACLE's own gradients by ~1% of the
    tensor maximum (measured: tests/probes/grad_sensitivity.py), so element-wise max-norm bounds are not meaningful for
    them; the L2 norm is."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(name, value):
    """Measured parity errors are appended to gpurun_out/parity_errors.log (the bf16 gates are set to ~2x what this file
    shows on the MI355X; keep them honest when kernels change)."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write("%s %.6g\n" % (name, value))
    except OSError:
        pass
    return value


def check_grads(prefix, named, ref, t_tensor, t_slope=0.25, t_cos=None, t_norm=None):
    """Gradient gates of the network-level tests.  `named`: [(name, gradient)], `ref`: {name: reference gradient}.
      * tensors (filters, biases): relative L2 error < t_tensor (relerr2 explains why not max-norm);
      * the single PReLU slopes: each is ONE cancelling sum over a whole layer (the fp32 oracle itself moves some of them by
        more than 100 % between float32 and float64, tests/probes/conditioning_probe.py), so they are held to |error| < t_slope x
        the largest slope gradient of the network instead of their own magnitude;
      * t_cos: lower bound of the cosine between all tensor gradients concatenated and the reference;
      * t_norm = (lo, hi): bounds of every tensor's norm ratio |g| / |reference| (a scaled or vanishing gradient passes an
        L2 gate of tens of per cent unnoticed).
    Every value goes to gpurun_out/parity_errors.log.  Returns the list of violations."""
    named = list(named)
    bad = []
    slopes = [(n, g) for n, g in named if g.numel() == 1]
    smax = max([abs(float(ref[n])) for n, _ in slopes] + [1e-30])
    dot = na = nb = 0.0
    for n, g in named:
        r = ref[n]
        if g.numel() == 1:
            e = report("%s.slope.%s" % (prefix, n), abs(float(g) - float(r)) / smax)
            if not e < t_slope:
                bad.append((n, "slope", e))
            continue
        e = report("%s.%s" % (prefix, n), relerr2(g, r))
        if not e < t_tensor:
            bad.append((n, e))
        a, b = g.detach().double().cpu().reshape(-1), r.detach().double().cpu().reshape(-1)
        nr = report("%s.normratio.%s" % (prefix, n), float(a.norm() / b.norm().clamp_min(1e-300)))
        if t_norm is not None and not (t_norm[0] < nr < t_norm[1]):
            bad.append((n, "normratio", nr))
        dot, na, nb = dot + float(a @ b), na + float(a @ a), nb + float(b @ b)
    cos = report("%s.cosine" % prefix, dot / max((na * nb) ** 0.5, 1e-300))
    if t_cos is not None and not cos > t_cos:
        bad.append(("cosine", cos))
    return bad
