"""Operators of the Fast-SRGAN hot path: torch.autograd.Functions over the C ABI of libfsr_hip.so.

Layout contract: activations are NHWC tensors (N,H,W,C) in the compute dtype
(torch.bfloat16 for FSR_BF16, torch.float32 for FSR_F32) with C a multiple of CPAD (32 / 16);
3-channel images enter and leave as float32 NCHW-shaped tensors of any strides.  Every kernel is
enqueued on torch's current stream; nothing here synchronises, allocates pinned memory or falls
back to a torch implementation of the math.
"""
import ctypes
import os
import weakref

import torch

from . import _lib as L

_DT = {"f32": (L.FSR_F32, torch.float32, 16), "bf16": (L.FSR_BF16, torch.bfloat16, 32), "f16": (L.FSR_F16, torch.float16, 32),
       "x3": (L.FSR_X3, torch.float32, 32)}


class Compute:
    """Compute mode of a module: 'bf16' / 'f16' (16-bit MFMA, f32 accumulate; f16 = BASELINE configs[4]), 'f32' (exact-f32
    MFMA, the parity mode) or 'x3' (split bf16: every value is the pair hi = bf16(v), lo = bf16(v - hi), every product three
    bf16 MFMAs into one f32 accumulator -- the fast mode inside the reference's fp32 tolerance).

    An x3 activation lives in a float32 tensor of its logical shape (same bytes: per pixel and 32-channel group 64 bytes of
    hi, then 64 bytes of lo; include/fsr_hip.h).  The container is NOT float data: torch arithmetic on it is meaningless, only
    copies (cat along the batch, clone, slicing whole pixels) are legal -- x3_encode / x3_decode convert."""

    def __init__(self, name="f16"):
        if name not in _DT:
            raise ValueError("compute dtype must be 'bf16', 'f16', 'x3' or 'f32', got %r" % (name,))
        self.name = name
        self.code, self.torch_dtype, self.cpad = _DT[name]
        self.x3 = name == "x3"
        # packed filters: the MFMA operand type; an x3 pack holds 2 x K bf16 elements per row ([w_hi | w_lo] per 32 channels)
        self.pack_dtype = torch.bfloat16 if self.x3 else self.torch_dtype
        self.pack_kmul = 2 if self.x3 else 1
        self.is16 = name in ("bf16", "f16")     # plain 16-bit storage: the kernels that exist for those modes only

    def pad(self, c):
        return (c + self.cpad - 1) // self.cpad * self.cpad


def _empty(shape, dtype, device):
    """Activation storage.  x3 kernels address hi / lo halves through 128-byte blocks of the tensor: the GPU allocator hands out
    512-byte aligned blocks, the CPU allocator of the emulated test runs only 64-byte aligned ones."""
    if torch.device(device).type != "cpu":
        return torch.empty(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= d
    es = torch.empty((), dtype=dtype).element_size()
    flat = torch.empty(n + 128 // es, dtype=dtype, device=device)
    off = (-flat.data_ptr() % 128) // es
    return flat[off:off + n].view(shape)


def _empty_like(t):
    return _empty(tuple(t.shape), t.dtype, t.device)


def _aligned(t):
    """t, or a 128-byte aligned copy of it (CPU tensors of the emulated runs only)."""
    if t is None or t.is_cuda or t.data_ptr() % 128 == 0:
        return t
    out = _empty_like(t)
    out.copy_(t)
    return out


def x3_encode(x):
    """float32 (..., C) tensor, C % 32 == 0 -> its x3 storage form (a float32 tensor of the same shape; see Compute)."""
    if x.dtype != torch.float32 or x.shape[-1] % 32:
        raise ValueError("x3_encode expects a float32 tensor with a multiple of 32 channels, got %s %s" % (x.dtype, tuple(x.shape)))
    x = x.contiguous()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    g = x.shape[-1] // 32
    pair = torch.stack([hi.view(*x.shape[:-1], g, 32), lo.view(*x.shape[:-1], g, 32)], dim=-2)     # (..., g, 2, 32) bf16
    out = _empty(tuple(x.shape), torch.float32, x.device)
    out.view(torch.bfloat16).view(*x.shape[:-1], g, 2, 32).copy_(pair)
    return out


def x3_decode(t):
    """x3 storage (float32 container, (..., C)) -> the float32 values hi + lo."""
    t = t.contiguous()
    g = t.shape[-1] // 32
    pair = t.view(torch.bfloat16).view(*t.shape[:-1], g, 2, 32).float()
    return (pair[..., 0, :] + pair[..., 1, :]).reshape(t.shape)


def to_storage(cd, x):
    """float32 NHWC values -> the storage tensor of compute mode `cd` (tests, module boundaries)."""
    return x3_encode(x.float()) if cd.x3 else x.to(cd.torch_dtype)


def from_storage(cd, t):
    """storage tensor of compute mode `cd` -> float32 values."""
    return x3_decode(t) if cd.x3 else t.float()


class _StorageBoundaryFn(torch.autograd.Function):
    """Storage tensor of a compute mode -> float32 VALUES, differentiable: the backward re-encodes the cotangent.  The boundary
    every public forward() crosses before handing a tensor to arbitrary torch code -- an x3 activation is a float32 CONTAINER of
    interleaved bf16 hi / lo planes, and any torch arithmetic on it (autograd accumulation, .float(), a channel-wise cat)
    would be silent garbage (round-5 advisor, model.py:279)."""

    @staticmethod
    def forward(ctx, cd, t):
        ctx.cd = cd
        return from_storage(cd, t)

    @staticmethod
    def backward(ctx, g):
        return None, to_storage(ctx.cd, g.contiguous())


def values(cd, t):
    """float32 values of a storage tensor, with the cotangent routed back in storage form (module boundaries)."""
    return _StorageBoundaryFn.apply(cd, t) if cd.x3 else t


def add(cd, a, b):
    """a + b of two activation tensors (fsr_add): what autograd's accumulation would do, legal for x3 containers."""
    _check_dev(a, b)
    a, b = _aligned(a.contiguous()), _aligned(b.contiguous())
    out = _empty_like(a)
    L.check(L.lib().fsr_add(cd.code, _p(a), _p(b), _p(out), a.numel(), _stream()), "fsr_add")
    return out


def _stream():
    if L.is_emulation():
        return None
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _check_dev(*ts):
    for t in ts:
        if t is not None and not L.is_emulation() and not t.is_cuda:
            raise L.FsrError("fast-srgan_amd operators run on the GPU only (got a %s tensor); there is no CPU path" % t.device)


# ---------------------------------------------------------------------------------- packed filters
_pack_cache = {}   # id(weight) -> (weakref to the weight, {(mode, dtype, k_pad): (version, packed tensor)})


PACK_C3 = "c3"   # packed_filter mode of the first-layer (image-input) kernels
PACK_C3T = "c3t"  # ... of the head conv's data gradient run as a first-layer forward (transposed, tap-flipped filter)


class FilterSpec:
    """A filter handed to conv3x3_raw as (weight, pack mode, k_pad) instead of a packed tensor: the launch then asks the
    library which pack layout the kernel it will dispatch reads (fsr_conv3x3_pack_block: the standard [9][rows][K] image or
    the stage-contiguous one of the 128..512-channel kernels) and packs -- cached per weight version -- accordingly."""

    __slots__ = ("weight", "mode", "k_pad")

    def __init__(self, weight, mode, k_pad):
        self.weight, self.mode, self.k_pad = weight, mode, k_pad


def packed_filter(cd, weight, mode, k_pad, lin=0):
    """[9][rows_pad][k_pad] image of an OIHW float weight (fsr_pack_conv3x3), cached per weight OBJECT and version.
    lin = 64 / 128: the stage-contiguous layout of that channel-block size instead (fsr_pack_conv3x3_lin).

    Entries die with the weight (weak reference), so a recycled id()/address can never serve a stale filter.

    The image of a (weight, mode, dtype, k_pad) lives in ONE buffer for the lifetime of the weight: a newer weight
    version is re-packed IN PLACE.  A captured hipGraph therefore bakes in addresses that stay valid, and the re-pack
    launches it contains (the optimizer step inside the graph bumps the version, so the next use re-packs) refresh
    exactly the memory the next replay reads -- no replay can see the filters of capture time."""
    ver = (weight._version, getattr(weight, "_fsr_epoch", 0), weight.data_ptr())
    slot = _pack_cache.get(id(weight))
    if slot is None or slot[0]() is not weight:
        wid = id(weight)
        slot = (weakref.ref(weight, lambda _r, wid=wid: _pack_cache.pop(wid, None)), {})
        _pack_cache[wid] = slot
    key = (mode, cd.code, k_pad, lin)
    hit = slot[1].get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    cout, cin = weight.shape[0], weight.shape[1]
    fwd = mode in (L.PACK_FWD, L.PACK_FWD_PS)
    rows = cout if fwd else cin
    rows_pad = (rows + 15) // 16 * 16
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    _check_dev(w)
    if mode == PACK_C3T:
        numel = ((cin + 15) // 16 * 16) * 32
    elif mode == PACK_C3:
        numel = ((cout + 15) // 16 * 16) * 32
    else:
        numel = 9 * (rows if lin else rows_pad) * k_pad * cd.pack_kmul
    out = hit[1] if (hit is not None and hit[1].device == w.device) else None
    if out is None:
        # (x3: the first-layer kernels compute in exact f32 on x3 storage, their filter image is float)
        out = torch.empty(numel, dtype=torch.float32 if (cd.x3 and mode in (PACK_C3, PACK_C3T)) else cd.pack_dtype, device=w.device)
    if lin:
        if k_pad != (cin if fwd else cout) or mode in (PACK_C3, PACK_C3T):
            raise L.FsrError("the stage-contiguous filter pack has no padding")
        L.check(L.lib().fsr_pack_conv3x3_lin(cd.code, mode, _p(w), cout, cin, lin, _p(out), _stream()), "fsr_pack_conv3x3_lin")
    elif mode == PACK_C3:     # first-layer kernels: [rows_pad][32]
        L.check(L.lib().fsr_pack_conv3x3_c3(cd.code, _p(w), cout, _p(out), 0, _stream()), "fsr_pack_conv3x3_c3")
    elif mode == PACK_C3T:  # head conv [3][cin][3][3]: rows = its input channels
        L.check(L.lib().fsr_pack_conv3x3_c3(cd.code, _p(w), cin, _p(out), 1, _stream()), "fsr_pack_conv3x3_c3")
    else:
        L.check(L.lib().fsr_pack_conv3x3(cd.code, mode, _p(w), cout, cin, k_pad, _p(out), _stream()), "fsr_pack_conv3x3")
    slot[1][key] = (ver, out)
    return out


_ws = {}
_const = {}


class ZeroPool:
    """Per-iteration arena of zero-initialised float32 scratch (InstanceNorm statistics, reduction outputs, bias /
    PReLU gradients, loss accumulators): ONE memset per iteration instead of ~160 tiny fill kernels.  `reset()` at the
    start of an iteration re-zeroes the arena and rewinds it; tensors carved from it live until the next reset."""

    def __init__(self, device, nfloats=4 << 20):
        self.buf = torch.zeros(nfloats, dtype=torch.float32, device=device)
        self.off = 0
        self.active = False

    def reset(self):
        if self.off:
            self.buf[:self.off].zero_()
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4          # keep 16-byte alignment for vector loads
        if self.off + n4 > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n].view(shape)
        self.off += n4
        return t


_zero_pool = {}


def zero_pool_reset(device):
    """Called by the Trainer at the start of every iteration (creates the arena on first use)."""
    key = str(device)
    pool = _zero_pool.get(key)
    if pool is None:
        pool = _zero_pool[key] = ZeroPool(device)
    pool.reset()
    pool.active = True


def zero_pool_end(device):
    """End of the iteration: later allocations (inference, metrics, captured inference graphs) must get their own
    zero-filled memory -- the arena is only re-zeroed by the next `zero_pool_reset`."""
    pool = _zero_pool.get(str(device))
    if pool is not None:
        pool.active = False


def _zeros(shape, device):
    pool = _zero_pool.get(str(device))
    if pool is not None and pool.active:
        t = pool.take(tuple(shape))
        if t is not None:
            return t
    return torch.zeros(shape, dtype=torch.float32, device=device)


def _assigned(shape, device):
    """float32 scratch for a result the kernels ASSIGN (the order-fixed second-level reduce overwrites it): from the
    iteration's arena when one is open, else uninitialised -- inference then launches no fill kernel per statistics tensor."""
    pool = _zero_pool.get(str(device))
    if pool is not None and pool.active:
        t = pool.take(tuple(shape))
        if t is not None:
            return t
    return torch.empty(shape, dtype=torch.float32, device=device)


def _const_vec(values, device):
    """Small float constant vector on `device`, uploaded once (no host-to-device copies inside captured steps)."""
    key = (tuple(values), str(device))
    t = _const.get(key)
    if t is None:
        t = torch.tensor(values, dtype=torch.float32, device=device)
        _const[key] = t
    return t

USE_LIN_PACK = os.environ.get("FSR_PACK_LIN", "1") != "0"   # A/B switch: 0 = every launch reads the standard filter pack
# A/B switch: 0 = activation-gradient masks are always the saved tensors.  Only conv_s2d3 reads the packed bits, so switching
# that kernel off (FSR_S2D3=0) switches the bits off with it instead of leaving a mask no kernel takes
USE_SIGN_BITS = os.environ.get("FSR_SIGN_BITS", "1") != "0" and os.environ.get("FSR_S2D3", "1") != "0"
_pack_block_memo = {}     # (descriptor fields, optional-tensor mask) -> block size: the dispatch is deterministic per shape
_DISPATCH_ENV = ("FSR_PERSIST_CUS", "FSR_T3_ROWS", "FSR_S2D3", "FSR_T3N_G3", "FSR_CONV_STAGE", "FSR_CONV64_S2FWD",
                 "FSR_PACK_LIN")   # every switch the conv dispatch reads (csrc: getenv) + the Python-side ones
USE_POOL_ARGMAX = os.environ.get("FSR_POOL_ARGMAX", "1") != "0"   # A/B switch: 0 = the pool backward re-reads its input and output
USE_C3_KERNELS = True   # tests flip this to compare the first-layer kernels with the padded-tensor path

# bench.py sets this to a list to collect (start_event, end_event, algorithmic_flops, algorithmic_bytes, kernel name, kind)
# per convolution launch; kind is "fwd", "dgrad" or "wgrad"
PROFILE_CONV = None


def _last_kernel():
    name = L.lib().fsr_last_kernel()
    return name.decode() if name else "?"


_ws_retired = []    # outgrown scratch buffers that a captured hipGraph may still address: never freed
_capture_seen = {}  # (device, stream) -> the current buffer was handed out during a stream capture


def _workspace(nbytes, device):
    """Scratch (split-K partials of the weight gradients, per-workgroup partials of the reductions), one buffer per
    (device, stream): launches on different streams may overlap; on one stream a producer and the kernel that finishes
    its partials are enqueued back to back by the same C-ABI call, so the next call may reuse the buffer."""
    key = (str(device), _stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            # a captured hipGraph may have baked this address in: keep the buffer alive only in that case (a buffer that no
            # capture can have seen is simply dropped, so a run of growing requests does not pile up)
            if _capture_seen.get(key):
                _ws_retired.append(buf)
            # geometric growth (O(log) reallocations, not one per larger request), capped: doubling below 64 MB, +25 % above --
            # a retired generation stays pinned for as long as a captured graph may address it, so large split-K workspaces
            # must not double their way to hundreds of MB each
            old = buf.numel() * 4
            nbytes = max(nbytes, 2 * old if old < (64 << 20) else old + old // 4)
        buf = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _ws[key] = buf
        _capture_seen[key] = False
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        _capture_seen[key] = True
    return buf


# ---------------------------------------------------------------------------------- weight gradients on their own stream
# A weight gradient depends only on (x, dz) and nothing downstream depends on it before the optimizer step, so it can run
# beside the data-gradient chain instead of inside it.  `wgrad_stream_begin(device)` (the trainer, per iteration) turns
# this on; Conv3x3Fn.backward then launches its weight gradient on the side stream after an event of the producing
# stream, and `wgrad_stream_join()` makes the current stream wait for all of them (before gradients are reduced / used).
# The tensors a queued launch reads are kept referenced until the join, so the caching allocator cannot hand their
# memory to later work on the producing stream.
#
# Layers of ONE shape (the generator's 64 -> 64 stem: 17 of them, model.py:62-80) are additionally DEFERRED while the
# trainer's iteration is open: backward only queues (x, dz, arena) and the join launches the whole group as one
# fsr_conv3x3_wgrad_grouped call -- the layers share one launch's workgroups (15 slabs each instead of 256), so the
# split-K partials shrink 17x and 17 reduce launches become one.
_wgrad_state = {"stream": None, "active": False, "refs": [], "group": False, "pending": {}}
USE_WGRAD_STREAM = os.environ.get("FSR_WGRAD_STREAM", "1") != "0"
USE_WGRAD_GROUP = os.environ.get("FSR_WGRAD_GROUP", "1") != "0"
WGRAD_GROUP_MAX = 32


def wgrad_stream_begin(device):
    _wgrad_state["group"] = USE_WGRAD_GROUP
    if not USE_WGRAD_STREAM or L.is_emulation() or not torch.cuda.is_available():
        return
    if _wgrad_state["stream"] is None:
        _wgrad_state["stream"] = torch.cuda.Stream(device=device)
    _wgrad_state["active"] = True


def _wgrad_defer(cd, xin, dz, cout, cin, cfg, arena):
    """Queue a weight gradient for the grouped launch; False if this layer is not of the groupable class."""
    st = _wgrad_state
    if not st["group"] or cfg.stride != 1 or cfg.pixel_shuffle or cd.x3:
        return False
    if not (cout == cin == 64 and xin.shape[3] == 64 and dz.shape[3] == 64):
        return False
    st["pending"].setdefault((cd.code, tuple(xin.shape), str(xin.device)), []).append((cd, xin, dz, arena))
    return True


def _wgrad_flush():
    st = _wgrad_state
    pending, st["pending"] = st["pending"], {}
    lib = L.lib()
    for (_code, shape, _dev), items in pending.items():
        for i0 in range(0, len(items), WGRAD_GROUP_MAX):
            grp = items[i0:i0 + WGRAD_GROUP_MAX]
            cd, x0, dz0, _ = grp[0]
            if len(grp) == 1:
                with _on_wgrad_stream(x0, dz0):
                    conv3x3_wgrad_raw(cd, x0, dz0, 64, 64, 1, out=grp[0][3])
                continue
            n, ih, iw, _c = shape
            d = L.WgradDesc(cd.code, n, ih, iw, 64, 64, ih, iw, 64, 64, 1, 0)
            need = lib.fsr_conv3x3_wgrad_grouped_workspace(ctypes.byref(d), len(grp))
            if need == 0:
                L.check(-2, "fsr_conv3x3_wgrad_grouped_workspace")
            arr = ctypes.c_void_p * len(grp)
            xs, dys, dws = arr(*[_p(g[1]) for g in grp]), arr(*[_p(g[2]) for g in grp]), arr(*[_p(g[3]) for g in grp])
            with _on_wgrad_stream(*[t for g in grp for t in (g[1], g[2])]):
                prof = PROFILE_CONV
                if prof is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                L.check(lib.fsr_conv3x3_wgrad_grouped(ctypes.byref(d), len(grp), xs, dys, dws, _p(_workspace(need, x0.device)),
                                                      _stream()), "fsr_conv3x3_wgrad_grouped")
                if prof is not None:
                    ev1.record()
                    nbytes = len(grp) * (2 * x0.numel() * x0.element_size() + 64 * 64 * 9 * 4)
                    prof.append((ev0, ev1, len(grp) * 2.0 * n * ih * iw * 64 * 64 * 9, nbytes, "conv_wgrad_kernel", "wgrad"))


def wgrad_stream_join():
    st = _wgrad_state
    if st["pending"]:
        _wgrad_flush()
    if st["active"] and st["refs"]:
        torch.cuda.current_stream().wait_stream(st["stream"])
    st["refs"] = []


def wgrad_stream_end():
    # an iteration that ran to its optimizer steps has joined (and flushed) already; what is still queued here belongs to an
    # iteration that raised: drop it instead of launching on its tensors
    _wgrad_state["pending"] = {}
    wgrad_stream_join()
    _wgrad_state["active"] = False
    _wgrad_state["group"] = False


class _on_wgrad_stream:
    """with _on_wgrad_stream(tensors): the body's launches go to the weight-gradient stream (if active), ordered after
    everything queued so far on the current stream."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        st = _wgrad_state
        if st["active"] and self.tensors[0].is_cuda:
            st["stream"].wait_stream(torch.cuda.current_stream())
            st["refs"].append(self.tensors)
            self.ctx = torch.cuda.stream(st["stream"])
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# ---------------------------------------------------------------------------------- raw launches
def conv3x3_raw(cd, x, wpk, cout, *, mode=L.CONV_FWD, out_hw=None, stride=1, bias=None, act=L.ACT_NONE, slope=0.0,
                prelu=None, oscale=None, pixel_shuffle=False, in_pixel_shuffled=False, out_f32=False, want_stats=False,
                want_preact=False, alg_k=None, dact_mask=None, dact_slope=0.0, out_u8=False, pool2=False, dact_add=False, dact_bits=False):
    """One fsr_conv3x3 launch.  x: (N,IH,IW,Cin) [or its depth-to-space form when in_pixel_shuffled].
    out_u8 (tanh heads): the output is the finished uint8 HWC image of inference.py:53-56.
    pool2 (no-grad passes): the output is MaxPool2d(2,2) of the activated result; the full-resolution tensor is never written.
    dact_add: `dact_mask` is ADDED to the result (the gradient of a skip connection) instead of gating it.
    dact_bits: `dact_mask` is the packed sign-bit tensor (N,OH,OW,Cout/8) uint8 of the producing layer's output (stride-2 data gradients)."""
    _check_dev(x)
    if cd.x3:
        x, dact_mask = _aligned(x), (_aligned(dact_mask) if dact_mask is not None else None)
    n = x.shape[0]
    if in_pixel_shuffled:
        ih, iw, cin = x.shape[1] // 2, x.shape[2] // 2, x.shape[3] * 4
    else:
        ih, iw, cin = x.shape[1], x.shape[2], x.shape[3]
    if mode == L.CONV_FWD:
        oh, ow = (ih - 1) // stride + 1, (iw - 1) // stride + 1
    else:
        oh, ow = out_hw
    odt = torch.uint8 if out_u8 else (torch.float32 if out_f32 else cd.torch_dtype)
    oshape = (n, 2 * oh, 2 * ow, cout // 4) if pixel_shuffle else ((n, oh // 2, ow // 2, cout) if pool2 else (n, oh, ow, cout))
    out = _empty(oshape, odt, x.device)
    pre = _empty(oshape, odt, x.device) if want_preact else None
    stats = _assigned((n, cout, 2), x.device) if want_stats else None
    d = L.ConvDesc(cd.code, mode, n, ih, iw, cin, oh, ow, cout, stride, act, float(slope), int(pixel_shuffle),
                   int(in_pixel_shuffled), L.OUT_U8 if out_u8 else int(out_f32), int(pool2), 2 if dact_bits else int(bool(dact_add)), 0)
    if isinstance(wpk, FilterSpec):     # pack in the layout the kernel this launch dispatches reads
        opt = ((L.OPT_BIAS if bias is not None else 0) | (L.OPT_PRELU if prelu is not None else 0) | (L.OPT_OSCALE if oscale is not None else 0)
               | (L.OPT_MASK if dact_mask is not None else 0) | (L.OPT_PREACT if want_preact else 0) | (L.OPT_STATS if want_stats else 0))
        blk = 0
        if USE_LIN_PACK:
            # one ctypes call (a walk of the whole dispatch chain) per SHAPE, not per launch: eager paths (validation,
            # inference, f32 / x3 without graphs) otherwise pay it for every convolution
            # (the key carries everything the C-side dispatch reads besides the descriptor: the device -- its CU count sizes the
            # persistent grids --, the library in use and every dispatch switch of the environment; round-5 advisor)
            key = (tuple(getattr(d, f) for f, _ in d._fields_), opt, L.is_emulation(), str(x.device), L.LIB_PATH,
                   tuple(os.environ.get(v) for v in _DISPATCH_ENV))
            blk = _pack_block_memo.get(key)
            if blk is None:
                blk = L.lib().fsr_conv3x3_pack_block(ctypes.byref(d), opt)
                if blk < 0:
                    L.check(blk, "fsr_conv3x3_pack_block")
                _pack_block_memo[key] = blk
        d.pack_lin = blk
        wpk = packed_filter(cd, wpk.weight, wpk.mode, wpk.k_pad, lin=blk)
    scratch = _workspace(L.lib().fsr_conv3x3_scratch(ctypes.byref(d)), x.device) if want_stats else None
    prof = PROFILE_CONV
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    L.check(L.lib().fsr_conv3x3(ctypes.byref(d), _p(x), _p(wpk), _p(bias), _p(prelu), _p(oscale), _p(dact_mask),
                                float(dact_slope), _p(out), _p(pre), _p(stats), _p(scratch), _stream()), "fsr_conv3x3")
    if prof is not None:
        ev1.record()
        k = cin if alg_k is None else alg_k
        pix = oh * ow if mode == L.CONV_FWD else ih * iw      # dgrad: one MAC per forward MAC
        # every tensor the launch has to read or write once: input, output, filter, and the fused extras (activation-gradient
        # mask read, pre-activation copy written)
        nbytes = x.numel() * x.element_size() + out.numel() * out.element_size() + wpk.numel() * wpk.element_size()
        if dact_mask is not None:
            nbytes += dact_mask.numel() * dact_mask.element_size()
        if pre is not None:
            nbytes += pre.numel() * pre.element_size()
        prof.append((ev0, ev1, 2.0 * n * pix * cout * k * 9, nbytes, _last_kernel(), "fwd" if mode == L.CONV_FWD else "dgrad"))
    return out, pre, stats


def conv3x3_wgrad_raw(cd, x, dy, cout, cin, stride, dy_pixel_shuffled=False, out=None):
    """OIHW float weight gradient.  x (N,IH,IW,CinPad); dy (N,OH,OW,CoutPad) or its depth-to-space form."""
    if cd.x3:
        x, dy = _aligned(x), _aligned(dy)
    n, ih, iw, cin_pad = x.shape
    if dy_pixel_shuffled:
        oh, ow, cout_pad = dy.shape[1] // 2, dy.shape[2] // 2, dy.shape[3] * 4
    else:
        oh, ow, cout_pad = dy.shape[1], dy.shape[2], dy.shape[3]
    d = L.WgradDesc(cd.code, n, ih, iw, cin_pad, cin, oh, ow, cout_pad, cout, stride, int(dy_pixel_shuffled))
    need = L.lib().fsr_conv3x3_wgrad_workspace(ctypes.byref(d))
    if need == 0:
        L.check(-2, "fsr_conv3x3_wgrad_workspace")
    ws = _workspace(need, x.device)
    dw = out if out is not None else torch.zeros((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    prof = PROFILE_CONV
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    L.check(L.lib().fsr_conv3x3_wgrad(ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(ws), _stream()), "fsr_conv3x3_wgrad")
    if prof is not None:
        ev1.record()
        nbytes = x.numel() * x.element_size() + dy.numel() * dy.element_size() + cout * cin * 9 * 4
        prof.append((ev0, ev1, 2.0 * n * oh * ow * cout * cin * 9, nbytes, "conv_wgrad_kernel", "wgrad"))
    return dw


def image_to_nhwc(cd, img, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
    """(N,3,H,W) float32 of any strides -> zero-padded (N,H,W,CPAD) compute-dtype tensor."""
    _check_dev(img)
    if img.dtype != torch.float32:
        img = img.float()
    n, c, h, w = img.shape
    if c != 3:
        raise ValueError("expected a 3-channel image batch, got %s" % (tuple(img.shape),))
    out = _empty((n, h, w, cd.cpad), cd.torch_dtype, img.device)
    sn, sc, sh, sw = img.stride()
    L.check(L.lib().fsr_image_to_nhwc(cd.code, _p(img), sn, sc, sh, sw, n, h, w, scale[0], scale[1], scale[2], shift[0],
                                      shift[1], shift[2], _p(out), cd.cpad, _stream()), "fsr_image_to_nhwc")
    return out


# ---------------------------------------------------------------------------------- autograd: convolution
class ConvCfg:
    """Static description of one fused convolution (kept out of autograd's tensor arguments)."""

    def __init__(self, cd, *, stride=1, act=L.ACT_NONE, slope=0.0, pixel_shuffle=False, stats=False, image_in=False,
                 in_scale=(1.0, 1.0, 1.0), in_shift=(0.0, 0.0, 0.0), tanh_head=False, input_act_bwd=None,
                 act_bwd_by_consumer=False, u8_head=False, pool_after=False, n_alias=0, emit_signs=False):
        # u8_head (inference only, with tanh_head): the head stores the finished uint8 HWC frame instead of float
        # input_act_bwd = slope: the data-gradient launch also applies the backward of the ReLU (0.0) / LeakyReLU
        #   that produced this conv's input (the mask is the saved input itself), so the tensor it returns is
        #   already the producer's pre-activation gradient;
        # act_bwd_by_consumer: the incoming gradient already went through this conv's own activation backward
        #   (its consumer -- a conv with input_act_bwd or a pool with relu_mask -- did it); a bias gradient, if
        #   needed, then costs one read pass (column sums) instead of the three passes of a separate act_bwd.
        self.input_act_bwd, self.act_bwd_by_consumer = input_act_bwd, act_bwd_by_consumer
        self.cd, self.stride, self.act, self.slope = cd, stride, act, slope
        self.pixel_shuffle, self.stats, self.image_in = pixel_shuffle, stats, image_in
        self.in_scale, self.in_shift, self.tanh_head = in_scale, in_shift, tanh_head
        self.u8_head = u8_head and tanh_head
        # pool_after (no-grad passes only): the MaxPool2d(2,2) that follows this conv + ReLU is taken in the epilogue and
        # only the pooled tensor is stored (the full-resolution tensor is what a backward pass would need)
        self.pool_after = pool_after
        # n_alias (1 or 2): forward also returns that many ALIASES of its input for the caller's skip connections
        # (model.py:69, :115).  Their gradients come back to THIS function's backward, whose data-gradient launch adds the
        # first one in its epilogue (fsr_conv_desc.mask_is_addend): autograd has nothing left to accumulate on the block input.
        self.n_alias = n_alias
        # emit_signs (first-layer kernels, training, 16-bit): the forward also writes the SIGN BITS of its output (N,H,W,Cout/8 uint8)
        # and hangs them on the output tensor (`_fsr_signs`); a stride-2 consumer with input_act_bwd then reads those -- a sixteenth
        # of the bytes -- as its activation-gradient mask instead of the tensor itself
        self.emit_signs = emit_signs


class Conv3x3Fn(torch.autograd.Function):
    """Conv2d(k=3,p=1) + bias + activation [+ PixelShuffle(2)] [+ InstanceNorm statistics].

    forward(x, weight, bias, prelu_weight, cfg) -> (out, stats)
      x      : NHWC activation, or (image_in) a float32 NCHW image batch of any strides
      out    : NHWC activation; (tanh_head) a float32 (N,3,H,W) view of the NHWC head output
      stats  : (N,Cout,2) float32 sums of the pre-activation, or an empty tensor
    """

    @staticmethod
    def forward(ctx, x, weight, bias, prelu, cfg, grad_on=True):
        # grad_on: torch.is_grad_enabled() at the call site (grad mode is always off INSIDE Function.forward, and
        # ctx.needs_input_grad reports the parameters' requires_grad even under no_grad): inference must not pay for the
        # tensors only a backward pass reads (the pre-activation copy of the PReLU layers)
        ctx.grad_on = grad_on
        cd = cfg.cd
        cout, cin = weight.shape[0], weight.shape[1]
        if (cfg.image_in and cfg.stride == 1 and cout % 16 == 0 and not (cfg.pixel_shuffle or cfg.stats or cfg.tanh_head)
                and USE_C3_KERNELS):
            return Conv3x3Fn._forward_c3(ctx, x, weight, bias, prelu, cfg)
        ctx.c3 = False
        # sign bits of the input (hung on it by the first-layer kernel that produced it): the stride-2 data gradient's mask
        ctx.in_signs = getattr(x, "_fsr_signs", None) if (cfg.input_act_bwd is not None and cfg.stride == 2 and cd.is16) else None
        if cfg.image_in:
            xin = image_to_nhwc(cd, x, cfg.in_scale, cfg.in_shift)
        else:
            xin = x if x.is_contiguous() else x.contiguous()
            if xin.dtype != cd.torch_dtype:
                raise L.FsrError("activation dtype %s does not match the module's compute dtype %s" % (xin.dtype, cd.name))
        cin_pad = xin.shape[3]
        wpk = FilterSpec(weight, L.PACK_FWD_PS if cfg.pixel_shuffle else L.PACK_FWD, cin_pad)
        training = ctx.grad_on and any(ctx.needs_input_grad)
        if (cfg.u8_head or cfg.pool_after) and training:
            raise L.FsrError("the uint8 head / fused max-pool epilogues are inference-only: they have no gradient")
        act = L.ACT_TANH if cfg.tanh_head else cfg.act
        want_pre = training and act == L.ACT_PRELU
        b32 = bias if bias is None or bias.dtype == torch.float32 else bias.float()
        out, pre, stats = conv3x3_raw(cd, xin, wpk, cout, stride=cfg.stride, bias=b32, act=act, slope=cfg.slope,
                                      prelu=prelu, pixel_shuffle=cfg.pixel_shuffle, out_f32=cfg.tanh_head,
                                      want_stats=cfg.stats, want_preact=want_pre, alg_k=cin, out_u8=cfg.u8_head,
                                      pool2=cfg.pool_after)
        ctx.cfg = cfg
        ctx.dims = (cout, cin, tuple(xin.shape))
        ctx.has_bias = bias is not None
        ctx.x_is_image = cfg.image_in
        saved_act = pre if want_pre else (out if act in (L.ACT_RELU, L.ACT_LEAKY, L.ACT_TANH) else None)
        ctx.save_for_backward(xin, weight, prelu, saved_act)
        if stats is None:
            stats = torch.empty(0, device=out.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)    # no zero-filled gradient tensor for the statistics output per backward call
        if cfg.u8_head:
            return out, stats               # (N,H,W,3) uint8: already the frame layout inference.py:55 permutes to
        if cfg.tanh_head:
            return out.permute(0, 3, 1, 2), stats
        if cfg.n_alias:
            if cfg.image_in or cfg.input_act_bwd is not None or cfg.pixel_shuffle or cfg.stride != 1:
                raise L.FsrError("skip aliases are for plain stride-1 convolutions on NHWC activations")
            return (out, stats) + tuple(xin.view_as(xin) for _ in range(cfg.n_alias))
        return out, stats

    @staticmethod
    def _forward_c3(ctx, x, weight, bias, prelu, cfg):
        """First layer straight from the (N,3,H,W) float image (fsr_conv3x3_c3_fwd): no padded NHWC copy."""
        cd = cfg.cd
        _check_dev(x)
        if x.dtype != torch.float32:
            x = x.float()
        n, c, h, w = x.shape
        if c != 3:
            raise ValueError("expected a 3-channel image batch, got %s" % (tuple(x.shape),))
        cout = weight.shape[0]
        wpk = packed_filter(cd, weight, PACK_C3, 32)
        training = ctx.grad_on and any(ctx.needs_input_grad)
        want_pre = training and cfg.act == L.ACT_PRELU
        b32 = bias if bias is None or bias.dtype == torch.float32 else bias.float()
        out = _empty((n, h, w, cout), cd.torch_dtype, x.device)
        pre = _empty_like(out) if want_pre else None
        signs = (torch.empty((n, h, w, cout // 8), dtype=torch.uint8, device=x.device)
                 if (cfg.emit_signs and USE_SIGN_BITS and training and cd.is16 and cout % 64 == 0) else None)
        sn, sc, sh, sw = x.stride()
        prof = PROFILE_CONV
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(L.lib().fsr_conv3x3_c3_fwd(cd.code, _p(x), sn, sc, sh, sw, n, h, w, *cfg.in_scale, *cfg.in_shift, _p(wpk),
                                           _p(b32), cfg.act, float(cfg.slope), _p(prelu), cout, _p(out), _p(pre), _p(signs), _stream()),
                "fsr_conv3x3_c3_fwd")
        if signs is not None:
            out._fsr_signs = signs
        if prof is not None:
            ev1.record()
            nbytes = x.numel() * 4 + out.numel() * out.element_size() * (2 if want_pre else 1) + wpk.numel() * wpk.element_size()
            prof.append((ev0, ev1, 2.0 * n * h * w * cout * 27, nbytes, "conv_c3_fwd_kernel", "fwd"))
        ctx.cfg = cfg
        ctx.c3 = True
        ctx.bias_param = bias
        ctx.dims = (cout, 3, (n, h, w, cd.cpad))
        ctx.has_bias = bias is not None
        ctx.x_is_image = True
        saved_act = pre if want_pre else (out if cfg.act in (L.ACT_RELU, L.ACT_LEAKY) else None)
        ctx.save_for_backward(x, weight, prelu, saved_act)
        stats = torch.empty(0, device=out.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)    # no zero-filled gradient tensor for the statistics output per backward call
        return out, stats

    @staticmethod
    def backward(ctx, g, _gstats, *g_alias):
        skips = [t for t in g_alias if t is not None]
        if g is None:       # (gradients are not materialised: only the statistics output -- or only a skip alias -- was used)
            dx = None
            for t in skips:
                dx = t if dx is None else add(ctx.cfg.cd, dx, t)
            return dx, None, None, None, None, None
        cfg, cd = ctx.cfg, ctx.cfg.cd
        xin, weight, prelu, saved = ctx.saved_tensors
        cout, cin, xshape = ctx.dims
        n, ih, iw, cin_pad = xshape
        lib, st = L.lib(), _stream()
        dbias = _zeros((cout,), xin.device) if ctx.has_bias else None
        dprelu = None
        act = L.ACT_TANH if cfg.tanh_head else cfg.act
        if cfg.tanh_head and USE_C3_KERNELS and cout == 3 and cin_pad % 16 == 0 and cin == cin_pad:
            return Conv3x3Fn._backward_head_c3(ctx, g)
        if cfg.tanh_head:
            # g: (N,3,H,W) float of any strides; saved: head output (N,H,W,3) float
            if g.dtype != torch.float32:
                g = g.float()
            _, _, h, w = g.shape
            dz = _empty((n, h, w, cd.cpad), cd.torch_dtype, xin.device)
            sn, sc, sh, sw = g.stride()
            L.check(lib.fsr_tanh_bwd_to_nhwc(cd.code, _p(g), sn, sc, sh, sw, _p(saved), n, h, w, _p(dz), cd.cpad, _p(dbias),
                                             _p(_workspace(lib.fsr_tanh_bwd_scratch(), xin.device)), st), "fsr_tanh_bwd_to_nhwc")
        else:
            g = g if g.is_contiguous() else g.contiguous()
            if cd.x3:
                g, saved = _aligned(g), _aligned(saved)
            # first-layer kernels with arena gradients: the weight-gradient launch also produces the bias gradient (a column
            # of ones in its padded K dimension), straight into the bias' arena slice
            bias_arena = getattr(getattr(ctx, "bias_param", None), "_fsr_grad", None) if ctx.c3 else None
            fused_dbias = (bias_arena is not None and cfg.act_bwd_by_consumer and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
                           and getattr(weight, "_fsr_grad", None) is not None)
            if cfg.act_bwd_by_consumer:
                dz = g          # already multiplied by this layer's act'() in the consumer's data-gradient epilogue
                if ctx.has_bias and ctx.needs_input_grad[2] and not fused_dbias:    # bias gradient: column sums of dz, one read pass
                    _, h, w, c = g.shape
                    scr = _workspace(lib.fsr_act_bwd_scratch(n, h, w, c, int(cfg.pixel_shuffle)), xin.device)
                    L.check(lib.fsr_act_bwd(cd.code, _p(g), None, L.ACT_NONE, 0.0, None, None, _p(dbias), None, _p(scr), n, h, w,
                                            c, int(cfg.pixel_shuffle), st), "fsr_act_bwd")
            elif act != L.ACT_NONE or ctx.has_bias:
                if act == L.ACT_PRELU:
                    dprelu = _zeros((1,), xin.device)
                dz = _empty_like(g)
                _, h, w, c = g.shape
                scr = _workspace(lib.fsr_act_bwd_scratch(n, h, w, c, int(cfg.pixel_shuffle)), xin.device)
                L.check(lib.fsr_act_bwd(cd.code, _p(g), _p(saved), act, float(cfg.slope), _p(prelu), _p(dz), _p(dbias),
                                        _p(dprelu), _p(scr), n, h, w, c, int(cfg.pixel_shuffle), st), "fsr_act_bwd")
            else:
                dz = g
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.x_is_image:
                wpk = packed_filter(cd, weight, L.PACK_DGRAD, dz.shape[3] * (4 if cfg.pixel_shuffle else 1))
                osc = _const_vec(cfg.in_scale, xin.device) if tuple(cfg.in_scale) != (1.0, 1.0, 1.0) else None
                dx, _, _ = conv3x3_raw(cd, dz, wpk, 3, mode=L.CONV_DGRAD, out_hw=(ih, iw), stride=cfg.stride, oscale=osc,
                                       out_f32=True, in_pixel_shuffled=cfg.pixel_shuffle, alg_k=cout)
                dx = dx.permute(0, 3, 1, 2)
            else:
                kpad = dz.shape[3] * (4 if cfg.pixel_shuffle else 1)
                wpk = FilterSpec(weight, L.PACK_DGRAD_PS if cfg.pixel_shuffle else L.PACK_DGRAD, kpad)
                mask = xin if cfg.input_act_bwd is not None else None
                bits = getattr(ctx, "in_signs", None) if (mask is not None and cin_pad % 64 == 0) else None
                if bits is not None:
                    mask = bits
                addend = None
                if skips:       # dL/dx = conv_dgrad(dz) + dL/d(skip): the first skip gradient rides in the launch's epilogue
                    addend = skips[0] if skips[0].is_contiguous() else skips[0].contiguous()
                dx, _, _ = conv3x3_raw(cd, dz, wpk, cin_pad, mode=L.CONV_DGRAD, out_hw=(ih, iw), stride=cfg.stride,
                                       in_pixel_shuffled=cfg.pixel_shuffle, alg_k=cout, dact_mask=addend if addend is not None else mask,
                                       dact_slope=cfg.input_act_bwd or 0.0, dact_add=addend is not None,
                                       dact_bits=(bits is not None and addend is None))
                for t in skips[1:]:
                    dx = add(cd, dx, t)
        dw = None
        if ctx.needs_input_grad[1]:
            arena = getattr(weight, "_fsr_grad", None)  # optim.ArenaAdamW: accumulate in place, hand autograd nothing

            def launch_wgrad():
                if ctx.c3:      # xin is the float image itself
                    out = arena if arena is not None else torch.zeros((cout, 3, 3, 3), dtype=torch.float32, device=xin.device)
                    need = lib.fsr_conv3x3_c3_wgrad_workspace(n, ih, iw, cout)
                    sn, sc, sh, sw = xin.stride()
                    L.check(lib.fsr_conv3x3_c3_wgrad(cd.code, _p(xin), sn, sc, sh, sw, n, ih, iw, *cfg.in_scale, *cfg.in_shift,
                                                     _p(dz), cout, _p(out), _p(bias_arena) if fused_dbias else None,
                                                     _p(_workspace(need, xin.device)), 0, _stream()),
                            "fsr_conv3x3_c3_wgrad")
                    return out
                return conv3x3_wgrad_raw(cd, xin, dz, cout, cin, cfg.stride, dy_pixel_shuffled=cfg.pixel_shuffle, out=arena)

            if arena is not None:   # nothing downstream reads the arena before the optimizer: own stream (see above)
                if ctx.c3 or not _wgrad_defer(cd, xin, dz, cout, cin, cfg, arena):
                    with _on_wgrad_stream(xin, dz):
                        launch_wgrad()
            else:
                dw = launch_wgrad()
            if arena is not None:
                dw = None
        db = dbias if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if not cfg.tanh_head and ctx.c3 and fused_dbias:
            db = None       # accumulated in the arena by the weight-gradient launch
        dp = dprelu if (dprelu is not None and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dp, None, None


def _head_backward_c3(ctx, g):
    """Backward of the head Conv2d(nf -> 3) + Tanh (model.py:102-110) on the first-layer kernels: dz = g (1 - y^2) is written
    as a 3-channel float image; the data gradient is the 3 -> nf convolution of that image with the transposed, tap-flipped
    filter (fsr_conv3x3_c3_fwd), the weight gradient the first-layer weight gradient with the roles of image and gradient
    swapped.  (The generic path pads dz to 32 channels: 10x the bytes for 3 channels of data.)"""
    cfg, cd = ctx.cfg, ctx.cfg.cd
    xin, weight, _prelu, saved = ctx.saved_tensors
    cout, cin, xshape = ctx.dims
    n, ih, iw, cin_pad = xshape
    lib, st = L.lib(), _stream()
    if g.dtype != torch.float32:
        g = g.float()
    if cd.x3:
        xin = _aligned(xin)
    dbias = _zeros((cout,), xin.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    dz_img = torch.empty((n, ih, iw, 3), dtype=torch.float32, device=xin.device)
    sn, sc, sh, sw = g.stride()
    L.check(lib.fsr_tanh_bwd_image(_p(g), sn, sc, sh, sw, _p(saved), n, ih, iw, _p(dz_img), _p(dbias),
                                   _p(_workspace(lib.fsr_tanh_bwd_scratch(), xin.device)), st), "fsr_tanh_bwd_image")
    img_strides = (ih * iw * 3, 1, iw * 3, 3)     # the NHWC float image viewed as (N,3,H,W)
    one, zero = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
    dx = None
    if ctx.needs_input_grad[0]:
        wpk = packed_filter(cd, weight, PACK_C3T, 32)
        dx = _empty((n, ih, iw, cin_pad), cd.torch_dtype, xin.device)
        prof = PROFILE_CONV
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        L.check(lib.fsr_conv3x3_c3_fwd(cd.code, _p(dz_img), *img_strides, n, ih, iw, *one, *zero, _p(wpk), None, L.ACT_NONE, 0.0, None,
                                       cin_pad, _p(dx), None, None, st), "fsr_conv3x3_c3_fwd")
        if prof is not None:
            ev1.record()
            prof.append((ev0, ev1, 2.0 * n * ih * iw * cin * 27, dz_img.numel() * 4 + dx.numel() * dx.element_size(), "conv_c3_fwd_kernel", "dgrad"))
    dw = None
    if ctx.needs_input_grad[1]:
        arena = getattr(weight, "_fsr_grad", None)

        def launch():
            out = arena if arena is not None else torch.zeros((cout, cin, 3, 3), dtype=torch.float32, device=xin.device)
            need = lib.fsr_conv3x3_c3_wgrad_workspace(n, ih, iw, cin_pad)
            L.check(lib.fsr_conv3x3_c3_wgrad(cd.code, _p(dz_img), *img_strides, n, ih, iw, *one, *zero, _p(xin), cin_pad, _p(out), None,
                                             _p(_workspace(need, xin.device)), 1, _stream()), "fsr_conv3x3_c3_wgrad")
            return out

        if arena is not None:
            with _on_wgrad_stream(xin, dz_img):
                launch()
        else:
            dw = launch()
    return dx, dw, dbias, None, None, None


Conv3x3Fn._backward_head_c3 = staticmethod(_head_backward_c3)


def conv3x3(x, weight, bias, prelu, cfg):
    return Conv3x3Fn.apply(x, weight, bias, prelu, cfg, torch.is_grad_enabled())


# ---------------------------------------------------------------------------------- autograd: InstanceNorm + act + residual
class InstNormActFn(torch.autograd.Function):
    """out = act(InstanceNorm(x)) + res, from the raw conv output and the statistics its epilogue made."""

    @staticmethod
    def forward(ctx, x, stats, res, prelu, cd, act, slope):
        _check_dev(x)
        n, h, w, c = x.shape
        if res is not None and not res.is_contiguous():
            res = res.contiguous()
        if cd.x3:
            x, res = _aligned(x), _aligned(res)
        out = _empty_like(x)
        L.check(L.lib().fsr_instnorm_act_fwd(cd.code, _p(x), _p(stats), _p(res), act, float(slope), _p(prelu), _p(out), n,
                                             h * w, c, _stream()), "fsr_instnorm_act_fwd")
        ctx.meta = (cd, act, slope, res is not None)
        ctx.save_for_backward(x, stats, prelu)
        return out

    @staticmethod
    def backward(ctx, g):
        cd, act, slope, has_res = ctx.meta
        x, stats, prelu = ctx.saved_tensors
        n, h, w, c = x.shape
        g = g if g.is_contiguous() else g.contiguous()
        if cd.x3:
            g = _aligned(g)
        lib, st = L.lib(), _stream()
        sums = _zeros((n, c, 2), x.device)
        dprelu = _zeros((1,), x.device) if act == L.ACT_PRELU else None
        scr = _workspace(lib.fsr_instnorm_act_bwd_scratch(n, h * w, c), x.device)
        L.check(lib.fsr_instnorm_act_bwd_reduce(cd.code, _p(g), _p(x), _p(stats), act, float(slope), _p(prelu), _p(sums),
                                                _p(dprelu), _p(scr), n, h * w, c, st), "fsr_instnorm_act_bwd_reduce")
        dx = _empty_like(x)
        L.check(lib.fsr_instnorm_act_bwd_apply(cd.code, _p(g), _p(x), _p(stats), _p(sums), act, float(slope), _p(prelu),
                                               _p(dx), n, h * w, c, st), "fsr_instnorm_act_bwd_apply")
        return dx, None, (g if has_res else None), dprelu, None, None, None


def instnorm_act(x, stats, res, prelu, cd, act=L.ACT_NONE, slope=0.0):
    return InstNormActFn.apply(x, stats, res, prelu, cd, act, slope)


# ---------------------------------------------------------------------------------- autograd: MaxPool2d(2,2)
class MaxPool2Fn(torch.autograd.Function):
    """relu_mask: backward also applies the backward of the ReLU that produced x (see ConvCfg.act_bwd_by_consumer).
    With gradients enabled the forward also writes one arg-max byte per pooled element (position of the maximum + its sign);
    the backward pass reads the pooled gradient and those bytes -- not the full-resolution input and the pooled output."""

    @staticmethod
    def forward(ctx, x, cd, relu_mask=False, grad_on=True):
        _check_dev(x)
        n, h, w, c = x.shape
        if cd.x3:
            x = _aligned(x)
        y = _empty((n, h // 2, w // 2, c), x.dtype, x.device)
        idx = torch.empty((n, h // 2, w // 2, c), dtype=torch.uint8, device=x.device) if (grad_on and USE_POOL_ARGMAX) else None
        L.check(L.lib().fsr_maxpool2_fwd(cd.code, _p(x), _p(y), _p(idx), n, h, w, c, _stream()), "fsr_maxpool2_fwd")
        ctx.cd, ctx.relu_mask, ctx.xshape = cd, relu_mask, tuple(x.shape)
        ctx.by_idx = idx is not None
        if idx is not None:
            ctx.save_for_backward(idx)
        else:
            ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.xshape
        g = g if g.is_contiguous() else g.contiguous()
        if ctx.cd.x3:
            g = _aligned(g)
        dx = _empty((n, h, w, c), g.dtype, g.device)
        if ctx.by_idx:
            (idx,) = ctx.saved_tensors
            L.check(L.lib().fsr_maxpool2_bwd_argmax(ctx.cd.code, _p(g), _p(idx), _p(dx), n, h, w, c, int(ctx.relu_mask), _stream()),
                    "fsr_maxpool2_bwd_argmax")
        else:
            x, y = ctx.saved_tensors
            L.check(L.lib().fsr_maxpool2_bwd(ctx.cd.code, _p(g), _p(x), _p(y), _p(dx), n, h, w, c, int(ctx.relu_mask), _stream()),
                    "fsr_maxpool2_bwd")
        return dx, None, None, None


def maxpool2(x, cd, relu_mask=False):
    return MaxPool2Fn.apply(x, cd, relu_mask, torch.is_grad_enabled())


# ---------------------------------------------------------------------------------- autograd: Conv2d(C -> 1, k=1)
class Conv1x1ToLogitsFn(torch.autograd.Function):
    """x (N,H,W,C) -> float32 logits (N,1,H,W) (model.py:184-186)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cd):
        _check_dev(x)
        n, h, w, c = x.shape
        x = x if x.is_contiguous() else x.contiguous()
        if cd.x3:
            x = _aligned(x)
        wv = weight.detach().reshape(-1)
        logits = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        L.check(L.lib().fsr_conv1x1_c1_fwd(cd.code, _p(x), _p(wv), _p(bias), _p(logits), n * h * w, c, _stream()),
                "fsr_conv1x1_c1_fwd")
        ctx.cd = cd
        ctx.save_for_backward(x, weight)
        return logits

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        n, h, w, c = x.shape
        g = g.contiguous().float()
        dx = _empty_like(x) if ctx.needs_input_grad[0] else None
        dw = _zeros(tuple(weight.shape), x.device)
        db = _zeros((1,), x.device)
        scr = _workspace(L.lib().fsr_conv1x1_c1_bwd_scratch(c), x.device)
        L.check(L.lib().fsr_conv1x1_c1_bwd(ctx.cd.code, _p(g), _p(x), _p(weight.detach().reshape(-1)), _p(dx), _p(dw), _p(db),
                                           _p(scr), n * h * w, c, _stream()), "fsr_conv1x1_c1_bwd")
        return dx, dw, db, None


def conv1x1_to_logits(x, weight, bias, cd):
    return Conv1x1ToLogitsFn.apply(x, weight, bias, cd)


# ---------------------------------------------------------------------------------- autograd: losses
class BCEWithLogitsFn(torch.autograd.Function):
    """torch.nn.BCEWithLogitsLoss() (mean) on float32 logits/targets (trainer.py:41)."""

    @staticmethod
    def forward(ctx, x, t):
        _check_dev(x, t)
        x = x.contiguous().float()
        t = t.contiguous().float()
        loss = _zeros((1,), x.device).view(())
        scr = _workspace(L.lib().fsr_loss_scratch(), x.device)
        L.check(L.lib().fsr_bce_logits_fwd(_p(x), _p(t), _p(loss), _p(scr), x.numel(), _stream()), "fsr_bce_logits_fwd")
        ctx.save_for_backward(x, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        L.check(L.lib().fsr_bce_logits_bwd(_p(x), _p(t), _p(g), _p(dx), x.numel(), _stream()), "fsr_bce_logits_bwd")
        return dx, None


def _is_dense(t):
    """True when t's elements tile its storage without gaps or overlap (e.g. a permuted contiguous tensor)."""
    pairs = sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1)
    expect = 1
    for st, sz in pairs:
        if st != expect:
            return False
        expect *= sz
    return True


class SmoothL1Fn(torch.autograd.Function):
    """torch.nn.SmoothL1Loss() (beta 1, mean) (trainer.py:43); the second argument is the target."""

    @staticmethod
    def forward(ctx, a, b, x3=False):
        _check_dev(a, b)
        if a.dtype != b.dtype or a.shape != b.shape:
            raise L.FsrError("SmoothL1: operands must share dtype and shape")
        if not (a.stride() == b.stride() and _is_dense(a)):  # elementwise: any shared dense layout will do
            a, b = a.contiguous(), b.contiguous()
        code = {torch.bfloat16: L.FSR_BF16, torch.float16: L.FSR_F16, torch.float32: L.FSR_F32}.get(a.dtype)
        if code is None:
            raise L.FsrError("SmoothL1: unsupported dtype %s" % a.dtype)
        if x3:      # x3 containers are float32 tensors: the caller says what they hold
            if a.dtype != torch.float32:
                raise L.FsrError("SmoothL1: x3 operands live in float32 containers")
            code = L.FSR_X3
            a, b = _aligned(a), _aligned(b)
        loss = _zeros((1,), a.device).view(())
        scr = _workspace(L.lib().fsr_loss_scratch(), a.device)
        L.check(L.lib().fsr_smooth_l1_fwd(code, _p(a), _p(b), _p(loss), _p(scr), a.numel(), _stream()), "fsr_smooth_l1_fwd")
        ctx.code = code
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous().float()
        da = _empty_like(a) if ctx.code == L.FSR_X3 else torch.empty_like(a)
        L.check(L.lib().fsr_smooth_l1_bwd(ctx.code, _p(a), _p(b), _p(g), _p(da), a.numel(), _stream()), "fsr_smooth_l1_bwd")
        return da, None, None


def bce_with_logits(x, t):
    return BCEWithLogitsFn.apply(x, t)


def smooth_l1(a, b, cd=None):
    """cd: the compute mode of activation operands (needed for x3, whose containers are float32 tensors); images: None."""
    return SmoothL1Fn.apply(a, b, bool(cd is not None and cd.x3))


# ---------------------------------------------------------------------------------- validation metrics
def ssim_sse(a, b):
    """Per-image (sum of the SSIM map, sum of squared errors) of two (N,3,H,W) float32 batches in [-1,1], any strides
    (fsr_ssim_sse; torchmetrics' SSIM / PSNR defaults on (1+x)/2, trainer.py:46-69).  Returns a float32 (N,2) tensor."""
    _check_dev(a, b)
    if a.shape != b.shape or a.dim() != 4 or a.shape[1] != 3:
        raise ValueError("ssim_sse expects two (N,3,H,W) batches of one shape, got %s and %s" % (tuple(a.shape), tuple(b.shape)))
    a = a if a.dtype == torch.float32 else a.float()
    b = b if b.dtype == torch.float32 else b.float()
    n, _, h, w = a.shape
    out = torch.empty((n, 2), dtype=torch.float32, device=a.device)
    scr = _workspace(L.lib().fsr_ssim_sse_scratch(n, h, w), a.device)
    L.check(L.lib().fsr_ssim_sse(_p(a), *a.stride(), _p(b), *b.stride(), n, h, w, _p(out), _p(scr), _stream()), "fsr_ssim_sse")
    return out


def u8_to_image(frames):
    """(N,H,W,3) uint8 frames -> float32 (N,3,H,W) VIEW in [-1,1] (x / 127.5 - 1, inference.py:48) of an NHWC buffer: the
    first-layer kernels read it in place through its strides."""
    _check_dev(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or not frames.is_contiguous():
        raise ValueError("u8_to_image expects a contiguous (N,H,W,3) uint8 tensor")
    img = torch.empty(frames.shape, dtype=torch.float32, device=frames.device)
    L.check(L.lib().fsr_u8_to_image(_p(frames), _p(img), frames.numel(), _stream()), "fsr_u8_to_image")
    return img.permute(0, 3, 1, 2)
