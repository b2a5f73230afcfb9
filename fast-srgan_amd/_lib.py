"""ctypes binding of libfsr_hip.so (the C ABI declared in include/fsr_hip.h).

The product path has exactly one backend: the hipcc-built gfx950 library next to this file.
If it is missing, `lib()` raises -- there is no CPU or PyTorch fallback.  The only other
library that can ever be installed here is the thread-per-lane emulation build used by the CPU
test-suite (tests/emu), and only through the explicit `_install_for_testing` hook.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSR_HIP_LIB: another build of the same sources (kernel A/B experiments, tools/ab.py); the default is the in-tree library
LIB_PATH = os.environ.get("FSR_HIP_LIB") or os.path.join(_HERE, "libfsr_hip.so")

FSR_F32, FSR_BF16, FSR_F16, FSR_X3 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_PRELU, ACT_TANH = 0, 1, 2, 3, 4
CONV_FWD, CONV_DGRAD = 0, 1
PACK_FWD, PACK_FWD_PS, PACK_DGRAD, PACK_DGRAD_PS = 0, 1, 2, 3
OUT_DTYPE, OUT_F32, OUT_U8 = 0, 1, 2
OPT_BIAS, OPT_PRELU, OPT_OSCALE, OPT_MASK, OPT_PREACT, OPT_STATS = 1, 2, 4, 8, 16, 32   # fsr_conv3x3_pack_block
ABI_VERSION = 10

c_int, c_float, c_void_p, c_size_t, c_ll = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_longlong


class ConvDesc(ctypes.Structure):
    """struct fsr_conv_desc (include/fsr_hip.h)."""
    _fields_ = [
        ("dtype", c_int), ("mode", c_int),
        ("n", c_int), ("ih", c_int), ("iw", c_int), ("cin", c_int),
        ("oh", c_int), ("ow", c_int), ("cout", c_int),
        ("stride", c_int), ("act", c_int), ("slope", c_float),
        ("pixel_shuffle", c_int), ("in_pixel_shuffled", c_int), ("out_f32", c_int), ("pool2", c_int), ("mask_is_addend", c_int), ("pack_lin", c_int),
    ]


class WgradDesc(ctypes.Structure):
    """struct fsr_wgrad_desc (include/fsr_hip.h)."""
    _fields_ = [
        ("dtype", c_int),
        ("n", c_int), ("ih", c_int), ("iw", c_int), ("cin_pad", c_int), ("cin", c_int),
        ("oh", c_int), ("ow", c_int), ("cout_pad", c_int), ("cout", c_int),
        ("stride", c_int), ("dy_pixel_shuffled", c_int),
    ]


P = c_void_p
# name -> (restype, argtypes); every symbol include/fsr_hip.h declares must appear here
SIGNATURES = {
    "fsr_version": (c_int, []),
    "fsr_last_error": (ctypes.c_char_p, []),
    "fsr_last_kernel": (ctypes.c_char_p, []),
    "fsr_device_info": (c_int, [ctypes.c_char_p, c_size_t]),
    "fsr_pack_conv3x3": (c_int, [c_int, c_int, P, c_int, c_int, c_int, P, P]),
    "fsr_pack_conv3x3_lin": (c_int, [c_int, c_int, P, c_int, c_int, c_int, P, P]),
    "fsr_conv3x3_pack_block": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "fsr_conv3x3_scratch": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "fsr_conv3x3": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, c_float, P, P, P, P, P]),
    "fsr_conv3x3_wgrad_workspace": (c_size_t, [ctypes.POINTER(WgradDesc)]),
    "fsr_conv3x3_wgrad": (c_int, [ctypes.POINTER(WgradDesc), P, P, P, P, P]),
    "fsr_conv3x3_wgrad_grouped_workspace": (c_size_t, [ctypes.POINTER(WgradDesc), c_int]),
    "fsr_conv3x3_wgrad_grouped": (c_int, [ctypes.POINTER(WgradDesc), c_int, P, P, P, P, P]),
    "fsr_instnorm_act_fwd": (c_int, [c_int, P, P, P, c_int, c_float, P, P, c_int, c_int, c_int, P]),
    "fsr_instnorm_act_bwd_scratch": (c_size_t, [c_int, c_int, c_int]),
    "fsr_instnorm_act_bwd_reduce": (c_int, [c_int, P, P, P, c_int, c_float, P, P, P, P, c_int, c_int, c_int, P]),
    "fsr_instnorm_act_bwd_apply": (c_int, [c_int, P, P, P, P, c_int, c_float, P, P, c_int, c_int, c_int, P]),
    "fsr_act_bwd_scratch": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fsr_act_bwd": (c_int, [c_int, P, P, c_int, c_float, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "fsr_add": (c_int, [c_int, P, P, P, c_ll, P]),
    "fsr_image_to_nhwc": (c_int, [c_int, P, c_ll, c_ll, c_ll, c_ll, c_int, c_int, c_int, c_float, c_float, c_float,
                                  c_float, c_float, c_float, P, c_int, P]),
    "fsr_u8_to_image": (c_int, [P, P, c_ll, P]),
    "fsr_pack_conv3x3_c3": (c_int, [c_int, P, c_int, P, c_int, P]),
    "fsr_tanh_bwd_image": (c_int, [P, c_ll, c_ll, c_ll, c_ll, P, c_int, c_int, c_int, P, P, P, P]),
    "fsr_conv3x3_c3_fwd": (c_int, [c_int, P, c_ll, c_ll, c_ll, c_ll, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                   c_float, c_float, P, P, c_int, c_float, P, c_int, P, P, P, P]),
    "fsr_conv3x3_c3_wgrad_workspace": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "fsr_conv3x3_c3_wgrad": (c_int, [c_int, P, c_ll, c_ll, c_ll, c_ll, c_int, c_int, c_int, c_float, c_float, c_float,
                                     c_float, c_float, c_float, P, c_int, P, P, P, c_int, P]),
    "fsr_tanh_bwd_scratch": (c_size_t, []),
    "fsr_tanh_bwd_to_nhwc": (c_int, [c_int, P, c_ll, c_ll, c_ll, c_ll, P, c_int, c_int, c_int, P, c_int, P, P, P]),
    "fsr_maxpool2_fwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    "fsr_png_to_npy": (c_int, [P, P, c_int, c_int, P]),
    "fsr_png_decode_chw": (c_int, [ctypes.c_char_p, P, c_size_t, P, P]),
    "fsr_maxpool2_bwd_argmax": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "fsr_maxpool2_bwd": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "fsr_conv1x1_c1_fwd": (c_int, [c_int, P, P, P, P, c_int, c_int, P]),
    "fsr_conv1x1_c1_bwd_scratch": (c_size_t, [c_int]),
    "fsr_conv1x1_c1_bwd": (c_int, [c_int, P, P, P, P, P, P, P, c_int, c_int, P]),
    "fsr_loss_scratch": (c_size_t, []),
    "fsr_bce_logits_fwd": (c_int, [P, P, P, P, c_ll, P]),
    "fsr_bce_logits_bwd": (c_int, [P, P, P, P, c_ll, P]),
    "fsr_smooth_l1_fwd": (c_int, [c_int, P, P, P, P, c_ll, P]),
    "fsr_smooth_l1_bwd": (c_int, [c_int, P, P, P, P, c_ll, P]),
    "fsr_ssim_sse_scratch": (c_size_t, [c_int, c_int, c_int]),
    "fsr_ssim_sse": (c_int, [P, c_ll, c_ll, c_ll, c_ll, P, c_ll, c_ll, c_ll, c_ll, c_int, c_int, c_int, P, P, P]),
    "fsr_adamw_step": (c_int, [P, P, P, P, c_ll, c_float, c_float, c_float, c_float, c_float, P, c_float, P]),
    "fsr_grad_nonfinite": (c_int, [P, c_ll, P, P]),
    "fsr_adamw_step_scaled": (c_int, [P, P, P, P, c_ll, c_float, c_float, c_float, c_float, c_float, P, c_float, P, P]),
    "fsr_loss_scale_update": (c_int, [P, c_float, c_float, c_float, P]),
    "fsr_crop_resize": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, P, P, c_int, P, P, P, P]),
}

_lib = None
_is_emulation = False


class FsrError(RuntimeError):
    pass


def _bind(path):
    handle = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return handle


def lib():
    """The loaded kernel library.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FsrError(
                "libfsr_hip.so is missing (%s). Build it with `python fast-srgan_amd/build.py` "
                "(hipcc, gfx950). There is no fallback path." % LIB_PATH)
        _lib = _bind(LIB_PATH)
    return _lib


def is_emulation():
    return _is_emulation


def _install_for_testing(path):
    """TEST HOOK: route calls to the host-emulation build of the same sources (tests/emu)."""
    global _lib, _is_emulation
    if path is None:
        _lib, _is_emulation = None, False
    else:
        _lib, _is_emulation = _bind(path), True


def check(rc, what=""):
    if rc != 0:
        msg = lib().fsr_last_error()
        raise FsrError("%s failed (%d): %s" % (what or "fsr call", rc, msg.decode() if msg else "?"))
