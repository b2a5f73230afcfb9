"""Design tools that the kernels' layouts were derived with."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lds_bank_model_accepts_conv_tall3_layouts_and_rejects_unswizzled_ones():
    """tools/lds_swizzle_check.py: the ds_read_b128 pass / bank model finds conv_tall3.hip's swizzled filter rows and halo pixels
    conflict-free for every tap column, tap row and K half (what SQ_LDS_BANK_CONFLICT = 0 says on the GPU) and the same
    layouts without the swizzle conflicting; 32-byte pixels have conflict-free swizzles at the halo pitches a 16-channel-chunk
    form would use."""
    m = _load("lds_swizzle_check")
    assert m.self_test()
    assert m.conflicts(m.tall3_pixels(1, 0)) == 1
    assert m.conflicts(lambda lane: ((lane >> 4 & 1) * 18 + (lane & 15)) * 64 + ((lane >> 5) << 4)) > 1
    res = m.search_32byte()
    assert 0x8 in res["filter"] and (0, 1) in res["pixels_xy"][18] and (2, 0) in res["pixels_xy"][18]
