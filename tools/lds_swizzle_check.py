"""Bank-conflict model of gfx950's ds_read_b128 and a search for conflict-free LDS layouts.

Model (MI355X_MICROARCH guide, confirmed on conv_tall3 by SQ_LDS_BANK_CONFLICT = 0 and 4.00 LDS cycles per read,
profiles/r03_pmc_tall3.txt): 64 banks of 4 bytes; a ds_read_b128 (16 bytes = 4 consecutive banks per lane) is served in
FOUR passes of 16 lanes -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same two sets + 32 -- and a pass is
conflict-free when its 16 lanes touch 64 different banks (lanes reading the SAME 16 bytes are a broadcast, not a conflict).

Checked here:
  * conv_tall3's two fragment reads (csrc/conv_tall3.hip: filter rows of 64 bytes with unit u at u ^ ((R >> 2) & 3); halo
    pixels of 64 bytes, 18 per row, unit u at u ^ ((x >> 1) & 3)) for every tap column and both K halves -- the self-test;
  * the layout a stride-2 / 16-channel-chunk form would need (32-byte pixels and filter rows: DESIGN.md section 8): which
    swizzles f make  address = index * 32 + ((half ^ f(index)) << 4)  conflict-free for the filter fragment (32 rows x 2
    halves) and for the pixel fragment (2 rows x 16 columns x 2 halves, any row pitch).

python tools/lds_swizzle_check.py
"""
import itertools

PASSES = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
PASSES = PASSES + [[l + 32 for l in p] for p in PASSES]


def conflicts(addr_of_lane):
    """Worst number of distinct 16-byte addresses mapped onto one bank in any pass (1 = conflict-free)."""
    worst = 1
    for lanes in PASSES:
        per_bank = {}
        for l in lanes:
            a = addr_of_lane(l)
            assert a % 16 == 0
            for b in range(4):
                per_bank.setdefault(((a >> 2) + b) & 63, set()).add(a)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


# ---------------------------------------------------------------------------------------- conv_tall3 (self-test)
def tall3_filter(j, wco=0):
    def addr(lane):
        l31, hi = lane & 31, lane >> 5
        R = wco * 64 + l31
        return R * 64 + (((2 * j + hi) ^ ((R >> 2) & 3)) << 4)
    return addr


def tall3_pixels(kx, j, mb=4, wpx=0, m=0, ky=0):
    def addr(lane):
        l15, lrow, hi = lane & 15, (lane >> 4) & 1, lane >> 5
        x = l15 + kx
        return ((wpx * 2 * mb + lrow + 2 * m + ky) * 18 + x) * 64 + (((2 * j + hi) ^ ((x >> 1) & 3)) << 4)
    return addr


def self_test():
    for j in (0, 1):
        for wco in (0, 1):
            assert conflicts(tall3_filter(j, wco)) == 1, ("filter", j, wco)
        for kx in (0, 1, 2):
            for ky in (0, 1, 2):
                assert conflicts(tall3_pixels(kx, j, ky=ky)) == 1, ("pixels", kx, ky, j)
    # and the model does see a conflict where there is one: the same layouts without their swizzle
    assert conflicts(lambda lane: (lane & 31) * 64 + ((lane >> 5) << 4)) > 1
    return True


# ---------------------------------------------------------------------------------------- 32-byte rows / pixels
def bit_functions(nbits=6):
    """f(index) = XOR of a subset of the index's low bits (one output bit: the 16-byte half)."""
    for mask in range(1 << nbits):
        yield mask, (lambda idx, mask=mask: bin(idx & mask).count("1") & 1)


def search_32byte():
    out = {"filter": [], "pixels": {}}
    for mask, f in bit_functions():
        def filt(lane, f=f):
            r, hi = lane & 31, lane >> 5
            return r * 32 + ((hi ^ f(r)) << 4)
        if conflicts(filt) == 1:
            out["filter"].append(mask)
    for pitch in (16, 17, 18, 20):
        good = []
        for mask, f in bit_functions():
            ok = True
            for kx, row0 in itertools.product((0, 1), (0, 1, 2, 3)):
                def pix(lane, f=f, kx=kx, row0=row0):
                    l15, lrow, hi = lane & 15, (lane >> 4) & 1, lane >> 5
                    p = (row0 + lrow) * pitch + l15 + kx
                    return p * 32 + ((hi ^ f(p)) << 4)
                if conflicts(pix) != 1:
                    ok = False
                    break
            if ok:
                good.append(mask)
        out["pixels"][pitch] = good
    # the same with the swizzle taken from the COLUMN and the ROW separately (conv_tall3's halo swizzle is a function of the column)
    out["pixels_xy"] = {}
    for pitch in (17, 18, 20):
        good = []
        for mx, mr in itertools.product(range(32), range(4)):
            ok = True
            for kx, row0 in itertools.product((0, 1), (0, 1, 2, 3)):
                def pix(lane, kx=kx, row0=row0):
                    l15, lrow, hi = lane & 15, (lane >> 4) & 1, lane >> 5
                    x, r = l15 + kx, row0 + lrow
                    f = (bin(x & mx).count("1") + bin(r & mr).count("1")) & 1
                    return (r * pitch + x) * 32 + ((hi ^ f) << 4)
                if conflicts(pix) != 1:
                    ok = False
                    break
            if ok:
                good.append((mx, mr))
        out["pixels_xy"][pitch] = good
    return out


def main():
    self_test()
    print("conv_tall3's filter and halo fragment reads: conflict-free under the model (as the PMC counters say)")
    res = search_32byte()
    print("32-byte filter rows, half ^ parity(row & mask): conflict-free masks %s" % (["0x%x" % m for m in res["filter"]] or "none"))
    for pitch, good in res["pixels"].items():
        print("32-byte pixels, row pitch %2d, half ^ parity(pixel index & mask): conflict-free masks %s" % (
            pitch, ["0x%x" % m for m in good] or "none"))
    for pitch, good in res["pixels_xy"].items():
        print("32-byte pixels, row pitch %2d, half ^ parity(x & mx) ^ parity(row & mr): conflict-free (mx, mr) %s" % (
            pitch, ["(0x%x, %d)" % g for g in good] or "none"))


if __name__ == "__main__":
    main()
