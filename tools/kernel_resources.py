"""Registers / scratch / occupancy per kernel, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/kernel_resources.py                  every kernel of the last build (fast-srgan_amd/_obj/*.res, written by build.py)
    python tools/kernel_resources.py FILE.hip         compile that file now and print its kernels

`resources()` returns {normalised kernel name: {"vgpr", "agpr", "sgpr", "scratch", "occupancy", "lds", "file"}}; names are spelled
the way bench.py / fsr_last_kernel spell them (tools/pmc_traffic.norm).  tests/test_tools.py asserts the hot kernels spill nothing."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OBJ = os.path.join(ROOT, "fast-srgan_amd", "_obj")
_KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize": "scratch", "Occupancy": "occupancy", "LDS Size": "lds",
         "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill"}


def parse(text, origin=""):
    from pmc_traffic import norm
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"file": origin, "mangled": m.group(1)}
            out[norm(m.group(1))] = cur
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None and m.group(1).strip() in _KEYS:
            cur[_KEYS[m.group(1).strip()]] = int(m.group(2))
    return out


def resources():
    """Every kernel of the last in-tree build."""
    out = {}
    for f in sorted(glob.glob(os.path.join(OBJ, "*.res"))):
        out.update(parse(open(f).read(), os.path.basename(f)[:-4]))
    return out


def compile_now(src):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-I", ROOT + "/include",
           "-I", ROOT + "/fast-srgan_amd/csrc", "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
    return parse(subprocess.run(cmd, capture_output=True, text=True).stderr, os.path.basename(src))


if __name__ == "__main__":
    rows = compile_now(sys.argv[1]) if len(sys.argv) > 1 else resources()
    for name, r in sorted(rows.items(), key=lambda kv: (kv[1]["file"], kv[0])):
        print("%-22s %-64s vgpr=%3d agpr=%3d sgpr=%3d scratch=%4d sgpr_spill=%4d occ=%d" % (
            r["file"], name[:64], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("sgpr_spill", -1),
            r.get("occupancy", -1)))
