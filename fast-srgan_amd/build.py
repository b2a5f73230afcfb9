"""Builds libfsr_hip.so (gfx950 only) from csrc/*.hip with hipcc, in-tree.

`python fast-srgan_amd/build.py` or `build_hip()` from __graft_entry__.build().  Objects are
cached by mtime; the shared library is written next to this file so that it travels with the
repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libfsr_hip.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libfsr_hip.so cannot be built")


def _newest_header():
    t = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build_hip(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))     # .cpp: host-only code (ingest.cpp), same compiler driver
    hdr_t = _newest_header()
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    todo = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if (force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)
                or not os.path.exists(os.path.splitext(obj)[0] + ".res")):
            todo.append((src, obj))
    hipcc = _hipcc()

    def cc(job):
        src, obj = job
        # -Rpass-analysis=kernel-resource-usage: the registers / scratch / occupancy of every kernel, kept next to the object
        # (<name>.res) -- tools/kernel_resources.py prints them, tests/test_tools.py asserts the hot kernels spill nothing
        cmd = [hipcc] + (flags if src.endswith(".hip") else ["-x", "hip"] + flags) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-8000:]))
        with open(os.path.splitext(obj)[0] + ".res", "w") as f:
            f.write(r.stderr)
        return src

    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            for done in ex.map(cc, todo):
                if verbose:
                    print("[fsr build] compiled", os.path.basename(done), flush=True)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in srcs]
    if todo or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]      # zlib: csrc/ingest.cpp (PNG inflate)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
        if verbose:
            print("[fsr build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
