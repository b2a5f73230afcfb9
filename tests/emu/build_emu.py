"""TEST INFRASTRUCTURE: compiles the kernel sources of fast-srgan_amd/csrc with the HOST clang++
against tests/emu/include/hip/hip_runtime.h (a thread-per-lane emulation of the HIP device
surface) into tests/emu/_build/libfsr_emu.so.  Used only by `pytest -m "not gpu"` to debug index
arithmetic without a GPU; the package never loads this library."""
import concurrent.futures
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "fast-srgan_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libfsr_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emu(force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    runtime_src, runtime_obj = os.path.join(HERE, "emu_runtime.cpp"), os.path.join(OUT, "emu_runtime.o")
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(ROOT, "include", "fsr_hip.h"), os.path.join(HERE, "include", "hip", "hip_runtime.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    flags = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-pthread", "-Wno-unused-value",
             "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    todo = []
    if force or not os.path.exists(runtime_obj) or os.path.getmtime(runtime_obj) < max(os.path.getmtime(runtime_src), hdr_t):
        todo.append((runtime_src, runtime_obj))
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT, os.path.splitext(s)[0] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([CLANG] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu compile failed for %s:\n%s" % (src, r.stderr[-8000:]))

    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(cc, todo))
    objs = [os.path.join(OUT, os.path.splitext(s)[0] + ".o") for s in srcs] + [runtime_obj]
    if todo or not os.path.exists(LIB):
        r = subprocess.run([CLANG, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-lz"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu link failed:\n" + r.stderr[-8000:])
    return LIB


if __name__ == "__main__":
    print(build_emu())
