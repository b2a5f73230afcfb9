"""HBM traffic of the 3x3 conv forward / data-gradient kernels per iteration from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; counter_collection.csv) over `bench.py --steps 1 --warmup 1 --no-graph --no-f32` (= 4 iterations).
FETCH_SIZE on gfx950 counts half the bytes (MI355X_MICROARCH.md): x2.  Units: KB.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv ITERS LAUNCHES_PER_ITER [OUT.json]

With OUT.json (profiles/conv_traffic.json) the per-launch figure is recorded together with the hash of the kernel
sources it was measured on; bench.py quotes it as roofline.traffic only while that hash still matches."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNELS = ("conv_igemm_kernel", "conv_tall3_kernel", "conv_s2d3_kernel", "conv64_", "conv_c3_fwd_kernel")      # every forward / data-gradient convolution kernel (conv64_*: all persistent forms)


def total_kb(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS):
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


_cxa = None


def demangle(name):
    """rocprofv3 leaves a name mangled when its demangler does not know a type -- `DF16_` (_Float16), i.e. every fp16 kernel.
    Demangle with libstdc++ after spelling _Float16 as `unsigned short` (the bf16 storage type), then put the type back."""
    global _cxa
    if not name.startswith("_Z"):
        return name
    import ctypes
    if _cxa is None:
        _cxa = ctypes.CDLL("libstdc++.so.6").__cxa_demangle
        _cxa.restype = ctypes.c_void_p
    f16 = "DF16_" in name
    st = ctypes.c_int()
    r = _cxa(name.replace("DF16_", "t").encode(), None, None, ctypes.byref(st))
    if not r or st.value != 0:
        return name
    out = ctypes.string_at(r).decode()
    return out.replace("unsigned short", "_Float16") if f16 else out


def norm(name):
    """rocprofv3's demangled kernel name -> the spelling bench.py / fsr_last_kernel use."""
    name = demangle(name)
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    n = n.replace("unsigned short", "bf16").replace("_Float16", "f16").replace("float", "f32").replace(" ", "")
    if n.startswith("conv_tall3_kernel<"):     # its trailing STATS / stride / x3 flags: the library's note prints "<x3|bf16,...>" / "<...,stats>" / "<...,s2>" / "<...,stats,s2>"
        psm = ""
        if n.count(",") >= 10:                                  # the trailing PSM value (round 6): ...,STATS,S,X3,PSM>
            head, last = n[:-1].rsplit(",", 1)
            psm = {"0": "", "1": ",ps_in", "2": ",up"}.get(last, "," + last)
            n = head + ">"
        if n.endswith(",true>") and n.count(",") >= 9:          # the trailing X3 flag (round 5): ...,STATS,S,X3>
            n = n[:-len(",true>")].replace("<bf16,", "<x3,", 1) + ">"
        elif n.endswith(",false>") and n.count(",") >= 9:
            n = n[:-len(",false>")] + ">"
        n = n.replace(",false,1>", ">").replace(",true,1>", ",stats>").replace(",false,2>", ",s2>").replace(",true,2>", ",stats,s2>")
        n = n.replace(",false>", ">").replace(",true>", ",stats>")
        n = n[:-1] + psm + ">"
    if n.startswith("conv_s2d3_kernel<") or n.startswith("conv_igemm_kernel<"):
        if n.startswith("conv_s2d3_kernel<bf16,true>"):
            n = "conv_s2d3_kernel<x3>"
        elif n.startswith("conv_s2d3_kernel<"):
            n = n.replace(",false>", ">")
        if n.startswith("conv_igemm_kernel<") and n.endswith(",1>") and n.count(",") == 9:      # trailing X3 = 1
            n = n[:-len(",1>")].replace("<bf16,", "<x3,", 1) + ">"
        elif n.startswith("conv_igemm_kernel<") and n.endswith(",0>") and n.count(",") == 9:
            n = n[:-len(",0>")] + ">"
    return n


def per_kernel_kb(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS):
            e = out.setdefault(norm(r["Kernel_Name"]), [0.0, 0])
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return out


def main():
    fetch, n1 = total_kb(sys.argv[1], "FETCH_SIZE")
    write, n2 = total_kb(sys.argv[2], "WRITE_SIZE")
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    api = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    byt = (2 * fetch + write) * 1024 / iters
    print("conv kernel dispatches per iteration: %.1f / %.1f" % (n1 / iters, n2 / iters))
    if api and (abs(n1 / iters - api) > 0.01 or abs(n2 / iters - api) > 0.01):
        sys.exit("the kernel-name filter matched %.1f / %.1f dispatches per iteration, the bench counted %d launches: update KERNELS" % (n1 / iters, n2 / iters, api))
    print("FETCH_SIZE %.0f KB (x2) + WRITE_SIZE %.0f KB per iteration -> %.3e bytes per iteration" % (fetch / iters, write / iters, byt))
    if api:
        print("per API-level conv launch (%d per iteration): %.3e bytes" % (api, byt / api))
    if len(sys.argv) > 5 and api:
        import bench
        rec = {"bytes_per_launch": round(byt / api), "bytes_per_iteration": round(byt), "launches_per_step": api, "iterations": iters,
               "fetch_size_kb_per_iteration": round(fetch / iters), "write_size_kb_per_iteration": round(write / iters),
               "formula": "2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request) + WRITE_SIZE, conv forward + data-gradient kernels",
               "dtype": os.environ.get("PMC_DTYPE", "bf16"), "batch": 32, "kernel_sources_sha16": bench.kernel_sources_hash(),
               "files": "%s, %s" % (os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2]))}
        fk, wk = per_kernel_kb(sys.argv[1], "FETCH_SIZE"), per_kernel_kb(sys.argv[2], "WRITE_SIZE")
        rec["per_kernel"] = {k: {"dispatches_per_iteration": fk[k][1] / iters,
                                 "bytes_per_dispatch": round((2 * fk[k][0] / fk[k][1] + wk[k][0] / wk[k][1]) * 1024)}
                             for k in sorted(fk) if k in wk and fk[k][1] and wk[k][1]}
        json.dump(rec, open(sys.argv[5], "w"), indent=1)
        print("wrote", sys.argv[5], rec)


if __name__ == "__main__":
    main()
