"""Summarises the rocprofv3 PMC passes of tools/ubench/pmc3.sh (conv_tall3 on two layers) into profiles/r03_pmc_tall3.txt:
raw counters per dispatch plus the derived figures DESIGN.md 3.1e quotes (shader clock from GRBM_GUI_ACTIVE, share of the launch
the waves are alive, matrix-pipe utilisation inside a wave's lifetime and over the launch, LDS cycles per read, L2 hit rate)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03")
out = []
for d in sorted(glob.glob(os.path.join(src, "pmc_tall3_*"))):
    if not os.path.isdir(d):
        continue
    agg, cnt, dur, name = collections.defaultdict(float), collections.defaultdict(int), [], "?"
    for f in glob.glob(os.path.join(d, "pmc_*", "pmc_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "tall3" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
                name = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
    for f in glob.glob(os.path.join(d, "pmc_*", "pmc_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "tall3" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not agg:
        continue
    c = {k: agg[k] / cnt[k] for k in agg}
    us = sum(dur) / len(dur)
    n, h, w, cin, cout = [int(x) for x in os.path.basename(d).replace("pmc_tall3_", "").split("_")]
    flop = 2.0 * n * h * w * cout * cin * 9
    cyc = c["GRBM_GUI_ACTIVE"] / 8                      # the counter sums the 8 XCDs
    ghz = cyc / us / 1e3
    alive = c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"]     # quad-cycles -> cycles per wave
    mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024         # per SIMD (256 CUs x 4)
    out.append("%s   (%d x %d x %d, %d -> %d channels, uniform(-1,1) operands)" % (name, n, h, w, cin, cout))
    out.append("  launch %.1f us under the profiler = %.0f TFLOP/s; shader clock %.2f GHz -> clock-adjusted MFMA peak %.0f TFLOP/s" % (
        us, flop / us / 1e6, ghz, 2500 * ghz / 2.4))
    out.append("  waves alive %.1f %% of the launch; matrix pipe busy %.1f %% of a wave's lifetime, %.1f %% of the launch" % (
        100 * alive / cyc, 100 * mfma / alive, 100 * mfma / cyc))
    out.append("  per wave-cycle: parked (waitcnt / barrier) %.3f, issue-stalled %.3f, issuing %.3f" % (
        c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    out.append("  instructions per MFMA: LDS %.2f, other VALU %.2f, SALU %.2f, VMEM %.3f;  LDS cycles per read %.2f, bank-conflict cycles %d" % (
        c["SQ_INSTS_LDS"] / c["SQ_INSTS_MFMA"], (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
        c["SQ_INSTS_SALU"] / c["SQ_INSTS_MFMA"], c["SQ_INSTS_VMEM"] / c["SQ_INSTS_MFMA"], c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"],
        c["SQ_LDS_BANK_CONFLICT"]))
    out.append("  L2: hit rate %.3f of %.3g requests; TCP -> L2 read requests %.3g" % (
        c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), c["TCC_HIT_sum"] + c["TCC_MISS_sum"], c["TCP_TCC_READ_REQ_sum"]))
    out.append("  raw: " + ", ".join("%s %.4g" % (k, c[k]) for k in sorted(c)))
    out.append("")
text = ("conv_tall3 under rocprofv3 --pmc (four counter passes, kernel-trace only; tools/ubench/pmc3.sh + tools/pmc_tall3_summary.py),\n"
        "the shipped 4-wave form through the C ABI (tools/ubench/conv_bench3), 5 dispatches per pass averaged.\n\n" + "\n".join(out))
open(os.path.join(ROOT, "profiles", "r03_pmc_tall3.txt"), "w").write(text)
print(text)
