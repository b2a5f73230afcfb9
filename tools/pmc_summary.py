"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files (several passes), with the ratios that matter."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for path in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(ConvK")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:70]
        if "conv" not in k and "instnorm" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[(k, path)] += 1
for k, v in agg.items():
    n = max(c for (kk, _), c in cnt.items() if kk == k)
    print(k, "dispatches/pass", n)
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   per wave-cycle: wait_any %.3f  wait_inst_any %.3f  active_any %.3f  wait_inst_lds %.3f" % (
            v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_WAIT_INST_LDS"] / wc))
        print("   mfma_busy/busy_cycles %.3f (x4 SIMDs per CU-ish)  mfma insts %.3g" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CYCLES"], 1), v.get("SQ_INSTS_MFMA", 0)))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print("   lds conflict/active %.3f  active_inst lds %.3g valu %.3g vmem %.3g misc %.3g sca %.3g" % (
            v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], v["SQ_ACTIVE_INST_LDS"], v["SQ_ACTIVE_INST_VALU"], v["SQ_ACTIVE_INST_VMEM"],
            v["SQ_ACTIVE_INST_MISC"], v["SQ_ACTIVE_INST_SCA"]))
    if v.get("TCC_REQ"):
        print("   L2 hit rate %.3f (req %.3g)" % (v["TCC_HIT"] / max(v["TCC_HIT"] + v["TCC_MISS"], 1), v["TCC_REQ"]))
    if v.get("GRBM_GUI_ACTIVE"):
        print("   GRBM_GUI_ACTIVE %.4g  TA_BUSY %.4g" % (v["GRBM_GUI_ACTIVE"], v.get("GRBM_TA_BUSY", 0)))
