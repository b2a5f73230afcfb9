"""Prints the figures of a bench.py result: the lean stdout line (profiles/r06_bench_steps20.json.log) or the full object (bench_detail.json)."""
import json
import sys

text = open(sys.argv[1]).read()
d = json.loads(text) if text.lstrip().startswith("{\n") else json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
r = d["roofline"]
print("%s: %.1f %s, %.2f ms/step; dominant kernel %s: %.1f %s = %.3f of %.0f, traffic %s" % (
    d["dtype"], d["value"], d["unit"], d["ms_per_step"], r["kernel"], r["achieved"], r["unit"], r["frac"], r["peak"], r.get("traffic")))
for k in ("legs_images_per_s", "legs_dominant_kernel_frac", "cfg5", "inference_fps", "cpu_baseline"):
    if k in d:
        print(k, d[k])
for m in ("x3v_mode", "x3_mode", "f16_mode", "bf16_mode", "f32_mode"):      # the full object only
    if m in d:
        x = d[m]
        print(m, x["value"], x["ms_per_step"], x["roofline"]["kernel"], x["roofline"]["frac"])
        for k in x["roofline"]["kernels"][:8]:
            print("   ", k["kernel"], k["launches_per_step"], k["ms_per_step"], k["mfma_frac"])
