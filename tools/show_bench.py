import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["dtype"], d["value"], d["ms_per_step"], d.get("sustained",{}).get("value"), d["roofline"]["frac"])
for m in ("bf16_mode","x3_mode","f32_mode"):
    if m not in d: continue
    x=d[m]; print(m, x["value"], x["ms_per_step"], x.get("sustained",{}).get("value"), x["roofline"]["kernel"], x["roofline"]["frac"])
    for k in x["roofline"]["kernels"][:10]: print("   ", k["kernel"], k["launches_per_step"], k["ms_per_step"], k["mfma_frac"])
    print("   wgrad", x["roofline"]["weight_gradient"])
if "cfg5" in d: print("cfg5", d["cfg5"]["value"])
i=d.get("inference")
if i:
    print({k:v for k,v in i.items() if k.startswith("fps") or k.startswith("e2e_fps")})
    print(i["timed"]); print(i.get("roofline",{}).get("kernel"), i.get("roofline",{}).get("frac"), i.get("roofline",{}).get("whole_forward"))
    for m,x in i["modes"].items():
        print(m, {k:v for k,v in x.items() if k.startswith("fps")}, x.get("roofline",{}).get("kernel"), x.get("roofline",{}).get("frac"), x.get("roofline",{}).get("whole_forward"))
        for k in x.get("roofline",{}).get("kernels",[]): print("    ", k)
    print(i["e2e_spread"])
print(d.get("cpu_baseline"))
