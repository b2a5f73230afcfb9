"""Prints VGPR/AGPR/scratch/occupancy per kernel for one .hip file (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-I", ROOT + "/include",
       "-I", ROOT + "/fast-srgan_amd/csrc", "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print("%-90s vgpr=%3d agpr=%3d sgpr=%3d scratch=%4d occ=%d lds=%d" % (
        name[:90], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
        r.get("Occupancy", -1), r.get("LDS Size", -1)))
