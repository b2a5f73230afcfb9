cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04_inf
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_inf/inf_stats -o inf -- python $R/tools/infer_kernel_profile.py > $R/gpurun_out/r04_inf/inf_probe.log 2>&1
tail -5 $R/gpurun_out/r04_inf/inf_probe.log
python - <<'EOP'
import csv,os
p=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04_inf/inf_stats/inf_kernel_stats.csv"
rows=list(csv.DictReader(open(p)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-90s calls %5s avg %8.1f us  %5.1f %%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
EOP
