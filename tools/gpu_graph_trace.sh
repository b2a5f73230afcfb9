#!/bin/bash
# kernel trace of hipGraph-replayed iterations: how much of an iteration's wall time has NO kernel running, how much has one / several
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/graph_trace -o g -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-inference --no-f32 --no-sustained > $O/graph_trace.log 2>&1
tail -1 $O/graph_trace.log | cut -c1-200
python - <<P
import csv
rows=list(csv.DictReader(open('$O/graph_trace/g_kernel_trace.csv')))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id']) for r in rows)
t_end=max(e[1] for e in ev)
# the last ~150 ms: graph replays (timed steps) -- find by taking the final 6 iterations worth: use events within last 170 ms before the roofline instrumentation? simpler: analyse the window of the timed region printed by bench is unknown; take the densest 100 ms window
import bisect
starts=[e[0] for e in ev]
# choose window: 100 ms ending 60% through the trace's last second
lo=ev[0][0]; hi=t_end
# sweep-line over entire trace, then report for 10 ms buckets
pts=[]
for s,e,_,_ in ev: pts.append((s,1)); pts.append((e,-1))
pts.sort()
import collections
bucket=collections.defaultdict(lambda:[0,0,0])   # idle, single, multi ns per 10ms bucket
cur=0; prev=pts[0][0]
for t,d in pts:
    dt=t-prev
    if dt>0:
        b=(prev-lo)//10_000_000
        k=0 if cur==0 else (1 if cur==1 else 2)
        bucket[b][k]+=dt
    cur+=d; prev=t
out=[]
for b in sorted(bucket):
    i,s,m=bucket[b]; tot=i+s+m
    out.append((b,i/1e6,s/1e6,m/1e6))
# print the 30 busiest consecutive buckets
best=max(range(0,max(1,len(out)-18)), key=lambda k: sum(o[2]+o[3] for o in out[k:k+18]))
print("10-ms buckets (idle / one kernel / several kernels, ms):")
for o in out[best:best+18]: print("  bucket %5d: idle %.2f  single %.2f  multi %.2f"%o)
tot=[sum(o[j] for o in out[best:best+18]) for j in (1,2,3)]
print("window total: idle %.1f ms, single %.1f ms, multi %.1f ms"%tuple(tot))
P
