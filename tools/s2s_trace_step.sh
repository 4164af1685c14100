#!/bin/bash
# per-launch durations of ONE seq2seq step in launch order (rocprofv3 kernel trace of the device rollout): tools/s2s_trace_step.sh <tag> [rollout args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $root/tools/s2s_rollout_only.py --reps 2 "$@" > $out/time.txt 2>&1
python3 - <<PY
import csv,re,glob
f=glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# last complete step: find last k_s2s_rff, take the step before it
idx=[k for k,r in enumerate(rows) if "k_s2s_rff" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
prev_end=None
tot=0
for r in rows[a:b]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'\(.*','',n)[:50]
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev_end)/1e3 if prev_end else 0
    print(f"{n:52s} grid={r['Grid_Size_X']:>8s} dur={(e-s)/1e3:7.1f}us gap={gap:5.1f}us")
    prev_end=e; tot+=(e-s)/1e3
print("kernel sum %.1f us, span %.1f us" % (tot,(int(rows[b-1]["End_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e3))
PY
rm -rf $out/trace
