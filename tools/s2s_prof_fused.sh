#!/bin/bash
# rocprofv3 kernel stats of the fused seq2seq rollout: tools/s2s_prof_fused.sh <tag> [s2s_rollout_only.py args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/s2s_rollout_only.py "$@" > $out/time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
steps=float(open("$out/time.txt").read().split("steps=")[1].split()[0])
tot=0
for r in rows[:45]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:58]
    c=int(r['Calls']); a=float(r['AverageNs'])/1e3
    tot+=c*a/steps
    print(f"{n:60s} calls/step={c/steps:5.1f} avg={a:8.1f}us step-us={c*a/steps:7.1f}")
print("sum of kernel time per step: %.1f us" % tot)
PY
tail -1 $out/time.txt
