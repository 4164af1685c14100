#!/bin/bash
# Kernels of one captured training step at cfg2: rocprofv3 kernel trace of tools/train_graph_run.py; kernels that run at
# least once per replayed step are listed with their per-step cost.  tools/train_graph_prof.sh <tag>
tag=${1:-train_graph}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/tools/train_graph_run.py 300 > $out/time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/train_graph_run.py 300 > $out/prof_time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
steps=320.0      # 300 timed + 20 warm replays (the eager warm-up and the capture add a few launches per kernel)
tot=0
for r in rows[:60]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n)[:70]
    c=int(r['Calls']); a=float(r['AverageNs'])/1e3
    if c < steps: continue
    tot+=c*a/steps
    print(f"{n:72s} calls/step={c/steps:5.2f} avg={a:8.1f}us step-us={c*a/steps:7.1f}")
print("sum of kernel time per step: %.1f us" % tot)
PY
tail -n 1 $out/time.txt
