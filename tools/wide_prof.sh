#!/bin/bash
# rocprofv3 kernel stats of the hidden_size > 64 path (tools/wide_time.py H): run on the GPU box from the repo root
root=${GRAFT_REPO_ROOT:-$(pwd)}
H=${1:-256}
out=$root/gpurun_out/wideprof_$H
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/wide_time.py $H > $out/time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:18]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:60]
    print(f"{n:62s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:9.1f}us total={float(r['TotalDurationNs'])/1e6:8.2f}ms {r['Percentage']}%")
PY
cat $out/time.txt | tail -2
