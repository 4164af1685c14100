#!/bin/bash
# rocprofv3 kernel stats of tools/dyn_batch_time.py (64 scenes per prediction step): run on the GPU box from the repo root
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/dynbatch
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/dyn_batch_time.py > $out/time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.1f ms over %d launches" % (tot/1e6, sum(int(r['Calls']) for r in rows)))
for r in rows[:22]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:60]
    print(f"{n:62s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:9.1f}us total={float(r['TotalDurationNs'])/1e6:8.2f}ms {r['Percentage']}%")
PY
tail -3 $out/time.txt
