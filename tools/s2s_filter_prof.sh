#!/bin/bash
# rocprofv3 kernel stats of tools/s2s_filter_time.py: tools/s2s_filter_prof.sh <tag> [args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/s2s_filter_time.py "$@" > $out/time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:12]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:60]
    print(f"{n:62s} calls={int(r['Calls']):4d} avg={float(r['AverageNs'])/1e3:9.1f}us min={float(r['MinNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
tail -2 $out/time.txt
