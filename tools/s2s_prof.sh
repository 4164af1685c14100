#!/bin/bash
# rocprofv3 kernel stats of the seq2seq step: tools/s2s_prof.sh <tag> [s2s_step_time.py args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/${S2S_TOOL:-s2s_step_time.py} "$@" > $out/step_time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:40]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:60]
    print(f"{n:62s} calls={int(r['Calls']):4d} avg={float(r['AverageNs'])/1e3:9.1f}us pct={r['Percentage']}")
PY
tail -1 $out/step_time.txt
