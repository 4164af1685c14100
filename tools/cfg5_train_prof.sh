#!/bin/bash
# rocprofv3 kernel stats of the config-5 shard's training steps.   usage: tools/cfg5_train_prof.sh [edge_acc option 0|1|2|3]
root=${GRAFT_REPO_ROOT:-$(pwd)}
acc=${1:-2}
out=$root/gpurun_out/cfg5train_$acc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py --config cfg5shard --no-cpu-baseline --no-rollout --no-other-configs --no-graph --steps 3 --warmup 1 --opt edge_acc=$acc > $out/bench.json 2> $out/err.txt
for f in $(find $out/trace -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
rm -rf $out/trace
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:24]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:64]
    print(f"{n:66s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:10.1f}us total={float(r['TotalDurationNs'])/1e6:9.1f}ms")
PY
