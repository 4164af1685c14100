#!/bin/bash
# rocprofv3 kernel stats (+ memory copies) of the config-5 shard's training steps
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/cfg5train
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py --config cfg5shard --no-cpu-baseline --no-rollout --no-graph --steps 3 --warmup 1 > $out/bench.json 2> $out/err.txt
for f in $(find $out/trace -name "*stats.csv"); do cp $f $out/$(basename $f); done
rm -rf $out/trace
python3 - <<PY
import csv,re,glob
for f in sorted(glob.glob("$out/*stats.csv")):
    print("==", f.split('/')[-1])
    rows=list(csv.DictReader(open(f)))
    for r in rows[:16]:
        n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)[:64]
        print(f"{n:66s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:10.1f}us total={float(r['TotalDurationNs'])/1e6:9.1f}ms")
PY
tail -c 600 $out/bench.json
