#!/bin/bash
# Round-3 final evidence with the library as committed: default bench line (+ under rocprofv3 with kernel stats), the B = 256
# line, the config-5 shard line with kernel stats.   usage: tools/profile_round3_final.sh <tag>  -> gpurun_out/<tag>/
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
echo "[1] default bench"; python3 bench.py > $out/default_bench.json 2> $out/bench.err || exit 1
echo "[2] B = 256"; python3 bench.py --batch 256 --no-cpu-baseline --no-train --no-rollout > $out/b256_bench.json 2>> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
echo "[3] rocprofv3 stats of the default command"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py > $out/default_bench_under_rocprof.json 2> $out/trace.err || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/default_bench_kernel_stats.csv
rm -rf $out/trace
echo "[4] config-5 shard under rocprofv3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace5 -o t -- python3 $root/bench.py --config cfg5shard --no-cpu-baseline > $out/cfg5shard_bench_under_rocprof.json 2> $out/trace5.err || exit 1
cp $(find $out/trace5 -name "*kernel_stats.csv" | head -1) $out/cfg5shard_kernel_stats.csv
rm -rf $out/trace5
echo done
