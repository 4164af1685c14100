#!/bin/bash
# rocprofv3 kernel stats of the seq2seq step at the reference's own sizes (5 objects per graph, 3-D).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/s2s_prof_small
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/s2s_step_time.py --dims 3 --nodes 5 --decoder-hidden 256 > $out/step_time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
