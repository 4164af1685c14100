#!/bin/bash
# rocprofv3 kernel stats of the seq2seq autoregressive step (run on the GPU box from the repo root).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/s2s_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/s2s_step_time.py > $out/step_time.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 $root/tools/s2s_field_time.py > $out/field_time.txt 2>&1
python3 $root/tools/s2s_localizer_time.py > $out/localizer_time.txt 2>&1
