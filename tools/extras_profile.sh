#!/bin/bash
# rocprofv3 kernel stats + timing output of the components beyond the headline path (run on the GPU box from the repo
# root): dynamic-field variant, variable-N model step, kNN edge builder, data-set simulators.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/extras
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for name in dynfield_time dyn_decoder_time knn_time sim_time s2s_dynfield_time; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o t -- python3 $root/tools/$name.py > $out/$name.txt 2>&1
    cp $(find $out/$name -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv
done
