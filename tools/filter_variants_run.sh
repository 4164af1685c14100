#!/bin/bash
# kernel time of k_s2s_filter_split in the product library and in the timing-only variants (tools/filter_variants.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/filt_var
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in hip filtvar1 filtvar2 filtvar3 filtvar4 filtvar5; do
    rm -rf $out/trace
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/_alt_lib_run.py libaether_$v.so $root/tools/s2s_rollout_only.py --reps 2 "$@" > $out/$v.log 2>&1 || { echo "$v failed"; tail -3 $out/$v.log; exit 1; }
    f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
    echo "$v: $(grep filter_split $f | head -1 | cut -d, -f2-4)"
done
