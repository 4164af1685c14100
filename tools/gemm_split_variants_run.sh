#!/bin/bash
# kernel time of k_s2s_gemm_split<1|2> in the product library and in the timing-only variants (tools/gemm_split_variants.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/gs_var
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in ${GSLIBS:-hip gsvar1 gsvar2 gsvar3 gsvar4 gsvar5 gsvar6 gsvar7}; do
    rm -rf $out/trace
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/tools/_alt_lib_run.py libaether_$v.so $root/tools/s2s_rollout_only.py --reps 2 "$@" > $out/$v.log 2>&1 || { echo "$v failed"; tail -3 $out/$v.log; exit 1; }
    f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
    echo "$v: $(grep 'gemm_split<' $f | cut -d, -f1-4 | tr '\n' ' ')"
done
