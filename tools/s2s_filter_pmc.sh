#!/bin/bash
# PMC passes over the fused seq2seq rollout (tools/s2s_rollout_only.py args): where k_s2s_filter_split's cycles go.
# usage: tools/s2s_filter_pmc.sh <tag> [rollout args]   -> gpurun_out/<tag>/*.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$n -- python3 $root/tools/s2s_rollout_only.py --reps 2 "$@" > $out/$n.log 2>&1 || { tail -3 $out/$n.log; exit 1; }
    python3 $root/tools/pmc_summary.py $out/$n > $out/$n.txt
    rm -rf $out/$n
done
grep -h "k_s2s_filter_split\|k_s2s_gemm_split" $out/*.txt
