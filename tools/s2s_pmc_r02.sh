#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) of the fused seq2seq rollout: tools/s2s_pmc_r02.sh <tag> [s2s_rollout_only.py args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
pass() {
    local n=$1; shift; local ctr=$1; shift
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$n -- python3 $root/tools/s2s_rollout_only.py "$@" > $out/pmc_$n.log 2>&1
    cd $root
    python3 tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt 2>&1
    rm -rf $out/pmc_$n
}
pass fetch "FETCH_SIZE" "$@"
pass write "WRITE_SIZE" "$@"
pass mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES" "$@"
pass lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "$@"
cat $out/pmc_fetch.txt $out/pmc_write.txt $out/pmc_mfma.txt $out/pmc_lds.txt | grep -E "filter_split|linear_jobs<4, 4|filter_bimg" 
