#!/bin/bash
# PMC passes of the config-5 shard's training step for one edge_acc option: where kb_edge_acc / kb_edge_acc8 spend their cycles.
# usage: tools/cfg5_acc_pmc.sh <edge_acc option> -> gpurun_out/cfg5acc<opt>/pmc_*.txt
acc=${1:-3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/cfg5acc$acc
mkdir -p $out
export TMPDIR=/tmp
ARGS="--config cfg5shard --no-cpu-baseline --no-rollout --no-other-configs --no-graph --steps 2 --warmup 1 --opt edge_acc=$acc"
pass() {
    local n=$1; shift; local ctr=$1; shift
    echo "[pmc] $n: $ctr"
    cd /tmp
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$n -- python3 $root/bench.py $ARGS > $out/pmc_$n.log 2>&1 || { tail -5 $out/pmc_$n.log; return 1; }
    cd $root
    python3 tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt 2>&1
    rm -rf $out/pmc_$n
}
pass issue "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" && \
pass wait "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" && \
pass lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
echo done
