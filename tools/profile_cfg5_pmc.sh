#!/bin/bash
# PMC passes of the config-5 shard (forward + training step) with the library as committed: HBM bytes and issue / wait
# counters of k_edge_layer, k_edge_layer1, kb_edge_acc.  Each counter set in its own run, --kernel-trace only.
# usage: tools/profile_cfg5_pmc.sh <tag>  -> gpurun_out/<tag>/pmc_*.txt
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
ARGS="--config cfg5shard --no-cpu-baseline --no-rollout --no-graph --steps 4 --warmup 1"
pass() {
    local n=$1; shift; local ctr=$1; shift
    echo "[pmc] $n: $ctr"
    cd /tmp
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$n -- python3 $root/bench.py $ARGS > $out/pmc_$n.log 2>&1 || return 1
    cd $root
    python3 tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt 2>&1
    rm -rf $out/pmc_$n
}
pass fetch "FETCH_SIZE" && pass write "WRITE_SIZE" && \
pass issue "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" && \
pass wait "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"
echo done
