#!/bin/bash
# PMC passes over the seq2seq step (run on the GPU box from the repo root): matrix-pipe busy cycles and memory traffic of
# k_s2s_filter / k_s2s_linear.  Separate --pmc passes, counters only with --kernel-trace (MI355X_MICROARCH.md).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/s2s_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES"; do
    n=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$n -- python3 $root/tools/s2s_step_time.py > $out/$n.log 2>&1
    python3 $root/tools/pmc_summary.py $out/$n > $out/$n.txt
done
cat $out/*.txt | grep "k_s2s_filter\|k_s2s_linear<0, 4, 4>\|k_s2s_linear<4, 4, 4>" > $out/summary.txt
