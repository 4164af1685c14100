#!/bin/bash
# Collect the artifacts profiles/ is refreshed from (run on the GPU box from the repo root):
#   bench line, rocprofv3 kernel stats (forward + training step, eager launches), PMC passes.
# usage: tools/profile_round.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py --no-cpu-baseline --no-graph --steps 100 --warmup 10 > $out/trace.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cd $root
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    n=pmc_$(echo $ctr | cut -d' ' -f1)
    bash tools/pmc_pass.sh $tag/$n "$ctr" --no-cpu-baseline --no-train --no-graph --steps 100 --warmup 10 > $out/$n.txt 2>&1 || echo "pass $n failed" >> $out/errors.txt
done
grep -h "^k_fused\|^k_edge\|^k_node\|^k_segment" $out/pmc_*.txt > $out/pmc_summary.txt || true
