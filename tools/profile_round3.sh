#!/bin/bash
# Round-3 evidence (run on the GPU box from the repo root): the default bench line, rocprofv3 kernel stats of the default
# command, PMC passes (each in its own run, --kernel-trace only) for the forward, the training step, the config-5 shard
# (forward + the new kb_edge_acc training kernels) and the unsplit B = 256 shape (k_fused<*, 8, 3, *>).
# usage: tools/profile_round3.sh <tag>      -> gpurun_out/<tag>/
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
echo "[1] default bench"; python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
echo "[2] rocprofv3 stats of the default command"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py > $out/bench_under_rocprof.json 2> $out/trace.err
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/default_bench_kernel_stats.csv
rm -rf $out/trace
cd $root
pass() {   # name, counters, bench args...
    local n=$1; shift; local ctr=$1; shift
    echo "[pmc] $n: $ctr"
    cd /tmp
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$n -- python3 $root/bench.py "$@" > $out/pmc_$n.log 2>&1
    cd $root
    python3 tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt 2>&1
    rm -rf $out/pmc_$n
}
FW="--no-cpu-baseline --no-train --no-rollout --no-graph --steps 100 --warmup 10"
pass fwd_fetch "FETCH_SIZE" $FW
pass fwd_write "WRITE_SIZE" $FW
pass fwd_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES" $FW
pass fwd_wait "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $FW
TR="--no-cpu-baseline --no-rollout --no-graph --steps 40 --warmup 5"
pass train_fetch "FETCH_SIZE" $TR
pass train_write "WRITE_SIZE" $TR
pass train_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" $TR
echo "[3] training step kernel stats (eager)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_t -o t -- python3 $root/bench.py --no-cpu-baseline --no-rollout --no-graph --steps 40 --warmup 5 > $out/train_eager_bench.json 2> $out/trace_t.err
cp $(find $out/trace_t -name "*kernel_stats.csv" | head -1) $out/train_step_kernel_stats.csv
rm -rf $out/trace_t
cd $root
echo "[4] unsplit shape B = 256 (k_fused<2, 8, 3, false>)"
python3 bench.py --batch 256 --no-cpu-baseline --no-train > $out/b256_bench.json 2> $out/b256_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_b -o t -- python3 $root/bench.py --batch 256 --no-cpu-baseline --no-train --no-rollout > $out/b256_under_rocprof.json 2> $out/trace_b.err
cp $(find $out/trace_b -name "*kernel_stats.csv" | head -1) $out/b256_kernel_stats.csv
rm -rf $out/trace_b
cd $root
B2="--batch 256 --no-cpu-baseline --no-train --no-rollout --no-graph --steps 100 --warmup 10"
pass b256_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" $B2
echo "[5] config-5 shard"
python3 bench.py --config cfg5shard > $out/cfg5shard_bench.json 2> $out/cfg5shard_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_5 -o t -- python3 $root/bench.py --config cfg5shard --no-graph --steps 6 --warmup 2 > $out/cfg5shard_under_rocprof.json 2> $out/trace_5.err
cp $(find $out/trace_5 -name "*kernel_stats.csv" | head -1) $out/cfg5shard_kernel_stats.csv
rm -rf $out/trace_5
cd $root
C5="--config cfg5shard --no-graph --steps 4 --warmup 1"
pass cfg5_fetch "FETCH_SIZE" $C5
pass cfg5_write "WRITE_SIZE" $C5
pass cfg5_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" $C5
echo done
