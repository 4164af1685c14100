#!/bin/bash
# Round-4 evidence (run on the GPU box from the repo root): the default bench line (+ under rocprofv3 with kernel stats), PMC
# passes of the training step (each counter set in its own run, --kernel-trace only), eager training-step kernel stats, the
# hidden_size > 64 path, the 64-scene variable-N step, the config-5 shard line.
# usage: tools/profile_round4.sh <tag> [parts]     -> gpurun_out/<tag>/     parts: any of  bench stats pmc train wide dyn cfg5
tag=$1; parts=${2:-"bench stats pmc train wide dyn cfg5"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
has() { [[ " $parts " == *" $1 "* ]]; }
pass() {   # name, counters, bench args...
    local n=$1; shift; local ctr=$1; shift
    echo "[pmc] $n: $ctr"
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$n -- python3 $root/bench.py "$@" > $out/pmc_$n.log 2>&1
    cd $root
    python3 tools/pmc_summary.py $out/pmc_$n > $out/pmc_$n.txt 2>&1
    rm -rf $out/pmc_$n
}
cd $root
if has bench; then echo "[1] default bench"; python3 bench.py > $out/default_bench.json 2> $out/bench.err || exit 1; fi
if has stats; then
    echo "[2] rocprofv3 stats of the default command"
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py > $out/default_bench_under_rocprof.json 2> $out/trace.err || exit 1
    cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/default_bench_kernel_stats.csv
    rm -rf $out/trace
    cd $root
fi
if has pmc; then
    FW="--no-cpu-baseline --no-train --no-rollout --no-other-configs --no-graph --steps 100 --warmup 10"
    pass fwd_fetch "FETCH_SIZE" $FW
    pass fwd_write "WRITE_SIZE" $FW
    pass fwd_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES" $FW
    pass fwd_wait "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $FW
    TR="--no-cpu-baseline --no-rollout --no-other-configs --no-graph --steps 40 --warmup 5"
    pass train_fetch "FETCH_SIZE" $TR
    pass train_write "WRITE_SIZE" $TR
    pass train_mfma "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" $TR
    pass train_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $TR
fi
if has train; then
    echo "[3] training step kernel stats (eager + graph replays)"
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_t -o t -- python3 $root/bench.py --no-cpu-baseline --no-rollout --no-other-configs --no-graph --steps 40 --warmup 5 > $out/train_eager_bench.json 2> $out/trace_t.err
    cp $(find $out/trace_t -name "*kernel_stats.csv" | head -1) $out/train_step_kernel_stats.csv
    rm -rf $out/trace_t
    cd $root
fi
if has wide; then
    echo "[4] hidden_size > 64"
    tools/wide_prof.sh 128 > $out/wide_128.txt 2>&1; cp gpurun_out/wideprof_128/kernel_stats.csv $out/wide_128_kernel_stats.csv
    tools/wide_prof.sh 256 > $out/wide_256.txt 2>&1; cp gpurun_out/wideprof_256/kernel_stats.csv $out/wide_256_kernel_stats.csv
    python3 tools/wide_time.py 128 192 256 > $out/wide_time.txt 2>&1
fi
if has dyn; then
    echo "[5] 64-scene variable-N step"
    python3 tools/dyn_batch_time.py > $out/dyn_batched_timing.txt 2>&1
    python3 tools/dyn_decoder_time.py > $out/dyn_timing_one_call.txt 2>&1
fi
if has cfg5; then
    echo "[6] config-5 shard"
    python3 bench.py --config cfg5shard > $out/cfg5shard_bench.json 2> $out/cfg5shard_bench.err
fi
echo done
