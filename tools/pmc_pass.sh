#!/bin/bash
# One rocprofv3 PMC pass over a bench.py invocation (run on the GPU box from the repo root).
# usage: tools/pmc_pass.sh <outdir-name> "<counters>" <bench args...>
set -e
name=$1; shift; ctr=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $root/gpurun_out/$name -- python3 $root/bench.py "$@" > $root/gpurun_out/$name.log 2>&1
python3 $root/tools/pmc_summary.py $root/gpurun_out/$name
