#!/bin/bash
# rocprofv3 kernel summary of the default bench command (no forward sub-record, no CPU baseline, no sustained record:
# exactly 33 training steps of kernels) + the per-family table.   usage (on the GPU box): tools/prof_step.sh TAG [bench args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 "$@" > $root/gpurun_out/${tag}_bench.json 2> $root/gpurun_out/${tag}.err
cd $root
python tools/kernel_categories.py $(ls gpurun_out/$tag/*kernel_stats.csv gpurun_out/$tag/*/*kernel_stats.csv 2>/dev/null | head -1) 33 > gpurun_out/${tag}_cat.txt
cat gpurun_out/${tag}_cat.txt
