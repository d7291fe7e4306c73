#!/bin/bash
# baseline of the round: tests, bench line, forward-only kernel table, full-step kernel table
set -e
tag=${1:-r4_base}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
if [ "$2" != "notests" ]; then
python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
fi
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_fwd -o p -- python3 $root/bench.py --forward-only --steps 27 > $out/fwd_under_rocprof.json 2> $out/trace_fwd.err
cp $(ls $out/trace_fwd/*kernel_stats.csv $out/trace_fwd/*/*kernel_stats.csv 2>/dev/null | head -1) $out/fwd_kernel_stats.csv
python3 $root/tools/kernel_categories.py $out/fwd_kernel_stats.csv 32 > $out/fwd_kernel_categories.txt
cat $out/fwd_kernel_categories.txt | head -16
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 > $out/bench_under_rocprof.json 2> $out/trace.err
cp $(ls $out/trace/*kernel_stats.csv $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv
python3 $root/tools/kernel_categories.py $out/kernel_stats.csv 33 > $out/kernel_categories.txt
cat $out/kernel_categories.txt | head -16
rm -rf $out/trace/*kernel_trace.csv $out/trace/*/*kernel_trace.csv $out/trace_fwd/*kernel_trace.csv $out/trace_fwd/*/*kernel_trace.csv 2>/dev/null || true
