#!/bin/bash
# quick check of a change: GPU tests (optional), the bench line without the CPU baseline, optional kernel tables
# usage: tools/probes/r4_quick.sh TAG [tests] [prof]
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
if [[ " $* " == *" tests "* ]]; then
python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
fi
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("$out/bench.json"))
print("img/s", d["value"], "ms/step", d["ms_per_step"], "fwd ms", d["forward"]["ms"], "sustained", d.get("sustained",{}).get("ms_per_step"), "igemm", d["roofline"]["families"]["igemm"]["ms_per_step"], "wgrad", d["roofline"]["families"]["wgrad"]["ms_per_step"])
PY
if [[ " $* " == *" prof "* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_fwd -o p -- python3 $root/bench.py --forward-only --steps 27 > $out/fwd_under_rocprof.json 2> $out/trace_fwd.err
cp $(ls $out/trace_fwd/*kernel_stats.csv $out/trace_fwd/*/*kernel_stats.csv 2>/dev/null | head -1) $out/fwd_kernel_stats.csv
python3 $root/tools/kernel_categories.py $out/fwd_kernel_stats.csv 32 > $out/fwd_kernel_categories.txt
head -9 $out/fwd_kernel_categories.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 > $out/bench_under_rocprof.json 2> $out/trace.err
cp $(ls $out/trace/*kernel_stats.csv $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv
python3 $root/tools/kernel_categories.py $out/kernel_stats.csv 33 > $out/kernel_categories.txt
head -14 $out/kernel_categories.txt
rm -rf $out/trace $out/trace_fwd
fi
