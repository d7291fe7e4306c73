#!/bin/bash
# same-box A/B of two source TREES (when the change crosses the C ABI, so that YMI_LIB cannot switch it): a = build_ab/head_tree (git archive of
# the previous commit with its library), b = the working tree.  usage: tools/probes/r4_tree_ab.sh TAG
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for l in a b a b; do
  if [ $l = a ]; then t=$root/build_ab/head_tree; else t=$root; fi
  python3 $t/bench.py --no-cpu-baseline --no-kernel-timing --sustained 100 > $out/bench_$l.json 2> $out/bench_$l.err || { tail -5 $out/bench_$l.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$out/bench_$l.json"))
print("$l", "img/s", d["value"], "ms/step", d["ms_per_step"], "fwd ms", d["forward"]["ms"], "sustained", d.get("sustained",{}).get("ms_per_step"))
PY
done
for l in a b; do
  if [ $l = a ]; then t=$root/build_ab/head_tree; else t=$root; fi
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$l -o p -- python3 $t/bench.py --no-cpu-baseline --no-forward --no-kernel-timing --sustained 0 --steps 6 --warmup 2 > /dev/null 2> $out/trace_$l.err
  cp $(ls $out/trace_$l/*kernel_trace.csv $out/trace_$l/*/*kernel_trace.csv 2>/dev/null | head -1) $out/trace_$l.csv
  rm -rf $out/trace_$l
done
python3 $root/tools/step_timeline.py $out/trace_a.csv --compare $out/trace_b.csv > $out/timeline.txt
grep -A 30 "per kernel name" $out/timeline.txt
