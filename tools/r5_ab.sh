#!/bin/bash
# same-box A/B of two bench.py command lines (hooks, options, libraries): bench line twice per arm, interleaved a b a b, then one step's kernel trace per arm
# usage: tools/r5_ab.sh TAG "ARGS_A" "ARGS_B" [ENV_A] [ENV_B]     e.g.  tools/r5_ab.sh r5_mlp "--hook fused_swin_mlp=0" ""
set -e
tag=$1; aa=$2; ab=$3; ea=$4; eb=$5
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for l in a b a b; do
  if [ $l = a ]; then args=$aa; envs=$ea; else args=$ab; envs=$eb; fi
  env $envs python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --sustained 100 $args > $out/bench_$l.json 2> $out/bench_$l.err || { tail -5 $out/bench_$l.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$out/bench_$l.json"))
print("$l", "img/s", d["value"], "ms/step", d["ms_per_step"], "fwd ms", d["forward"]["ms"], "sustained", d.get("sustained",{}).get("ms_per_step"))
PY
done
for l in a b; do
  if [ $l = a ]; then args=$aa; envs=$ea; else args=$ab; envs=$eb; fi
  [ -n "$envs" ] && export $envs
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$l -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --no-kernel-timing --sustained 0 --steps 6 --warmup 2 $args > /dev/null 2> $out/trace_$l.err
  [ -n "$envs" ] && for kv in $envs; do unset ${kv%%=*}; done
  cp $(ls $out/trace_$l/*kernel_trace.csv $out/trace_$l/*/*kernel_trace.csv 2>/dev/null | head -1) $out/trace_$l.csv
  rm -rf $out/trace_$l
done
python3 $root/tools/step_timeline.py $out/trace_a.csv --compare $out/trace_b.csv > $out/timeline.txt
grep -A 40 "per kernel name" $out/timeline.txt
