#!/bin/bash
# in-situ sweep of ONE environment knob: tools/probes/r4_env_sweep.sh VAR "v1 v2 ..."   (three passes, interleaved)
root=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; list=$2
cd /tmp
for rep in 1 2 3; do
for v in $list; do
  export $var=$v
  python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --no-forward --sustained 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var $v :', d['ms_per_step'], d['sustained']['ms_per_step'])"
done
done
