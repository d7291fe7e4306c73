#!/bin/bash
# in-situ sweep of choose_tile's two thresholds (needs a diagnostic build that reads YMI_IGEMM_ENOUGH / YMI_IGEMM_PP_MIN in choose_tile; the shipped library does not): ms/step per setting, interleaved twice
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for rep in 1 2; do
for cfg in "400 300" "200 300" "800 300" "400 150" "400 600" "300 200" "600 450"; do
  set -- $cfg
  YMI_IGEMM_ENOUGH=$1 YMI_IGEMM_PP_MIN=$2 python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --no-forward --sustained 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enough $1 pp_min $2 :', d['ms_per_step'], d['sustained']['ms_per_step'])"
done
done
