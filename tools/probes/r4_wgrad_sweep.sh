#!/bin/bash
# in-situ sweep of the weight-gradient split-K workgroup targets (YMI_WGRAD_BLOCKS for the 64-row tile, YMI_WGRAD_BLOCKS128 for the 128-row tile)
root=${GRAFT_REPO_ROOT:-/root/repo}
list=${1:-"1024,640 768,512 1024,512 1280,768 1024,768 512,512 768,640"}
cd /tmp
for rep in 1 2 3; do
for cfg in $list; do
  IFS=, read a b <<< "$cfg"
  YMI_WGRAD_BLOCKS=$a YMI_WGRAD_BLOCKS128=$b python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --no-forward --sustained 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks64 $a blocks128 $b :', d['ms_per_step'], d['sustained']['ms_per_step'])"
done
done
