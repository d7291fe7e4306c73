#!/bin/bash
# in-situ sweep of the streaming passes' grid sizing (needs a diagnostic build that reads YMI_EW_PPT / YMI_EW_CAP / YMI_RED_CAP; the shipped
# library does not).  usage: tools/probes/r4_ew_sweep.sh "ppt,cap,redcap ppt,cap,redcap ..."  (three passes over the list, interleaved)
root=${GRAFT_REPO_ROOT:-/root/repo}
list=${1:-"8,2048,1024 4,2048,1024 16,2048,1024 8,1024,1024 8,4096,1024 8,2048,768 8,2048,1536"}
cd /tmp
for rep in 1 2 3; do
for cfg in $list; do
  IFS=, read a b c <<< "$cfg"
  YMI_EW_PPT=$a YMI_EW_CAP=$b YMI_RED_CAP=$c python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --no-forward --sustained 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ppt $a cap $b redcap $c :', d['ms_per_step'], d['sustained']['ms_per_step'])"
done
done
