#!/bin/bash
# in-situ sweep over lines of environment assignments (file, one setting per line; "@" = repository root): three passes, interleaved
root=${GRAFT_REPO_ROOT:-/root/repo}
file=$1; case $file in /*) ;; *) file=$root/$file;; esac
cd /tmp
for rep in 1 2 3; do
while IFS= read -r line; do
  [ -z "$line" ] && continue
  l=${line//@/$root}
  env $l python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --no-forward --sustained 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('''$line :''', d['ms_per_step'], d['sustained']['ms_per_step'])"
done < $file
done
