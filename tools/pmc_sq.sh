#!/bin/bash
# SQ counter passes (two --pmc passes of 8 / 8 counters) of the dominant GEMM kernel instantiations on model layer shapes -> gpurun_out/<TAG>/pmc_sq.txt
# usage: tools/pmc_sq.sh TAG
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
: > $out/pmc_sq.txt
run() {  # name, conv_bench arguments
  name=$1; shift
  rm -rf $out/p1 $out/p2
  rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $root/tools/conv_bench.py "$@" --iters 8 > /dev/null 2> $out/p1.err
  rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $root/tools/conv_bench.py "$@" --iters 8 > /dev/null 2> $out/p2.err
  echo "# ===== $name: tools/conv_bench.py $* --iters 8 (bs 32 layer shape; per-launch means)" >> $out/pmc_sq.txt
  python3 $root/tools/pmc_sq.py $out/p1 $out/p2 --filter "$FILTER" >> $out/pmc_sq.txt
  rm -rf $out/p1 $out/p2
}
FILTER=igemm_kernel run "det.cv3[0] second conv 128->128 3x3 @80x80: forward (256x128 ping-pong tile, statistics epilogue) and data gradient (the same tile, no statistics)" --only "det.cv3[0]" --ops fwd,dgrad
FILTER=igemm_kernel run "c2f4.m 64->64 3x3 @80x80: forward and data gradient (128x64 tile)" --only "c2f4.m" --ops fwd,dgrad
FILTER=wgrad_kernel run "det.cv3[0] 128->128 3x3 @80x80: weight gradient (128-row tile)" --only "det.cv3[0]" --ops wgrad
cat $out/pmc_sq.txt
