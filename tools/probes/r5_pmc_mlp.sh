#!/bin/bash
# SQ counter passes of the fused SwinBlock MLP kernels at the model's shape (tools/probes/swin_mlp_probe.py) -> gpurun_out/<TAG>/pmc_mlp.txt
set -e
tag=${1:-r5_pmc_mlp}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
rm -rf $out/p1 $out/p2 $out/p3
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $out/p1 -o p -- python3 $root/tools/probes/swin_mlp_probe.py > /dev/null 2> $out/p1.err
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $out/p2 -o p -- python3 $root/tools/probes/swin_mlp_probe.py > /dev/null 2> $out/p2.err
rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d $out/p3 -o p -- python3 $root/tools/probes/swin_mlp_probe.py > /dev/null 2> $out/p3.err || true
python3 $root/tools/pmc_sq.py $out/p1 $out/p2 $out/p3 --filter swin_mlp > $out/pmc_mlp.txt
rm -rf $out/p1 $out/p2 $out/p3
cat $out/pmc_mlp.txt
