set -e
cd $GRAFT_REPO_ROOT
for tile in 128,128 256,128; do
  echo "== tile $tile"
  for cfg in "32 32" "32 40" "64 32" "32 48" "32 56" "32 64"; do
    set -- $cfg
    YMI_IGEMM_TILE=$tile python tools/conv_bench.py --ops fwd --batch $1 --iters 30 --shape "m128_b$1_h$2,128,128,3,1,$2" | grep m128
  done
done
