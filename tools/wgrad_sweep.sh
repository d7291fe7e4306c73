# development sweep of the split-K workgroup targets (env knobs of csrc/wgrad.hip) over the model's layer list
for b128 in ${B128S:-512}; do for b64 in ${B64S:-384 512 640 768}; do
echo "== B128=$b128 B64=$b64"; YMI_WGRAD_BLOCKS128=$b128 YMI_WGRAD_BLOCKS=$b64 python tools/conv_bench.py --ops wgrad --iters 10 2>/dev/null | tail -1
done; done
