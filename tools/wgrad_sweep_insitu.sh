# in-situ sweep of the split-K workgroup targets: the whole training step (graph replay) per setting
for b128 in ${B128S:-640 896 1280}; do for b64 in ${B64S:-1024 1536}; do
echo "== B128=$b128 B64=$b64"; YMI_WGRAD_BLOCKS128=$b128 YMI_WGRAD_BLOCKS=$b64 python bench.py --no-cpu-baseline --no-forward --sustained 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['families']['wgrad'])"
done; done
