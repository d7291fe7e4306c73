"""Group a rocprofv3 --kernel-trace --stats kernel_stats.csv by kernel family (ms per step).
usage: kernel_categories.py STATS.csv STEPS"""
import csv
import sys
from collections import defaultdict

CATS = [
    ("igemm", ("igemm_kernel",)),
    ("wgrad", ("wgrad_kernel",)),
    ("wgrad_reduce", ("wgrad_reduce", "wgrad_stage")),
    ("bn_fwd", ("scale_shift_act", "bn_finalize", "stat_rows")),
    ("bn_bwd", ("bn_act_bwd", "chan_reduce")),
    ("pool", ("pool", "sppf_")),
    ("cbam", ("cbam",)),
    ("swin_mlp", ("swin_mlp_",)),   # round 5: the fused LayerNorm-2 + MLP kernels (forward, backward data path, weight-image pack)
    ("first_conv", ("first_conv_kernel",)),
    ("swin", ("window", "layernorm", "attn", "gelu", "token")),
    ("loss", ("decode_kernel", "metric_kernel", "topk_kernel", "assign_kernel", "posmax", "finalize_kernel", "loss_kernel", "loss_final", "targets_kernel")),
    ("move", ("move_kernel", "nchw_to", "nhwc_to", "upsample", "add_inplace", "copy_kernel")),
    ("pack", ("pack_",)),
    ("optimizer", ("opt_",)),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    tot = defaultdict(float)
    cnt = defaultdict(int)
    other = []
    for r in csv.DictReader(open(path)):
        name, ns, calls = r["Name"], float(r["TotalDurationNs"]), int(r["Calls"])
        for cat, keys in CATS:
            if any(k in name for k in keys):
                break
        else:
            cat = "aten/other"
            other.append((ns, calls, name))
        tot[cat] += ns
        cnt[cat] += calls
    total = sum(tot.values())
    for cat, ns in sorted(tot.items(), key=lambda kv: -kv[1]):
        print(f"{cat:14s} {ns / steps / 1e6:8.3f} ms/step  {cnt[cat] / steps:7.1f} launches/step  {100 * ns / total:5.1f}%")
    print(f"{'total':14s} {total / steps / 1e6:8.3f} ms/step  {sum(cnt.values()) / steps:7.1f} launches/step")
    print("largest aten/other kernels:")
    for ns, calls, name in sorted(other, reverse=True)[:14]:
        print(f"  {ns / steps / 1e6:7.3f} ms/step {calls / steps:6.1f}/step  {name[:130]}")


if __name__ == "__main__":
    main()
