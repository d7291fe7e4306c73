import torch, time
dev=torch.device("cuda:0")
for mb in (26, 115, 420):
    n = mb*1024*1024//2
    a=torch.randn(n,device=dev).to(torch.bfloat16); b=torch.empty_like(a)
    for name,fn in (("copy", lambda: b.copy_(a)), ("fill", lambda: b.zero_()), ("read(sum)", lambda: a.float().sum() if False else torch.sum(a))):
        for _ in range(3): fn()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)*1e3/20
        print(f"{mb} MB {name}: {us:.1f} us -> {mb*1.048576/us*1e-6*1e6/1e3:.2f} TB/s (one-way bytes)")
