import torch
dev="cuda:0"
n = 600*1024*1024  # bf16 elements -> 1.2 GB
a = torch.empty(n, dtype=torch.bfloat16, device=dev)
b = torch.empty(n, dtype=torch.bfloat16, device=dev)
def t(f, reps=5):
    best=1e9
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1))
    return best
gb = n*2/1e9
x=t(lambda: a.fill_(1.0)); print("fill  %.1f us  %.2f TB/s (write)"%(x*1e3, gb/x))
x=t(lambda: b.copy_(a)); print("copy  %.1f us  %.2f TB/s (r+w)"%(x*1e3, 2*gb/x))
x=t(lambda: a.float().sum() if False else torch.sum(a.view(torch.int16)[:n//1].to(torch.int16))); print("sum   %.1f us  %.2f TB/s (read)"%(x*1e3, gb/x))
x=t(lambda: torch.add(a, b, out=b)); print("add   %.1f us  %.2f TB/s (2r+w)"%(x*1e3, 3*gb/x))
