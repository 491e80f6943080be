import torch,time
N=10_000_000; D=128
X=torch.empty(N,D,device="cuda").uniform_(-1,1)
Y=torch.empty_like(X)
def t(f,n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
b=X.numel()*4
ms=t(lambda: X.sum()); print(f"sum      {ms:.3f} ms  read {b/ms/1e9:.2f} TB/s")
ms=t(lambda: X.abs().max() if False else torch.max(X)); print(f"max      {ms:.3f} ms  read {b/ms/1e9:.2f} TB/s")
ms=t(lambda: Y.copy_(X)); print(f"copy     {ms:.3f} ms  r+w  {2*b/ms/1e9:.2f} TB/s")
ms=t(lambda: Y.fill_(1.0)); print(f"fill     {ms:.3f} ms  write {b/ms/1e9:.2f} TB/s")
