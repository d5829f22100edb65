import torch, time
dev=torch.device("cuda:0")
Z=10125
def timeit(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for (mi,mo,d) in [(256,64,5),(256,64,3),(192,192,1),(64,64,5),(64,64,3)]:
    x=torch.randn(Z,mi*d,device=dev,requires_grad=True); W=torch.randn(mi,mo,device=dev)
    g=torch.randn(Z,mo*d,device=dev)
    def f_transpose():
        return torch.matmul(x.reshape(Z,mi,d).transpose(1,2),W).transpose(1,2).reshape(Z,-1)
    def f_bmm():
        return torch.matmul(W.t(), x.view(Z,mi,d)).reshape(Z,-1)
    def f_einsum():
        return torch.einsum('zum,uw->zwm', x.view(Z,mi,d), W).reshape(Z,-1)
    for name,f in [("transpose",f_transpose),("bmm_bcast",f_bmm),("einsum",f_einsum)]:
        tf=timeit(lambda: f())
        o=f()
        tb=timeit(lambda: torch.autograd.grad(o,x,g,retain_graph=True))
        print(f"mi={mi} mo={mo} d={d} {name:10s} fwd {tf:7.1f} us bwd {tb:7.1f} us")
# slicing vs split backward
x=torch.randn(Z,2240,device=dev,requires_grad=True)
def f_slice():
    return x[:, :192].sum()+x[:, 192:960].sum()+x[:,960:].sum()
def f_split():
    a,b,c=torch.split(x,[192,768,1280],dim=1); return a.sum()+b.sum()+c.sum()
for name,f in [("slice",f_slice),("split",f_split)]:
    o=f(); print(name, timeit(lambda: torch.autograd.grad(o,x,retain_graph=True)))
