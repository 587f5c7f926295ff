import torch, time
dev='cuda:0'
N,B,H=100,1024,128
d=torch.randn(N,B,H,device=dev); a=torch.randn(N,B,H,device=dev); a2=torch.randn(N,B,2*H,device=dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-s)/n*1e3
print('flat  d.t()@a        %.3f ms'%t(lambda: d.reshape(-1,H).t() @ a.reshape(-1,H)))
print('bmm   sum            %.3f ms'%t(lambda: torch.bmm(d.transpose(1,2), a).sum(0)))
print('flat  d.t()@a2       %.3f ms'%t(lambda: d.reshape(-1,H).t() @ a2.reshape(-1,2*H)))
print('bmm   sum a2         %.3f ms'%t(lambda: torch.bmm(d.transpose(1,2), a2).sum(0)))
print('sum(0) bias          %.3f ms'%t(lambda: d.reshape(-1,H).sum(0)))
print('elementwise mul      %.3f ms'%t(lambda: d*a))
print('tanh                 %.3f ms'%t(lambda: torch.tanh(d)))
w=torch.randn(H,H,device=dev)
print('fwd gemm NBxH @ HxH  %.3f ms'%t(lambda: d.reshape(-1,H) @ w))
