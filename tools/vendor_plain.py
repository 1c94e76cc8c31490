import torch
n=8192
a=torch.randn(n,n,device="cuda").to(torch.bfloat16); b=torch.randn(n,n,device="cuda").to(torch.bfloat16)
for _ in range(6): torch.matmul(a,b.t())
ev=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize(); ev[0].record()
for _ in range(20): torch.matmul(a,b.t())
ev[1].record(); torch.cuda.synchronize()
us=ev[0].elapsed_time(ev[1])/20*1e3
print({"vendor_matmul_nt_8192": round(us,1), "PFLOPs": round(2*n**3/us*1e-9,3)})
