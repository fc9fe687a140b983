import time, torch, numpy as np
n=6000
v=torch.randn(n,n,dtype=torch.float64,device='cuda')
torch.cuda.synchronize()
for rep in range(3):
    t0=time.perf_counter(); a=np.empty((n,n)); ta=torch.from_numpy(a); ta.copy_(v); torch.cuda.synchronize(); t1=time.perf_counter()
    print(f"fresh pageable numpy: {(t1-t0)*1e3:.1f} ms")
    del a, ta
p=torch.empty((n,n),dtype=torch.float64,pin_memory=True)
for rep in range(3):
    t0=time.perf_counter(); p.copy_(v, non_blocking=True); torch.cuda.synchronize(); t1=time.perf_counter()
    print(f"pinned (reused): {(t1-t0)*1e3:.1f} ms")
t0=time.perf_counter(); q=torch.empty((n,n),dtype=torch.float64,pin_memory=True); t1=time.perf_counter(); print(f"pinned alloc 288 MB: {(t1-t0)*1e3:.1f} ms")
del q
t0=time.perf_counter(); q=torch.empty((n,n),dtype=torch.float64,pin_memory=True); t1=time.perf_counter(); print(f"pinned alloc again (cached): {(t1-t0)*1e3:.1f} ms")
