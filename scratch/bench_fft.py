import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'
def bench(fn, reps=50):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for N,C,H,W in [(8,1,320,320),(64,1,320,320),(8,4,320,320)]:
    k=torch.randn(N,C,H,W,dtype=torch.complex64,device=dev); k0=torch.randn_like(k); s=torch.randn_like(k)
    out=torch.empty(N,3,H,W,device=dev); r=torch.randn(N,2,H,W,device=dev)
    mask=(torch.rand(W,device=dev)>0.5).float(); dcw=torch.ones(1,device=dev); ko=torch.empty_like(k)
    E=H*W*8
    tA=bench(lambda: ops.sens_reduce(k,s,out)); tB=bench(lambda: ops.sens_expand_dc(r,s,k,k0,mask,dcw,ko)); tF=bench(lambda: ops.fft2c(k)); tR=bench(lambda: ops.ifft2_rss(k))
    print(f"N={N} C={C}: sens_reduce {tA:.1f} us ({(2*C+1)*N*E/tA/1e3:.0f} GB/s)  expand_dc {tB:.1f} us ({(4*C+1)*N*E/tB/1e3:.0f} GB/s)  fft2 {tF:.1f} us ({2*C*N*E/tF/1e3:.0f} GB/s) ifft2_rss {tR:.1f} us; A+B algorithmic {(6*C+2)*N*E/(tA+tB)/1e3:.0f} GB/s")
