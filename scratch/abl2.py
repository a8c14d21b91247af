import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'; N=8
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for cin,cout,s,ks in [(18,18,320,3),(18,36,160,3),(72,36,160,3),(72,72,80,3),(64,64,160,3),(144,144,40,3),(288,288,20,3)]:
    x=torch.randn(N,cin,s,s,device=dev); w=torch.randn(cout,cin,ks,ks,device=dev)*0.05
    sc=torch.rand(N,cin,device=dev)+0.5; sh=torch.randn(N,cin,device=dev)
    y=torch.empty(N,cout,s,s,device=dev)
    xa=ops.Act(x,0,cin,sc,sh,0.2); ya=ops.full(y)
    res=[]
    for wcmax in [4,2,1]:
        os.environ['SAN_WCMAX']=str(wcmax)
        for dbg in [0,15]:
            os.environ['SAN_DBG']=str(dbg)
            res.append((wcmax,dbg, bench(lambda: ops.conv2d(xa,w,None,ya,stats=True))))
    print((cin,cout,s), ' '.join(f"wc{c}/dbg{d}={t:.0f}" for c,d,t in res))
