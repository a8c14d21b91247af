import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'
N=8
cfgs=[(3,18,320,3),(18,18,320,3),(36,18,320,3),(18,36,160,3),(36,36,160,3),(72,36,160,3),(36,72,80,3),(72,72,80,3),(144,72,80,3),
      (72,144,40,3),(144,144,40,3),(288,144,40,3),(144,288,20,3),(288,288,20,3),
      (2,32,320,3),(32,32,320,3),(96,32,320,3),(64,64,160,3),(128,64,160,3),(64,64,80,3),(64,64,40,3),(64,64,20,3),(18,2,320,1),(64,64,160,1)]
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
tot=0
for cin,cout,s,ks in cfgs:
    x=torch.randn(N,cin,s,s,device=dev); w=torch.randn(cout,cin,ks,ks,device=dev)*0.05
    sc=torch.rand(N,cin,device=dev)+0.5; sh=torch.randn(N,cin,device=dev)
    y=torch.empty(N,cout,s,s,device=dev)
    xa=ops.Act(x,0,cin,sc,sh,0.2); ya=ops.full(y)
    us=bench(lambda: ops.conv2d(xa,w,None,ya,stats=True))
    fl=2.0*N*s*s*cout*cin*ks*ks
    tiles=ops.lib().query("san_conv_stat_tiles", N,s,s,cin,cout,ks)
    print(f"conv{ks} {cin:3d}->{cout:3d} @{s:3d}: {us:8.1f} us  {fl/us/1e6:6.1f} TF  tiles={tiles}")
for cin,cout,s in [(288,144,20),(144,72,40),(72,36,80),(36,18,160)]:
    x=torch.randn(N,cin,s,s,device=dev); w=torch.randn(cin,cout,2,2,device=dev)*0.05
    y=torch.empty(N,cout,2*s,2*s,device=dev)
    us=bench(lambda: ops.tconv2x2(ops.full(x),w,ops.full(y),stats=True))
    fl=2.0*N*s*s*4*cout*cin
    print(f"tconv {cin:3d}->{cout:3d} @{s:3d}: {us:8.1f} us  {fl/us/1e6:6.1f} TF")
