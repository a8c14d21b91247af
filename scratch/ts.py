import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'; N=8
for cin,cout,s,ks in [(18,18,320,3),(72,72,80,3),(288,288,20,3)]:
    x=torch.randn(N,cin,s,s,device=dev); w=torch.randn(cout,cin,ks,ks,device=dev)*0.05
    sc=torch.rand(N,cin,device=dev)+0.5; sh=torch.randn(N,cin,device=dev)
    y=torch.empty(N,cout,s,s,device=dev)
    xa=ops.Act(x,0,cin,sc,sh,0.2); ya=ops.full(y)
    for _ in range(10): ops.conv2d(xa,w,None,ya,stats=True)
    torch.cuda.synchronize()
    ts=torch.zeros(256,dtype=torch.int64,device=dev)
    os.environ['SAN_DBG_TS']=str(ts.data_ptr())
    ops.conv2d(xa,w,None,ya,stats=True)
    torch.cuda.synchronize()
    os.environ.pop('SAN_DBG_TS')
    t=[v for v in ts.cpu().tolist() if v]
    d=[t[i+1]-t[i] for i in range(len(t)-1)]
    print((cin,cout,s),'total',t[-1]-t[0],'n',len(t))
    print('  setup+first prefetch', d[0])
    body=d[1:-3]
    per=[body[i:i+5] for i in range(0,len(body),5)]
    for p_ in per[:5]: print('  chunk: barrier1 %d  ldswrite %d  barrier2 %d  prefetch %d  compute %d'%tuple(p_) if len(p_)==5 else p_)
    print('  epilogue: stats', d[-2], ' stores', d[-1], ' (last compute->epi marker', d[-3], ')')
