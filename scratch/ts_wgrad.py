import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'; N=8
for cin,cout,s,ks in [(18,18,320,3),(72,72,80,3),(288,288,20,3)]:
    x=torch.randn(N,cin,s,s,device=dev); dy=torch.randn(N,cout,s,s,device=dev)
    sc=torch.rand(N,cin,device=dev)+0.5; sh=torch.randn(N,cin,device=dev)
    dw=torch.empty(cout,cin,ks,ks,device=dev)
    xa=ops.Act(x,0,cin,sc,sh,0.2); da=ops.full(dy)
    for _ in range(5): ops.conv2d_wgrad(xa,da,dw)
    torch.cuda.synchronize()
    ts=torch.zeros(256,dtype=torch.int64,device=dev)
    os.environ['SAN_DBG_TS']=str(ts.data_ptr())
    ops.conv2d_wgrad(xa,da,dw)
    torch.cuda.synchronize()
    os.environ.pop('SAN_DBG_TS')
    t=[v for v in ts.cpu().tolist() if v]
    d=[t[i+1]-t[i] for i in range(len(t)-1)]
    print((cin,cout,s),'total',t[-1]-t[0],'n',len(t))
    print('  prologue', d[0], ' per tile (compute+stage, barrier):', [(d[i],d[i+1]) for i in range(1,len(d)-1,2)][:10], ' epilogue', d[-1])
