import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'; N=8
for cin,cout,s,ks in [(18,18,320,3),(72,36,160,3),(64,64,160,3)]:
    x=torch.randn(N,cin,s,s,device=dev); w=torch.randn(cout,cin,ks,ks,device=dev)*0.05
    sc=torch.rand(N,cin,device=dev)+0.5; sh=torch.randn(N,cin,device=dev)
    y=torch.empty(N,cout,s,s,device=dev)
    xa=ops.Act(x,0,cin,sc,sh,0.2); ya=ops.full(y)
    for _ in range(30): ops.conv2d(xa,w,None,ya,stats=True)
    torch.cuda.synchronize()
    ts=torch.zeros(256,dtype=torch.int64,device=dev)
    os.environ['SAN_DBG_TS']=str(ts.data_ptr())
    ops.conv2d(xa,w,None,ya,stats=True)
    torch.cuda.synchronize()
    os.environ.pop('SAN_DBG_TS')
    t=ts.cpu().tolist(); t=[v for v in t if v]
    d=[t[i+1]-t[i] for i in range(len(t)-1)]
    print((cin,cout,s), 'total', t[-1]-t[0], 'n', len(t))
    print('  prologue-prefetch', d[0], 'first', d[1:2])
    # per chunk: [top->barrier1 wait][write][barrier2 wait][prefetch issue][compute]
    per=[d[i:i+5] for i in range(1,len(d)-2,5)]
    for p_ in per[:6]: print('  chunk: wait1', p_[0], 'write', p_[1] if len(p_)>1 else None, 'wait2', p_[2] if len(p_)>2 else None, 'prefetch', p_[3] if len(p_)>3 else None, 'compute', p_[4] if len(p_)>4 else None)
    print('  tail (stats, ...)', d[-2:])
