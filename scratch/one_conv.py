import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'; N=8
cin,cout,s,ks = [int(v) for v in sys.argv[1:5]]
x=torch.randn(N,cin,s,s,device=dev); w=torch.randn(cout,cin,ks,ks,device=dev)*0.05
sc=torch.rand(N,cin,device=dev)+0.5; sh=torch.randn(N,cin,device=dev)
y=torch.empty(N,cout,s,s,device=dev)
xa=ops.Act(x,0,cin,sc,sh,0.2); ya=ops.full(y)
for _ in range(5): ops.conv2d(xa,w,None,ya,stats=True)
torch.cuda.synchronize()
