import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops
dev='cuda:0'
N,cin,cout,s,ks=1,4,4,40,3
x=torch.ones(N,cin,s,s,device=dev)
dw=torch.empty(cout,cin,ks,ks,device=dev)
out=[]
for px in range(0,40):
    dy=torch.zeros(N,cout,s,s,device=dev); dy[0,0,5,px]=1
    ops.conv2d_wgrad(ops.full(x),ops.full(dy),dw)
    out.append(int(dw[0,0,1,1].item()))
print('center tap per impulse x in row 5:',out)
out=[]
for co in range(4):
    dy=torch.zeros(N,cout,s,s,device=dev); dy[0,co,5,:]=1
    ops.conv2d_wgrad(ops.full(x),ops.full(dy),dw)
    out.append(dw[:,0,1,1].tolist())
print('row5 ones in channel co -> dw[:,0,center]:',out)
