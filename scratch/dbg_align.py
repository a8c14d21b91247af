import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialalignmentnetwork_amd import ops, synth, cross
from oracle import cpu_ref as O
from tests.conftest import philox, rel_err
dev='cuda:0'
for (n,c,h,w) in [(2,1,32,32),(2,1,64,64),(1,1,64,64),(2,1,48,80),(2,1,64,32),(2,1,32,64),(2,3,32,32),(1,1,128,128)]:
    net = cross.SpatialTransformer(c)
    p = synth.fill_params([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=41)
    net.load_state_dict(p); net.to(dev).eval()
    mv = philox('mv', (n,c,h,w), lo=0, hi=1); fx = philox('fx', (n,c,h,w), lo=0, hi=1)
    with torch.no_grad():
        off, grid = net(mv.to(dev), fx.to(dev))
        woff, wgrid = O.spatial_transformer_forward(p, mv, fx)
    print((n,c,h,w), 'offset err', rel_err(off.cpu(), woff), 'grid err', rel_err(grid.cpu(), wgrid))
# conv with small cout and multiple tiles + channel offsets
for (cin,cout,h,w,ks,xoff,yoff) in [(32,2,64,64,3,0,0),(32,2,64,64,3,3,1),(6,32,48,80,3,0,0),(32,64,24,40,1,0,5),(64,64,12,20,3,2,0),(64,64,3,5,3,0,0),(64,64,6,10,1,0,0)]:
    n=2
    x = philox('x',(n,cin+xoff+2,h,w)); wt = philox('w',(cout,cin,ks,ks))*0.1; b = philox('b',(cout,))
    y = torch.zeros((n,cout+yoff+1,h,w),device=dev)
    ops.conv2d(ops.Act(x.to(dev),xoff,cin), wt.to(dev), b.to(dev), ops.Act(y,yoff,cout))
    want = torch.nn.functional.conv2d(x[:,xoff:xoff+cin].double(), wt.double(), b.double(), padding=ks//2).float()
    print((cin,cout,h,w,ks,xoff,yoff), rel_err(y[:,yoff:yoff+cout].cpu(), want), y[:, :yoff].abs().max().item() if yoff else 0, y[:, yoff+cout:].abs().max().item())
