"""CPU study for the next round's conv kernels: how far does a split-bf16 convolution (operands
decomposed into 2 or 3 bf16 parts, products accumulated in fp32 -- what bf16 MFMAs would compute) move
the 12-cascade output away from the fp32 reference?  Uses the oracle (test infrastructure) with F.conv2d /
F.conv_transpose2d monkey-patched; weights and inputs are the bench's.  Prints rel-L2 vs the fp32 oracle
run and vs the reference's fp64 arbiter (tests/golden/e2e_full_320.npz)."""
import os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as O
from spatialalignmentnetwork_amd import synth

torch.set_num_threads(8)
conv_real, tconv_real = F.conv2d, F.conv_transpose2d


def parts(t, k):
    out, r = [], t
    for _ in range(k):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def make(kparts, terms):
    def conv(x, w, b=None, **kw):
        if w.shape[-1] == 7 or x.shape[1] == 1 and w.shape[0] == 1:      # loss windows: leave exact
            return conv_real(x, w, b, **kw)
        xs, ws = parts(x, kparts), parts(w, kparts)
        y = None
        for i, j in terms:
            t = conv_real(xs[i], ws[j], None, **kw)
            y = t if y is None else y + t
        return y if b is None else y + b.view(1, -1, 1, 1)
    def tconv(x, w, b=None, **kw):
        xs, ws = parts(x, kparts), parts(w, kparts)
        y = None
        for i, j in terms:
            t = tconv_real(xs[i], ws[j], None, **kw)
            y = t if y is None else y + t
        return y
    return conv, tconv


def run():
    n, h, w, casc = 1, 320, 320, 12
    pruned = synth.equispaced_pruned(w, 0.25, 0)
    full, aux = synth.phantom_pair(n, 1, h, w, seed=1234)
    # weights exactly as the full-size golden uses them (seeds 1235 / 1236)
    from spatialalignmentnetwork_amd.cross import SpatialTransformer
    from spatialalignmentnetwork_amd.varnet import VarNet
    T, R = SpatialTransformer(channels=1), VarNet(num_cascades=casc, sens_chans=8, sens_pools=4, chans=18, pools=4, use_ref=True)
    pT = synth.fill_params([(k, tuple(v.shape)) for k, v in T.state_dict().items()], seed=1235)
    pR = synth.fill_params([(k, tuple(v.shape)) for k, v in R.state_dict().items()], seed=1236)
    kw = dict(shape=w, sparsity=0.25, num_cascades=casc)
    with torch.no_grad():
        ref = O.recon_align_forward(pT, pR, full, aux, pruned, **kw)["img_rec"]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "e2e_full_320.npz"))
    ref64 = torch.from_numpy(gold["img_rec_f64"]).float()
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    print("fp32 oracle vs fp64 arbiter: %.2e" % rel(ref, ref64))
    variants = {
        "bf16 x1 (plain bf16 operands, fp32 accumulate)": (1, [(0, 0)]),
        "bf16 x2, 3 products (hh, hl, lh)": (2, [(0, 0), (0, 1), (1, 0)]),
        "bf16 x2, 4 products": (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
        "bf16 x3, 6 products": (3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
    }
    for name, (k, terms) in variants.items():
        F.conv2d, F.conv_transpose2d = make(k, terms)
        t0 = time.time()
        with torch.no_grad():
            out = O.recon_align_forward(pT, pR, full, aux, pruned, **kw)["img_rec"]
        F.conv2d, F.conv_transpose2d = conv_real, tconv_real
        print("%-48s rel-L2 vs fp32 oracle %.2e   vs fp64 arbiter %.2e   (%.0f s)" % (name, rel(out, ref), rel(out, ref64), time.time() - t0), flush=True)


if __name__ == "__main__":
    run()
