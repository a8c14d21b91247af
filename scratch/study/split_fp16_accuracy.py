"""CPU study (round 2): would a split-fp16 convolution (two fp16 parts per operand = 22 mantissa bits, three products
a1w1 + a1w2 + a2w1, fp32 accumulate) stay inside the 1e-4 end-to-end bar, with and without fp16 denormals being flushed
by the matrix core?  Same harness as split_bf16_accuracy.py (oracle with F.conv2d patched, full-size golden weights)."""
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
MINN = 2.0 ** -14


def h(t, ftz):
    p = t.half().float()
    if ftz:
        p = torch.where(p.abs() < MINN, torch.zeros_like(p), p)
    return p


def parts16(t, ftz, scale):
    t = t * scale
    a = h(t, ftz)
    b = h(t - a, ftz)
    return [a, b]


def make(ftz, sa, sw):
    terms = [(0, 0), (0, 1), (1, 0)]
    def conv(x, w, b=None, **kw):
        if w.shape[-1] == 7 or x.shape[1] == 1 and w.shape[0] == 1:
            return conv_real(x, w, b, **kw)
        xs, ws = parts16(x, ftz, sa), parts16(w, ftz, sw)
        y = None
        for i, j in terms:
            t = conv_real(xs[i], ws[j], None, **kw)
            y = t if y is None else y + t
        y = y / (sa * sw)
        return y if b is None else y + b.view(1, -1, 1, 1)
    def tconv(x, w, b=None, **kw):
        xs, ws = parts16(x, ftz, sa), parts16(w, ftz, sw)
        y = None
        for i, j in terms:
            t = tconv_real(xs[i], ws[j], None, **kw)
            y = t if y is None else y + t
        return y / (sa * sw)
    return conv, tconv


def run():
    n, hh, w, casc = 1, 320, 320, 12
    pruned = synth.equispaced_pruned(w, 0.25, 0)
    full, aux = synth.phantom_pair(n, 1, hh, w, seed=1234)
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
    ref32 = torch.from_numpy(gold["img_rec"]).float()
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    print("fp32 oracle vs fp64 arbiter: %.2e" % rel(ref, ref64))
    for name, (ftz, sa, sw) in {"fp16 x2, denormals kept": (False, 1.0, 1.0), "fp16 x2, denormals flushed": (True, 1.0, 1.0),
                                "fp16 x2, flushed, activations x 2^4, weights x 2^8": (True, 16.0, 256.0)}.items():
        F.conv2d, F.conv_transpose2d = make(ftz, sa, sw)
        t0 = time.time()
        with torch.no_grad():
            out = O.recon_align_forward(pT, pR, full, aux, pruned, **kw)["img_rec"]
        F.conv2d, F.conv_transpose2d = conv_real, tconv_real
        print("%-52s vs fp32 oracle %.2e  vs reference fp32 %.2e  vs fp64 arbiter %.2e  (%.0f s)" % (name, rel(out, ref), rel(out, ref32), rel(out, ref64), time.time() - t0), flush=True)


if __name__ == "__main__":
    run()
