"""Round 6: the one-pass cluster forms of the norm + activation backward (san_act_bwd_in, san_bn_act_bwd) against the multi-launch
forms and float64, their timings, and a determinism loop.  Run on the GPU box under `timeout`."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spatialalignmentnetwork_amd import ops
from spatialalignmentnetwork_amd._lib import lib
dev = "cuda:0"
torch.manual_seed(0)


def tune(on=-1, min_hw=0, v=0, bn=-1):
    lib().call("san_act_bwd_cluster_set_tuning", on, min_hw, v, bn)


def bench(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ref_in(g, y, sc, sh, slope, g2=None):
    g, y, sc, sh = g.double(), y.double(), sc.double(), sh.double()
    if g2 is not None:
        g = g + 0.25 * torch.nn.functional.interpolate(g2.double(), scale_factor=2, mode="nearest")
    yh = y * sc[:, :, None, None] + sh[:, :, None, None]
    u = g * torch.where(yh >= 0, 1.0, slope)
    m1 = u.mean((2, 3), keepdim=True)
    m2 = (u * yh).mean((2, 3), keepdim=True)
    return sc[:, :, None, None] * (u - m1 - yh * m2)


def case_in(n, c, h, w, with_g2, min_hw, v):
    g = torch.randn(n, c, h, w, device=dev) * 1e-3
    y = torch.randn(n, c, h, w, device=dev)
    sc = torch.rand(n, c, device=dev) + 0.5
    sh = torch.randn(n, c, device=dev) * 0.3
    g2 = torch.randn(n, c, h // 2, w // 2, device=dev) * 1e-3 if with_g2 else None
    want = ref_in(g, y, sc, sh, 0.2, g2)
    outs = {}
    for name, on in (("multi", 0), ("cluster", 1)):
        tune(on=on, min_hw=min_hw, v=v)
        ops.AMAX.reset(dev)
        dy = torch.full_like(g, float("nan"))
        ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, g2=None if g2 is None else ops.full(g2))
        torch.cuda.synchronize()
        outs[name] = dy
    used = lib().query("san_act_bwd_in_sync_words", n, c, h * w)
    scale = want.abs().max().item()
    e_m = (outs["multi"].double() - want).abs().max().item() / scale
    e_c = (outs["cluster"].double() - want).abs().max().item() / scale
    return e_m, e_c, used


print("== InstanceNorm: error vs float64 (multi-launch, cluster), sync words")
for (n, c, h, w, g2, mn, v) in [(8, 18, 320, 320, False, 26625, 7), (8, 18, 320, 320, True, 26625, 7), (2, 3, 320, 320, False, 26625, 4),
                               (8, 36, 160, 160, False, 6401, 7), (8, 36, 160, 160, True, 6401, 4), (1, 5, 640, 368, False, 26625, 7),
                               (3, 7, 96, 72, True, 1025, 2), (2, 4, 48, 80, False, 513, 1)]:
    e_m, e_c, used = case_in(n, c, h, w, g2, mn, v)
    print(f"  {n}x{c}x{h}x{w} g2={int(g2)} v={v}: multi {e_m:.2e}  cluster {e_c:.2e}  words {used}", flush=True)
    assert used > 0 and e_c < 5e-6, (e_c, used)

print("== unshuffled / accumulated destinations: cluster == multi-launch")
for flags in ("unshuffle", "accumulate"):
    n, c, h, w = 4, 18, 320, 320
    g = torch.randn(n, c, h, w, device=dev)
    y = torch.randn(n, c, h, w, device=dev)
    sc = torch.rand(n, c, device=dev) + 0.5
    sh = torch.randn(n, c, device=dev) * 0.3
    res = []
    for on in (0, 1):
        tune(on=on, min_hw=26625, v=7)
        ops.AMAX.reset(dev)
        if flags == "unshuffle":
            dy = torch.zeros(n, 4 * c, h // 2, w // 2, device=dev)
            ops.act_bwd_ex(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, unshuffle=True)
        else:
            dy = torch.ones(n, c, h, w, device=dev)
            ops.act_bwd_ex(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, accumulate=True)
        torch.cuda.synchronize()
        res.append(dy)
    d = (res[0] - res[1]).abs().max().item() / res[0].abs().max().item()
    print(f"  {flags}: max rel diff {d:.2e}", flush=True)
    assert d < 5e-6

print("== BatchNorm: cluster vs three launches")
for (n, c, h, w) in [(8, 32, 320, 320), (8, 64, 160, 160), (8, 64, 80, 80), (8, 64, 20, 20), (3, 5, 40, 24)]:
    g = torch.randn(n, c, h, w, device=dev) * 1e-2
    y = torch.randn(n, c, h, w, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.2
    # the lazy affine of a training BN layer: yh = gamma * (y - mean) / sqrt(var + eps) + beta
    mean = y.mean((0, 2, 3))
    var = y.var((0, 2, 3), unbiased=False)
    sc1 = gamma / torch.sqrt(var + 1e-5)
    sc = sc1[None].repeat(n, 1).contiguous()
    sh = (beta - mean * sc1)[None].repeat(n, 1).contiguous()
    res = []
    for bn in (0, 1):
        tune(bn=bn)
        ops.AMAX.reset(dev)
        dy = torch.full_like(g, float("nan"))
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        ops.bn_act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.01), gamma, beta, dg, db, ops.full(dy))
        torch.cuda.synchronize()
        res.append((dy, dg, db))
    dd = [((a - b).abs().max() / a.abs().max()).item() for a, b in zip(res[0], res[1])]
    print(f"  {n}x{c}x{h}x{w}: dy {dd[0]:.2e} dgamma {dd[1]:.2e} dbeta {dd[2]:.2e}  words {lib().query('san_bn_act_bwd_sync_words', n, c, h * w)}", flush=True)
    assert max(dd) < 2e-5, dd

print("== determinism: 300 launches, identical bits")
n, c, h, w = 8, 18, 320, 320
g = torch.randn(n, c, h, w, device=dev)
y = torch.randn(n, c, h, w, device=dev)
sc = torch.rand(n, c, device=dev) + 0.5
sh = torch.randn(n, c, device=dev)
tune(on=1, min_hw=26625, v=7)
first = None
dy = torch.empty_like(g)
for it in range(300):
    ops.AMAX.reset(dev)
    ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True)
    if it % 50 == 0:
        torch.cuda.synchronize()
        cur = dy.clone()
        if first is None:
            first = cur
        assert torch.equal(first, cur) and torch.isfinite(cur).all()
print("  ok")

print("== timings (us per call)")
for (n, c, h, w) in [(8, 18, 320, 320), (8, 36, 160, 160), (8, 72, 80, 80)]:
    g = torch.randn(n, c, h, w, device=dev)
    y = torch.randn(n, c, h, w, device=dev)
    sc = torch.rand(n, c, device=dev) + 0.5
    sh = torch.randn(n, c, device=dev)
    dy = torch.empty_like(g)
    g2 = torch.randn(n, c, h // 2, w // 2, device=dev)
    ops.AMAX.reset(dev)
    row = []
    for name, on, mn, v in (("multi/plane", 0, 26625, 7), ("cluster v7", 1, 1025, 7), ("cluster v4", 1, 1025, 4), ("cluster v2", 1, 1025, 2)):
        tune(on=on, min_hw=mn, v=v)
        t = bench(lambda: ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True))
        t2 = bench(lambda: ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dy), instance_norm=True, g2=ops.full(g2)))
        row.append(f"{name} {t:.1f} / {t2:.1f}")
    gb = 3 * g.numel() * 4 / 1e9
    print(f"  IN {n}x{c}x{h}x{w} ({gb * 1e3:.0f} MB one pass): " + "; ".join(row), flush=True)
for (n, c, h, w) in [(8, 32, 320, 320), (8, 64, 160, 160), (8, 64, 80, 80), (8, 64, 20, 20)]:
    g = torch.randn(n, c, h, w, device=dev)
    y = torch.randn(n, c, h, w, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.2
    sc = torch.rand(n, c, device=dev) + 0.5
    sh = torch.randn(n, c, device=dev)
    dy = torch.empty_like(g)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    row = []
    for name, bn, v in (("three launches", 0, 7), ("cluster v7", 1, 7), ("cluster v4", 1, 4)):
        tune(bn=bn, v=v)
        t = bench(lambda: ops.bn_act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.01), gamma, beta, dg, db, ops.full(dy)))
        row.append(f"{name} {t:.1f}")
    print(f"  BN {n}x{c}x{h}x{w}: " + "; ".join(row), flush=True)

print("== staleness: two alternating input sets through ONE sync buffer, every launch checked against its own reference")
n, c, h, w = 2, 32, 160, 160
sets = []
for k in range(2):
    g = torch.randn(n, c, h, w, device=dev) * (1.0 + 3.0 * k)
    y = torch.randn(n, c, h, w, device=dev) + 0.5 * k
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.2
    sc = torch.rand(n, c, device=dev) + 0.5
    sh = torch.randn(n, c, device=dev)
    sets.append((g, y, gamma, beta, sc, sh))
tune(on=1, min_hw=1025, v=7, bn=0)
refs = []
for (g, y, gamma, beta, sc, sh) in sets:        # references from the three-launch / multi-launch forms
    dy = torch.empty_like(g)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.bn_act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.01), gamma, beta, dg, db, ops.full(dy))
    tune(on=0)
    dyi = torch.empty_like(g)
    ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dyi), instance_norm=True)
    tune(on=1, min_hw=1025, v=7)
    torch.cuda.synchronize()
    refs.append((dy.clone(), dg.clone(), db.clone(), dyi.clone()))
tune(bn=1)
worst = [0.0, 0.0, 0.0, 0.0]
for it in range(200):
    k = it & 1
    g, y, gamma, beta, sc, sh = sets[k]
    dy = torch.empty_like(g)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.bn_act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.01), gamma, beta, dg, db, ops.full(dy))
    dyi = torch.empty_like(g)
    ops.act_bwd(ops.full(g), ops.Act(y, 0, c, sc, sh, 0.2), ops.full(dyi), instance_norm=True)
    for j, (a, b) in enumerate(zip((dy, dg, db, dyi), refs[k])):
        worst[j] = max(worst[j], ((a - b).abs().max() / b.abs().max()).item())
print("  worst relative differences (BN dy, dgamma, dbeta, IN dy):", " ".join(f"{v:.2e}" for v in worst))
assert max(worst) < 2e-5
