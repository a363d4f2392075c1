import torch, torch.nn.functional as F
from gligen_amd.engine import Engine
eng = Engine(0, arena_gb=2.0)
def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()
bf = lambda t: t.to(torch.bfloat16)
def rel_err(y, ref):
    y, ref = y.float(), ref.float()
    return ((y - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
C, M = 320, 256
x = bf(rnd(M, C, seed=1))
pre_w, pre_b = rnd(C, C, scale=C ** -0.5, seed=2), 0.1 * rnd(C, seed=3)
pre_res = bf(rnd(M, C, seed=4) * 1.3 + 0.2)
gamma, beta = 1.0 + 0.3 * rnd(C, seed=5), 0.2 * rnd(C, seed=6)
w1, b1 = rnd(8 * C, C, scale=C ** -0.5, seed=7), 0.5 * rnd(8 * C, seed=8)
w2, b2 = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=9), 0.1 * rnd(C, seed=10)
for pg, g in [(None, None), (0.37, None), (None, 0.6), (None, -0.6), (0.37, 0.6), (0.37, -0.6), (None, 1e-30), (None, 2.0), (None, 0.1)]:
    g1 = None if pg is None else torch.tensor([pg], device="cuda")
    g2 = None if g is None else torch.tensor([g], device="cuda")
    y = eng.op_ff_chain(x, pre_w, pre_b, pre_res, gamma, beta, w1, b1, w2, b2, pre_gate=g1, gate=g2)
    y2 = eng.op_ff_chain(x, pre_w, pre_b, pre_res, gamma, beta, w1, b1, w2, b2, pre_gate=g1, gate=g2)
    t = bf(pre_res.float() + (pg or 1.0) * (x.float() @ bf(pre_w).float().t() + pre_b)).float()
    h = F.layer_norm(t, (C,), gamma, beta, 1e-5) @ w1.t() + b1
    val, gg = h.chunk(2, dim=-1)
    ffv = (val * F.gelu(gg)) @ w2.t() + b2
    u = t + (1.0 if g is None else g) * ffv
    d = (y.float() - u)
    print(f"pre_gate {pg} gate {g}: rel_err {rel_err(y, u):.4f} deterministic {torch.equal(y, y2)}  max|ff| {ffv.abs().max():.2f} max|t| {t.abs().max():.2f} "
          f"err/ff corr {float((d * ffv).sum() / (ffv * ffv).sum()):.4f} err/t corr {float((d * t).sum() / (t * t).sum()):.4f}")

print("---- localisation: chain vs plain row-local kernel on the same t")
pg, g = None, None
y = eng.op_ff_chain(x, pre_w, pre_b, pre_res, gamma, beta, w1, b1, w2, b2)
t = bf(pre_res.float() + (x.float() @ bf(pre_w).float().t() + pre_b))
yp, _, used = eng.op_feedforward(t, w1, b1, w2, b2, gamma, beta, t, None)
d = (y.float() - yp.float()).abs()
print("used_rows", used, "max diff", float(d.max()), "n > 0.05:", int((d > 0.05).sum()), "of", d.numel())
bad = (d > 0.05)
rows = bad.any(dim=1).nonzero().flatten().tolist()
print("bad rows", len(rows), rows[:40])
cols = bad.any(dim=0).nonzero().flatten().tolist()
print("bad cols", len(cols), cols[:60])
r0 = rows[0] if rows else 0
print("row", r0, "bad cols", bad[r0].nonzero().flatten().tolist()[:40])
print("diff sample", d[r0, :16].tolist())
