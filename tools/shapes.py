"""Enumerate the kernel launches of one UNetModel.forward / AutoencoderKL.decode as the engine
schedules them (gligen_amd/csrc/engine.hip), grouped by distinct shape, for gligen_amd/build/kbench.

Line formats (count = launches per forward):
  gemm  M N K epi count        epi: 0 plain+bias(+res)  1 GEGLU  2 qk-heads  3 vt-heads (transposed launch)  4 fused q,k,v^T heads
  conv  B H W C0 C1 Cout stride ups count
  attn  B H d Nq Nk count
  gn    B HW C0 C1 silu count
  ln    B N1 N2 Tpad C count
"""
import sys
from collections import OrderedDict


def rup(a, b):
    return (a + b - 1) // b * b


class Rec:
    def __init__(self):
        self.d = OrderedDict()

    def add(self, *key):
        self.d[key] = self.d.get(key, 0) + 1

    def dump(self, f):
        for k, n in self.d.items():
            f.write(" ".join(str(x) for x in k) + f" {n}\n")


def unet(B=8, hw=64, Ng=30, mc=320, mult=(1, 2, 4, 4), nres=2, attn_res=(4, 2, 1), heads=8, ctx=77):
    r = Rec()

    def res(Cin0, Cin1, Cout, H):
        HW = H * H
        r.add("gn", B, HW, Cin0, Cin1, 1)
        # the engine normalises the concat into one tensor first, so the conv reads a single source
        r.add("conv", B, H, H, Cin0 + Cin1, 0, Cout, 1, 0)
        r.add("gn", B, HW, Cout, 0, 1)
        if Cin0 + Cin1 != Cout:
            r.add("gemm", B * HW, Cout, Cin0 + Cin1, 0)
        r.add("conv", B, H, H, Cout, 0, Cout, 1, 0)

    def st(C, H):
        HW = H * H
        M = B * HW
        d = C // heads
        Tp = rup(HW, 64)
        Tf = rup(HW + Ng, 64)
        r.add("gn", B, HW, C, 0, 0)
        r.add("gemm", M, C, C, 0)                 # proj_in
        r.add("ln", B, HW, 0, Tp, C)
        r.add("gemm", B * Tp, 3 * C, C, 4)        # attn1 q, k, v^T (one EPI_QKV_HEADS launch)
        r.add("attn", B, heads, d, HW, HW)
        r.add("gemm", M, C, C, 0)                 # attn1 out + res
        # (round 5: the grounding tokens' K / V are projected once per prompt -- no [x ; objs] LayerNorm pass, HW rows per sample)
        r.add("gemm", B * HW, 3 * C, C, 4)        # fuser q, k, v^T of the visual rows
        r.add("attn", B, heads, d, HW, HW + Ng)
        r.add("gemm", M, C, C, 0)
        r.add("ln", B, HW, 0, HW, C)
        r.add("gemm", M, 8 * C, C, 1)             # fuser GEGLU
        r.add("gemm", M, C, 4 * C, 0)
        r.add("ln", B, HW, 0, Tp, C)
        r.add("gemm", B * Tp, C, C, 2)            # attn2 q
        r.add("attn", B, heads, d, HW, ctx)
        r.add("gemm", M, C, C, 0)
        r.add("ln", B, HW, 0, HW, C)
        r.add("gemm", M, 8 * C, C, 1)
        r.add("gemm", M, C, 4 * C, 0)
        r.add("gemm", M, C, C, 0)                 # proj_out

    H = hw
    r.add("gemm", B * H * H, mc, 64, 0)           # first conv via small im2col
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            res(ch, 0, m * mc, H)
            ch = m * mc
            if ds in attn_res:
                st(ch, H)
            chans.append(ch)
        if level != len(mult) - 1:
            r.add("conv", B, H, H, ch, 0, ch, 2, 0)
            chans.append(ch)
            ds *= 2
            H //= 2
    res(ch, 0, ch, H)
    st(ch, H)
    res(ch, 0, ch, H)
    for level in reversed(range(len(mult))):
        m = mult[level]
        for i in range(nres + 1):
            ich = chans.pop()
            res(ch, ich, mc * m, H)
            ch = mc * m
            if ds in attn_res:
                st(ch, H)
            if level and i == nres:
                r.add("conv", B, H, H, ch, 0, ch, 1, 1)
                ds //= 2
                H *= 2
    r.add("gn", B, H * H, ch, 0, 1)
    r.add("conv", B, H, H, ch, 0, 32, 1, 0)
    return r


def vae(B=4, hw=64, chn=128, mult=(1, 2, 4, 4), nres=2):
    r = Rec()
    H = hw
    C = chn * mult[-1]
    r.add("gemm", B * H * H, C, 64, 0)

    def res(Cin, Cout, H):
        HW = H * H
        r.add("gn", B, HW, Cin, 0, 1)
        r.add("conv", B, H, H, Cin, 0, Cout, 1, 0)
        r.add("gn", B, HW, Cout, 0, 1)
        if Cin != Cout:
            r.add("gemm", B * HW, Cout, Cin, 0)
        r.add("conv", B, H, H, Cout, 0, Cout, 1, 0)

    res(C, C, H)
    HW = H * H
    r.add("gn", B, HW, C, 0, 0)
    r.add("gemm", B * HW, C, C, 0)
    r.add("gemm", B * HW, C, C, 0)
    for _ in range(B):
        r.add("gemm", C, HW, C, 0)
        r.add("gemm", HW, HW, C, 0)
        r.add("gemm", HW, C, HW, 0)
    r.add("gemm", B * HW, C, C, 0)
    res(C, C, H)
    for level in reversed(range(len(mult))):
        out = chn * mult[level]
        for _ in range(nres + 1):
            res(C, out, H)
            C = out
        if level:
            r.add("conv", B, H, H, C, 0, C, 1, 1)
            H *= 2
    r.add("gn", B, H * H, C, 0, 1)
    r.add("conv", B, H, H, C, 0, 32, 1, 0)
    return r


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "unet"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else (8 if which == "unet" else 4)
    (unet(B=B) if which == "unet" else vae(B=B)).dump(sys.stdout)
