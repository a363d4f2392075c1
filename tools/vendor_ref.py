#!/usr/bin/env python
"""Reference points from the vendor stack on the same box: what PyTorch-ROCm's own kernels (hipBLASLt matmul: tools/blaslt_ref.py; here the
flash-attention behind scaled_dot_product_attention, MIOpen's 3x3 convolutions, ATen's GroupNorm + SiLU) take for the problems of one UNet
evaluation at batch 8, replayed from a captured graph (eager torch is host-bound below ~19 us per op), hot operands, bf16, channels-last.
They answer "is 0.3 of the MFMA roof low for these shapes on this chip?" with the numbers of the kernels AMD ships -- no epilogue fusion on
their side (no bias / time-embedding / residual in the conv, GroupNorm and SiLU as two ops unless ATen fuses them).
Developer tool:  gpurun -- 'python tools/vendor_ref.py'   (prints one table; kbench's numbers for the same shapes are in profiles/)"""
import sys
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(reps):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        except Exception as e:           # (an op that cannot be captured: eager timing, marked)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return -(e0.elapsed_time(e1) * 1e3 / reps)


print("# attention: scaled_dot_product_attention, q/k/v [B*H? no: B, H, T, d] bf16;  us per call, algorithmic TFLOP/s (4 B H Nq Nk d)")
for (B, H, d, Nq, Nk, cnt) in [(8, 8, 40, 4096, 4096, 5), (8, 8, 40, 4096, 4126, 5), (8, 8, 80, 1024, 1024, 5), (8, 8, 80, 1024, 1054, 5),
                               (8, 8, 160, 256, 256, 5), (8, 8, 160, 256, 286, 5), (8, 8, 40, 4096, 77, 5), (8, 8, 80, 1024, 77, 5), (8, 8, 160, 256, 77, 5)]:
    q = torch.randn(B, H, Nq, d, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, Nk, d, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, H, Nk, d, device=dev, dtype=torch.bfloat16)
    us = timed(lambda: F.scaled_dot_product_attention(q, k, v))
    print(f"attn B{B} H{H} d{d} Nq{Nq} Nk{Nk} x{cnt}: {abs(us):9.1f} us{' (eager)' if us < 0 else ''}  {4.0 * B * H * Nq * Nk * d / abs(us) / 1e6:7.1f} TF/s", flush=True)

print("# conv 3x3 stride 1 pad 1, channels-last bf16 (MIOpen), no bias;  us per call, TFLOP/s (2 M N 9 Cin)")
for (B, HW, Cin, Cout, cnt) in [(8, 64, 320, 320, 7), (8, 64, 640, 320, 2), (8, 64, 960, 320, 1), (8, 32, 640, 640, 6), (8, 32, 1280, 640, 1), (8, 32, 1920, 640, 1),
                                (8, 16, 1280, 1280, 6), (8, 16, 2560, 1280, 2), (8, 8, 1280, 1280, 11), (8, 8, 2560, 1280, 3)]:
    x = torch.randn(B, Cin, HW, HW, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev, dtype=torch.bfloat16) * (9 * Cin) ** -0.5).contiguous(memory_format=torch.channels_last)
    us = timed(lambda: F.conv2d(x, w, None, 1, 1))
    print(f"conv B{B} {HW}x{HW} {Cin}->{Cout} x{cnt}: {abs(us):9.1f} us{' (eager)' if us < 0 else ''}  {2.0 * B * HW * HW * Cout * 9 * Cin / abs(us) / 1e6:7.1f} TF/s", flush=True)

print("# GroupNorm(32) + SiLU, channels-last bf16 (ATen native_group_norm + silu);  us per pair, GB/s over 2 x tensor bytes")
for (B, HW, C, cnt) in [(8, 64, 320, 13), (8, 32, 640, 11), (8, 16, 1280, 11), (8, 8, 1280, 12)]:
    x = torch.randn(B, C, HW, HW, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g, b = torch.ones(C, device=dev, dtype=torch.bfloat16), torch.zeros(C, device=dev, dtype=torch.bfloat16)
    us = timed(lambda: F.silu(F.group_norm(x, 32, g, b, 1e-5)))
    print(f"gn+silu B{B} {HW}x{HW} C{C} x{cnt}: {abs(us):9.1f} us{' (eager)' if us < 0 else ''}  {2.0 * B * HW * HW * C * 2 / abs(us) / 1e3:7.1f} GB/s", flush=True)
