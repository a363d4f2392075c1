#!/usr/bin/env python
"""Reference point for the row-GEMM class: what the vendor library (hipBLASLt / rocBLAS behind torch.nn.functional.linear, bf16)
achieves on the same problems, same box, same timing method as kbench (back-to-back launches of one shape, hot operands). The launches
are replayed from a captured graph (eager torch is host-bound below ~19 us per op). Plain C = A W^T: no bias / residual / head-layout /
GEGLU epilogue, so the library's numbers are a LOWER bound on what it would cost in place of gemm_u_kernel (whose epilogues are fused).
Under rocprofv3 --kernel-trace --stats the library's kernel names (macro tile, split) appear next to their durations.
Developer tool:  gpurun -- 'python tools/blaslt_ref.py tools/rowgemm.shapes'"""
import sys
import torch

dev = torch.device("cuda:0")
shapes = []
for line in open(sys.argv[1]):
    p = line.split()
    if p and p[0] == "gemm":
        shapes.append((int(p[1]), int(p[2]), int(p[3]), int(p[4]), int(p[5])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
print(f"{'M':>6s} {'N':>6s} {'K':>6s} {'epi':>3s} {'cnt':>3s} {'linear us':>10s} {'TF/s':>7s}")
tot = 0.0
for M, N, K, epi, cnt in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            torch.mm(x, w.t(), out=y)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                torch.mm(x, w.t(), out=y)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us * cnt / 1e3
    print(f"{M:6d} {N:6d} {K:6d} {epi:3d} {cnt:3d} {us:10.1f} {2.0 * M * N * K / us / 1e6:7.1f}", flush=True)
print(f"TOTAL ms (count-weighted, plain matmul): {tot:.3f}")
