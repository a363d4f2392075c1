#!/usr/bin/env python
"""The chip's dense bf16 MFMA ceiling with the clock that explains it: gl_mfma_calibrate over both MFMA shapes,
1 / 2 waves per SIMD, 4 / 8 independent accumulators, random and all-zero operands, each >= 20 ms; the shader clock is measured INSIDE the
loop by the kernel (s_memtime / s_memrealtime) and sampled from sysfs (pp_dpm_sclk, every 2 ms) while the loop runs.
    gpurun -- 'bash tools/gpu_run.sh calib'     -> gpurun_out/<out>/calib_mfma.txt"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import ClockSampler  # noqa: E402
from gligen_amd.engine import Engine  # noqa: E402


class FastSampler(ClockSampler):
    def run(self):
        while not self.stop_flag.is_set():
            for k, p in self.files.items():
                v = self._current(p)
                if v is not None:
                    self.samples[k].append(v)
            self.stop_flag.wait(0.002)


def main():
    eng = Engine(0, arena_gb=2.0)
    ms = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    print(f"# gl_mfma_calibrate, target {ms:.0f} ms per row; peak = 2.5 PFLOP/s dense bf16 = 256 CUs x 4 SIMDs x 2.4 GHz x 1024 FLOP/clk/SIMD")
    print(f"{'shape':9s} {'waves/SIMD':>10s} {'acc':>4s} {'data':>7s} {'TFLOP/s':>9s} {'of 2.5PF':>9s} {'sclk in loop':>13s} {'sysfs sclk min/mean/max':>24s} {'cyc/MFMA':>9s} {'ms':>7s}")
    rows = [(s, w, a, z) for s in (0, 1) for w in (1, 2) for a in (4, 8) for z in (False,)] + [(0, 2, 8, True), (1, 2, 8, True), (0, 1, 8, True)]
    for rep in range(2):          # twice: a cold chip clocks differently from one that has been running for a second
        for (s, w, a, z) in rows:
            smp = FastSampler(0)
            smp.start()
            time.sleep(0.004)
            r = eng.mfma_calibrate(s, w, a, z, ms)
            torch.cuda.synchronize()
            c = smp.summary() or {}
            sc = c.get("sclk_mhz")
            sysfs = f"{sc['min']:.0f}/{sc['mean']:.0f}/{sc['max']:.0f} ({sc['samples']})" if sc else "n/a"
            print(f"{r['shape']:9s} {r['waves_per_simd']:10d} {r['accumulators']:4d} {r['data']:>7s} {r['TFLOPs']:9.1f} {r['frac_of_2.5PF']:9.3f} "
                  f"{r['sclk_MHz']:10.0f} MHz {sysfs:>24s} {r['cycles_per_mfma']:9.2f} {r['ms']:7.2f}", flush=True)
        print("# ---- second pass (warm chip)" if rep == 0 else "# done")
    # what it implies: TFLOP/s = 2500 x (sclk / 2400) x (ideal cycles per MFMA / measured cycles per MFMA)


if __name__ == "__main__":
    main()
