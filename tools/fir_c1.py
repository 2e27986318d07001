"""255-tap polyphase FIR, few channels: time per feed by the kernel timer (dispatch-bound event pairs), 4 Mi and 16 Mi blocks.
   FIR_C (channels, default 1), FIR_D (decimation, default 16), FIR_T (taps, default 255)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth

ctx = engine.Context(0)
C, D, T = int(os.environ.get("FIR_C", 1)), int(os.environ.get("FIR_D", 16)), int(os.environ.get("FIR_T", 255))
taps = ctx.lpf_design(T, 0.75 / D)
for log2l in [int(v) for v in os.environ.get("FIR_LOG2L", "22,24").split(",")]:
    L = 1 << log2l
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_()
    bank = engine.ChannelBank(ctx, synth.raster(C, 0.25) if C > 1 else [0.25], D, taps)
    out = torch.empty((C, L // D + 8), dtype=torch.complex64, device="cuda")
    bank.feed(x, out=out)
    torch.cuda.synchronize()
    engine.kernel_timing(True)
    n = 20
    for _ in range(n):
        bank.feed(x, out=out)
    torch.cuda.synchronize()
    engine.kernel_timing(False)
    r = engine.kernel_timing_read()
    alg = 8 * L + 8 * C * L / D
    us = r["sum_ms"] / r["launches"] * 1e3
    print(f"C={C} D={D} T={T} L=2^{log2l}: {us:8.1f} us avg  (min {r['min_ms']*1e3:.1f})  {alg / us / 1e3:8.1f} GB/s algorithmic = {alg / us / 1e3 / 8000:.3f} of HBM peak  launches {r['launches']}")
