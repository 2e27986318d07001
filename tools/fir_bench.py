"""Isolated timing of the channel bank (translate + 255-tap polyphase FIR + decimate)."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth

ctx = engine.Context(0)
L = 1 << 22
x = torch.randn(L, dtype=torch.complex64, device="cuda")
for C, D in ((1, 16), (1, 64), (4, 16), (16, 64), (32, 64), (64, 64), (256, 64)):
    fn = synth.raster(C, 1.0 / (C + 1))
    bank = engine.ChannelBank(ctx, fn, D, ctx.lpf_design(255, 0.75 / D))
    out = engine.time_major(C, L // D + 4, "cuda")
    bank.feed(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        bank.feed(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = 8 * L + 8 * C * (L // D)
    fl = C * (L // D) * 255 * 8
    print(f"C={C} D={D}: {ms*1e3:.1f} us  {byt/ms/1e6:.0f} GB/s ({byt/ms/1e6/8000*100:.1f}% HBM)  {fl/ms/1e9:.1f} TF ({fl/ms/1e9/157.3*100:.1f}% FP32)")
