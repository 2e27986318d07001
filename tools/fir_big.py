import os, sys, torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
for lg in (22, 24, 26, 28):
    L = 1 << lg
    x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
    for C, D in ((1, 16), (1, 64), (4, 16)):
        bank = engine.ChannelBank(ctx, synth.raster(C, 1.0 / (C + 1)), D, ctx.lpf_design(255, 0.75 / D))
        out = engine.time_major(C, L // D + 4, "cuda")
        bank.feed(x, out=out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): bank.feed(x, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        byt = 8 * L + 8 * C * (L // D)
        print(f"L=2^{lg} C={C} D={D}: {ms*1e3:.1f} us {byt/ms/1e6:.0f} GB/s ({byt/ms/1e6/80:.1f}% HBM)")
    del x
