"""FIR variants next to a resident one-wavefront kernel (see corun.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
L = 1 << 22
x = torch.randn(L, dtype=torch.complex64, device="cuda")
cos = engine.CostasBank(ctx, 64, 2, 0.0, 0.125, 3, 0.005)
y = engine.time_major(64, L // 64, "cuda"); z = engine.time_major(64, L // 64, "cuda")
side = torch.cuda.Stream()
for C, D in ((64, 64), (16, 64), (4, 64), (1, 64), (1, 16)):
    bank = engine.ChannelBank(ctx, synth.raster(C, 0.0036), D, ctx.lpf_design(255, 0.75 / D))
    out = engine.time_major(C, L // D + 4, "cuda")
    bank.feed(x, out=out); torch.cuda.synchronize()
    r = []
    for busy in (False, True):
        ts = []
        for k in range(6):
            if busy:
                cos.feed(y, out=z, stream=side); time.sleep(0.002)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); bank.feed(x, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        r.append(sum(ts[1:]) / 5 * 1e3)
    print(f"C={C:3d} D={D:3d} env={os.environ.get('SUAMD_FIR_SMEM_TAPS','-')}/{os.environ.get('SUAMD_FIR_NOUT','-')}: alone {r[0]:6.0f} us, co-run {r[1]:6.0f} us")
