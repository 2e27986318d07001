"""Does a resident one-wavefront kernel slow down other kernels on MI355X?  Times a memory-bound torch
copy, the PSD kernel and the FIR bank alone and next to a running Costas (1 wave) kernel."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth

ctx = engine.Context(0)
L = 1 << 22
x = torch.randn(L, dtype=torch.complex64, device="cuda")
big = torch.randn(1 << 27, device="cuda")           # 512 MB
big2 = torch.empty_like(big)
bank = engine.ChannelBank(ctx, synth.raster(64, 0.0036), 64, ctx.lpf_design(255, 0.75 / 64))
out = engine.time_major(64, L // 64 + 4, "cuda")
psd = engine.PSD(ctx, 8192)
pout = torch.empty((512, 8192), device="cuda")
cos = engine.CostasBank(ctx, 64, 2, 0.0, 0.125, 3, 0.005)
y = engine.time_major(64, L // 64, "cuda"); y.copy_(torch.randn(64, L // 64, dtype=torch.complex64, device="cuda"))
z = engine.time_major(64, L // 64, "cuda")
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")

ops = {"copy 512MB": lambda: big2.copy_(big), "psd 512x8192": lambda: psd.feed(x, nframes=512, navg=1, out=pout),
       "fir C=64": lambda: bank.feed(x, out=out), "sgemm 4096": lambda: torch.mm(a, b)}
for name, fn in ops.items():
    fn(); torch.cuda.synchronize()
    for busy in (False, True):
        ts = []
        for k in range(6):
            if busy:
                cos.feed(y, out=z, stream=side)
                time.sleep(0.002)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts = ts[1:]
        print(f"{name:14s} {'next to 1-wave kernel' if busy else 'alone                '}: {sum(ts)/len(ts)*1e3:8.0f} us")
