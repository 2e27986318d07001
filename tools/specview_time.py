"""SpectrumView.feed_sweep alone (512 dwells x 8192 bins) on an idle stream: events around the call."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sigdigger_amd import engine
ctx = engine.Context(0)
dwells, N = 512, 8192
frames = torch.randn((dwells, N), dtype=torch.float32, device="cuda")
fs, rel, f0 = 20e6, 0.5, 100e6
view = engine.SpectrumView(ctx)
view.set_range(f0, f0 + dwells * fs * rel); view.set_fft(fs, rel)
centers = f0 + (np.arange(dwells) + 0.5) * fs * rel
for _ in range(3): view.feed_sweep(frames, centers)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); view.feed_sweep(frames, centers); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("feed_sweep us: median %.1f min %.1f" % (np.median(ts), min(ts)))
ts = []
for k in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c = centers[k * 7 % dwells]
    e0.record(); view.feed(frames[k], c - fs / 2, c + fs / 2); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("single feed (feed_linear + interpolate) us: median %.1f min %.1f" % (np.median(ts), min(ts)))
