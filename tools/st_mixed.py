"""A bank of 64-bin channels with several pass-band widths (several responses) on a 4 Mi-sample block, by the kernel timer."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
L = 1 << 22
x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
for nw in (1, 2, 5, 8, 10, 12, 20):
    widths = np.linspace(0.4, 0.9, nw) if nw > 1 else [0.75]
    st = engine.SpectTuner(ctx, 4096)
    for c, f in enumerate(synth.raster(64, 1.8 / 64)):
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * float(widths[c % nw]) / 64)
    out = engine.time_major(64, L // 64 + 64, "cuda")
    st.feed(x, out=out); torch.cuda.synchronize()
    engine.kernel_timing_read(); engine.kernel_timing(True)
    for _ in range(20):
        st.feed(x, out=out)
    torch.cuda.synchronize(); engine.kernel_timing(False)
    r = engine.kernel_timing_read()
    print(f"{nw:2d} pass-band widths: {r['sum_ms'] / r['launches'] * 1e3:6.1f} us")
    st.close()
