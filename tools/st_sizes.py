"""Narrow channel sizes (8 .. 64 bins, one response) on a 4 Mi-sample block: kernel time by the kernel timer."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
L = 1 << 22
x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
for C, D in ((64, 64), (128, 128), (256, 256), (512, 512), (64, 128), (64, 512)):
    st = engine.SpectTuner(ctx, 4096)
    for f in synth.raster(C, 1.8 / max(C, 2)):
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
    out = engine.time_major(C, L // D + 64, "cuda")
    st.feed(x, out=out); torch.cuda.synchronize()
    engine.kernel_timing_read(); engine.kernel_timing(True)
    for _ in range(20):
        st.feed(x, out=out)
    torch.cuda.synchronize(); engine.kernel_timing(False)
    r = engine.kernel_timing_read()
    us = r["sum_ms"] / r["launches"] * 1e3
    alg = 8 * L + 8 * C * L / D
    print(f"C={C:4d} D={D:4d} ({4096 // D:2d} bins): {us:6.1f} us  {alg / us / 1e3:7.1f} GB/s")
    st.close()
