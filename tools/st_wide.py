"""Wide channel sizes (128 .. 4096 bins: the workgroup kernel of specttuner.hip) on a 4 Mi-sample block, by the kernel timer."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
L = 1 << 22
x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
for C, D in ((1, 16), (16, 16), (32, 16), (64, 16), (128, 16), (1, 32), (32, 32), (64, 32), (128, 32), (16, 8), (1, 1), (1, 4)):
    st = engine.SpectTuner(ctx, 4096)
    for f in synth.raster(C, 1.8 / max(C, 2)):
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
    out = engine.time_major(C, L // D + 64, "cuda")
    st.feed(x, out=out); torch.cuda.synchronize()
    engine.kernel_timing_read(); engine.kernel_timing(True)
    for _ in range(10):
        st.feed(x, out=out)
    torch.cuda.synchronize(); engine.kernel_timing(False)
    r = engine.kernel_timing_read()
    us = r["sum_ms"] / 10 * 1e3
    alg = 8 * L + 8 * C * L / D
    print(f"C={C:3d} D={D:3d} ({4096 // D:4d} bins): {us:7.1f} us per feed ({r['launches']} launches)  {alg / us / 1e3:7.1f} GB/s = {alg / us / 8e6:.3f} of HBM peak")
    st.close()
