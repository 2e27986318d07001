"""Channeliser launch time against the run length at a fixed number of workgroups: slope = time per window, intercept =
everything else (dispatch, first samples, seam, drain)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
for nwg in (256, 512, 683, 1024):
    row = []
    for R in (2, 3, 4, 6, 8):
        L = nwg * R * 2048
        x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
        st = engine.SpectTuner(ctx, 4096); st.set_run(R)
        for f in synth.raster(64, 1.8 / 64):
            st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / 64)
        out = engine.time_major(64, L // 64 + 64, "cuda")
        st.feed(x, out=out); torch.cuda.synchronize()
        engine.kernel_timing(True)
        for _ in range(20):
            st.feed(x, out=out)
        torch.cuda.synchronize()
        engine.kernel_timing(False)
        r = engine.kernel_timing_read()
        row.append(r["sum_ms"] / r["launches"] * 1e3)
        st.close(); del x, out
    slope = (row[-1] - row[0]) / 6
    print(f"{nwg:5d} workgroups: " + " ".join(f"R={R}: {v:6.1f}" for R, v in zip((2, 3, 4, 6, 8), row)) + f" us | per window {slope:.2f} us, intercept {row[0] - 2 * slope:.1f} us")
