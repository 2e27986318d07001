"""Is a kernel slower when the chip idled before it?  The headline's channeliser (64 channels x 64 bins, 16 Mi-sample block) and the
8192-point PSD by the kernel timer: launched back to back, with 20 ms of idle chip before every launch, and with 20 ms of a
single busy wavefront (what the pipeline's recurrence stages leave the chip doing) before and during every launch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth

ctx = engine.Context(0)
L = 1 << 24
x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
st = engine.SpectTuner(ctx, 4096)
for f in synth.raster(64, 1.8 / 64):
    st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / 64)
out = engine.time_major(64, L // 64 + 64, "cuda")
psd = engine.PSD(ctx, 8192)
pout = psd.feed(x, nframes=L // 8192, navg=256)
st.feed(x, out=out); torch.cuda.synchronize()
# a serial kernel of ~20 ms: one Costas wavefront over 64 channels
nser = 1 << 18
cos = engine.CostasBank(ctx, 64, 2, 0.0, 0.25, 3, 0.005)
zin = engine.time_major(64, nser, "cuda"); torch.view_as_real(zin).normal_()
zout = engine.time_major(64, nser, "cuda")
side = torch.cuda.Stream()

for mode in ("back to back", "idle", "serial", "back to back"):
    engine.kernel_timing_read(); engine.kernel_timing(True)
    for _ in range(12):
        if mode == "idle":
            torch.cuda.synchronize(); time.sleep(0.02)
        elif mode == "serial":
            torch.cuda.synchronize()
            cos.feed(zin, out=zout, stream=side)
            time.sleep(0.005)
        st.feed(x, out=out)
        psd.feed(x, nframes=L // 8192, navg=256, out=pout)
    torch.cuda.synchronize(); engine.kernel_timing(False)
    rs = engine.kernel_timing_read("stp_kernel"); rp = engine.kernel_timing_read("psd_kernel")
    f = lambda r: f"{r['sum_ms'] / max(r['launches'], 1) * 1e3:7.1f} us avg (min {r['min_ms'] * 1e3:.1f}, max {r['max_ms'] * 1e3:.1f}, {r['launches']} launches)"
    print(f"{mode:13s}: stp_kernel {f(rs)}   psd_kernel {f(rp)}")
