"""FFT channeliser micro-benchmark: C channels of W/D bins on a resident block (HIP events on the launch stream)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sigdigger_amd import engine, synth

L = 1 << int(__import__('os').environ.get('ST_LOG2L', 22))
ctx = engine.Context(0)
x = torch.empty(L, dtype=torch.complex64, device="cuda")
torch.view_as_real(x).normal_()
import os
CASES = [(64, 64, int(os.environ.get('ST_RUN', 0)))] if os.environ.get('ST_ONE') else [(64, 64, 0), (64, 64, 2), (64, 64, 3), (64, 64, 4), (64, 32, 0), (128, 128, 0), (512, 512, 0), (16, 64, 0)] if os.environ.get('ST_WAVE') else [(64, 64, 8), (64, 64, 4), (64, 64, 3), (64, 64, 2), (64, 64, 1)] if os.environ.get('ST_QUICK') else [(64, 64, 0), (64, 64, 2), (64, 64, 3), (64, 64, 4), (16, 64, 0), (128, 128, 0), (512, 512, 0), (64, 32, 0), (64, 16, 0), (1, 1, 0)]
for C, D, run in CASES:
    st = engine.SpectTuner(ctx, 4096)
    if run:
        st.set_run(run)
    fn = synth.raster(C, float(os.environ.get('ST_SPACING', 1.8 / max(C, 2))))
    for f in fn:
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
    # ST_LAYOUT=tm (default): time-major output, what the inspector loops stream; cm: channel-major rows
    out = (engine.time_major(C, L // D + 64, "cuda") if os.environ.get('ST_LAYOUT', 'tm') == 'tm'
           else torch.empty((C, L // D + 64), dtype=torch.complex64, device="cuda"))
    st.feed(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        st.feed(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    alg = 8 * L + 8 * C * L / D
    print(f"C={C:4d} D={D:3d} run={run:3d}: {ms * 1e3:8.1f} us  {alg / ms / 1e6:8.1f} GB/s algorithmic  {L / ms / 1e3:8.1f} MS/s")
    st.close()
