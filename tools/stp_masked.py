"""stp_kernel alone, 64 x 64 bins, 16 Mi samples: launch plans on an unmasked stream and on the partition's transform stream
    python tools/stp_masked.py   (GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sigdigger_amd import engine, pipeline, synth

torch.cuda.init()
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
n = ctx.cu_count()
Lb, D = 1 << int(os.environ.get("LOG2", "24")), 64
fn = synth.raster(64, 2 * 90e3 / 50e6)
x = torch.empty(Lb, dtype=torch.complex64, device=dev)
torch.view_as_real(x).normal_()
nbytes = 8.0 * Lb + 8.0 * 64 * Lb / D


def run(tag, slots, stream):
    st = engine.SpectTuner(ctx, 4096)
    st.set_slots(slots)
    for f in fn:
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
    out = engine.time_major(64, Lb // D + 64, dev)
    with torch.cuda.stream(stream):
        st.feed(x, out=out)
        torch.cuda.synchronize(dev)
        engine.kernel_timing_read()
        engine.kernel_timing(True)
        for _ in range(12):
            st.feed(x, out=out)
        torch.cuda.synchronize(dev)
        engine.kernel_timing(False)
    r = engine.kernel_timing_read()
    ms = r["sum_ms"] / max(r["launches"], 1)
    st.close()
    print(f"{tag:44s} slots {slots:5d}: {ms * 1e3:7.1f} us (min {r['min_ms'] * 1e3:.1f} max {r['max_ms'] * 1e3:.1f})  frac {nbytes / ms / 1e6 / 8000:.3f}", flush=True)


cur = torch.cuda.current_stream()
for s in (768, 1024):
    run("unmasked", s, cur)
for per in (1, 2):
    res, tr = pipeline.cu_partition(n, per)
    ms = ctx.masked_stream(tr)
    for s in (4 * len(tr), 4 * len(tr) - 32, 3 * len(tr), 768):
        run(f"masked, {per} CU/XCD reserved ({len(tr)} CUs)", s, ms)
# a mask that takes the CUs from all four shader engines of an XCD alike is impossible with 1 or 2 CUs; 4 per XCD (one per SE):
res = [i for i in range(n) if (i // 8) % 8 == 0][:32]          # bits 0..7, 64..71, 128..135, 192..199?  (probe says bit 8 -> another SE)
