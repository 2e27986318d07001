"""PSD of frames beyond the LDS (N = 32768 .. 1048576: fft.hip, Stockham passes through HBM) on a 2^28-sample capture."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine
ctx = engine.Context(0)
L = 1 << 28
x = torch.empty(L, dtype=torch.complex64, device='cuda'); torch.view_as_real(x).normal_()
for n in (16384, 32768, 65536, 262144, 1048576):
    psd = engine.PSD(ctx, n); nf = L // n
    navg = max(1, nf // 64)
    out = psd.feed(x, nframes=nf, navg=navg); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): psd.feed(x, nframes=nf, navg=navg, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    print(f"N={n} navg={navg}: {best*1e3:.1f} us {(8*L)/best/1e6:.0f} GB/s in ({(8*L)/best/1e6/80:.1f} % of HBM peak)")
