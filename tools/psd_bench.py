"""PSD kernel timings: the 4 Mi-sample analyzer block (launch/occupancy-limited) and a 1 Gi-sample
capture (HBM-limited steady state)."""
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine
ctx = engine.Context(0)
big = len(sys.argv) > 1 and sys.argv[1] == "big"
for L in ((1 << 22, 1 << 28) if not big else (1 << 22, 1 << 30)):
    x = torch.empty(L, dtype=torch.complex64, device='cuda')
    torch.view_as_real(x).normal_()
    for n, navg in ((8192, 256), (8192, 1), (16384, 128), (4096, 512), (2048, 64), (1024, 16), (512, 16)):
        psd = engine.PSD(ctx, n)
        nf = L // n
        out = psd.feed(x, nframes=nf, navg=navg)
        torch.cuda.synchronize()
        reps = 20 if L <= (1 << 24) else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            psd.feed(x, nframes=nf, navg=navg, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = (8 * L + 4 * n * (nf // navg)) / ms / 1e6
        print(f"L=2^{L.bit_length()-1} N={n} navg={navg}: {ms*1e3:.1f} us  {gbs:.0f} GB/s ({gbs/80:.1f}% HBM)")
    del x
