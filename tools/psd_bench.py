import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine
ctx = engine.Context(0)
L = 1 << 22
x = torch.randn(L, dtype=torch.complex64, device='cuda')
for n, navg in ((8192, 256), (8192, 1), (16384, 128), (4096, 512), (8192, 512)):
    psd = engine.PSD(ctx, n)
    nf = L // n
    out = psd.feed(x, nframes=nf, navg=navg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        psd.feed(x, nframes=nf, navg=navg, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"N={n} navg={navg}: {ms*1e3:.1f} us  {(8*L + 4*n*(nf//navg))/ms/1e6:.0f} GB/s")
