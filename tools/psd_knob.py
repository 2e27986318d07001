import torch, sys, os
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine
ctx = engine.Context(0)
L = 1 << int(os.environ.get("LOG2L", "28"))
x = torch.empty(L, dtype=torch.complex64, device='cuda'); torch.view_as_real(x).normal_()
for n, navg in ((8192, 256), (8192, L // 8192), (16384, 128)):
    psd = engine.PSD(ctx, n); nf = L // n
    out = psd.feed(x, nframes=nf, navg=navg); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): psd.feed(x, nframes=nf, navg=navg, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(os.environ.get("SUAMD_PSD_SPLIT_TARGET"), os.environ.get("SUAMD_PSD_MIN_FRAMES"), f"N={n} navg={navg}: {ms*1e3:.1f} us {(8*L)/ms/1e6:.0f} GB/s")
