import os, sys, torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine
ctx = engine.Context(0)
for lg in (22, 23, 24, 25):
    L = 1 << lg
    x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
    for n in (8192, 16384, 4096):
        psd = engine.PSD(ctx, n); nf = L // n; navg = min(256, nf)
        out = psd.feed(x, nframes=nf, navg=navg); torch.cuda.synchronize()
        engine.kernel_timing_read(); engine.kernel_timing(True)
        for _ in range(10): psd.feed(x, nframes=nf, navg=navg, out=out)
        torch.cuda.synchronize(); engine.kernel_timing(False)
        a = engine.kernel_timing_read("psd_kernel"); b = engine.kernel_timing_read("psd_reduce_kernel")
        print(f"L=2^{lg} N={n}: psd_kernel {a['sum_ms']/max(a['launches'],1)*1e3:6.1f} us + reduce {b['sum_ms']/max(b['launches'],1)*1e3:5.1f} us")
