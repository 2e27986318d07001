"""Phase clock stamps of one PSD wavefront (library built with SUAMD_BUILD_DEFS=-DPSD_TSTAMP)."""
import torch, sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
from sigdigger_amd import engine, lib as _lib
ctx = engine.Context(0)
L = 1 << 28
x = torch.empty(L, dtype=torch.complex64, device='cuda'); torch.view_as_real(x).normal_()
n = int(os.environ.get("PSD_N", "8192")); navg = 256
psd = engine.PSD(ctx, n); nf = L // n
out = psd.feed(x, nframes=nf, navg=navg); torch.cuda.synchronize()
psd.feed(x, nframes=nf, navg=navg, out=out); torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 512)()
print("rc", lib.suamd_debug_psd_ts(buf))
t = np.array(buf[:], dtype=np.int64).reshape(64, 8)
names = ["wait nxt", "window", "pass0", "-", "pass1+req", "pass2+req"]
t[:, 4] = t[:, 3]; d = np.diff(t[:, :7], axis=1)
for f in (0, 1, 2, 8, 16, 32, 48, 63):
    print(f, " ".join(f"{nm}={v}" for nm, v in zip(names, d[f])), "frame->frame", t[f, 0] - t[f - 1, 0] if f else 0)
print("mean (frames 8..63):", " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, d[8:].mean(axis=0))), "period", np.diff(t[8:, 0]).mean())
