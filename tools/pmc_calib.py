"""Known-byte-count launches in the access widths our kernels use, to calibrate rocprofv3's
FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md section HBM: FETCH_SIZE reads 1/2 for 16 B/lane streams;
other widths uncalibrated).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine

ctx = engine.Context(0)
L = 1 << 25                                   # 256 MiB of SUCOMPLEX: past the 256 MiB Infinity Cache with in+out
x = torch.randn(L, dtype=torch.complex64, device="cuda")
y = torch.empty_like(x)
for _ in range(3):
    ctx.xlate(x, 0, 12345, 0, out=y)          # 16 B/lane loads and stores: 8L bytes in, 8L bytes out
    ctx.quad_demod(x.unsqueeze(0), out=y.unsqueeze(0))   # 8 B/lane loads (x2: p and p-1) and stores
torch.cuda.synchronize()
print("bytes per launch: in", 8 * L, "out", 8 * L)
