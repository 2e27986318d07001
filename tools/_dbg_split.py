import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from sigdigger_amd import engine
import test_gpu_specttuner as T
ctx = engine.Context(0)
H = T.H
x = T.cnoise(H * 48, 5)
chans = [(0.3, 2 * np.pi / 64 * 0.7, 1.0, False), (2.1, 2 * np.pi / 64 * 0.5, 1.0, True), (4.0, 2 * np.pi / 16 * 0.8, 1.0, False)]
one = T.run_gpu(ctx, x, chans)
for splits, run in (([H], 8), ([H * 3, H * 4, H * 30], 8), ([H * 10], 3), ([], 1), ([H * 7], 64), ([], 2), ([], 3), ([], 47)):
    parts = T.run_gpu(ctx, x, chans, splits, run)
    for c, (a, b) in enumerate(zip(one, parts)):
        bad = np.nonzero(a.view(np.uint32).reshape(-1, 2) != b.view(np.uint32).reshape(-1, 2))[0]
        if a.size != b.size or bad.size:
            hs = a.size // 47
            print(splits, run, "chan", c, "sizes", a.size, b.size, "nbad", bad.size, "first", bad[:6], "blocks", sorted(set((bad // hs).tolist()))[:12],
                  "maxdiff", float(np.max(np.abs(a - b))) if a.size == b.size else None)
print("done")
