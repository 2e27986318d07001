"""PLL bank / gang timing: 64 channels x 65536 samples."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
C_, M = 64, 65536
x = synth.psk_carriers(M, [0.01], sps=16, order=2, seed=2)
rows = np.stack([np.roll(x, 97 * c) for c in range(C_)]).astype(np.complex64)
xt = engine.time_major(C_, M, "cuda"); xt.copy_(torch.from_numpy(rows).cuda())
yt = engine.time_major(C_, M, "cuda")
pb = engine.PLLBank(ctx, C_, 0.0, 0.01)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("pll bank  %.2f ms" % timeit(lambda: pb.feed(xt, out=yt)))
