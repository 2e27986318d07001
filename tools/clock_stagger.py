"""The bank clock kernel (suamd_clock_bank_feed) on 64 channels whose symbol clocks are aligned / staggered: ns per channel
sample.  SUAMD_CLOCK_MODE picks the kernel's schedule (read once per process)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sigdigger_amd import engine

ctx = engine.Context(0)
C, M, sps = 64, 1 << 18, 15.625
rng = np.random.default_rng(3)


def rows(stagger):
    t = np.arange(M)
    out = np.empty((C, M), np.complex64)
    for c in range(C):
        off = rng.random() * sps if stagger else 0.0
        s = sps * (1 + (2 * rng.random() - 1) * 1e-4) if stagger else sps
        sym = rng.integers(0, 4, M // 15 + 4)
        out[c] = np.exp(1j * (np.pi / 2 * sym[np.floor((t + off) / s).astype(np.int64)] + np.pi / 4)).astype(np.complex64)
    out += (0.05 * (rng.standard_normal((C, M)) + 1j * rng.standard_normal((C, M)))).astype(np.complex64)
    return out


for name, stagger in (("aligned", False), ("staggered", True)):
    z = engine.time_major(C, M, "cuda")
    z.copy_(torch.from_numpy(rows(stagger)).cuda())
    bank = engine.ClockBank(ctx, C, 0.2, 1.0 / sps)
    sym = torch.zeros((C, M), dtype=torch.complex64, device="cuda")
    cnt = torch.zeros(C, dtype=torch.int32, device="cuda")
    bank.feed(z, sym, cnt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 4
    e0.record()
    for _ in range(reps):
        cnt.zero_()
        bank.feed(z, sym, cnt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"clock bank mode={os.environ.get('SUAMD_CLOCK_MODE', 'default')} {name}: {ms:.3f} ms per {C} x {M} samples = {ms * 1e6 / M:.1f} ns per sample; symbols {int(cnt.sum())}", flush=True)
