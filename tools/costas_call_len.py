"""The bank Costas kernel at different call lengths, back to back and with idle gaps between calls: ns per channel sample.
(Is the live analyzer's 91 ns per sample -- 8192-sample sub-ranges, a lightly loaded chip -- the gang kernel's own, a per-call
cost, or the clock the chip runs a lone wavefront at between bursts?)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigdigger_amd import engine, synth

ctx = engine.Context(0)
C, sps = 64, 15.625
for M, gap_ms in ((1 << 18, 0), (1 << 16, 0), (1 << 13, 0), (1 << 13, 2.0), (1 << 11, 0)):
    x = engine.time_major(C, M, "cuda")
    x.copy_(torch.from_numpy(np.stack([synth.psk_carriers(M, [0.001 * (c % 7)], sps=16, order=4, seed=c) for c in range(C)])).cuda())
    y = engine.time_major(C, M, "cuda")
    bank = engine.CostasBank(ctx, C, engine.COSTAS_QPSK, 0.0, 2.0 / sps, 3, 0.005)
    bank.feed(x, out=y)
    torch.cuda.synchronize()
    reps = max(4, (1 << 20) // M)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record()
        bank.feed(x, out=y)
        e1.record()
        if gap_ms:
            torch.cuda.synchronize()
            time.sleep(gap_ms * 1e-3)
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    print(f"costas bank 64 x {M} samples per call, {reps} calls, gap {gap_ms} ms: median {np.median(t) * 1e6 / M:.1f} ns per sample "
          f"(min {t.min() * 1e6 / M:.1f}, max {t.max() * 1e6 / M:.1f}); per call {np.median(t) * 1e3:.0f} us", flush=True)
