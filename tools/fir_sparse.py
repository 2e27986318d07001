"""Why is chan_fir_kernel ~2.5x slower inside the pipeline than back to back?  Times single launches
(a) back to back, (b) 7 ms apart on an idle GPU, (c) 7 ms apart while a one-wavefront Costas kernel
runs on another stream, (d) like (a) but on a different input buffer every launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth

ctx = engine.Context(0)
L = 1 << 22
xs = [torch.randn(L, dtype=torch.complex64, device="cuda") for _ in range(12)]
fn = synth.raster(64, 0.0036)
bank = engine.ChannelBank(ctx, fn, 64, ctx.lpf_design(255, 0.75 / 64))
out = engine.time_major(64, L // 64 + 4, "cuda")
cos = engine.CostasBank(ctx, 64, 2, 0.0, 0.125, 3, 0.005)
y = engine.time_major(64, L // 64, "cuda")
y.copy_(torch.randn(64, L // 64, dtype=torch.complex64, device="cuda"))
z = engine.time_major(64, L // 64, "cuda")
side = torch.cuda.Stream()


def timed(x):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    bank.feed(x, out=out)
    e1.record()
    return e0, e1


def run(label, gap, busy, rotate):
    res = []
    for k in range(10):
        if busy:
            cos.feed(y, out=z, stream=side)
        if gap:
            time.sleep(gap)
        res.append(timed(xs[k % len(xs)] if rotate else xs[0]))
        torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in res][2:]
    print(f"{label}: avg {sum(ms)/len(ms)*1e3:.0f} us  min {min(ms)*1e3:.0f}  max {max(ms)*1e3:.0f}")


bank.feed(xs[0], out=out)
torch.cuda.synchronize()
run("(a) back to back, same buffer      ", 0, False, False)
run("(d) back to back, rotating buffers ", 0, False, True)
run("(b) 7 ms apart, idle GPU           ", 0.007, False, False)
run("(c) 7 ms apart, Costas running     ", 0.003, True, False)
run("(e) 7 ms apart, Costas, rotating   ", 0.003, True, True)

# (f)/(g): stream priorities
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)


def run_prio(label, fir_stream, cos_stream):
    res = []
    for k in range(10):
        cos.feed(y, out=z, stream=cos_stream)
        time.sleep(0.003)
        with torch.cuda.stream(fir_stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(fir_stream)
            bank.feed(xs[0], out=out, stream=fir_stream)
            e1.record(fir_stream)
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    ms = res[2:]
    print(f"{label}: avg {sum(ms)/len(ms)*1e3:.0f} us  min {min(ms)*1e3:.0f}  max {max(ms)*1e3:.0f}")


run_prio("(f) FIR on high-priority stream, Costas normal", hi, lo)
run_prio("(g) FIR normal, Costas on high-priority stream", lo, hi)
run_prio("(h) both on non-default normal streams        ", lo, side)
