"""bank vs gang on identical work: 64 channels x 65536 samples (Costas QPSK, AGC, Gardner)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
C_, M = 64, 65536
x = synth.psk_carriers(M, [0.0], sps=16, order=4, seed=2)
rows = np.stack([np.roll(x, 97 * c) for c in range(C_)]).astype(np.complex64)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
xt = engine.time_major(C_, M, "cuda"); xt.copy_(torch.from_numpy(rows).cuda())
yt = engine.time_major(C_, M, "cuda")
cb = engine.CostasBank(ctx, C_, 2, 0.0, 0.125, 3, 0.005)
print("costas bank  %.2f ms" % timeit(lambda: cb.feed(xt, out=yt)))
ab = engine.AGCBank(ctx, C_, tau=16.0)
print("agc bank     %.2f ms" % timeit(lambda: ab.feed(xt, out=yt)))
kb = engine.ClockBank(ctx, C_, 0.2, 1 / 16)
sym = torch.zeros((C_, M), dtype=torch.complex64, device="cuda"); cnt = torch.zeros(C_, dtype=torch.int32, device="cuda")
def clk_bank():
    cnt.zero_(); kb.feed(xt, sym, cnt)
print("clock bank   %.2f ms" % timeit(clk_bank))
xs = [torch.from_numpy(rows[c]).cuda() for c in range(C_)]
ys = [torch.empty_like(v) for v in xs]
cg = [engine.CostasBank(ctx, 1, 2, 0.0, 0.125, 3, 0.005) for _ in range(C_)]
print("costas gang  %.2f ms" % timeit(lambda: engine.gang_costas(ctx, cg, xs, ys)))
ag = [engine.AGCBank(ctx, 1, tau=16.0) for _ in range(C_)]
print("agc gang     %.2f ms" % timeit(lambda: engine.gang_agc(ctx, ag, xs, ys)))
kg = [engine.ClockBank(ctx, 1, 0.2, 1 / 16) for _ in range(C_)]
syms = [torch.zeros(M, dtype=torch.complex64, device="cuda") for _ in range(C_)]
cnts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(C_)]
def clk_gang():
    for c in cnts: c.zero_()
    engine.gang_clock(ctx, kg, xs, syms, cnts)
print("clock gang   %.2f ms" % timeit(clk_gang))
# the same items as columns of a time-major slab (round 6): no gather / scatter, the recurrence streams the slab
Y = torch.zeros((M + 128, C_), dtype=torch.complex64, device="cuda"); Y[:M] = torch.from_numpy(np.ascontiguousarray(rows.T)).cuda()
A = torch.zeros_like(Y); Z = torch.zeros_like(Y)
cols = list(range(C_))
cs = [engine.CostasBank(ctx, 1, 2, 0.0, 0.125, 3, 0.005) for _ in range(C_)]
print("costas slab  %.2f ms" % timeit(lambda: engine.gang_costas_slab(ctx, cs, Y, cols, Z, cols, [M] * C_)))
work = torch.zeros(2 * (M + 128) * C_, dtype=torch.float32, device="cuda")
asl = [engine.AGCBank(ctx, 1, tau=16.0) for _ in range(C_)]
print("agc slab     %.2f ms" % timeit(lambda: engine.gang_agc_slab(ctx, asl, Y, cols, A, cols, [M] * C_, work, parts=1)))
ks = [engine.ClockBank(ctx, 1, 0.2, 1 / 16) for _ in range(C_)]
def clk_slab():
    for c in cnts: c.zero_()
    engine.gang_clock_slab(ctx, ks, Y, cols, [M] * C_, syms, cnts)
print("clock slab   %.2f ms" % timeit(clk_slab))
Y2 = torch.zeros((M + 128, 2 * C_), dtype=torch.complex64, device="cuda"); Y2[:M, ::2] = Y[:M]
Z2 = torch.zeros_like(Y2)
cols2 = list(range(0, 2 * C_, 2))
cs2 = [engine.CostasBank(ctx, 1, 2, 0.0, 0.125, 3, 0.005) for _ in range(C_)]
print("costas slab, pitch 128 (run-time pitch loop)  %.2f ms" % timeit(lambda: engine.gang_costas_slab(ctx, cs2, Y2, cols2, Z2, cols2, [M] * C_)))
