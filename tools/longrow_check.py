import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth
from oracle import sdo as sdo_mod
sdo = sdo_mod; sdo.lib()
ctx = engine.Context(0)
rng = np.random.default_rng(3)
def first_diff(a, b):
    a = a.view(np.uint32); b = b.view(np.uint32)
    n = min(a.size, b.size); d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) // 2, d.size) if d.size else None
for n in (1 << 18, 1 << 19, (1 << 20) - 1024, 1 << 20, 1 << 21):
    sps = 5.0
    x = synth.psk_carriers(n, [0.01], sps=int(sps), order=4, seed=5, snr_db=25).astype(np.complex64)
    for nch in (1, 2):
        xx = np.stack([x] * nch)
        agc = engine.AGCBank(ctx, nch, tau=sps)
        cos = engine.CostasBank(ctx, nch, 2, 0.0, 2.0 / sps, 3, 0.005)
        clk = engine.ClockBank(ctx, nch, 0.2, 1.0 / sps)
        y = torch.from_numpy(xx).cuda()
        tm = engine.time_major(nch, n, "cuda"); tm.copy_(y)
        a = agc.feed(tm, out=engine.time_major(nch, n, "cuda"))
        z = cos.feed(a, out=engine.time_major(nch, n, "cuda"))
        sym = torch.zeros((nch, n), dtype=torch.complex64, device="cuda"); cnt = torch.zeros(nch, dtype=torch.int32, device="cuda")
        clk.feed(z, sym, cnt)
        torch.cuda.synchronize()
        ra = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), x)
        rz = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, 2.0 / sps, 3, 0.005), ra)
        rs = sdo.clock_feed_bulk(sdo.clock_new(0.2, 1.0 / sps), rz)
        c = nch - 1
        print(n, nch, "agc", first_diff(a[c].cpu().numpy().copy(), ra), "costas", first_diff(z[c].cpu().numpy().copy(), rz),
              "clock", int(cnt[c]), rs.size, first_diff(sym[c, :int(cnt[c])].cpu().numpy().copy(), rs))
