"""HIP path against the COMMITTED fixtures (tests/golden/oracle_v1.npz, oracle_v2.npz, oracle_v3.npz) -- not against a live
oracle run: inputs and expected outputs both come out of the .npz files.  Bit exact for the inspector chain and
the integer / byte work, the FFT-based entries within the stated tolerance."""
import os

import numpy as np
import pytest
import torch

from sigdigger_amd import engine

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def same_bits(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.array_equal(a, b), f"{what}: {int(np.sum(a != b))} of {a.size} values differ"      # -0 == +0 accepted


def rows1(x):
    return dev(x).reshape(1, -1)


def test_v1_chain(ctx):
    g = np.load(os.path.join(GOLDEN, "oracle_v1.npz"))
    x = g["input_iq"]
    psd = engine.PSD(ctx, 1024, engine.WINDOW_BLACKMANN_HARRIS)
    got = host(psd.feed(dev(x), nframes=4, navg=2, scale=1.0 / 1024))
    ref = g["psd_bh_1024"]
    assert np.max(np.abs(got - ref) / np.max(ref, axis=1, keepdims=True)) < 1e-5
    assert np.max(np.abs(host(ctx.psd_shift_db(dev(ref[:1].copy())))[0] - g["psd_shift_db"])) < 5e-5
    dp = ctx.fnor_to_dphase(-0.2)
    same_bits(host(ctx.xlate(dev(x), 7, dp, 1000)), g["xlate"], "xlate")
    taps = ctx.lpf_design(63, 0.1)
    same_bits(taps, g["taps"], "lpf taps")
    bank = engine.ChannelBank(ctx, [0.2], 8, taps)
    y = bank.feed(dev(x))
    same_bits(host(y)[0], g["chan_D8"], "channel bank")
    same_bits(host(ctx.quad_demod(y))[0], g["quad"], "quad demod")
    same_bits(host(ctx.delayed_conj(y[0].contiguous(), 3)), g["delayed_conj"], "delayed conj")
    a = engine.AGCBank(ctx, 1, tau=2.0).feed(y)
    same_bits(host(a)[0], g["agc"], "agc")
    cb = engine.CostasBank(ctx, 1, 2, 0.0, 1.0, 3, 0.02)
    z = cb.feed(a)
    same_bits(host(z)[0], g["costas_qpsk"], "costas")
    om, ph = cb.state()
    assert ph[0] == g["costas_state"][0] and om.view(np.uint32)[0] == g["costas_state"][1]
    same_bits(host(engine.PLLBank(ctx, 1, 0.0, 0.05).feed(y))[0], g["pll"], "pll")
    clk = engine.ClockBank(ctx, 1, 0.5, 0.5)
    sym = torch.zeros((1, z.shape[1] + 1), dtype=torch.complex64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    clk.feed(z, sym, cnt)
    same_bits(host(sym[0, :int(cnt.cpu()[0])]), g["gardner"], "gardner")


def test_v2_stages(ctx):
    g = np.load(os.path.join(GOLDEN, "oracle_v2.npz"))
    x = g["input_iq"]
    dx = dev(x)
    same_bits(host(ctx.sample_zero_crossing(dx, 1.0 / 8, 0, False, 0.05 + 0j, 1 + 0j)), g["zc_amplitude"], "zc amplitude")
    same_bits(host(ctx.sample_zero_crossing(dx, 1.0 / 8, 1, False, 0j, -1j)), g["zc_phase"], "zc phase")
    same_bits(host(ctx.sample_zero_crossing(dx, 1.0 / 8, 2)), g["zc_frequency"], "zc frequency")
    same_bits(host(ctx.conj_prev(dev(x[:2048]), 0.5 - 0.25j)), g["conj_prev"], "conj prev")
    same_bits(host(ctx.sample_manual(dev(x[:4096]), 500.0, 3, 0)), g["manual_amp"], "manual sampler (amplitude)")
    same_bits(host(ctx.sample_manual(dev(x[:4096]), 500.0, 3, 2)), g["manual_freq"], "manual sampler (frequency)")
    h = ctx.rrc_design(8.0, 0.35)
    same_bits(h, g["rrc_taps"], "rrc taps")
    m = engine.FIRBank(ctx, 1, h).feed(rows1(x[:3000]))
    same_bits(host(m)[0], g["matched"], "matched filter")
    clk = engine.ClockBank(ctx, 1, 0.0, 1.0 / 8)
    clk.set_phase(float(np.float32(0.5) * np.float32(0.3)))
    sym = torch.zeros((1, 3001), dtype=torch.complex64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    clk.feed(m, sym, cnt)
    n = int(cnt.cpu()[0])
    same_bits(host(sym[0, :n]), g["manual_clock"], "fixed-baud sampler")
    cma = engine.CMABank(ctx, 1, 8, 2e-3)
    same_bits(host(cma.feed(sym, count=cnt))[0, :n], g["cma"], "cma")
    same_bits(cma.weights()[:, 0].view(np.float32), g["cma_weights"], "cma weights")
    same_bits(host(ctx.rows_scale(rows1(x[:1000]), 0.37))[0], g["scale"], "fixed gain")
    same_bits(host(ctx.ingest(dev(g["raw_u8"]), engine.FORMAT_U8)), g["ingest_u8"], "ingest u8")
    same_bits(host(ctx.ingest(dev(g["raw_u8"].view(np.int8)), engine.FORMAT_S8)), g["ingest_s8"], "ingest s8")
    same_bits(host(ctx.ingest(dev(g["raw_s16"]), engine.FORMAT_S16)), g["ingest_s16"], "ingest s16")
    names = [ctx.lib.suamd_spectsrc_name(k).decode() for k in range(1, ctx.lib.suamd_spectsrc_count() + 1)]
    for k, name in enumerate(names, start=1):
        same_bits(host(ctx.spectsrc_preproc(k, dev(x[:2048]), 0.1 + 0.2j)), g["spectsrc_" + name], f"spectsrc {name}")
    fac = engine.FAC(ctx, 1024, 0.5)
    fac.feed(dev(x[:1024]), 2, 500)
    fac.feed(dev((x[1024:2048] * np.complex64(2)).astype(np.complex64)), 2, 500)
    assert np.max(np.abs(fac.array() - g["fac_1024"])) < 2e-5
    mn, mx = fac.range()
    assert abs(mx - g["fac_range"][1]) <= 1e-5 * g["fac_range"][1]



def test_v3_fft_channeliser_and_the_default_chain(ctx):
    """oracle_v3.npz: six channels of the FFT channeliser (sizes 64, 64, 256, 512 -> 8 ..., both arithmetic forms, precise and
    not) -- every row bit for bit --, the product's channel designs, and AGC -> Costas -> Gardner behind row 0"""
    import ctypes as C
    from sigdigger_amd import lib
    from tests.golden.make_golden import V3_CHANNELS
    g = np.load(os.path.join(GOLDEN, "oracle_v3.npz"))
    x = g["input_iq"]
    st = engine.SpectTuner(ctx, 4096)
    ids = [st.open_channel(*c) for c in V3_CHANNELS]
    out, counts = st.feed(dev(x))
    torch.cuda.synchronize()
    rows = [host(out[c, :counts[c]]) for c in ids]
    st.close()
    for k, r in enumerate(rows):
        same_bits(r.view(np.uint32), g[f"st32_row{k}"].view(np.uint32), f"channeliser row {k}")
    L = lib.load()
    for k, (f0, bw, guard, _) in enumerate(V3_CHANNELS):
        geom = (C.c_uint32 * 6)()
        hk = np.empty(int(g["st_geometry"][k][0]), np.complex64)
        assert L.suamd_specttuner_design(4096, f0, bw, guard, geom, hk.ctypes.data_as(C.c_void_p))
        assert list(geom) == list(g["st_geometry"][k])
        if k == 0:
            same_bits(hk.view(np.uint32), g["st_response_64"].view(np.uint32), "response, 64 bins")
        if k == 2:
            same_bits(hk.view(np.uint32), g["st_response_256"].view(np.uint32), "response, 256 bins")
    y = rows1(rows[0])
    a = engine.AGCBank(ctx, 1, tau=4.0).feed(y)
    z = engine.CostasBank(ctx, 1, 2, 0.0, 0.5, 3, 0.01).feed(a)
    sym = torch.zeros((1, z.shape[1] + 1), dtype=torch.complex64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    engine.ClockBank(ctx, 1, 0.2, 0.25).feed(z, sym, cnt)
    torch.cuda.synchronize()
    n = int(cnt[0])
    assert n == g["st32_chain_symbols"].size
    same_bits(host(sym)[0, :n].view(np.uint32), g["st32_chain_symbols"].view(np.uint32), "symbols behind the channeliser")
