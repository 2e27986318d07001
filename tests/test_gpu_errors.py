"""Error behaviour at the C ABI (the reference's convention: SUBOOL / NULL + a message, every call wrapped in
SU_ATTEMPT, include/Suscan/Compat.h:28-36): bad arguments are refused loudly, never computed around, and
degenerate-but-legal inputs (empty blocks, blocks shorter than the filter, zero symbols) are no-ops."""
import ctypes as C

import numpy as np
import pytest
import torch

from sigdigger_amd import engine, lib as L, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_constructors_refuse_bad_arguments(ctx):
    lib = ctx.lib
    for n in (0, 100, 256, 3000, 1 << 21):
        assert not lib.suamd_psd_new(ctx.h, n, 4)
        assert b"window_size" in lib.suamd_last_error() or lib.suamd_last_error()
    assert not lib.suamd_psd_new(ctx.h, 1024, 99)                        # unknown window
    taps = np.ones(4, np.float32)
    fn = np.zeros(1, np.float64)
    assert not lib.suamd_chanbank_new(ctx.h, 0, fn.ctypes.data_as(C.c_void_p), 4, taps.ctypes.data_as(C.c_void_p), 4)
    assert not lib.suamd_chanbank_new(ctx.h, 1, fn.ctypes.data_as(C.c_void_p), 0, taps.ctypes.data_as(C.c_void_p), 4)
    assert not lib.suamd_chanbank_new(ctx.h, 1, fn.ctypes.data_as(C.c_void_p), 4, taps.ctypes.data_as(C.c_void_p), 0)
    assert not lib.suamd_chanbank_new(ctx.h, 1, None, 4, taps.ctypes.data_as(C.c_void_p), 4)
    assert not lib.suamd_costas_bank_new(ctx.h, 4, 0, 0.0, 0.1, 3, 0.01)     # SU_COSTAS_KIND_NONE
    assert not lib.suamd_costas_bank_new(ctx.h, 4, 2, 0.0, 0.1, 9, 0.01)     # arm filter order too high
    assert not lib.suamd_clock_bank_new(ctx.h, 4, 0.2, 0.0)                  # su_clock_detector_init returns -1 for bhint <= 0
    assert not lib.suamd_cma_bank_new(ctx.h, 4, 17, 1e-3)
    assert not lib.suamd_fac_new(ctx.h, 1000, 0.5)
    assert not lib.suamd_fir_bank_new(ctx.h, 4, taps.ctypes.data_as(C.c_void_p), 0)
    assert not lib.suamd_ctx_new(12345)                                       # no such GPU: no silent fallback
    assert lib.suamd_last_error()


def test_calls_refuse_bad_arguments(ctx):
    lib = ctx.lib
    x = dev(synth.tone_noise(4096, seed=1))
    out = torch.empty(4096, dtype=torch.complex64, device="cuda")
    assert not lib.suamd_xlate_bulk(ctx.h, None, out.data_ptr(), 4096, 0, 1, 0, None)
    assert not lib.suamd_ingest_iq(ctx.h, 9, x.data_ptr(), 16, out.data_ptr(), None)
    assert not lib.suamd_ingest_iq(ctx.h, engine.FORMAT_S16, x.data_ptr() + 2, 16, out.data_ptr(), None)    # misaligned
    assert not lib.suamd_fft_forward_bulk(ctx.h, x.data_ptr(), out.data_ptr(), out.data_ptr(), 3, None)
    assert not lib.suamd_fft_forward_bulk(ctx.h, x.data_ptr(), out.data_ptr(), out.data_ptr(), 25, None)
    assert lib.suamd_sample_zero_crossing_bulk(ctx.h, x.data_ptr(), 4096, 0.1, 0, 0, 0.0, 0.0, 1.0, 0.0,
                                               out.data_ptr(), 100, None) == -1    # capacity too small
    with pytest.raises(L.SigDiggerAmdError):
        ctx.histogram_feed(x, 7)
    view = engine.SpectrumView(ctx)
    with pytest.raises(L.SigDiggerAmdError):                                     # range never set
        view.feed(torch.zeros(1024, device="cuda"), 0.0, 1e6)
    psd = engine.PSD(ctx, 1024)
    with pytest.raises(L.SigDiggerAmdError):
        psd.feed(x, nframes=5)                                                   # 5 frames do not fit 4096 samples
    bank = engine.FIRBank(ctx, 2, np.ones(5, np.float32))
    rows = torch.zeros((2, 64), dtype=torch.complex64, device="cuda")
    with pytest.raises(L.SigDiggerAmdError):
        bank.feed(rows, out=rows)                                                # in place
    with pytest.raises(L.SigDiggerAmdError):
        bank.feed(torch.zeros((3, 64), dtype=torch.complex64, device="cuda"))    # wrong number of rows


def test_degenerate_inputs_are_no_ops(ctx, sdo):
    # empty blocks
    x0 = torch.empty(0, dtype=torch.complex64, device="cuda")
    assert ctx.ingest(torch.empty(0, dtype=torch.int16, device="cuda"), engine.FORMAT_S16).numel() == 0
    assert ctx.sample_zero_crossing(x0, 0.1, 0).numel() == 0
    # a channel bank fed fewer samples than one decimation step emits nothing but remembers them
    taps = ctx.lpf_design(31, 0.1)
    bank = engine.ChannelBank(ctx, [0.1], 16, taps)
    x = synth.tone_noise(1000, seed=3)
    outs = []
    for a, b in ((0, 0), (0, 5), (5, 5), (5, 6), (6, 40), (40, 1000)):
        y = bank.feed(dev(x[a:b])) if b > a else bank.feed(x0)
        outs.append(y.cpu().numpy().reshape(1, -1))
    got = np.concatenate(outs, axis=1)[0]
    dp = sdo.fnor_to_dphase(-0.1)
    ref = sdo.chan_feed(np.zeros(30, np.complex64), x, 0, sdo.chan_modulate_taps(sdo.lpf_design(31, 0.1), dp), 16, 0, dp)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # clock bank on an empty block leaves the counters alone
    clk = engine.ClockBank(ctx, 3, 0.2, 0.1)
    sym = torch.zeros((3, 8), dtype=torch.complex64, device="cuda")
    cnt = torch.zeros(3, dtype=torch.int32, device="cuda")
    clk.feed(torch.zeros((3, 0), dtype=torch.complex64, device="cuda"), sym, cnt)
    assert int(cnt.sum()) == 0
    # an equalizer told that no symbols arrived
    cma = engine.CMABank(ctx, 3, 4, 1e-3)
    cma.feed(sym, count=cnt)
    assert np.array_equal(cma.weights()[0], np.ones(3, np.complex64)) and not cma.weights()[1:].any()


def test_gang_calls_refuse_bad_arguments(ctx):
    """gangs: only 1-channel banks, no aliased AGC rows, sub-ranges inside the block; empty gangs / blocks are no-ops"""
    lib = ctx.lib
    taps = ctx.lpf_design(63, 0.1)
    wide = engine.ChannelBank(ctx, [0.1, -0.1], 4, taps)
    one = engine.ChannelBank(ctx, [0.1], 4, taps)
    x = dev(synth.tone_noise(4096, seed=2))
    y = torch.empty(2048, dtype=torch.complex64, device="cuda")
    with pytest.raises(L.SigDiggerAmdError, match="1-channel"):
        engine.gang_chan(ctx, [wide], x, [y])
    with pytest.raises(L.SigDiggerAmdError):
        engine.gang_chan(ctx, [one], x, [torch.empty(8, dtype=torch.complex64, device="cuda")])        # row too short
    assert [t.numel() for t in engine.gang_chan(ctx, [one], x[:0], [y])] == [0]                        # empty block
    assert engine.gang_chan(ctx, [], x, []) == []                                                      # empty gang
    ptrs = (C.c_void_p * 1)(one.h)
    assert not lib.suamd_chanbank_gang_feed(ctx.h, ptrs, 1, x.data_ptr(), 4096, None, None, None)     # no output rows
    assert b"null" in lib.suamd_last_error()
    agc = engine.AGCBank(ctx, 1, tau=8.0)
    with pytest.raises(L.SigDiggerAmdError, match="alias"):
        engine.gang_agc(ctx, [agc], [x], [x])
    two = engine.AGCBank(ctx, 2, tau=8.0)
    with pytest.raises(L.SigDiggerAmdError, match="1-channel"):
        engine.gang_agc(ctx, [two], [x], [y.new_empty(4096)])
    b = (C.c_void_p * 1)(agc.h)
    lens, m0, m1 = (C.c_uint64 * 1)(4096), (C.c_uint64 * 1)(100), (C.c_uint64 * 1)(5000)
    assert not lib.suamd_agc_gang_level(ctx.h, b, 1, lens, m0, m1, None)                               # sub-range past the block
    assert b"sub-range" in lib.suamd_last_error()
    m1[0] = 50
    assert not lib.suamd_agc_gang_level(ctx.h, b, 1, lens, m0, m1, None)                               # m0 > m1
    out = torch.zeros(16, dtype=torch.complex64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert not lib.suamd_rows_deliver(ctx.h, 1, (C.c_void_p * 1)(x.data_ptr()), None, None, (C.c_void_p * 1)(out.data_ptr()),
                                      (C.c_void_p * 1)(cnt.data_ptr()), None)                          # neither counter nor length
    engine.rows_deliver(ctx, [], [], [], [])                                                           # nothing to hand over
    torch.cuda.synchronize()
