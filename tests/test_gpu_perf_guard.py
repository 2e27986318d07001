"""Performance guards: loose thresholds (>= 30 % above what the pool's boxes measure) on the kernels whose speed has depended
on what the optimiser makes of the source -- a silent codegen regression must fail a test, not wait for a profile.
Kernel times are the kernels' own durations (suamd_kernel_timing), the minimum over the launches."""
import numpy as np
import pytest
import torch

from sigdigger_amd import engine, synth

pytestmark = pytest.mark.gpu


def _timed(kernel, fn, reps=12):
    fn()
    torch.cuda.synchronize()
    engine.kernel_timing_read()
    engine.kernel_timing(True)
    try:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    finally:
        engine.kernel_timing(False)
    r = engine.kernel_timing_read(kernel)
    engine.kernel_timing_read()
    assert r["launches"] == reps, (kernel, r)
    return r["min_ms"] * 1e3


def test_chan_pair_kernel_keeps_its_tile_loop(ctx):
    """BASELINE configs[1]'s FIR stage (one channel, 255 taps, D = 16) on 4 Mi samples: 17.4 - 20 us.  Round 4 found the
    kernel at 33 us whenever the optimiser could fold the tile loop's structural branches (chan_stream.hip opaque_zero)."""
    L, D, T = 1 << 22, 16, 255
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_()
    bank = engine.ChannelBank(ctx, [0.25], D, ctx.lpf_design(T, 0.75 / D))
    out = torch.empty((1, L // D + 8), dtype=torch.complex64, device="cuda")
    us = _timed("chan_pair_kernel", lambda: bank.feed(x, out=out))
    assert us < 26.0, f"chan_pair_kernel {us:.1f} us per 4 Mi samples (17.4 - 20 expected)"


def test_stp_kernel_on_a_4_mi_block(ctx):
    """the headline's channeliser (64 channels of 64 bins) alone on 4 Mi samples: 27 - 31 us"""
    L, D = 1 << 22, 64
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_()
    st = engine.SpectTuner(ctx, 4096)
    for f in synth.raster(64, 1.8 / 64):
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
    out = engine.time_major(64, L // D + 64, "cuda")
    us = _timed("stp_kernel", lambda: st.feed(x, out=out))
    st.close()
    assert us < 41.0, f"stp_kernel {us:.1f} us per 4 Mi samples (27 - 31 expected)"


def test_staggered_symbol_clocks_cost_what_aligned_ones_do(ctx):
    """the bank clock recovery's round-by-round schedule (loops.hip clock_ring): 64 Gardner detectors whose crossings fall at
    unrelated instants run within 30 % of 64 aligned ones (the lock-step schedule: 2.5 x) and under 75 ns per sample"""
    C, M, sps = 64, 1 << 16, 15.625
    rng = np.random.default_rng(3)
    t = np.arange(M)
    ms = {}
    for name, stagger in (("aligned", False), ("staggered", True)):
        rows = np.empty((C, M), np.complex64)
        for c in range(C):
            off = rng.random() * sps if stagger else 0.0
            sym = rng.integers(0, 4, M // 15 + 8)
            rows[c] = np.exp(1j * (np.pi / 2 * sym[np.floor((t + off) / sps).astype(np.int64)] + np.pi / 4))
        z = engine.time_major(C, M, "cuda")
        z.copy_(torch.from_numpy(rows).cuda())
        bank = engine.ClockBank(ctx, C, 0.2, 1.0 / sps)
        sym = torch.zeros((C, M), dtype=torch.complex64, device="cuda")
        cnt = torch.zeros(C, dtype=torch.int32, device="cuda")
        bank.feed(z, sym, cnt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            cnt.zero_()
            e0.record()
            bank.feed(z, sym, cnt)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        ms[name] = best
    assert ms["staggered"] < 1.3 * ms["aligned"], ms
    assert ms["staggered"] * 1e6 / M < 75.0, ms
