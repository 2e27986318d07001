"""Gangs on time-major slabs (suamd_*_gang_*_slab, suamd_rows_deliver_strided): the live analyzer's inspectors of the FFT
filter bank are columns of one slab, and their recurrences stream it where it lies.  Every item must still equal the CPU
oracle bit for bit (SPEC.md section D: the chain is a fixed binary32 operation sequence) -- whatever column it sits in,
whatever the pitch, in sub-ranges as the analyzer feeds them, over two blocks (state, history and delay lines carry)."""
import numpy as np
import pytest
import torch

from sigdigger_amd import engine, synth


pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def assert_bits(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
        # -0.0 == +0.0 is accepted; anything else is a failure
        assert np.array_equal(a, b), f"{what}: {int(np.sum(a != b))} of {a.size} values differ (max abs {np.max(np.abs(a - b))})"

SLACK = 128            # rows past the longest item a slab must be readable for (the kernels look one tile ahead)


def _slab(rows, pitch, fill=None):
    t = torch.zeros((rows + SLACK, pitch), dtype=torch.complex64, device="cuda")
    if fill is not None:                                       # whatever lies in unused columns / past an item's end must not matter
        t.copy_(dev((fill.standard_normal((rows + SLACK, pitch)) + 1j * fill.standard_normal((rows + SLACK, pitch))).astype(np.complex64)))
    return t


def _put(slab, col, v):
    if v.size:
        slab[:v.size, col] = dev(v)


@pytest.mark.parametrize("n,pitch,parts", [(70, 128, 4), (64, 64, 1), (9, 64, 3), (130, 192, 2)])
def test_chain_on_slabs_equals_the_oracle_bit_for_bit(ctx, sdo, n, pitch, parts):
    """AGC -> Costas -> Gardner on three slabs, items in shuffled columns, heterogeneous loops and lengths, the serial stages
    in `parts` sub-ranges (row offsets into the slabs, as enqueue_inspectors does), two blocks."""
    rng = np.random.default_rng(1000 + n + pitch)
    lens = rng.integers(0, 5000, n)
    lens[:4] = [0, 1, 17, 4999]
    kinds = rng.integers(1, 4, n)
    arm = rng.integers(1, 6, n)
    lbw = rng.uniform(0.002, 0.02, n)
    sps = rng.choice([4, 8, 16], n)
    cols = rng.permutation(pitch)[:n]
    xs_h = [synth.psk_carriers(max(int(L_), 1), [0.002 * (i % 7 - 3)], sps=int(sps[i]), order=int(2 ** kinds[i]), seed=100 + i)[:int(L_)]
            for i, L_ in enumerate(lens)]
    cuts = [(0, int(L_) // 3) for L_ in lens], [(int(L_) // 3, int(L_)) for L_ in lens]
    agc = [engine.AGCBank(ctx, 1, tau=float(sps[i])) for i in range(n)]
    cos = [engine.CostasBank(ctx, 1, int(kinds[i]), 0.0, 2.0 / sps[i], int(arm[i]), float(lbw[i])) for i in range(n)]
    clk = [engine.ClockBank(ctx, 1, 0.2, 1.0 / sps[i]) for i in range(n)]
    syms = [torch.zeros(int(L_) + 2, dtype=torch.complex64, device="cuda") for L_ in lens]
    cnts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(n)]
    outs = [[] for _ in range(n)]
    rows = int(max(lens))
    work = torch.zeros(2 * (rows + SLACK) * pitch, dtype=torch.float32, device="cuda")
    for rnd in cuts:
        ln = [b - a for a, b in rnd]
        Y, A, Z = _slab(rows, pitch, rng), _slab(rows, pitch, rng), _slab(rows, pitch, rng)
        for i, (a, b) in enumerate(rnd):
            _put(Y, int(cols[i]), xs_h[i][a:b])
        engine.gang_agc_slab(ctx, agc, Y, cols, A, cols, ln, work, parts=parts)
        for j in range(parts):
            r0 = [L_ * j // parts for L_ in ln]
            sub = [L_ * (j + 1) // parts - L_ * j // parts for L_ in ln]
            engine.gang_costas_slab(ctx, cos, A, cols, Z, cols, sub, r0=r0)
            engine.gang_clock_slab(ctx, clk, Z, cols, sub, syms, cnts, r0=r0)
        torch.cuda.synchronize()
        zh = host(Z)
        for i in range(n):
            outs[i].append(zh[:ln[i], int(cols[i])].copy())
    for i in range(n):
        a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(float(sps[i]))), xs_h[i]) if lens[i] else np.zeros(0, np.complex64)
        st = sdo.costas_new(int(kinds[i]), 0.0, 2.0 / sps[i], int(arm[i]), float(lbw[i]))
        z = sdo.costas_feed_bulk(st, a) if lens[i] else a
        assert_bits(np.concatenate(outs[i]), z, f"slab item {i} (column {cols[i]}): costas output")
        om, ph = cos[i].state()
        assert ph[0] == st.phase and np.float32(st.omega) == om[0]
        ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, 1.0 / sps[i]), z) if lens[i] else z
        k = int(cnts[i].cpu()[0])
        assert k == ref.size, f"slab item {i}: symbol count"
        assert_bits(host(syms[i][:k]), ref, f"slab item {i}: symbols")


def test_slab_and_row_gangs_agree_and_share_their_banks(ctx, sdo):
    """a block through the row gangs, the next through the slab gangs, with the SAME bank objects: the state one form leaves
    is the state the other continues from (an analyzer that changes layout mid-stream -- the 65th inspector -- loses nothing)"""
    rng = np.random.default_rng(5)
    n, pitch, sps = 40, 64, 8
    L_ = 3000
    xs_h = [synth.psk_carriers(2 * L_, [0.002 * (i % 5 - 2)], sps=sps, order=4, seed=700 + i) for i in range(n)]
    agc = [engine.AGCBank(ctx, 1, tau=float(sps)) for _ in range(n)]
    cos = [engine.CostasBank(ctx, 1, 2, 0.0, 2.0 / sps, 3, 0.01) for _ in range(n)]
    clk = [engine.ClockBank(ctx, 1, 0.2, 1.0 / sps) for _ in range(n)]
    syms = [torch.zeros(2 * L_, dtype=torch.complex64, device="cuda") for _ in range(n)]
    cnts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(n)]
    xs = [dev(v[:L_]) for v in xs_h]
    ya, yz = [torch.empty_like(x) for x in xs], [torch.empty_like(x) for x in xs]
    engine.gang_agc(ctx, agc, xs, ya)
    engine.gang_costas(ctx, cos, ya, yz)
    engine.gang_clock(ctx, clk, yz, syms, cnts)
    cols = list(range(n))
    Y, A, Z = _slab(L_, pitch), _slab(L_, pitch), _slab(L_, pitch)
    for i in range(n):
        _put(Y, i, xs_h[i][L_:])
    work = torch.zeros(2 * (L_ + SLACK) * pitch, dtype=torch.float32, device="cuda")
    engine.gang_agc_slab(ctx, agc, Y, cols, A, cols, [L_] * n, work, parts=2)
    engine.gang_costas_slab(ctx, cos, A, cols, Z, cols, [L_] * n)
    engine.gang_clock_slab(ctx, clk, Z, cols, [L_] * n, syms, cnts)
    torch.cuda.synchronize()
    zh = host(Z)
    for i in range(n):
        a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(float(sps))), xs_h[i])
        z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, 2.0 / sps, 3, 0.01), a)
        assert_bits(np.concatenate([host(yz[i]), zh[:L_, i]]), z, f"item {i}: rows then slab")
        ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, 1.0 / sps), z)
        k = int(cnts[i].cpu()[0])
        assert k == ref.size
        assert_bits(host(syms[i][:k]), ref, f"item {i}: symbols")


def test_pll_gang_on_a_slab_bit_exact(ctx, sdo):
    rng = np.random.default_rng(7)
    n, pitch = 40, 64
    lens = rng.integers(0, 3000, n); lens[:3] = [0, 1, 2999]
    fcs = rng.uniform(0.005, 0.05, n)
    cols = rng.permutation(pitch)[:n]
    xs_h = [(np.exp(1j * (np.pi * 0.004 * (i % 5 + 1) * np.arange(L_) + i)) + 0.05 * synth.tone_noise(max(int(L_), 1), seed=i)[:int(L_)]).astype(np.complex64)
            for i, L_ in enumerate(lens)]
    plls = [engine.PLLBank(ctx, 1, 0.0, float(fcs[i])) for i in range(n)]
    got = [[] for _ in range(n)]
    rows = int(max(lens))
    for a_, b_ in ((0.0, 0.4), (0.4, 1.0)):
        X, Y = _slab(rows, pitch, rng), _slab(rows, pitch, rng)
        ln = [int(b_ * L_) - int(a_ * L_) for L_ in lens]
        for i, L_ in enumerate(lens):
            _put(X, int(cols[i]), xs_h[i][int(a_ * L_):int(b_ * L_)])
        engine.gang_pll_slab(ctx, plls, X, cols, Y, cols, ln)
        torch.cuda.synchronize()
        yh = host(Y)
        for i in range(n):
            got[i].append(yh[:ln[i], int(cols[i])].copy())
    for i in range(n):
        rp = sdo.pll_track_bulk(sdo.pll_new(0.0, float(fcs[i])), xs_h[i]) if lens[i] else np.zeros(0, np.complex64)
        assert_bits(np.concatenate(got[i]), rp, f"pll slab item {i}")


def test_rows_deliver_strided_hands_columns_over(ctx):
    rng = np.random.default_rng(3)
    pitch, rows, n = 96, 4000, 50
    slab_h = (rng.standard_normal((rows, pitch)) + 1j * rng.standard_normal((rows, pitch))).astype(np.complex64)
    slab = dev(slab_h)
    cols = rng.permutation(pitch)[:n]
    lens = rng.integers(0, rows, n); lens[:3] = [0, 1, rows]
    row = dev(slab_h[:, 0].copy())                             # one contiguous row among the columns (stride 1)
    ptrs = [slab.data_ptr() + int(c) * 8 for c in cols] + [row.data_ptr()]
    strides = [pitch] * n + [1]
    counters = [torch.tensor([int(L_)], dtype=torch.int32, device="cuda") if i % 2 else int(L_) for i, L_ in enumerate(lens)] + [rows]
    dsts = [torch.zeros(rows + 3, dtype=torch.complex64).pin_memory() for _ in range(n + 1)]
    outs = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(n + 1)]
    engine.rows_deliver_strided(ctx, ptrs, strides, counters, dsts, outs)
    torch.cuda.synchronize()
    for i, L_ in enumerate(lens):
        assert int(outs[i][0]) == L_
        assert_bits(host(dsts[i][:L_]), slab_h[:L_, int(cols[i])], f"column {cols[i]}")
        assert not host(dsts[i][L_:]).any()
    assert_bits(host(dsts[n][:rows]), slab_h[:, 0], "the contiguous row")


def test_slab_gangs_refuse_what_they_cannot_serve(ctx):
    import ctypes as C
    lib = ctx.lib
    b = engine.AGCBank(ctx, 1, tau=8.0)
    X = torch.zeros((256, 64), dtype=torch.complex64, device="cuda")
    Y = torch.zeros((256, 64), dtype=torch.complex64, device="cuda")
    work = torch.zeros(2 * 256 * 64, dtype=torch.float32, device="cuda")
    hb = (C.c_void_p * 1)(b.h)
    ln = (C.c_uint64 * 1)(100)
    # a "column" beyond the pitch
    bad = (C.c_void_p * 1)(X.data_ptr() + 64 * 8)
    assert not lib.suamd_agc_gang_pre_slab(ctx.h, hb, 1, X.data_ptr(), 64, bad, ln, work.data_ptr(), 256, None)
    # work slabs shorter than the row
    ok = (C.c_void_p * 1)(X.data_ptr() + 8)
    assert not lib.suamd_agc_gang_pre_slab(ctx.h, hb, 1, X.data_ptr(), 64, ok, ln, work.data_ptr(), 50, None)
    # aliased apply
    m0, m1 = (C.c_uint64 * 1)(0), (C.c_uint64 * 1)(100)
    assert not lib.suamd_agc_gang_apply_slab(ctx.h, hb, 1, X.data_ptr(), 64, ok, X.data_ptr(), 64, ok, ln, m0, m1, work.data_ptr(), 256, None)
    # zero pitch
    cb = engine.CostasBank(ctx, 1, 2, 0.0, 0.25, 3, 0.01)
    hc = (C.c_void_p * 1)(cb.h)
    yk = (C.c_void_p * 1)(Y.data_ptr() + 8)
    assert not lib.suamd_costas_gang_feed_slab(ctx.h, hc, 1, ok, 0, yk, 64, ln, None)
    assert lib.suamd_costas_gang_feed_slab(ctx.h, hc, 0, None, 64, None, 64, None, None)        # nothing to do
