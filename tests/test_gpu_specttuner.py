"""FFT channeliser (rows T2 / N2; SPEC.md section C2): csrc/specttuner.hip against the oracle's restatement
(oracle/sdo.c sdo_specttuner_*), through both front ends of the same object -- suamd_specttuner_* (device block
interface) and su_specttuner_* (the libsigutils names and callback contract of Tasks/LPFTask.cpp).

Tolerance: the device transforms are binary32, the oracle's binary64: |y - y_ref| <= TOL * max|y_ref| per channel with
TOL = 1e-5 (the north star's bound) on inputs whose channels carry comparable power; the measured error is ~1e-6.
Index geometry (sizes, centre bins, block counts, which samples a feed yields) is exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from sigdigger_amd import engine, sigutils, synth

pytestmark = pytest.mark.gpu

W, H = 4096, 2048
TOL = 1e-5


def cnoise(n, seed):
    r = np.random.default_rng(seed)
    return ((r.standard_normal(n) + 1j * r.standard_normal(n)) / np.sqrt(2)).astype(np.complex64)


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def run_gpu(ctx, x, chans, splits=None, run=None, time_major=False):
    """chans: list of (f0, bw, guard, precise).  Returns the per-channel streams."""
    st = engine.SpectTuner(ctx, W)
    if run:
        st.set_run(run)
    ids = [st.open_channel(*c) for c in chans]
    dx = torch.from_numpy(x).cuda()
    cuts = [0] + list(splits or []) + [x.size]
    outs = [[] for _ in ids]
    for a, b in zip(cuts[:-1], cuts[1:]):
        buf = engine.time_major(len(ids), (b - a) + 16, "cuda") if time_major else None
        out, counts = st.feed(dx[a:b], out=buf)
        torch.cuda.synchronize()
        for k, c in enumerate(ids):
            outs[k].append(out[c, :counts[c]].cpu().numpy())
    st.close()
    return [np.concatenate(o) for o in outs]


@pytest.mark.parametrize("dec", [1, 2, 4, 16, 64, 128, 256, 512, 1024, 2048])
def test_one_channel_every_size_matches_oracle(ctx, sdo, dec):
    size = W // dec
    bw = 2 * np.pi * (0.75 * size / W)
    f0 = 2 * np.pi * 0.137
    x = cnoise(H * 21, 100 + dec)
    g = sdo.specttuner_geometry(W, f0, bw, 1.0)
    assert g.size == size and g.decimation == dec and g.center % 2 == 0
    ref = sdo.specttuner_run(x, W, f0, bw, 1.0)
    got = run_gpu(ctx, x, [(f0, bw, 1.0, False)])[0]
    assert got.size == ref.size == 20 * (size // 2)
    assert relerr(got, ref) <= TOL


def test_bank_of_64_psk_channels_d64(ctx, sdo):
    """the C4 slice's shape: 64 channels of 64 bins on a raster, here with real carriers in them"""
    nch = 64
    fn = synth.raster(nch, 2 * 700e3 / 50e6)
    x = synth.psk_carriers(H * 40, fn, sps=100, order=4, seed=7, snr_db=30)
    bw = 2 * np.pi * 0.75 / 64
    chans = [(np.pi * f % (2 * np.pi), bw, 1.0, False) for f in fn]
    got = run_gpu(ctx, x, chans)
    for c in (0, 1, 31, 32, 63):
        ref = sdo.specttuner_run(x, W, chans[c][0], bw, 1.0)
        assert got[c].size == ref.size
        assert relerr(got[c], ref) <= TOL, c
    # time-major rows ([m][channel] in memory, what the recurrence kernels stream) go through the LDS tile: same bits
    tm = run_gpu(ctx, x, chans, splits=[H * 9], time_major=True)
    for a, b in zip(got, tm):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_any_split_of_the_stream_gives_the_same_samples(ctx):
    x = cnoise(H * 48, 5)
    chans = [(0.3, 2 * np.pi / 64 * 0.7, 1.0, False), (2.1, 2 * np.pi / 64 * 0.5, 1.0, True), (4.0, 2 * np.pi / 16 * 0.8, 1.0, False)]
    one = run_gpu(ctx, x, chans)
    for splits, run in (([H], 8), ([H * 3, H * 4, H * 30], 8), ([H * 10], 3), ([], 1), ([H * 7], 64)):
        parts = run_gpu(ctx, x, chans, splits, run)
        for a, b in zip(one, parts):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (splits, run)


def test_a_run_that_cannot_get_its_seam_payload_transforms_the_seam_window_itself(ctx, tune):
    """The wavefront kernel hands the block at the seam of two runs over through global memory; a run whose successor has
    not published in time (not resident: the device full of something else) computes that block itself.  Forced here with
    a zero wait budget: every run takes the fallback, and the samples are those of the normal path bit for bit."""
    x = cnoise(H * 40, 21)
    chans = [(0.3 + 0.09 * c, 2 * np.pi / 64 * 0.7, 1.0, bool(c % 3 == 0)) for c in range(70)]       # two blocks of 64 lanes
    chans += [(1.1, 2 * np.pi / 16 * 0.8, 1.0, False)]
    ref = run_gpu(ctx, x, chans, splits=[H * 13], run=3)
    tune.setenv("SUAMD_ST_SEAM_POLLS", "0")
    for run in (1, 3):
        got = run_gpu(ctx, x, chans, splits=[H * 13], run=run)
        for a, b in zip(ref, got):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), run


def test_precise_channel_corrects_the_bin_rounding(ctx, sdo):
    f0 = 2 * np.pi * 411 / W                      # an odd bin: as far from the even centre bins as it gets
    n = H * 60
    x = np.exp(1j * f0 * np.arange(n)).astype(np.complex64)
    bw = 2 * np.pi / 64 * 0.75
    ref = sdo.specttuner_run(x, W, f0, bw, 1.0, precise=True)
    got = run_gpu(ctx, x, [(f0, bw, 1.0, True)], splits=[H * 20])[0]
    assert relerr(got, ref) <= TOL
    tail = got[200:]
    assert np.max(np.abs(np.angle(tail[1:] * np.conj(tail[:-1])))) < 2e-3     # the tone lands on DC
    coarse = run_gpu(ctx, x, [(f0, bw, 1.0, False)])[0][200:]
    assert np.median(np.abs(np.angle(coarse[1:] * np.conj(coarse[:-1])))) > 5e-2   # one bin, times the decimation


def test_more_channels_than_one_workgroup_serves_and_mixed_responses(ctx, sdo):
    nch = 150                                                     # 64 per workgroup at 64 bins: grid.y = 3
    r = np.random.default_rng(9)
    x = cnoise(H * 12, 9)
    chans = [(float(r.uniform(0, 2 * np.pi)), 2 * np.pi / 64 * float(r.uniform(0.3, 0.9)), 1.0, bool(c % 2)) for c in range(nch)]
    got = run_gpu(ctx, x, chans)
    for c in (0, 63, 64, 127, 128, 149):
        ref = sdo.specttuner_run(x, W, chans[c][0], chans[c][1], 1.0, chans[c][3])
        assert relerr(got[c], ref) <= TOL, c
    tm = run_gpu(ctx, x, chans, time_major=True)
    for a, b in zip(got, tm):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_channels_of_different_sizes_and_open_close_between_feeds(ctx, sdo):
    x = cnoise(H * 30, 11)
    st = engine.SpectTuner(ctx, W)
    a = st.open_channel(1.0, 2 * np.pi / 64 * 0.8)
    b = st.open_channel(2.0, 2 * np.pi / 8 * 0.8)
    dx = torch.from_numpy(x).cuda()
    o1, c1 = st.feed(dx[:H * 10])
    ya, yb = [o1[a, :c1[a]].cpu().numpy()], [o1[b, :c1[b]].cpu().numpy()]
    c = st.open_channel(3.0, 2 * np.pi / 64 * 0.6)                 # joins a's size group mid-stream
    o2, c2 = st.feed(dx[H * 10:H * 20])
    ya.append(o2[a, :c2[a]].cpu().numpy()); yb.append(o2[b, :c2[b]].cpu().numpy())
    yc = [o2[c, :c2[c]].cpu().numpy()]
    st.close_channel(b)
    o3, c3 = st.feed(dx[H * 20:])
    ya.append(o3[a, :c3[a]].cpu().numpy()); yc.append(o3[c, :c3[c]].cpu().numpy())
    assert c3[b] == 0
    st.close()
    ra = sdo.specttuner_run(x, W, 1.0, 2 * np.pi / 64 * 0.8, 1.0)
    rb = sdo.specttuner_run(x[:H * 20], W, 2.0, 2 * np.pi / 8 * 0.8, 1.0)
    assert relerr(np.concatenate(ya), ra) <= TOL
    assert relerr(np.concatenate(yb), rb) <= TOL
    # the late channel starts with an empty previous block: from its second block on it equals a channel that had
    # been open all along (same windows: it joined on a window boundary)
    rc = sdo.specttuner_run(x, W, 3.0, 2 * np.pi / 64 * 0.6, 1.0)
    yc = np.concatenate(yc)
    off = 9 * 32                                                   # blocks 0..8 came out of the first feed
    assert yc.size == rc.size - off
    assert relerr(yc[32:], rc[off + 32:]) <= TOL


def test_lpftask_contract_through_the_sigutils_names(sdo):
    """Tasks/LPFTask.cpp: window 4096, f0 = 0, bw = pi * rel_bw, guard = 2 pi / bw ("no decimation"), fed 8192 samples
    per work(), then single zero samples until `length` outputs have arrived (:104-107)."""
    n = 40000
    x = (cnoise(n, 13) + np.exp(1j * 0.05 * np.arange(n))).astype(np.complex64)
    rel_bw = 0.1
    bw = np.float32(np.pi * rel_bw)
    guard = np.float32(2 * np.pi / bw)
    t = sigutils.SpectTuner(W)
    k = t.open_channel(0.0, float(bw), float(guard))
    assert t.L.su_specttuner_channel_get_decimation(t.channel(k)) == 1.0
    for p in range(0, n, 8192):
        t.feed(x[p:p + 8192])
    fed = n
    zero = np.zeros(1, np.complex64)
    while t.samples(k).size < n:
        t.feed(zero)
        fed += 1
        assert fed < n + 2 * W
    got = t.samples(k)[:n]
    t.close()
    ref = sdo.specttuner_run(np.concatenate([x, np.zeros(fed - n, np.complex64)]), W, 0.0, float(bw), float(guard))
    assert ref.size >= n
    assert relerr(got, ref[:n]) <= TOL
    # it is a low-pass: the tone at 0.05 rad/sample (inside pi * 0.1 / 2 ... ) survives, the wideband noise drops
    assert np.var(got[W:]) < 0.6 * np.var(x)                # the unit tone + 5 % of the unit noise, out of 2


def test_full_size_block_64_channels(ctx, sdo):
    """BASELINE's C4 slice: 4 Mi samples, 64 channels of 64 bins; oracle spot checks + split invariance at full size"""
    L = 1 << 22
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_(generator=g)
    fn = synth.raster(64, 2 * 90e3 / 50e6)
    bw = 2 * np.pi * 0.75 / 64
    st = engine.SpectTuner(ctx, W)
    for f in fn:
        st.open_channel(np.pi * f % (2 * np.pi), bw)
    out, counts = st.feed(x)
    torch.cuda.synchronize()
    assert counts == [(L // H - 1) * 32] * 64
    st.close()
    xh = x.cpu().numpy()
    for c, k0 in ((0, 0), (40, 1000), (63, L // H - 1 - 12)):
        # blocks k0 .. k0+9 depend on windows k0-1 .. k0+9 only
        lo = max(k0 - 1, 0)
        ref = sdo.specttuner_run(xh[lo * H:(k0 + 11) * H], W, np.pi * fn[c] % (2 * np.pi), bw, 1.0)
        skip = (k0 - lo) * 32
        got = out[c, k0 * 32:(k0 + 10) * 32].cpu().numpy()
        assert relerr(got, ref[skip:skip + 320]) <= TOL, (c, k0)
    st2 = engine.SpectTuner(ctx, W)
    for f in fn:
        st2.open_channel(np.pi * f % (2 * np.pi), bw)
    o1, c1 = st2.feed(x[:L // 2])
    first = o1[:, :c1[0]].clone()
    o2, c2 = st2.feed(x[L // 2:])
    both = torch.cat([first, o2[:, :c2[0]]], dim=1)
    assert torch.equal(torch.view_as_real(both.contiguous()), torch.view_as_real(out[:, :counts[0]].contiguous()))
    st2.close()


def test_reference_lpftask_unchanged_on_the_gpu_channeliser(sdo):
    """Tasks/LPFTask.cpp compiled UNCHANGED from the reference (oracle/_ref/libsdref.so, built where /root/reference is),
    linked against the product's su_specttuner_*: the whole task -- constructor, work() loop, zero flush -- against
    the oracle."""
    from oracle import sdref
    if not sdref.available():
        pytest.skip("oracle/_ref/libsdref.so was not built (it needs /root/reference)")
    n = 50000
    x = (cnoise(n, 17) + np.exp(1j * 0.02 * np.arange(n))).astype(np.complex64)
    rel_bw = np.float32(0.08)
    got = sdref.lpf_task(x, float(rel_bw))
    assert got is not None
    bw = np.float32(np.pi) * rel_bw                       # SU_NORM2ANG_FREQ(bw): PI * SU_ASFLOAT(bw) in double, stored as SUFLOAT
    bw = np.float32(np.pi * float(rel_bw))
    guard = np.float32(2 * np.pi / float(bw))
    ref = sdo.specttuner_run(np.concatenate([x, np.zeros(2 * W, np.complex64)]), W, 0.0, float(bw), float(guard))
    assert relerr(got, ref[:n]) <= TOL


# ---- bit for bit against the binary32 statement (SPEC.md C2 "binary32 arithmetic"; oracle/sdo.c sdo_specttuner_bank_f32) ----
def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("dec", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048])
@pytest.mark.parametrize("precise", [False, True])
def test_every_size_equals_the_binary32_statement_bit_for_bit(ctx, sdo, dec, precise):
    """sizes 8 .. 64 run specttuner_wave.hip (64 x 64 forward transform, fused response and cross-fade), every other size
    specttuner.hip (radix-16 passes, unfused): the oracle states both, operation for operation"""
    size = W // dec
    bw = 2 * np.pi * (0.75 * size / W)
    f0 = 2 * np.pi * 0.1371
    x = cnoise(H * 25, 300 + dec)
    ref = sdo.specttuner_run_f32(x, f0, bw, 1.0, precise)
    got = run_gpu(ctx, x, [(f0, bw, 1.0, precise)], splits=[H * 7])[0]
    assert got.size == ref.size == 24 * max(size // 2, 1)
    assert np.array_equal(_bits(got), _bits(ref))


def test_a_mixed_bank_equals_the_binary32_statement_bit_for_bit(ctx, sdo):
    """150 channels of 64 bins with different responses (the per-lane response table), precise and not, plus three other
    sizes in the same tuner, fed in three pieces with runs of 3 windows: every sample of every channel"""
    r = np.random.default_rng(19)
    x = cnoise(H * 30, 23)
    chans = [(float(r.uniform(0, 2 * np.pi)), 2 * np.pi / 64 * float(r.uniform(0.3, 0.9)), 1.0, bool(c % 2)) for c in range(150)]
    chans += [(1.1, 2 * np.pi / 16 * 0.8, 1.0, False), (4.4, 2 * np.pi / 256 * 0.7, 1.0, True), (2.9, 2 * np.pi / 512 * 0.75, 1.0, False)]
    got = run_gpu(ctx, x, chans, splits=[H * 4, H * 17], run=3)
    ref = sdo.specttuner_bank_f32(x, [c[0] for c in chans], [c[1] for c in chans], [c[2] for c in chans], [c[3] for c in chans], threads=8)
    for k in range(len(chans)):
        assert got[k].size == ref[k].size, k
        assert np.array_equal(_bits(got[k]), _bits(ref[k])), k


@pytest.mark.parametrize("kernel", ["pair", "wave"])
def test_the_bank_changes_between_no_some_and_all_precise_channels(ctx, sdo, tune, kernel):
    """The narrow-channel kernels come in three forms by which channels of a launch are precise (none / some / all: the host's
    flag per size group, csrc/specttuner_pair.hip ROTCAP).  One tuner whose bank goes none -> some -> all -> none between feeds:
    every channel's stream equals the oracle's for a channel opened where it was opened (a channel that opens mid-stream
    sees the half window before its first feed as history: a fresh oracle started one half window earlier)."""
    tune.setenv("st_kernel", 0 if kernel == "pair" else 1)
    x = cnoise(H * 26, 77)
    bw = 2 * np.pi / 64 * 0.8
    dx = torch.from_numpy(x).cuda()
    st = engine.SpectTuner(ctx, W)
    cuts = [0, H * 6, H * 12, H * 19, H * 26]
    got = {}

    def feed(k, live):
        out, cnt = st.feed(dx[cuts[k]:cuts[k + 1]])
        torch.cuda.synchronize()
        for name, idx in live.items():
            got.setdefault(name, []).append(out[idx, :cnt[idx]].cpu().numpy())

    a = st.open_channel(1.0, bw, precise=False)
    a2 = st.open_channel(2.2, bw * 0.7, precise=False)
    feed(0, {"a": a, "a2": a2})                                     # no precise channel
    b = st.open_channel(3.0, bw, precise=True)
    feed(1, {"a": a, "a2": a2, "b": b})                             # some
    st.close_channel(a); st.close_channel(a2)
    b2 = st.open_channel(4.1, bw * 0.6, precise=True)
    feed(2, {"b": b, "b2": b2})                                     # all
    st.close_channel(b); st.close_channel(b2)
    c = st.open_channel(5.0, bw, precise=False)
    feed(3, {"c": c})                                               # none again
    st.close()
    want = {"a": (0, cuts[2], 1.0, bw, False), "a2": (0, cuts[2], 2.2, bw * 0.7, False),
            "b": (cuts[1] - H, cuts[3], 3.0, bw, True), "b2": (cuts[2] - H, cuts[3], 4.1, bw * 0.6, True),
            "c": (cuts[3] - H, cuts[4], 5.0, bw, False)}
    for name, (lo, hi, f0, w_, precise) in want.items():
        ref = sdo.specttuner_run_f32(x[lo:hi], f0, w_, 1.0, precise)
        g = np.concatenate(got[name])
        assert g.size == ref.size, (name, g.size, ref.size)
        assert np.array_equal(_bits(g), _bits(ref)), name


def test_full_size_block_all_64_channels_equal_the_binary32_statement(ctx, sdo):
    """BASELINE's C4 slice as the bench builds it: 4 Mi samples, 64 channels of 64 bins on the 90 kHz raster, time-major
    rows -- ALL 64 rows against the oracle, bit for bit (the oracle transforms the 2047 windows on the host's cores)"""
    L = 1 << 22
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_(generator=g)
    fn = synth.raster(64, 2 * 90e3 / 50e6)
    bw = 2 * np.pi * 0.75 / 64
    st = engine.SpectTuner(ctx, W)
    for f in fn:
        st.open_channel(np.pi * f % (2 * np.pi), bw)
    out, counts = st.feed(x, out=engine.time_major(64, L // 64 + 16, "cuda"))
    torch.cuda.synchronize()
    st.close()
    import os
    ref = sdo.specttuner_bank_f32(x.cpu().numpy(), [np.pi * f % (2 * np.pi) for f in fn], [bw] * 64, [1.0] * 64,
                                  threads=min(32, os.cpu_count() or 1))
    got = out[:, :counts[0]].cpu().numpy()
    for c in range(64):
        assert counts[c] == ref[c].size
        assert np.array_equal(_bits(got[c]), _bits(ref[c])), c


def test_reset_forgets_the_stream_position_and_closing_the_higher_channel_is_safe(ctx, sdo):
    """suamd_specttuner_reset (seek / gap): the next feed starts like a first feed; and the advisor's case -- two channels,
    the higher one closed, then a retune of the other (close + open reuses slots) -- with counts sized by
    suamd_specttuner_channel_capacity()"""
    x = cnoise(H * 20, 31)
    bw = 2 * np.pi / 64 * 0.8
    st = engine.SpectTuner(ctx, W)
    a = st.open_channel(1.0, bw)
    b = st.open_channel(2.0, bw)
    dx = torch.from_numpy(x).cuda()
    st.feed(dx[:H * 6])
    st.close_channel(b)
    assert st.capacity() == 2
    o, cnt = st.feed(dx[H * 6:H * 8])
    assert len(cnt) == 2 and cnt[b] == 0 and cnt[a] == 2 * 32
    for f in (1.5, 2.5):                                           # retune twice: index a -> (new slot) -> a again
        st.close_channel(a)
        a = st.open_channel(f, bw)
        o, cnt = st.feed(dx[H * 8:H * 10])
        assert len(cnt) == st.capacity() and cnt[a] == 2 * 32
    st.reset()
    o, cnt = st.feed(dx[H * 10:])
    torch.cuda.synchronize()
    got = o[a, :cnt[a]].cpu().numpy()
    ref = sdo.specttuner_run_f32(x[H * 10:], 2.5, bw, 1.0)
    assert cnt[a] == 9 * 32 and np.array_equal(_bits(got), _bits(ref))   # like a fresh tuner on the rest of the stream
    st.close()
    # the sigutils front end sizes its counts the same way (advisor r2: flush() wrote past the vector)
    t = sigutils.SpectTuner(W)
    k0 = t.open_channel(0.5, float(bw), 1.0)
    k1 = t.open_channel(1.5, float(bw), 1.0)
    t.feed(x[:H * 4])
    t.close_channel(k1)
    t.feed(x[H * 4:H * 8])
    assert t.samples(k0).size == 7 * 32
    t.close()


def test_two_wavefronts_per_window_equal_one_bit_for_bit(ctx, tune):
    """64-bin channels with one response run on specttuner_pair.hip (two wavefronts per window, every 64-point DFT split
    between them); SUAMD_ST_KERNEL=wave keeps them on the one-wavefront kernel.  Same samples bit for bit -- runs of two
    and three windows, the seam hand-off, a residual NCO, more channels than one workgroup serves, 64-bit row
    addressing, and whatever slot budget plans the launch."""
    x = cnoise(H * 96, 77)
    chans = [(0.2 + 0.085 * c, 2 * np.pi / 64 * 0.75, 1.0, bool(c % 5 == 0)) for c in range(70)]
    tune.setenv("SUAMD_ST_KERNEL", "wave")
    ref = run_gpu(ctx, x, chans, splits=[H * 40], run=3)
    tune.delenv("SUAMD_ST_KERNEL")
    for run, tm in ((2, False), (3, True), (7, False)):
        got = run_gpu(ctx, x, chans, splits=[H * 40], run=run, time_major=tm)
        for a, b in zip(ref, got):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (run, tm)
    tune.setenv("SUAMD_ST_Y32", "0")
    got = run_gpu(ctx, x, chans, splits=[H * 40], run=2)
    for a, b in zip(ref, got):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    tune.delenv("SUAMD_ST_Y32")
    # slot budgets: 64 slots -> runs of two windows on a 96-window stream, 1024 -> runs of one (the one-wavefront kernel)
    for slots in (64, 1024, 0):
        st = engine.SpectTuner(ctx, W)
        st.set_slots(slots)
        ids = [st.open_channel(*c) for c in chans]
        out, counts = st.feed(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        one = run_gpu(ctx, x, chans)
        for k, c in enumerate(ids):
            assert np.array_equal(out[c, :counts[c]].cpu().numpy().view(np.uint32), one[k].view(np.uint32)), slots
        st.close()
    with pytest.raises(engine.SigDiggerAmdError):
        st2 = engine.SpectTuner(ctx, W)
        try:
            st2.set_slots(5)
        finally:
            st2.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_banks_and_splits_two_wavefront_kernel_against_one(ctx, tune, seed):
    """Seeded fuzz of the kernel choice: random channel counts (partial last workgroup, several workgroups), run lengths,
    feed boundaries and residual NCOs -- the two-wavefront kernel and the one-wavefront kernel must agree on every bit,
    whichever mixture of them the run lengths select."""
    r = np.random.default_rng(1000 + seed)
    nch = int(r.integers(1, 150))
    nwin = int(r.integers(20, 120))
    x = cnoise(H * nwin, 500 + seed)
    dec = int(r.choice([64, 128, 256, 512]))                     # channels of 64 / 32 / 16 / 8 bins
    nch = nch * (dec // 64)                                      # a lane serves dec / 64 channels: fill a few workgroups at every size
    # pass-band widths: one for the whole bank, a handful (their response tables live in LDS), or one per channel (more
    # tables than the two-wavefront kernel holds: the bank falls back to the one-wavefront kernel)
    widths = [[0.75], [0.75, 0.6, 0.5, 0.9, 0.4], list(np.linspace(0.3, 0.95, 40))][int(r.integers(0, 3))]
    chans = [(float(r.uniform(0, 2 * np.pi)), 2 * np.pi / dec * float(r.choice(widths)), 1.0, bool(r.integers(0, 4) == 0)) for _ in range(nch)]
    cuts = sorted(set(int(c) * H for c in r.integers(1, nwin, size=int(r.integers(0, 4)))))
    tune.setenv("SUAMD_ST_KERNEL", "wave")
    ref = run_gpu(ctx, x, chans, splits=cuts, run=int(r.integers(1, 6)))
    tune.delenv("SUAMD_ST_KERNEL")
    for _ in range(2):
        run = int(r.integers(2, 9))
        got = run_gpu(ctx, x, chans, splits=cuts, run=run, time_major=bool(r.integers(0, 2)))
        for a, b in zip(ref, got):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (seed, dec, nch, nwin, cuts, run)


@pytest.mark.parametrize("dec", [64, 128, 256, 512])
def test_row_tables_with_32_bit_offsets_give_the_same_samples(ctx, tune, dec):
    """suamd_specttuner_feed_rows_near: one row per channel, all of one arena -- the kernels address them with 32-bit offsets
    from the lowest row (and every narrow size runs on the two-wavefront kernel).  Same samples as the view and as the
    plain row table, bit for bit; rows in scrambled order, a hole in the channel table (a closed channel), two feeds."""
    x = cnoise(H * 64, 900 + dec)
    dx = torch.from_numpy(x).cuda()
    nch = 70 * (dec // 64) + 3
    chans = [(0.1 + 6.0 * c / nch, 2 * np.pi / dec * (0.75 if c % 3 else 0.6), 1.0, bool(c % 7 == 0)) for c in range(nch)]
    cap = x.size // dec + 64
    outs = {}
    # "near": 64-bin channels leave through stp_kernel's LDS transposition (round 6, ROWT: sixteen lanes per 256 bytes of a
    # row); "near_lane_per_row": the same rows with tuning().st_row_stage = 0 (lane = channel, 8 bytes per row and store)
    for mode in ("view", "rows", "near", "near_lane_per_row"):
        tune.setenv("SUAMD_ST_ROW_STAGE", 0 if mode == "near_lane_per_row" else 1)
        st = engine.SpectTuner(ctx, W)
        st.set_run(3)
        ids = [st.open_channel(*c) for c in chans]
        st.close_channel(ids[5])
        arena = torch.zeros((nch, cap), dtype=torch.complex64, device="cuda")
        order = np.random.default_rng(3).permutation(nch)                    # channel c's row is arena[order[c]]
        got = [[] for _ in ids]
        for lo, hi in ((0, H * 24), (H * 24, x.size)):
            if mode == "view":
                out, counts = st.feed(dx[lo:hi])
                rows = [out[c] for c in ids]
            else:
                rows = [arena[order[c]] for c in ids]
                counts = st.feed_rows(dx[lo:hi], [None if c == ids[5] else rows[c] for c in ids], near=mode.startswith("near"))
            torch.cuda.synchronize()
            for c in ids:
                if c != ids[5]:
                    got[c].append(rows[c][:counts[c]].cpu().numpy())
        outs[mode] = [np.concatenate(g) if g else None for g in got]
        st.close()
    for c in range(nch):
        if c == 5:
            continue
        assert outs["view"][c].size == x.size // dec - (W // dec) // 2, c
        assert np.array_equal(outs["view"][c].view(np.uint32), outs["rows"][c].view(np.uint32)), (dec, c)
        assert np.array_equal(outs["view"][c].view(np.uint32), outs["near"][c].view(np.uint32)), (dec, c)
        assert np.array_equal(outs["view"][c].view(np.uint32), outs["near_lane_per_row"][c].view(np.uint32)), (dec, c)


@pytest.mark.parametrize("dec,nch", [(512, 5), (512, 64), (256, 70), (128, 3)])
def test_few_narrow_channels_with_their_own_responses_on_the_one_wavefront_kernel(ctx, tune, dec, nch):
    """A bank far smaller than one wavefront serves (64 lanes x dec / 64 channels) with a pass-band width per channel, kept on
    the one-wavefront kernel: it loads a response block for EVERY lane group, present or not, so the [block][bin][lane]
    table must hold whole wavefronts' worth.  Rounds 2-3 sized it by the channel count -- an over-read of up to 28 KiB that
    faulted when the table ended a mapped region (eight analyzer shards on one device).  suamd_specttuner_feed now checks
    the table against the launch; the samples equal the two-wavefront kernel's (tables in LDS) bit for bit."""
    x = cnoise(H * 40, 4242)
    widths = np.linspace(0.35, 0.9, 7)
    chans = [(0.1 + 6.0 * c / nch, 2 * np.pi / dec * float(widths[c % 7]), 1.0, bool(c % 3 == 0)) for c in range(nch)]
    tune.setenv("SUAMD_ST_KERNEL", "wave")
    one = run_gpu(ctx, x, chans, splits=[H * 13])
    tune.setenv("SUAMD_ST_Y32", "0")
    one64 = run_gpu(ctx, x, chans, splits=[H * 13])
    tune.delenv("SUAMD_ST_Y32")
    tune.delenv("SUAMD_ST_KERNEL")
    two = run_gpu(ctx, x, chans, splits=[H * 13])
    for a, b, c in zip(one, one64, two):
        assert a.size > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), c.view(np.uint32))


@pytest.mark.parametrize("dec,nch", [(16, 1), (16, 5), (16, 40), (32, 70), (8, 3), (1, 1), (4, 2)])
def test_wide_channels_without_the_warm_up_window(ctx, tune, dec, nch):
    """The workgroup kernel's two ways over the seam of consecutive runs: a warm-up window per run (SUAMD_ST_SEAM=0: the
    window before the run is transformed again for its second half) and the hand-off (SUAMD_ST_SEAM=1: the run leaves its
    last second half in a buffer, st_seam_kernel completes the next run's first block) -- the same samples bit for bit,
    with and without a residual NCO, channel-major and time-major output, runs of 1, 2 and 5 windows, feeds cut at
    arbitrary half windows, one or two channel groups per workgroup."""
    x = cnoise(H * 47, 900 + dec + nch)
    bw = 2 * np.pi * 0.75 / dec
    chans = [(0.3 + 6.0 * c / max(nch, 2), bw * (0.6 + 0.4 * (c % 3) / 2), 1.0, bool(c % 2)) for c in range(nch)]
    def seam_launches(fn):
        """fn() with the library's kernel timer on: its result and how many st_seam_kernel launches it made"""
        engine.kernel_timing_read()
        engine.kernel_timing(True)
        try:
            out = fn()
            torch.cuda.synchronize()
        finally:
            engine.kernel_timing(False)
        n = engine.kernel_timing_read("st_seam_kernel")["launches"]
        engine.kernel_timing_read()
        return out, n
    for run in (1, 2, 5, None):
        for tm in (False, True):
            # (the knob is read on every feed -- ADVICE r4: cached in a function-local static it compared a mode with itself)
            tune.setenv("SUAMD_ST_SEAM", "0")
            ref, n0 = seam_launches(lambda: run_gpu(ctx, x, chans, splits=[H * 9, H * 10, H * 31], run=run, time_major=tm))
            tune.setenv("SUAMD_ST_SEAM", "1")
            got, n1 = seam_launches(lambda: run_gpu(ctx, x, chans, splits=[H * 9, H * 10, H * 31], run=run, time_major=tm))
            tune.delenv("SUAMD_ST_SEAM")
            assert n0 == 0, "SUAMD_ST_SEAM=0 still launched the seam kernel"
            if run in (1, 2, 5):
                assert n1 > 0, "SUAMD_ST_SEAM=1 never launched the seam kernel: the two runs took the same path"
            for a, b in zip(ref, got):
                assert a.size == b.size > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (run, tm)


def test_mixed_feed_narrow_channels_to_a_slab_wide_ones_to_rows(ctx):
    """suamd_specttuner_feed_mixed (the live analyzer's feed): channels of <= 64 bins land as columns of a time-major slab,
    wider ones in their own rows; the samples are the plain view feed's, bit for bit; a closed channel's column is left
    alone; two feeds."""
    x = cnoise(H * 64, 4242)
    dx = torch.from_numpy(x).cuda()
    decs = [64, 64, 16, 128, 64, 4, 512, 64, 64, 32, 64]
    chans = [(0.1 + 0.5 * c, 2 * np.pi / d * 0.7, 1.0, bool(c % 4 == 0)) for c, d in enumerate(decs)]
    ref = engine.SpectTuner(ctx, W)
    mix = engine.SpectTuner(ctx, W)
    for st in (ref, mix):
        st.set_run(3)
    ids = [ref.open_channel(*c) for c in chans]
    assert ids == [mix.open_channel(*c) for c in chans]
    ref.close_channel(ids[4]); mix.close_channel(ids[4])
    pitch = 64
    cap = x.size // min(decs) + 64
    slab = torch.full((x.size // 64 + 64, pitch), 7.0 + 0j, dtype=torch.complex64, device="cuda")
    arena = torch.zeros((len(ids), cap), dtype=torch.complex64, device="cuda")
    got_r, got_m = [[] for _ in ids], [[] for _ in ids]
    for lo, hi in ((0, H * 24), (H * 24, x.size)):
        out, counts = ref.feed(dx[lo:hi])
        rows = [None if (W // decs[c] <= 64 or c == ids[4]) else arena[c] for c in ids]
        cm = mix.feed_mixed(dx[lo:hi], slab, 64, rows)
        torch.cuda.synchronize()
        assert list(cm) == list(counts)
        for c in ids:
            if c == ids[4]:
                continue
            got_r[c].append(out[c][:counts[c]].cpu().numpy())
            got_m[c].append((slab[:cm[c], c] if W // decs[c] <= 64 else arena[c][:cm[c]]).cpu().numpy())
    for c in ids:
        if c == ids[4]:
            continue
        assert np.array_equal(np.concatenate(got_r[c]).view(np.uint32), np.concatenate(got_m[c]).view(np.uint32)), (c, decs[c])
    assert bool((slab[:, ids[4]] == 7.0).all()), "a closed channel's column is not written"
    assert bool((slab[:, len(ids):] == 7.0).all()), "columns beyond the channel table are not written"
    ref.close(); mix.close()
