"""BASELINE.json's full sizes (C4 slice: 4 Mi-sample blocks, 64 inspectors, D = 64, 255 taps, 8192-pt PSD; C5:
frames over a capture), checked through size-independent properties -- block-split invariance, exact power-of-two
scaling, Parseval, frame independence -- plus oracle spot checks on windows the oracle finishes in milliseconds."""
import numpy as np
import pytest
import torch

from sigdigger_amd import engine, synth

pytestmark = pytest.mark.gpu

L = 1 << 22
C, D, T = 64, 64, 255


def bits_equal(a, b):
    if a.is_complex():
        a, b = torch.view_as_real(a), torch.view_as_real(b)
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


@pytest.fixture(scope="module")
def block():
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_(generator=g)
    return x


def test_fir_bank_full_size_properties(ctx, sdo, block):
    fn = synth.raster(C, 2 * 90e3 / 50e6)
    taps = ctx.lpf_design(T, 0.75 / D)
    one = engine.ChannelBank(ctx, fn, D, taps).feed(block, out=engine.time_major(C, L // D + 4, "cuda"))
    assert one.shape == (C, L // D)
    # block-split invariance: the stream may be cut anywhere (also off the decimation grid)
    b2 = engine.ChannelBank(ctx, fn, D, taps)
    cut = (1 << 21) + 37
    parts = [b2.feed(block[:cut]), b2.feed(block[cut:])]
    assert bits_equal(torch.cat(parts, dim=1).contiguous(), one.contiguous())
    # scaling by a power of two is exact in binary32, through every fma of the chain and the de-rotation
    two = engine.ChannelBank(ctx, fn, D, taps).feed(block * 4.0)
    assert bits_equal(two.contiguous(), (one * 4.0).contiguous())
    # oracle spot checks: a few (channel, output range) windows
    xh = block.cpu().numpy()
    tp = sdo.lpf_design(T, 0.75 / D)
    for c, m0 in ((0, 0), (17, 4000), (63, L // D - 300)):
        dp = sdo.fnor_to_dphase(-fn[c])
        n_first = m0 * D
        lo = max(n_first - (T - 1), 0)
        hist = np.zeros(T - 1, np.complex64)
        hist[T - 1 - (n_first - lo):] = xh[lo:n_first]
        ref = sdo.chan_feed(hist, xh[n_first:n_first + 256 * D], n_first, sdo.chan_modulate_taps(tp, dp), D, 0, dp)
        got = one[c, m0:m0 + ref.size].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (c, m0)


def test_psd_full_size_properties(ctx, block):
    n = 8192
    psd = engine.PSD(ctx, n, engine.WINDOW_NONE)
    frames = psd.feed(block, nframes=L // n, navg=1, scale=1.0 / n)
    # Parseval per frame: sum_k |X_k|^2 / N = sum_n |x_n|^2
    e_time = (block.view(L // n, n).abs() ** 2).sum(dim=1, dtype=torch.float64)
    e_freq = frames.sum(dim=1, dtype=torch.float64)
    assert torch.max(torch.abs(e_freq / e_time - 1)).item() < 2e-6
    # frames are independent: any sub-range of the block gives the same frames, bit for bit
    sub = psd.feed(block[100 * n:], nframes=50, navg=1, scale=1.0 / n)
    assert bits_equal(sub, frames[100:150].contiguous())
    # x -> 2x is exact: every bin scales by exactly 4
    assert bits_equal(psd.feed(block * 2.0, nframes=L // n, navg=1, scale=1.0 / n), frames * 4.0)
    # Welch averaging (split over workgroups + deterministic reduce) against the mean of the single frames
    avg = psd.feed(block, nframes=L // n, navg=256, scale=1.0 / n)
    ref = frames.view(2, 256, n).to(torch.float64).mean(dim=1)
    assert torch.max(torch.abs(avg.to(torch.float64) - ref) / ref.max()).item() < 1e-6
    assert bits_equal(avg, psd.feed(block, nframes=L // n, navg=256, scale=1.0 / n))       # run-to-run identical


def test_recurrence_chain_full_size_split_invariance(ctx):
    """64 channels x 65536 samples through AGC -> Costas -> Gardner, once in one block and once in three."""
    M = L // D
    x = synth.psk_carriers(M, [0.0], sps=16, order=4, seed=2)
    rows = np.stack([np.roll(x, 97 * c) * np.exp(1j * 0.1 * c) for c in range(C)]).astype(np.complex64)
    xt = engine.time_major(C, M, "cuda")
    xt.copy_(torch.from_numpy(rows).cuda())

    def run(cuts):
        agc = engine.AGCBank(ctx, C, tau=16.0)
        cos = engine.CostasBank(ctx, C, 2, 0.0, 2.0 / 16, 3, 0.005)
        clk = engine.ClockBank(ctx, C, 0.2, 1.0 / 16)
        sym = torch.zeros((C, M // 8), dtype=torch.complex64, device="cuda")
        cnt = torch.zeros(C, dtype=torch.int32, device="cuda")
        outs = []
        for a, b in cuts:
            z = cos.feed(agc.feed(xt[:, a:b], out=engine.time_major(C, b - a, "cuda")), out=engine.time_major(C, b - a, "cuda"))
            clk.feed(z, sym, cnt)
            outs.append(z)
        return torch.cat(outs, dim=1), sym, cnt

    z1, s1, c1 = run(((0, M),))
    z3, s3, c3 = run(((0, 1), (1, 30001), (30001, M)))
    assert bits_equal(z1.contiguous(), z3.contiguous())
    assert torch.equal(c1, c3) and bits_equal(s1, s3)
    n = c1.cpu().numpy()
    assert np.all(np.abs(n - M / 16) <= 3)                         # one symbol per 16 samples
    tail = s1[:, int(n.min()) // 2:int(n.min())].cpu().numpy()
    assert np.all(np.abs(np.mean((tail / np.abs(tail)) ** 4, axis=1)) > 0.6)        # every channel locked (QPSK)


def test_capture_psd_frames_do_not_depend_on_the_launch_shape(ctx):
    """C5 shape on a 2^27-sample capture: dwell spectra from one launch over the capture equal those from launches
    over its halves (workgroup assignment and splitting must not leak into the result)."""
    n, tile, total = 8192, 256, 1 << 27
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.empty(total, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_(generator=g)
    psd = engine.PSD(ctx, n)
    dwells = total // (n * tile)
    whole = psd.feed(x, nframes=dwells * tile, navg=tile, scale=1.0 / n, mode=engine.PSD_DB_SHIFTED)
    h = total // 2
    halves = torch.cat([psd.feed(x[:h], nframes=dwells * tile // 2, navg=tile, scale=1.0 / n, mode=engine.PSD_DB_SHIFTED),
                        psd.feed(x[h:], nframes=dwells * tile // 2, navg=tile, scale=1.0 / n, mode=engine.PSD_DB_SHIFTED)])
    assert whole.shape == (dwells, n)
    # different splits sum the same frames in a different association: equal to rounding, not bitwise
    assert torch.max(torch.abs(whole - halves)).item() < 1e-4                                      # dB
    assert abs(whole.mean().item() - 10 * np.log10(2 * 0.35875 ** 2 + 2 * (0.48829 ** 2 + 0.14128 ** 2 + 0.01168 ** 2) / 2)) < 0.1


@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_capture_sized_psd_input_read_with_the_streaming_policy_gives_the_same_bits(ctx, n):
    """An input above 128 MiB is requested with the streaming cache policy, a smaller one with the default policy
    (psd.hip, launch_psd): 2^25 samples in one launch equal the two 2^24-sample halves bit for bit (every output has its
    own workgroup in both cases, so the arithmetic is the same)."""
    total, navg = 1 << 25, 2
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    x = torch.empty(total, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_(generator=g)
    psd = engine.PSD(ctx, n)
    nf = total // n
    whole = psd.feed(x, nframes=nf, navg=navg, scale=1.0 / n)
    h = total // 2
    halves = torch.cat([psd.feed(x[:h], nframes=nf // 2, navg=navg, scale=1.0 / n),
                        psd.feed(x[h:], nframes=nf // 2, navg=navg, scale=1.0 / n)])
    assert whole.shape == (nf // navg, n)
    assert bits_equal(whole, halves)
    assert abs(whole.mean().item() - 2 * (0.35875 ** 2 + (0.48829 ** 2 + 0.14128 ** 2 + 0.01168 ** 2) / 2)) < 0.01
