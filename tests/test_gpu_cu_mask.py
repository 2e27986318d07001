"""CU-masked streams (suamd_stream_new_cu_mask) and the AGC bank on two streams (suamd_agc_bank_feed_split): round 5's
CU-partition option of the stream pipeline (sigdigger_amd/pipeline.py, SUAMD_PIPELINE_CU_PARTITION)."""
import numpy as np
import pytest
import torch

from sigdigger_amd import engine, pipeline, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    torch.cuda.init()
    c = engine.Context(0)
    yield c
    c.close()


def test_a_cu_mask_confines_a_stream_and_consecutive_bits_sit_on_consecutive_xcds(ctx):
    n = ctx.cu_count()
    assert n == 256
    reserved, transform = pipeline.cu_partition(n, 1)
    assert reserved == list(range(8)) and len(transform) == 248
    s_res, s_tr = ctx.masked_stream(reserved), ctx.masked_stream(transform)
    try:
        w_res = ctx.probe_placement(s_res, 512, 20000)
        w_tr = ctx.probe_placement(s_tr, 8192, 20000)
        # one CU per XCD: bit i -> XCD i mod 8 (WHICH physical (se, cu) an XCD's first enabled CU is depends on the die's
        # harvesting: most boxes of the pool read (x, 0, 0) for every x, one read (1, 0, 1))
        assert len(set(w_res)) == 8 and sorted(x for x, _, _ in set(w_res)) == list(range(8)), sorted(set(w_res))
        assert len(set(w_tr)) == 248 and not (set(w_tr) & set(w_res))
        # the dispatcher deals the workgroups of a launch to the XCDs in turn whatever CUs they have enabled
        per_xcd = np.bincount([x for x, _, _ in w_tr], minlength=8)
        assert per_xcd.tolist() == [1024] * 8
    finally:
        ctx.destroy_stream(s_res)
        ctx.destroy_stream(s_tr)


def test_an_empty_mask_is_refused(ctx):
    import ctypes as C
    words = (C.c_uint32 * 8)()
    assert not ctx.lib.suamd_stream_new_cu_mask(ctx.h, words, 8)


@pytest.mark.parametrize("nchan", [1, 6, 64])
def test_the_agc_bank_on_two_streams_gives_the_same_samples(ctx, nchan):
    m, D = 20000, 16
    fn = synth.raster(nchan, 0.9 / max(nchan, 2))
    x = synth.psk_carriers((m + 1) * D, fn, sps=8 * D, order=4, seed=33, snr_db=20)
    y = torch.from_numpy(np.stack([np.roll(x[c % D::D][:m], 37 * c) * (0.1 + c) for c in range(nchan)]).astype(np.complex64)).cuda()
    ytm = engine.time_major(nchan, m, "cuda")
    ytm.copy_(y)
    one = engine.AGCBank(ctx, nchan, tau=8.0)
    two = engine.AGCBank(ctx, nchan, tau=8.0)
    s_level, s_wide = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    a1 = engine.time_major(nchan, m, "cuda")
    a2 = engine.time_major(nchan, m, "cuda")
    cuts = [0, 1, 4097, 12000, m]                               # the state carries over from call to call on both paths
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        one.feed(ytm[:, lo:hi], out=a1[:, lo:hi])
        two.feed(ytm[:, lo:hi], out=a2[:, lo:hi], stream=s_level, wide=s_wide)
    torch.cuda.synchronize()
    bits = lambda t: np.ascontiguousarray(t.cpu().numpy()).view(np.uint32)
    assert np.array_equal(bits(a1), bits(a2))
    assert float(a1.abs().max()) > 0
