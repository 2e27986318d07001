"""The live analyzer on its default channeliser: the FFT filter bank with su_specttuner's semantics (SPEC.md C2) -- one
forward FFT of every block shared by all open inspectors, as libsuscan does it -- through the suscan_analyzer_* ABI.
Channel samples are compared with the oracle's restatements: the binary64 one to 1e-5 and, since round 3, the binary32
one bit for bit -- so the stages behind the channel (AGC, Costas, Gardner) are compared exactly on this path too."""
import ctypes as C

import numpy as np
import pytest

from sigdigger_amd import suscan, synth
from tests.test_gpu_analyzer import FS, L, N, _pump, _start

pytestmark = pytest.mark.gpu
W, H = 4096, 2048
TOL = 1e-5


def _relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def _chan_params(fc, bw):
    D = 1
    while D * 2 <= FS / (2 * bw) and D < 4096:
        D *= 2
    f0 = (2 * np.pi * fc / FS) % (2 * np.pi)
    return D, f0, 2 * np.pi * bw / FS, FS / (D * bw)


def _build_rccl_standin(tmp_path):
    """tests/rccl_standin.cpp -> a shared library with RCCL's two entry points and single-process semantics on ONE device"""
    import os
    import subprocess
    so = str(tmp_path / "librccl_standin.so")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_standin.cpp")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src, "-lpthread"])
    return so


def _check_raw_channel(sdo, x, got, k, D, f0, bwa, guard, nblocks, last_open):
    """the samples a raw inspector delivered from its first block to the end of the capture, against the oracle"""
    # a shard's filter bank starts with the first block that finds an inspector on it; every later block of the
    # capture is delivered, so the count of samples tells which block that was: (nblocks - b) L / H - 1 channel blocks
    b = nblocks - (got.size // (W // D // 2) + 1) * H // L
    assert 0 <= b <= last_open + 2
    # (the first batches may have gone out under inspector id 0, before the id hand-shake -- the consumer answers the OPEN
    # message while the worker runs on: then the count is a block or two short and the bank started that much earlier;
    # compare the tails for each reading)
    errs = []
    for bb in range(b, max(b - 4, -1), -1):
        ref = sdo.specttuner_run(x[bb * L:], W, f0, bwa, guard, precise=(k % 2 == 0))
        n = min(got.size, ref.size)
        if n > 0.6 * ref.size:
            errs.append(_relerr(got[-n:], ref[-n:]))
    assert errs and min(errs) <= TOL, (k, b, errs)


def _real_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("devices", [None, "0,0", "0,0,0", "0,0,0:rccl-standin", "0,0:rccl-real-refuses", "0,1:rccl"])
def test_raw_inspectors_share_one_forward_fft_and_match_the_oracle(tmp_path, sdo, monkeypatch, devices):
    """devices: SUAMD_DEVICES.  "0,0" / "0,0,0" run two / three GPU shards (csrc/analyzer.cpp: BlockBus) on the one GPU of
    the test box: inspector handle h lives on shard h mod G, every shard gets every block from shard 0's pinned buffer
    and posts to the same queue -- channel for channel the same samples as on one shard, one PSD stream.
    ":rccl-standin": the block reaches the shards through the analyzer's RCCL branch (SUAMD_ANALYZER_BCAST=rccl: one
    ncclBroadcast per block rooted at shard 0) served by tests/rccl_standin.cpp on the one device -- setup_rccl, the
    root's call and the shards' matching calls all execute; "0,1:rccl": the real librccl over two real GPUs (skipped
    on a one-GPU box); "0,0:rccl-real-refuses": the REAL librccl is loaded and asked for two ranks on the one device, which it
    refuses (ncclCommInitAll fails) -- the analyzer says so and the blocks travel as per-GPU host copies: the dlopen, the
    symbol lookups and the refusal path run against the library the production box has."""
    standin = None
    refuses = False
    if devices and devices.endswith(":rccl-real-refuses"):
        devices = devices.split(":")[0]
        refuses = True
        monkeypatch.delenv("SUAMD_RCCL_LIB", raising=False)
        monkeypatch.setenv("SUAMD_RCCL_ALLOW_SAME_DEVICE", "1")
        monkeypatch.setenv("SUAMD_ANALYZER_BCAST", "rccl")
    if devices and devices.endswith(":rccl-standin"):
        devices = devices.split(":")[0]
        so = _build_rccl_standin(tmp_path)
        standin = C.CDLL(so)                                       # kept loaded: its counters outlive the analyzer's dlclose
        standin.standin_broadcasts.restype = C.c_ulonglong
        standin.standin_bytes.restype = C.c_ulonglong
        monkeypatch.setenv("SUAMD_RCCL_LIB", so)
        monkeypatch.setenv("SUAMD_RCCL_ALLOW_SAME_DEVICE", "1")
        monkeypatch.setenv("SUAMD_ANALYZER_BCAST", "rccl")
    elif devices and devices.endswith(":rccl"):
        devices = devices.split(":")[0]
        if _real_gpus() < 2:
            pytest.skip("the real RCCL broadcast needs two GPUs")
        # a box of the pool may show GPUs that belong to other tenants: the second device must be ours to use
        import torch
        try:
            free, total = torch.cuda.mem_get_info(1)
            torch.zeros(1 << 20, device="cuda:1").sum().item()
        except Exception as e:                                         # noqa: BLE001 -- whatever the runtime says
            pytest.skip(f"the second GPU is not usable from here: {e!r}")
        if free < (16 << 30) or free < 0.5 * total:
            pytest.skip(f"the second GPU is busy with somebody else's work ({free >> 30} of {total >> 30} GiB free)")
        monkeypatch.setenv("SUAMD_ANALYZER_BCAST", "rccl")
    if devices:
        monkeypatch.setenv("SUAMD_DEVICES", devices)
    nblocks = 10
    chans = [(125e3, 40e3), (-200e3, 40e3), (310e3, 9e3), (0.0, 300e3), (-50e3, 2.5e3)]          # D = 8, 8, 32, 1, 128
    x = synth.psk_carriers(L * nblocks, [2 * c[0] / FS for c in chans], sps=64, order=4, seed=8, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    for k, (fc, bw) in enumerate(chans):
        ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
        assert Lb.suscan_analyzer_open_ex_async(an, b"raw", C.byref(ch), int(k % 2 == 0), -1, 100 + k)
    st = {"psd": 0, "open_at": {}, "id": {}, "samples": {}, "efs": {}, "handles": {}, "status": []}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t in (suscan.MSG_INTERNAL, suscan.MSG_READ_ERROR, suscan.MSG_SOURCE_INIT) and ptr:
            # what the shards say about themselves (a GPU that could not be initialised, RCCL falling back to copies ...):
            # shown with every assertion below, so that a failure on a box this was never run on explains itself
            m = C.cast(ptr, C.POINTER(suscan.StatusMsg)).contents
            st["status"].append((t, m.code, (m.err_msg or b"").decode("utf-8", "replace")))
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                k = m.req_id - 100
                st["open_at"][k] = st["psd"]
                st["efs"][k] = m.equiv_fs
                st["handles"][k] = m.handle
                assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, 500 + k, 0)
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            a = np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64)
            st["samples"].setdefault(m.inspector_id - 500, []).append(a)

    _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert len(st["open_at"]) == len(chans) and st["psd"] == nblocks, (st["open_at"], st["psd"], st["status"])   # one PSD stream whatever the shard count
    if refuses:
        assert any("ncclCommInitAll failed" in m for _, _, m in st["status"]), st["status"]
    if standin is not None:
        # every block went out as ONE broadcast of the block's bytes, rooted at shard 0
        assert standin.standin_broadcasts() == nblocks and standin.standin_bytes() == nblocks * L * 8
    G = len(devices.split(",")) if devices else 1
    assert len(set(st["handles"].values())) == len(chans), st["status"]
    assert sorted(h % G for h in st["handles"].values()) == sorted(k % G for k in range(len(chans))), st["status"]   # dealt round the shards
    if G == 1:
        b0 = st["open_at"][0]
        assert all(b == b0 for b in st["open_at"].values()), "the five requests were posted together"
    assert max(st["open_at"].values()) < nblocks - 4
    for k, (fc, bw) in enumerate(chans):
        D, f0, bwa, guard = _chan_params(fc, bw)
        assert abs(st["efs"][k] - FS / D) < 1e-3
        _check_raw_channel(sdo, x, np.concatenate(st["samples"][k]), k, D, f0, bwa, guard, nblocks, max(st["open_at"].values()))


def test_a_shard_that_dies_mid_broadcast_does_not_hang_the_analyzer(tmp_path, sdo, monkeypatch):
    """VERDICT r4 #4.  Three shards, blocks exchanged by ncclBroadcast (the stand-in: a missing rank leaves the root's stream in
    a kernel that only ncclCommAbort releases, as librccl does).  Shard 1 dies at block 6 AFTER the publisher has chosen the
    broadcast for that block (SUAMD_ANALYZER_FAULT).  The root's watchdog must notice that its broadcast does not complete,
    abort its communicator, switch the bus to per-GPU host copies and report the dead shard -- and the analyzer must run to
    the end of the capture with the surviving shards' inspectors delivering every sample, exact against the oracle."""
    so = _build_rccl_standin(tmp_path)
    standin = C.CDLL(so)
    for f in ("standin_broadcasts", "standin_aborts", "standin_orphans"):
        getattr(standin, f).restype = C.c_ulonglong
    monkeypatch.setenv("SUAMD_RCCL_LIB", so)
    monkeypatch.setenv("SUAMD_RCCL_ALLOW_SAME_DEVICE", "1")
    monkeypatch.setenv("SUAMD_ANALYZER_BCAST", "rccl")
    monkeypatch.setenv("SUAMD_ANALYZER_BCAST_TIMEOUT_MS", "300")
    monkeypatch.setenv("STANDIN_GRACE_MS", "150")
    monkeypatch.setenv("SUAMD_DEVICES", "0,0,0")
    monkeypatch.setenv("SUAMD_TEST_HOOKS", "1")
    monkeypatch.setenv("SUAMD_ANALYZER_FAULT", "shard_dies:1:6")
    nblocks, dies_at = 14, 6
    chans = [(125e3, 40e3), (-200e3, 40e3), (310e3, 9e3), (0.0, 300e3), (-50e3, 2.5e3), (220e3, 40e3)]
    x = synth.psk_carriers(L * nblocks, [2 * c[0] / FS for c in chans], sps=64, order=4, seed=8, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    for k, (fc, bw) in enumerate(chans):
        ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
        assert Lb.suscan_analyzer_open_ex_async(an, b"raw", C.byref(ch), int(k % 2 == 0), -1, 100 + k)
    st = {"psd": 0, "open_at": {}, "samples": {}, "handles": {}, "status": [], "eos": 0}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_EOS:
            st["eos"] += 1
        elif t in (suscan.MSG_INTERNAL, suscan.MSG_READ_ERROR) and ptr:
            m = C.cast(ptr, C.POINTER(suscan.StatusMsg)).contents
            st["status"].append((t, m.code, (m.err_msg or b"").decode("utf-8", "replace")))
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                k = m.req_id - 100
                st["open_at"][k] = st["psd"]
                st["handles"][k] = m.handle
                assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, 500 + k, 0)
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            a = np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64)
            st["samples"].setdefault(m.inspector_id - 500, []).append(a)

    import time
    t0 = time.time()
    seen = _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    took = time.time() - t0
    texts = [m for _, _, m in st["status"]]
    assert seen[-1] == suscan.MSG_HALT and st["eos"] == 1 and st["psd"] == nblocks, (st["psd"], texts)     # ran to the end
    assert took < 15.0, f"{took:.1f} s: the analyzer sat in a stuck broadcast"
    assert len(st["open_at"]) == len(chans) and max(st["open_at"].values()) < dies_at - 1, (st["open_at"], texts)
    # the root noticed, aborted and said so; the dead shard was reported as a read error; no broadcast after that
    assert any("ncclBroadcast did not complete" in m and "shard 0" in m for m in texts), texts
    assert any(t == suscan.MSG_READ_ERROR and "shard 1 is gone" in m for t, _, m in st["status"]), st["status"]
    assert standin.standin_orphans() == 1 and standin.standin_aborts() >= 1
    assert standin.standin_broadcasts() == dies_at, standin.standin_broadcasts()      # blocks 0 .. dies_at-1 went out whole
    for k, (fc, bw) in enumerate(chans):
        D, f0, bwa, guard = _chan_params(fc, bw)
        got = np.concatenate(st["samples"][k])
        if st["handles"][k] % 3 == 1:                            # on the dead shard: nothing after the block it died at
            assert got.size <= (dies_at - st["open_at"][k] + 2) * L // D, (k, got.size)
            continue
        _check_raw_channel(sdo, x, got, k, D, f0, bwa, guard, nblocks, max(st["open_at"].values()))


def test_psk_chain_behind_the_fft_channel_and_config_change_keeps_the_channel(tmp_path, sdo):
    nblocks = 14
    fc, baud, bw = 125e3, 15625.0, 40e3
    x = synth.psk_carriers(L * nblocks, [2 * fc / FS], sps=int(FS / baud), order=4, seed=8, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 77)
    st = {"psd": 0, "samples": [], "open_at": None, "cfg_at": None}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                st["open_at"] = st["psd"]
                cfg = Lb.suscan_config_dup(m.config)
                Lb.suscan_config_set_integer(cfg, b"afc.costas-order", 2)
                Lb.suscan_config_set_float(cfg, b"afc.loop-bw", 40.0)
                Lb.suscan_config_set_integer(cfg, b"clock.type", 1)
                Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
                Lb.suscan_config_set_float(cfg, b"clock.gain", 0.2)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 82)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg_at"] = st["psd"]
                st["samples"] = []
        elif t == suscan.MSG_SAMPLES and st["cfg_at"] is not None:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    bs, b0 = st["open_at"], st["cfg_at"]
    assert bs is not None and b0 is not None and b0 < nblocks - 5
    D, f0, bwa, guard = _chan_params(fc, bw)
    # the channel opened at block bs and was NOT re-opened by the configuration change at b0: the new stages take its
    # stream from where it is.  Channel blocks delivered before b0: (b0 - bs) L / H - 1
    # the oracle's binary32 statement of the channeliser (SPEC.md C2): the channel samples are the device's bit for bit, so
    # everything behind them -- AGC, Costas, Gardner, with their hard decisions -- is compared exactly (round 2 had to
    # allow 2e-3 here: its only statement of the channeliser was the binary64 one)
    y = sdo.specttuner_run_f32(x[bs * L:], f0, bwa, guard, precise=True)
    skip = ((b0 - bs) * L // H - 1) * (W // D // 2) if b0 > bs else 0
    y = y[skip:]
    sps = (FS / D) / baud
    a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), y)
    z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, min(2.0 / sps, 0.95), 3, 2 * 40.0 / (FS / D)), a)
    ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, baud / (FS / D)), z)
    got = np.concatenate(st["samples"])
    assert got.size == ref.size
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    n = got.size
    tail = got[n // 2:n]
    assert abs(np.mean((tail / np.abs(tail)) ** 4)) > 0.7          # a locked QPSK constellation


def test_blocks_that_are_not_whole_half_windows_fall_back_to_the_fir_channeliser(tmp_path, sdo):
    """window 512 x 9 frames per update = 4608-sample blocks: not a multiple of 2048 -> translate + FIR per inspector"""
    nblocks = 40
    n, navg = 512, 9
    Lb_ = n * navg
    fc, bw = 100e3, 40e3
    x = synth.psk_carriers(Lb_ * nblocks, [2 * fc / FS], sps=64, order=4, seed=3, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb = suscan.load()
    mq = suscan.MQ()
    assert Lb.suscan_mq_init(C.byref(mq))
    cfg = Lb.suscan_source_config_new(b"file", 1)
    Lb.suscan_source_config_set_samp_rate(cfg, FS)
    assert Lb.suscan_source_config_set_path(cfg, str(path).encode())
    p = suscan.AnalyzerParams.default()
    p.detector_params.window_size = n
    p.psd_update_int = Lb_ / FS
    an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
    assert an
    Lb.suscan_source_config_destroy(cfg)
    Lb.suscan_analyzer_set_throttle_async(an, 2 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=0)
    assert Lb.suscan_analyzer_open_async(an, b"raw", C.byref(ch), 5)
    st = {"psd": 0, "open_at": None, "samples": []}
    while True:
        t, ptr = suscan.read_message(Lb, mq, 60.0)
        if t == suscan.MSG_HALT:
            break
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                st["open_at"] = st["psd"]
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))
        Lb.suscan_analyzer_dispose_message(t, ptr)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    b0 = st["open_at"]
    assert b0 is not None and b0 < nblocks - 5
    D = 8
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    ref = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * Lb_:], 0, sdo.chan_modulate_taps(sdo.lpf_design(255, bw / FS), dp), D, 0, dp)
    got = np.concatenate(st["samples"])
    assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n", [512, 1024, 4096, 16384])
def test_channel_detector_noise_floor_is_the_exact_median_and_the_walk_equals_the_oracle(sdo, ctx, n):
    """chandet.hip: the noise floor is the element of rank n / 2 -- a radix SELECT on the device, a sort in the oracle and in
    numpy: the same element whatever the values look like (ties, one value only, few distinct values, forty decades of
    dynamic range, negative numbers and zeros, which a power spectrum never has); and the channel walk, which reads the
    spectrum from LDS, gives the oracle's records bit for bit at the smallest and the largest size"""
    import torch
    from sigdigger_amd import engine
    rng = np.random.default_rng(n)
    cases = {
        "chi2": (rng.chisquare(8, n) / 8).astype(np.float32),
        "ties": rng.integers(0, 4, n).astype(np.float32),
        "one value": np.full(n, 0.25, np.float32),
        "two values": np.where(np.arange(n) % 2 == 0, 1.0, 2.0).astype(np.float32),
        "decades": (10.0 ** rng.uniform(-30, 30, n)).astype(np.float32),
        "signed": np.concatenate([rng.standard_normal(n - 8), np.zeros(8)]).astype(np.float32),
        "sorted": np.arange(n, dtype=np.float32),
        "reversed": np.arange(n, 0, -1).astype(np.float32),
    }
    for name, P in cases.items():
        det = engine.ChannelDetector(ctx, n, alpha=0.2, beta=0.0, gamma=0.5, snr=4.0)
        det.feed(torch.from_numpy(P).cuda())
        want = np.sort(P)[n // 2]
        got = np.float32(det.noise_floor())
        assert got.view(np.uint32) == want.view(np.uint32) or (got == want == 0), (name, got, want)
    # several updates: the smoothed floor and the records against the oracle object
    det = engine.ChannelDetector(ctx, n, alpha=0.2, beta=0.0, gamma=0.5, snr=4.0)
    ora = sdo.ChannelDetector(n, 0.2, 0.5, 4.0)
    base = np.full(n, 1e-3, np.float32)
    for lo, hi, lvl in ((n // 16, n // 16 + n // 64, 0.05), (n // 4, n // 4 + 3, 0.2), (n // 2 - 5, n // 2 + 9, 0.01), (n - n // 8, n - n // 8 + 1, 1.0),
                        (n - n // 16, n - 1, 0.004), (0, 6, 0.3)):
        base[lo:hi] += lvl
    for k in range(4):
        P = np.roll((base * rng.chisquare(8, n).astype(np.float32) / 8).astype(np.float32), n // 2)
        det.feed(torch.from_numpy(P).cuda())
        ora.feed(P)
        assert det.noise_floor() == np.float32(ora.N0), k
    got, ref = det.channels(1e6), ora.find()
    assert len(got) == len(ref) >= 4
    df = 1e6 / n
    for g, (first, last, width, peak, s_, ws) in zip(got, ref):
        assert g["f_lo"] == (first - n / 2 - 0.5) * df and g["f_hi"] == (last - n / 2 + 0.5) * df and g["fc"] == (ws / s_ - n / 2) * df


def test_channel_detector_bit_exact_and_channel_messages(tmp_path, sdo, ctx):
    """Row N1: su_channel_detector on the device -- (a) the object against the oracle on the same frames: smoothed spectrum,
    noise floor and every channel record bit for bit; (b) the analyzer's CHANNEL messages find the carriers of a capture."""
    import torch
    from sigdigger_amd import engine
    n = 8192
    rng = np.random.default_rng(4)
    det = engine.ChannelDetector(ctx, n, alpha=0.2, beta=1e-3, gamma=0.5, snr=4.0)
    ora = sdo.ChannelDetector(n, 0.2, 0.5, 4.0)
    base = np.full(n, 1e-3, np.float32)
    for lo, hi, lvl in ((300, 420, 0.05), (2000, 2003, 0.2), (4090, 4110, 0.01), (7000, 7001, 1.0), (7500, 7800, 0.004)):
        base[lo:hi] += lvl
    base[7600:7602] = 1e-3                                                  # a two-bin gap inside a channel: bridged
    for k in range(6):
        P = (base * rng.chisquare(8, n).astype(np.float32) / 8).astype(np.float32)
        P = np.roll(P, n // 2)                                              # natural FFT order
        det.feed(torch.from_numpy(P).cuda())
        ora.feed(P)
    assert det.noise_floor() == np.float32(ora.N0)
    got = det.channels(1e6)
    ref = ora.find()
    assert len(got) == len(ref) >= 4
    df = 1e6 / n
    for g, (first, last, width, peak, s, ws) in zip(got, ref):
        assert g["f_lo"] == (first - n / 2 - 0.5) * df and g["f_hi"] == (last - n / 2 + 0.5) * df
        assert g["fc"] == (ws / s - n / 2) * df
        assert abs(g["S0"] - 10 * np.log10(peak + 1e-8)) < 1e-5                 # dB formatting on the host: libm log10f
        assert g["age"] == 0                                                      # the first list: nothing to continue
    # beta (the detector's signal-level smoothing): a second list continues the channels of the first -- levels smoothed
    # towards the new peaks, ages counted; a detector with beta = 0 reports the new peaks themselves
    det0 = engine.ChannelDetector(ctx, n, alpha=0.2, beta=0.0, gamma=0.5, snr=4.0)
    detb = engine.ChannelDetector(ctx, n, alpha=0.2, beta=0.25, gamma=0.5, snr=4.0)
    lists0, listsb = [], []
    for k in range(3):
        P = np.roll((base * (1.0 + 0.5 * k) * rng.chisquare(8, n).astype(np.float32) / 8).astype(np.float32), n // 2)
        for d_, ls in ((det0, lists0), (detb, listsb)):
            d_.feed(torch.from_numpy(P).cuda())
            ls.append(d_.channels(1e6))
    prev = []
    for raw, sm in zip(lists0, listsb):
        assert [c["age"] for c in raw] == [0] * len(raw) and len(raw) == len(sm)
        want = sdo.chandet_track(prev, [(c["fc"], c["f_lo"], c["f_hi"], c["S0"]) for c in raw], 0.25)
        for c, (fc, s0, age) in zip(sm, want):
            assert c["fc"] == fc and c["age"] == age and np.float32(c["S0"]) == s0 and c["snr"] == np.float32(np.float32(c["S0"]) - np.float32(c["N0"]))
        prev = want
    assert max(c["age"] for c in listsb[-1]) == 2
    # (b)
    nblocks = 12
    fcs = [-300e3, 50e3, 220e3]
    x = synth.psk_carriers(L * nblocks, [2 * f / FS for f in fcs], sps=32, order=4, seed=9, snr_db=15)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    p = suscan.AnalyzerParams.default()
    p.detector_params.alpha, p.detector_params.gamma, p.detector_params.snr = 0.3, 0.5, 6.0
    p.channel_update_int = 3 * L / FS
    Lb, mq, an = _start(path, L, params=p)
    lists = []

    def on_msg(t, ptr):
        if t == suscan.MSG_CHANNEL:
            m = C.cast(ptr, C.POINTER(suscan.ChannelMsg)).contents
            lists.append([(m.channel_list[i].contents.fc, m.channel_list[i].contents.bw, m.channel_list[i].contents.snr)
                          for i in range(m.channel_count)])

    seen = _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert seen.count(suscan.MSG_CHANNEL) == nblocks // 3
    last = lists[-1]
    assert len(last) == 3
    for (fc, bw, snr), want in zip(last, fcs):
        assert abs(fc - want) < 3e3 and 25e3 < bw < 60e3 and snr > 8.0       # 31.25 kBd RRC carriers, 15 dB over the noise


def test_wide_spectrum_sweep_through_the_abi_feeds_a_spectrum_view(tmp_path, sdo, ctx):
    """Row P1: a WIDE_SPECTRUM analyzer (Panoramic/Scanner.cpp:295-370) on a capture: every block is a dwell, the PSD frames
    carry the hop frequencies of the sweep strategy, and what Scanner::onPSDMessage does with them (PSDMessage ctor, then
    SpectrumView::feed at msg.getFrequency(), :503-523) gives the panorama -- oracle SpectrumView on the analyzer's frames
    against the device SpectrumView on the same frames, and the progressive hop sequence itself."""
    import torch
    from sigdigger_amd import engine
    ndw = 24
    fs, n, navg = FS, 1024, 4
    Lw = n * navg
    rng = np.random.default_rng(12)
    x = (0.05 * (rng.standard_normal(Lw * ndw) + 1j * rng.standard_normal(Lw * ndw))).astype(np.complex64)
    for d in range(ndw):                                           # a line per dwell whose offset grows with the dwell
        t = np.arange(Lw)
        x[d * Lw:(d + 1) * Lw] += np.exp(2j * np.pi * (0.3 * (d / ndw) - 0.15) * t).astype(np.complex64)
    path = tmp_path / "sweep.raw"
    x.tofile(path)
    Lb = suscan.load()
    mq = suscan.MQ()
    assert Lb.suscan_mq_init(C.byref(mq))
    cfg = Lb.suscan_source_config_new(b"file", 1)
    Lb.suscan_source_config_set_samp_rate(cfg, fs)
    assert Lb.suscan_source_config_set_path(cfg, str(path).encode())
    p = suscan.AnalyzerParams.default()
    p.mode = 1                                                     # SUSCAN_ANALYZER_MODE_WIDE_SPECTRUM
    p.detector_params.window_size = n
    p.psd_update_int = Lw / fs
    fmin, fmax, rel = 400e6, 406e6, 0.5
    p.min_freq, p.max_freq = fmin, fmax
    an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
    assert an
    Lb.suscan_source_config_destroy(cfg)
    assert Lb.suscan_analyzer_set_rel_bandwidth(an, rel) and Lb.suscan_analyzer_set_sweep_stratrgy(an, 1)
    frames, fcs = [], []
    while True:
        t, ptr = suscan.read_message(Lb, mq, 60.0)
        if t == suscan.MSG_HALT:
            break
        if t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            frames.append(np.ctypeslib.as_array(m.psd_data, shape=(n,)).copy())
            fcs.append(float(m.fc))
        Lb.suscan_analyzer_dispose_message(t, ptr)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert len(frames) == ndw
    step = rel * fs
    nsteps = int(np.ceil((fmax - fmin) / step))
    # the setters land a block or two into the stream: from the first frame labelled by the progressive walk on, the
    # sequence is the slots in order
    k0 = next(i for i in range(ndw - nsteps) if all(abs(fcs[i + j + 1] - fcs[i + j] - step) < 1 or fcs[i + j + 1] < fcs[i + j] for j in range(nsteps)))
    walk = fcs[k0:]
    slots = [int(round((f - fmin) / step - 0.5)) for f in walk]
    assert all(0 <= s < nsteps for s in slots)
    assert all((b - a) % nsteps == 1 for a, b in zip(slots, slots[1:]))
    # Scanner::onPSDMessage on every frame, on the device and in the oracle
    view, oview = engine.SpectrumView(ctx), sdo.SpectrumView()
    for v in (view, oview):
        v.set_range(fmin, fmax)
    view.set_fft(fs, rel)
    oview.v.fftBandwidth, oview.v.fftRelBw = fs, rel
    for f, fc in zip(frames[k0:], walk):
        db = sdo.psd_shift_db(f)                                   # the PSDMessage constructor
        view.feed(torch.from_numpy(db).cuda(), fc - fs / 2, fc + fs / 2)
        oview.feed(db, fc - fs / 2, fc + fs / 2)
    psd, accum, count = view.arrays()
    assert np.array_equal(accum, oview.accum) and np.array_equal(count, oview.count) and np.array_equal(psd, oview.psd)
    assert np.count_nonzero(count) > 0.9 * view.spectrum_size          # the panorama is filled (6 MHz at 1 kHz: 8192 bins)


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_audio_demodulators_and_resampler_match_the_oracle(ctx, sdo, mode):
    """Row A6, the "audio" class's arithmetic (SPEC.md section Q): AM / FM / USB / LSB / RAW, cut-off low-pass and resampler,
    in one block and in ragged pieces."""
    import torch
    from sigdigger_amd import engine
    efs, fa, bw, cutoff = 250e3, 48e3, 40e3, 12e3
    n = 120000
    t = np.arange(n)
    tone = np.sin(2 * np.pi * 1000 * t / efs)
    if mode == 1:
        x = (1 + 0.5 * tone) * np.exp(1j * 0.3)
    elif mode == 2:
        x = np.exp(1j * np.cumsum(2 * np.pi * 5e3 * tone / efs))
    elif mode in (3, 4):
        sgn = 1 if mode == 3 else -1
        x = np.exp(sgn * 2j * np.pi * (1000 - bw / 2) * t / efs)            # a 1 kHz tone of the upper / lower sideband
    else:
        x = np.exp(2j * np.pi * 3e3 * t / efs)
    x = (x + 0.01 * (np.random.default_rng(mode).standard_normal(n) + 1j * np.random.default_rng(9).standard_normal(n))).astype(np.complex64)
    ref = sdo.audio_run(x, mode, efs, bw, fa, cutoff, 0.8)
    dx = torch.from_numpy(x).cuda()
    for cuts in ([0, n], [0, 777, 50000, 50001, 90000, n]):
        au = engine.Audio(ctx, efs, bw)
        au.configure(mode, fa, cutoff, volume=0.8)
        got = np.concatenate([au.feed(dx[a:b]).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:])])
        assert abs(got.size - ref.size) <= 0 and got.size > 0.99 * n * fa / efs - 300
        err = np.abs(got - ref)
        assert err.max() <= 3e-5 * max(1.0, np.abs(ref).max()), (int(np.argmax(err)), np.flatnonzero(err > 1e-5)[:20].tolist(), got[np.argmax(err)], ref[np.argmax(err)])
    if mode != 5:                                                      # the 1 kHz tone comes out (skip the filter's run-in)
        a = got.real[2000:] - np.mean(got.real[2000:])
        spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
        assert abs(np.argmax(spec) * fa / a.size - 1000) < 30


def test_audio_inspector_through_the_abi(tmp_path, sdo):
    """AudioProcessor's protocol: open "audio" on a 200 kHz channel, push the audio.* configuration, play what arrives"""
    nblocks = 16
    fc, dev_hz, ftone = 150e3, 5e3, 800.0
    t = np.arange(L * nblocks)
    x = (np.exp(1j * (2 * np.pi * fc * t / FS + (dev_hz / ftone) * np.sin(2 * np.pi * ftone * t / FS))) +
         0.01 * (np.random.default_rng(1).standard_normal(t.size) + 1j * np.random.default_rng(2).standard_normal(t.size))).astype(np.complex64)
    path = tmp_path / "fm.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    bw = 100e3
    ch = suscan.Channel(fc=fc, f_lo=-bw / 2, f_hi=bw / 2, bw=bw, ft=0)
    assert Lb.suscan_analyzer_open_async(an, b"audio", C.byref(ch), 31)
    st = {"audio": [], "cfg": False, "efs": None}

    def on_msg(t_, ptr):
        if t_ == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                assert m.class_name == b"audio" and m.estimator_count == 0
                st["efs"] = m.equiv_fs
                cfg = Lb.suscan_config_dup(m.config)
                assert Lb.suscan_config_set_float(cfg, b"audio.cutoff", 5000.0)
                assert Lb.suscan_config_set_float(cfg, b"audio.volume", 1.0)
                assert Lb.suscan_config_set_integer(cfg, b"audio.sample-rate", 48000)
                assert Lb.suscan_config_set_integer(cfg, b"audio.demodulator", 2)
                assert Lb.suscan_config_set_bool(cfg, b"audio.squelch", 0)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 32)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg"] = True
                st["audio"] = []
        elif t_ == suscan.MSG_SAMPLES and st["cfg"]:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["audio"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert st["cfg"] and abs(st["efs"] - FS / 4) < 1e-3                # D = pow2floor(1e6 / 2e5) = 4
    audio = np.concatenate(st["audio"]).real
    assert audio.size > 0.5 * nblocks * L / FS * 48000
    a = audio[4000:] - np.mean(audio[4000:])
    spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
    assert abs(np.argmax(spec) * 48000 / a.size - ftone) < 20           # the modulating tone
    assert 0.6 * dev_hz / (st["efs"] / 2) < np.max(np.abs(a)) < 1.4 * dev_hz / (st["efs"] / 2)   # (1/pi) arg: deviation / Nyquist



def test_end_of_stream_flushes_the_tail_of_every_channel(tmp_path, sdo):
    """A capture that does not end on a block boundary: what is left goes through the inspectors too (whole half windows
    of the filter bank) before EOS -- a file-source consumer loses nothing but the last incomplete half window."""
    nblocks, extra = 6, 5 * H + 123
    fc, bw = 125e3, 40e3
    x = synth.psk_carriers(L * nblocks + extra, [2 * fc / FS], sps=64, order=4, seed=12, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"raw", C.byref(ch), 0, -1, 9)
    st = {"psd": 0, "open_at": None, "samples": [], "eos": False, "after_eos": 0}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_EOS:
            st["eos"] = True
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                st["open_at"] = st["psd"]
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))
            st["after_eos"] += int(st["eos"])

    _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert st["eos"] and st["after_eos"] == 0                     # the tail's batch comes BEFORE the EOS message
    # the tail holds 5 half windows and one whole 4096-point PSD frame of the 16 a block averages: one more PSD message
    assert st["psd"] == nblocks + 1
    D, f0, bwa, guard = _chan_params(fc, bw)
    got = np.concatenate(st["samples"])
    hs = W // D // 2
    # the channel was opened at some block b: everything from there to the last whole half window was delivered
    nh = got.size // hs + 1
    b = (nblocks * L + 5 * H) // H - nh
    assert b % (L // H) == 0 and 0 <= b // (L // H) <= (st["open_at"] or 0) + 2
    ref = sdo.specttuner_run_f32(x[b * H:], f0, bwa, guard)
    assert got.size == ref.size == (nh - 1) * hs
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))



def test_watermark_sets_the_batch_size_and_the_stream_stays_the_same(tmp_path, sdo):
    """Analyzer::setInspectorWatermark (Suscan/Analyzer.cpp:528-537): SAMPLES batches of exactly `watermark` samples; what
    does not fill a batch waits, the rest goes out before EOS.  Concatenated, the batches are the stream without a
    watermark -- here the oracle's, bit for bit."""
    nblocks, wm = 6, 1000
    fc, bw = -150e3, 40e3
    x = synth.psk_carriers(L * nblocks, [2 * fc / FS], sps=64, order=4, seed=14, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"raw", C.byref(ch), 0, -1, 3)
    st = {"psd": 0, "open_at": None, "sizes": [], "samples": [], "ack": None, "eos_at": None}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_EOS:
            st["eos_at"] = len(st["sizes"])
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                st["open_at"] = st["psd"]
                assert Lb.suscan_analyzer_set_inspector_watermark_async(an, m.handle, wm, 4)
            elif m.kind == suscan.KIND_SET_WATERMARK:
                st["ack"] = m.watermark
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["sizes"].append(int(m.sample_count))
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert st["ack"] == wm and st["eos_at"] == len(st["sizes"])       # every batch, the flushed remainder included, before EOS
    sizes = st["sizes"]
    first_wm = next(i for i, n in enumerate(sizes) if n == wm)          # (a block may have gone out whole before the request took effect)
    assert all(n == wm for n in sizes[first_wm:-1]) and 0 < sizes[-1] <= wm and len(sizes) - first_wm > 20
    D, f0, bwa, guard = _chan_params(fc, bw)
    got = np.concatenate(st["samples"])
    b = nblocks - (got.size // (W // D // 2) + 1) * H // L
    ref = sdo.specttuner_run_f32(x[b * L:], f0, bwa, guard)
    assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("n", [32768, 65536])
def test_main_spectrum_of_large_frames_through_the_analyzer(tmp_path, sdo, n):
    """detector_params.window_size beyond the LDS -- the scanner's nextPow2(fs / 1 kHz) (Panoramic/Scanner.cpp:323): 32768
    points take one trip through HBM (psd.hip HALVES), 65536 two (psd_large.hip) -- as PSD messages of a live analyzer"""
    navg, nblocks = 4, 3
    Lb_ = n * navg
    x = synth.psk_carriers(Lb_ * nblocks + 777, [0.31, -0.12], sps=16, seed=n % 1000, snr_db=20)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb = suscan.load()
    mq = suscan.MQ()
    assert Lb.suscan_mq_init(C.byref(mq))
    cfg = Lb.suscan_source_config_new(b"file", 1)
    Lb.suscan_source_config_set_samp_rate(cfg, FS)
    assert Lb.suscan_source_config_set_path(cfg, str(path).encode())
    p = suscan.AnalyzerParams.default()
    p.detector_params.window_size = n
    p.detector_params.window = 4
    p.psd_update_int = Lb_ / FS
    an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
    assert an
    Lb.suscan_source_config_destroy(cfg)
    frames = []
    while True:
        t, ptr = suscan.read_message(Lb, mq, 60.0)
        if t == suscan.MSG_HALT:
            break
        if t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            assert m.psd_size == n
            frames.append(np.ctypeslib.as_array(m.psd_data, shape=(n,)).copy())
        Lb.suscan_analyzer_dispose_message(t, ptr)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert len(frames) == nblocks                                  # (the 777-sample tail holds no whole frame)
    ref = sdo.psd_frames(x, nblocks * navg, n, n, sdo.window(4, n), navg=navg, scale=1.0 / n)
    got = np.stack(frames)
    err = np.max(np.abs(got - ref), axis=1) / np.max(ref, axis=1)
    assert np.all(err < 1e-5), err


def test_the_slab_grows_under_running_inspectors(tmp_path, sdo):
    """Round 6: the narrow channels of the filter bank are columns of one time-major slab per shard, pitch = the tuner's
    channel table rounded up to 64.  Sixty narrow raw inspectors (and two wide ones, which keep rows of their own) run
    when ten more narrow ones are opened: the table passes 64 entries, the slabs are re-allocated at pitch 128 under the
    running streams -- and every stream, old and new, is the oracle's from its first block to the end of the capture."""
    nblocks = 12
    rng = np.random.default_rng(77)
    narrow_bw = [2.5e3, 5e3, 1.2e3]                                   # D = 128, 64, 256: channels of 32, 64, 16 bins
    first = [((k - 30) * 15e3 + 1e3, narrow_bw[k % 3]) for k in range(60)] + [(200e3, 40e3), (-150e3, 150e3)]
    late = [((k - 5) * 23e3 + 4e3, narrow_bw[k % 3]) for k in range(10)]
    chans = first + late
    x = synth.psk_carriers(L * nblocks, [0.11, -0.23, 0.37], sps=64, order=4, seed=9, snr_db=25)
    x = (x + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 2 * FS, 0)

    def open_(k):
        # (the late ones are not `precise`: a residual NCO counts from the window its channel was opened at, which this test
        # does not track -- tests/test_gpu_analyzer_fuzz.py does)
        fc, bw = chans[k]
        ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
        assert Lb.suscan_analyzer_open_ex_async(an, b"raw", C.byref(ch), int(k % 2 == 0 and k < len(first)), -1, 100 + k)

    for k in range(len(first)):
        open_(k)
    st = {"psd": 0, "open_at": {}, "samples": {}, "late_posted": False, "status": []}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
            if st["psd"] == 4 and not st["late_posted"]:
                st["late_posted"] = True
                for k in range(len(first), len(chans)):
                    open_(k)
        elif t in (suscan.MSG_INTERNAL, suscan.MSG_READ_ERROR) and ptr:
            m = C.cast(ptr, C.POINTER(suscan.StatusMsg)).contents
            st["status"].append((t, m.code, (m.err_msg or b"").decode("utf-8", "replace")))
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                k = m.req_id - 100
                st["open_at"][k] = st["psd"]
                assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, 500 + k, 0)
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            a = np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64)
            st["samples"].setdefault(m.inspector_id - 500, []).append(a)

    _pump(Lb, an, on_msg)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    assert not [s for s in st["status"] if s[0] == suscan.MSG_INTERNAL], st["status"]
    assert len(st["open_at"]) == len(chans) and st["psd"] == nblocks, (len(st["open_at"]), st["psd"], st["status"])
    assert min(st["open_at"][k] for k in range(len(first), len(chans))) >= 3, "the late ones were opened under running streams"
    for k, (fc, bw) in enumerate(chans):
        D, f0, bwa, guard = _chan_params(fc, bw)
        assert k in st["samples"], (k, st["status"])
        got = np.concatenate(st["samples"][k])
        if k < len(first):
            _check_raw_channel(sdo, x, got, k, D, f0, bwa, guard, nblocks, max(st["open_at"].values()))
            continue
        # a channel opened while the tuner runs starts on the window that straddles the block boundary, with a zero cross-fade
        # partner: the oracle started half a window before that boundary (and, as above, the first batch or two may have gone
        # out before the id hand-shake: the tails are compared for each reading of the count)
        b = nblocks - (got.size // (W // D // 2)) * H // L
        errs = []
        for bb in range(min(b + 1, nblocks - 1), max(b - 4, 0), -1):
            ref = sdo.specttuner_run(x[bb * L - H:], W, f0, bwa, guard, precise=False)
            n = min(got.size, ref.size)
            if n > 0.6 * ref.size:
                errs.append(_relerr(got[-n:], ref[-n:]))
        assert errs and min(errs) <= TOL, (k, b, got.size, errs)
