"""Live path through the suscan_analyzer_* C ABI (include/suscan_amd.h), driven the way
Suscan::Analyzer::AsyncThread::run does (Suscan/Analyzer.cpp:63-103): new -> read loop ->
dispose; inspector open / set-id / set-config protocol of AnalyzerRequestTracker
(Suscan/AnalyzerRequestTracker.cpp:70-183).  PSD frames and recovered symbols are checked
against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

from sigdigger_amd import suscan, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fir_channeliser(monkeypatch):
    """These tests pin the inspector chains bit for bit against the oracle's translate + 255-tap FIR channeliser
    (SPEC.md C); the analyzer's default, the FFT filter bank (SPEC.md C2), has its own file: test_gpu_analyzer_fft.py."""
    monkeypatch.setenv("SUAMD_ANALYZER_CHANNELISER", "fir")


FS = 1_000_000
N = 4096
NAVG = 16                       # psd_update_int = N*NAVG/FS
L = N * NAVG


def _start(path, L_, loop=False, params=None, fmt=1, fs=FS):
    Lb = suscan.load()
    mq = suscan.MQ()
    assert Lb.suscan_mq_init(C.byref(mq))
    cfg = Lb.suscan_source_config_new(b"file", fmt)
    Lb.suscan_source_config_set_samp_rate(cfg, fs)
    Lb.suscan_source_config_set_freq(cfg, 433.92e6)
    assert Lb.suscan_source_config_set_path(cfg, str(path).encode())
    Lb.suscan_source_config_set_loop(cfg, int(loop))
    p = params or suscan.AnalyzerParams.default()
    p.detector_params.window_size = N
    p.detector_params.window = 4
    p.psd_update_int = L_ / FS
    an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
    assert an
    _MQ_OF[an] = mq
    Lb.suscan_source_config_destroy(cfg)          # the analyzer keeps its own copy
    return Lb, mq, an


_MQ_OF = {}


def _pump(Lb, an, on_msg, limit=10000):
    """AsyncThread::run: read until HALT / EOS; every message disposed exactly once.  (Analyzer::read blocks for ever;
    here the queue is polled with a deadline so that a stalled worker fails the test instead of hanging the run.)"""
    seen = []
    mq = _MQ_OF[an]
    for _ in range(limit):
        t, ptr = suscan.read_message(Lb, mq, 60.0)
        seen.append(t)
        if t == suscan.MSG_HALT:
            break
        on_msg(t, ptr)
        Lb.suscan_analyzer_dispose_message(t, ptr)
    return seen


def test_psd_stream_matches_oracle_and_eos(tmp_path, sdo):
    nblocks = 5
    x = synth.psk_carriers(L * nblocks + 1000, [0.2, -0.31], sps=16, seed=3)     # ragged tail is dropped
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    frames, info = [], {}

    def on_msg(t, ptr):
        if t == suscan.MSG_SOURCE_INIT:
            info["init"] = C.cast(ptr, C.POINTER(suscan.StatusMsg)).contents.code
        elif t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            assert m.psd_size == N and m.samp_rate == FS and m.fc == 433920000
            frames.append(np.ctypeslib.as_array(m.psd_data, shape=(N,)).copy())
            info.setdefault("ts", []).append(m.timestamp.tv_sec + 1e-6 * m.timestamp.tv_usec)
        elif t == suscan.MSG_EOS:
            info["eos"] = True

    seen = _pump(Lb, an, on_msg)
    assert info["init"] == 0 and info.get("eos") and seen[-1] == suscan.MSG_HALT
    assert seen.count(suscan.MSG_PSD) == nblocks
    assert Lb.suscan_analyzer_get_samp_rate(an) == FS
    win = sdo.window(4, N)
    ref = sdo.psd_frames(x, nblocks * NAVG, N, N, win, navg=NAVG, scale=1.0 / N)
    got = np.stack(frames)
    err = np.max(np.abs(got - ref), axis=1) / np.max(ref, axis=1)
    assert np.all(err < 1e-5), err
    assert np.allclose(info["ts"], np.arange(nblocks) * L / FS, atol=1e-5)          # signal time stamps
    # what the consumer does next (PSDMessage ctor, in place on the message buffer)
    assert np.argmax(sdo.psd_shift_db(got[0])) == np.argmax(sdo.psd_shift_db(ref[0]))
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def _write_wav(path, raw, rate, bits, tag):
    import struct
    data = raw.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, tag, 2, rate, rate * 2 * bits // 8, 2 * bits // 8, bits))
        f.write(b"LIST" + struct.pack("<I", 4) + b"abcd")                     # a chunk to skip
        f.write(b"data" + struct.pack("<I", len(data)) + data)


@pytest.mark.parametrize("kind", ["u8", "s8", "s16", "wav16", "wav8", "sigmf-ci16", "auto-cu8", "exported-wav"])
def test_compact_file_formats_are_expanded_on_the_gpu(tmp_path, sdo, kind):
    """File sources in the compact formats of FileSourcePage.cpp:80-104: the PSD stream must equal the
    oracle's PSD of the oracle-converted samples (ingest is bit exact, so the same 1e-5 bound holds)."""
    nblocks = 3
    x = synth.psk_carriers(L * nblocks, [0.15], sps=8, seed=11)
    x = (0.5 * x / np.max(np.abs(x.view(np.float32)))).astype(np.complex64)
    iq = x.view(np.float32)
    fs = FS
    if kind in ("u8", "wav8", "auto-cu8"):
        raw, ofmt = np.clip(np.round(iq * 128 + 128), 0, 255).astype(np.uint8), 2
    elif kind == "s8":
        raw, ofmt = np.clip(np.round(iq * 128), -128, 127).astype(np.int8), 3
    else:
        raw, ofmt = np.clip(np.round(iq * 32768), -32768, 32767).astype(np.int16), 4
    if kind in ("u8", "s8", "s16"):
        path, fmt = tmp_path / "iq.bin", {"u8": 2, "s8": 3, "s16": 4}[kind]
        raw.tofile(path)
    elif kind == "auto-cu8":
        path, fmt = tmp_path / "capture.cu8", 0
        raw.tofile(path)
    elif kind == "exported-wav":                                            # round trip: suamd_export_capture's float WAV is a source
        from sigdigger_amd import lib as sdlib                               # plain ctypes + HIP: this module stays torch-free
        path, fmt, fs, raw, ofmt = tmp_path / "cap.wav", 0, 250_000, x, 1
        lb, hip = sdlib.load(), C.CDLL("libamdhip64.so")
        ctx, dptr = lb.suamd_ctx_new(0), C.c_void_p()
        hip.hipMalloc.argtypes, hip.hipMemcpy.argtypes, hip.hipFree.argtypes = [C.c_void_p, C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int], [C.c_void_p]
        assert ctx and hip.hipMalloc(C.byref(dptr), x.nbytes) == 0
        assert hip.hipMemcpy(dptr, x.ctypes.data, x.nbytes, 1) == 0
        assert lb.suamd_export_capture(ctx, str(path).encode(), b"wav", dptr, x.size, float(fs), None), lb.suamd_last_error()
        hip.hipFree(dptr)
        lb.suamd_ctx_destroy(ctx)
    elif kind.startswith("wav"):
        path, fmt, fs = tmp_path / "iq.wav", 5, 250_000                     # the header's rate wins
        _write_wav(path, raw, fs, 8 if kind == "wav8" else 16, 1)
    else:
        path, fmt, fs = tmp_path / "rec.sigmf-data", 6, 2_000_000
        raw.tofile(path)
        (tmp_path / "rec.sigmf-meta").write_text(
            '{"global": {"core:datatype": "ci16_le", "core:sample_rate": 2000000, "core:version": "1.0.0"}, '
            '"captures": [{"core:sample_start": 0}], "annotations": []}')
    Lb, mq, an = _start(path, L * FS / fs, fmt=fmt, fs=FS)      # psd_update_int = L / fs
    frames, rates = [], []

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            frames.append(np.ctypeslib.as_array(m.psd_data, shape=(N,)).copy())
            rates.append(m.samp_rate)

    seen = _pump(Lb, an, on_msg)
    assert seen[-1] == suscan.MSG_HALT and len(frames) == nblocks
    assert all(r == fs for r in rates) and Lb.suscan_analyzer_get_samp_rate(an) == fs
    xc = sdo.ingest_iq(ofmt, raw)
    ref = sdo.psd_frames(xc, nblocks * NAVG, N, N, sdo.window(4, N), navg=NAVG, scale=1.0 / N)
    got = np.stack(frames)
    err = np.max(np.abs(got - ref), axis=1) / np.max(ref, axis=1)
    assert np.all(err < 1e-5), err
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_psk_inspector_protocol_and_symbols(tmp_path, sdo):
    nblocks = 12
    fc, baud, bw = 125e3, 15625.0, 40e3
    x = synth.psk_carriers(L * nblocks, [2 * fc / FS], sps=int(FS / baud), order=4, seed=8, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)     # leave time for the requests to land early
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 77)
    assert Lb.suscan_analyzer_open_async(an, b"drm", C.byref(ch), 78)            # unsupported class
    bad = suscan.Channel(fc=0.0, bw=0.0)
    assert Lb.suscan_analyzer_open_async(an, b"psk", C.byref(bad), 79)
    assert Lb.suscan_analyzer_close_async(an, 1234, 80)                          # unknown handle
    st = {"psd": 0, "samples": [], "kinds": {}, "cfg_at": None}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            st["kinds"][m.req_id] = m.kind
            if m.kind == suscan.KIND_OPEN:
                assert m.class_name == b"psk" and m.fs == FS and m.config
                st["handle"], st["equiv_fs"] = m.handle, m.equiv_fs
                # AnalyzerRequestTracker: OPEN -> set id (:150) ; then the tab pushes its config
                assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, 4242, 81)
                cfg = Lb.suscan_config_dup(m.config)
                Lb.suscan_config_set_integer(cfg, b"afc.costas-order", 2)
                Lb.suscan_config_set_float(cfg, b"afc.loop-bw", 40.0)
                Lb.suscan_config_set_integer(cfg, b"clock.type", 1)
                Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
                Lb.suscan_config_set_float(cfg, b"clock.gain", 0.2)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 82)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg_at"] = st["psd"]                    # blocks fully processed before the new chain
                st["samples"] = []
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            if st["cfg_at"] is not None:
                assert m.inspector_id == 4242
                a = np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy()
                st["samples"].append(a.view(np.complex64))

    _pump(Lb, an, on_msg)
    k = st["kinds"]
    assert k[77] == suscan.KIND_OPEN and k[78] == suscan.KIND_WRONG_KIND
    assert k[79] == suscan.KIND_INVALID_CHANNEL and k[80] == suscan.KIND_WRONG_HANDLE
    assert k[81] == suscan.KIND_SET_ID and k[82] == suscan.KIND_SET_CONFIG
    b0 = st["cfg_at"]
    assert b0 is not None and b0 < nblocks - 4, "the configuration landed too late to test anything"
    # oracle: the rebuilt chain starts fresh at block b0
    D = 8                                               # pow2floor(fs / (2 bw)) = pow2floor(12.5)
    assert abs(st["equiv_fs"] - FS / D) < 1e-3
    taps = sdo.lpf_design(255, bw / FS)
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    xs = x[b0 * L:]
    y = sdo.chan_feed(np.zeros(254, np.complex64), xs, 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
    sps = (FS / D) / baud
    a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), y)
    z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, min(2.0 / sps, 0.95), 3, 2 * 40.0 / (FS / D)), a)
    ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, baud / (FS / D)), z)
    got = np.concatenate(st["samples"])
    assert len(got) == len(ref), "symbol count"
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "symbols differ from the oracle"
    tail = got[len(got) // 2:]
    assert abs(np.mean((tail / np.abs(tail)) ** 4)) > 0.7          # a locked QPSK constellation
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_manual_chain_config_vocabulary(tmp_path, sdo):
    """The rest of the inspector vocabulary (GainControl / AfcControl / MfControl / ClockRecovery /
    EqualizerControl): fixed gain, manual carrier offset, RRC matched filter, fixed-baud sampler with a
    phase, CMA equalizer -- the SAMPLES stream must equal the oracle chain bit for bit."""
    nblocks = 10
    fc, baud, bw, offs = -200e3, 15625.0, 40e3, 310.0
    x = synth.psk_carriers(L * nblocks, [2 * (fc + offs) / FS], sps=int(FS / baud), order=4, seed=5, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 7)
    st = {"psd": 0, "samples": [], "cfg_at": None}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                cfg = Lb.suscan_config_dup(m.config)
                Lb.suscan_config_set_bool(cfg, b"agc.enabled", 0)
                Lb.suscan_config_set_float(cfg, b"agc.gain", -20.0)
                Lb.suscan_config_set_integer(cfg, b"afc.costas-order", 0)
                Lb.suscan_config_set_float(cfg, b"afc.offset", offs)
                Lb.suscan_config_set_integer(cfg, b"mf.type", 1)
                Lb.suscan_config_set_float(cfg, b"mf.roll-off", 0.35)
                Lb.suscan_config_set_integer(cfg, b"clock.type", 0)
                Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
                Lb.suscan_config_set_float(cfg, b"clock.phase", 0.3)
                Lb.suscan_config_set_integer(cfg, b"equalizer.type", 1)
                Lb.suscan_config_set_float(cfg, b"equalizer.rate", 2e-3)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 8)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg_at"] = st["psd"]
                st["samples"] = []
        elif t == suscan.MSG_SAMPLES and st["cfg_at"] is not None:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    b0 = st["cfg_at"]
    assert b0 is not None and b0 < nblocks - 4
    D, efs = 8, FS / 8
    taps = sdo.lpf_design(255, bw / FS)
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    y = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * L:], 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
    y = sdo.scale(y, np.float32(10.0 ** (-20.0 / 20.0)))
    y = sdo.xlate_bulk(y, 0, sdo.fnor_to_dphase(-2.0 * offs / efs), 0)
    sps = efs / baud
    h = sdo.rrc_design(sps, float(np.float32(0.35)))            # config values are SUFLOAT
    y = sdo.fir_feed(np.zeros(h.size - 1, np.complex64), h, y)
    cd = sdo.clock_new(0.0, baud / efs)
    cd.phi = np.float32(0.5) * np.float32(0.3)
    sym = sdo.clock_feed_bulk(cd, y)
    # the equalizer sees the symbols block by block, but its state carries over: one run is the same
    ref = sdo.cma_feed_bulk(sdo.cma_new(8, 2e-3), sym)
    got = np.concatenate(st["samples"])
    assert len(got) == len(ref) and abs(len(got) - (nblocks - b0) * L / D / sps) <= 2, "symbol count"
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "symbols differ from the oracle"
    assert np.all(np.isfinite(got.view(np.float32))) and np.mean(np.abs(got)) > 0.01
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_ask_inspector_with_pll(tmp_path, sdo):
    """class "ask" (InspectorCtl/AskControl.cpp:53-76): AGC -> PLL (ask.use-pll, ask.loop-bw, ask.offset) ->
    Gardner; bit exact against the oracle chain."""
    nblocks = 9
    fc, baud, bw = 90e3, 7812.5, 30e3
    n = L * nblocks
    rng = np.random.default_rng(21)
    sps = int(FS / baud)
    env = np.repeat(0.25 + 0.75 * rng.integers(0, 2, n // sps + 1), sps)[:n]            # on-off keying with a floor
    x = (env * np.exp(1j * (2 * np.pi * (fc + 40.0) / FS * np.arange(n) + 0.7)) +
         0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"ask", C.byref(ch), 1, -1, 3)
    st = {"psd": 0, "samples": [], "cfg_at": None}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                assert m.class_name == b"ask"
                cfg = Lb.suscan_config_dup(m.config)
                Lb.suscan_config_set_bool(cfg, b"ask.use-pll", 1)
                Lb.suscan_config_set_float(cfg, b"ask.loop-bw", 200.0)
                Lb.suscan_config_set_float(cfg, b"ask.offset", 0.0)
                Lb.suscan_config_set_integer(cfg, b"clock.type", 1)
                Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 4)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg_at"] = st["psd"]
                st["samples"] = []
        elif t == suscan.MSG_SAMPLES and st["cfg_at"] is not None:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    b0 = st["cfg_at"]
    assert b0 is not None and b0 < nblocks - 4
    D = 16                                              # pow2floor(1e6 / 60e3)
    efs = FS / D
    taps = sdo.lpf_design(255, bw / FS)
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    y = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * L:], 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
    sps_c = efs / baud
    a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps_c)), y)
    z = sdo.pll_track_bulk(sdo.pll_new(0.0, 2 * 200.0 / efs), a)
    ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, baud / efs), z)
    got = np.concatenate(st["samples"])
    assert len(got) == len(ref) and len(ref) > 100
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "symbols differ from the oracle"
    tail = got[len(got) // 2:]
    assert np.mean(np.abs(tail.imag)) < 0.35 * np.mean(np.abs(tail.real))        # carrier locked: energy on I
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_inspector_spectrum_sources(tmp_path, sdo):
    """Analyzer::setSpectrumSource (Suscan/Analyzer.cpp:539-547): the OPEN reply lists the sources, id k selects
    spectsrc_list[k-1], every block then yields an INSPECTOR/SPECTRUM message (linear power, natural order) that
    the tab post-processes in place (GenericInspector.cpp:231-247).  Checked against the oracle for "psd" and,
    after switching, "exp_2" (a BPSK carrier squared is a spectral line at twice its offset)."""
    nblocks = 10
    fc, baud, bw, offs = 100e3, 15625.0, 60e3, 1500.0
    x = synth.psk_carriers(L * nblocks, [2 * (fc + offs) / FS], sps=int(FS / baud), order=2, seed=9, snr_db=30)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 1)
    st = {"psd": 0, "spec": [], "names": None, "acks": []}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                lst = C.cast(m.spectsrc_list, C.POINTER(C.c_char_p))
                st["names"] = [lst[i].decode() for i in range(m.spectsrc_count)]
                st["open_at"] = st["psd"]                   # blocks processed before the chain existed
                assert Lb.suscan_analyzer_inspector_set_spectrum_async(an, m.handle, 99, 2)      # no such source
                assert Lb.suscan_analyzer_inspector_set_spectrum_async(an, m.handle, 1, 3)       # "psd"
            elif m.kind == suscan.KIND_SPECTRUM:
                if not m.spectrum_data:
                    st["acks"].append((m.req_id, m.spectsrc_id, st["psd"]))
                else:
                    d = np.ctypeslib.as_array(C.cast(m.spectrum_data, C.POINTER(C.c_float)), shape=(m.spectrum_size,)).copy()
                    st["spec"].append((st["psd"], m.spectsrc_id, m.samp_rate, d))
                    if len(st["spec"]) == 3:                                                  # switch source mid-stream
                        assert Lb.suscan_analyzer_inspector_set_spectrum_async(an, m.handle, 7, 4)   # "exp_2"
            elif m.kind == suscan.KIND_INVALID_ARGUMENT:
                st["acks"].append((m.req_id, None, st["psd"]))

    _pump(Lb, an, on_msg)
    assert st["names"] == list(sdo.SPECTSRC)
    acks = {a[0]: a for a in st["acks"]}
    assert acks[2][1] is None and acks[3][1] == 1 and acks[4][1] == 7
    D, efs = 8, FS / 8
    taps = sdo.lpf_design(255, bw / FS)
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    b0 = st["open_at"]
    y = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * L:], 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
    mb = L // D                                                      # channel samples per block
    n = 8192
    win = sdo.window(4, n)
    seen = {1: 0, 7: 0}
    for npsd, sid, rate, d in st["spec"]:
        blk = npsd - 1 - b0                                          # the block's PSD message precedes its inspector output
        assert rate == int(efs) and d.size == n and blk >= 0
        seg = y[blk * mb:(blk + 1) * mb]
        prev = 0j if (sid == 7 and seen[7] == 0) or blk == 0 else complex(y[blk * mb - 1])
        pre = sdo.spectsrc_preproc(sid, seg, prev)
        ref = sdo.psd_frames(pre, mb // n, n, n, win, navg=mb // n, scale=1.0 / n)[0]
        assert np.max(np.abs(d - ref)) < 1e-5 * np.max(ref), (blk, sid)
        seen[sid] += 1
        if sid == 7:                                                 # what the tab does next: dB + rotate, then look
            shown = sdo.inspector_spectrum_db_shift(d)
            k = int(np.argmax(shown)) - n // 2
            assert abs(k * efs / n - 2 * offs) < 2 * efs / n
    assert seen[1] >= 3 and seen[7] >= 2
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_many_concurrent_inspectors(tmp_path, sdo):
    """20 inspectors opened on one analyzer (their chains run concurrently on the analyzer's inspector streams):
    every SAMPLES stream is attributed to the right inspector_id and equals that channel's oracle chain."""
    import time
    nblocks, nins = 8, 20
    baud, bw = 15625.0, 30e3
    fcs = (np.arange(nins) - nins / 2 + 0.5) * 40e3
    x = synth.psk_carriers(L * nblocks, list(2 * fcs / FS), sps=int(FS / baud), order=4, seed=3, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 3 * FS, 0)
    for k, fc in enumerate(fcs):
        ch = suscan.Channel(fc=float(fc), f_lo=float(fc - bw / 2), f_hi=float(fc + bw / 2), bw=bw, ft=433.92e6)
        assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 100 + k)
    st = {"psd": 0, "cfg_at": {}, "samples": {}, "handle_of": {}}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                k = m.req_id - 100
                st["handle_of"][m.handle] = k
                assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, 5000 + k, 200 + k)
                cfg = Lb.suscan_config_dup(m.config)
                Lb.suscan_config_set_integer(cfg, b"afc.costas-order", 2)
                Lb.suscan_config_set_float(cfg, b"afc.loop-bw", 40.0)
                Lb.suscan_config_set_integer(cfg, b"clock.type", 1)
                Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 300 + k)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                k = m.req_id - 300
                st["cfg_at"][k] = st["psd"]
                st["samples"][k] = []
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            k = m.inspector_id - 5000
            if 0 <= k < nins and k in st["cfg_at"]:
                st["samples"][k].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    t0 = time.time()
    _pump(Lb, an, on_msg)
    assert len(st["cfg_at"]) == nins and time.time() - t0 < 60
    D, efs = 16, FS / 16                                   # pow2floor(1e6 / 60e3)
    taps = sdo.lpf_design(255, bw / FS)
    sps = efs / baud
    checked = 0
    for k in range(nins):
        b0 = st["cfg_at"][k]
        if b0 > nblocks - 3:
            continue
        dp = sdo.fnor_to_dphase(-2 * fcs[k] / FS)
        y = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * L:], 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
        a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), y)
        z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, min(2.0 / sps, 0.95), 3, 2 * 40.0 / efs), a)
        ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, baud / efs), z)
        got = np.concatenate(st["samples"][k])
        assert len(got) == len(ref), f"inspector {k}: symbol count"
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"inspector {k}: symbols differ from the oracle"
        checked += 1
    assert checked >= nins // 2
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_narrow_channel_in_a_wide_capture(tmp_path, sdo):
    """bw = 800 Hz at 1 MS/s: decimation 512, beyond the channeliser's LDS window (its sparse-output kernel), 128
    channel samples per block split over the analyzer's stage sub-ranges -- symbols equal the oracle chain's"""
    nblocks = 24
    baud, bw, fc = 200.0, 800.0, 123e3
    D, efs = 512, FS / 512
    sps = efs / baud
    t = np.arange(L * nblocks)
    rng = np.random.default_rng(8)
    nsym = int(L * nblocks / (FS / baud)) + 2
    bits = (rng.integers(0, 4, nsym) * 2 + 1) * np.pi / 4
    ph = np.repeat(bits, int(FS / baud))[:L * nblocks]
    x = (0.5 * np.exp(1j * (2 * np.pi * fc / FS * t + ph)) + 0.01 * (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size))).astype(np.complex64)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 4 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 7)
    st = {"psd": 0, "cfg_at": None, "samples": [], "status": []}

    def on_msg(tp, ptr):
        if tp == suscan.MSG_PSD:
            st["psd"] += 1
        elif tp == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                assert abs(m.equiv_fs - efs) < 1e-3
                cfg = Lb.suscan_config_dup(m.config)
                Lb.suscan_config_set_integer(cfg, b"afc.costas-order", 2)
                Lb.suscan_config_set_float(cfg, b"afc.loop-bw", 5.0)
                Lb.suscan_config_set_integer(cfg, b"clock.type", 1)
                Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 8)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg_at"] = st["psd"]
        elif tp == suscan.MSG_SAMPLES and st["cfg_at"] is not None:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["samples"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    b0 = st["cfg_at"]
    assert b0 is not None and b0 < nblocks - 8
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    taps = sdo.lpf_design(255, bw / FS)
    y = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * L:], 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
    a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), y)
    z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, min(2.0 / sps, 0.95), 3, 2 * 5.0 / efs), a)
    ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, baud / efs), z)
    got = np.concatenate(st["samples"])
    assert len(ref) > 100 and len(got) == len(ref)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "symbols differ from the oracle"
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


@pytest.mark.parametrize("fmt", ["f32", "u8"])
def test_baseband_filters_and_source_controls(tmp_path, sdo, fmt):
    """registerBaseBandFilter (Suscan/Analyzer.cpp:127-142): filters run on the worker thread in priority order on
    SUCOMPLEX samples and what they write is what the PSD sees; setIQReverse / setDCRemove act on the samples;
    setFrequency re-labels the PSD frames and announces itself with a SOURCE_INFO message"""
    nblocks = 6
    x = synth.psk_carriers(L * nblocks, [0.2], sps=16, seed=5) * np.float32(0.4) + np.complex64(0.05 - 0.02j)
    if fmt == "u8":
        raw = np.clip(np.round(x.view(np.float32) * 128 + 128), 0, 255).astype(np.uint8)
        xs = sdo.ingest_iq(2, raw)
        path = tmp_path / "iq.u8"
        raw.tofile(path)
        Lb, mq, an = _start(path, L, fmt=2)
    else:
        xs = x
        path = tmp_path / "iq.raw"
        x.tofile(path)
        Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 2 * FS, 0)        # slow enough for the control calls below to land mid-stream
    calls = []

    @suscan.BASEBAND_FILTER
    def halve(priv, analyzer, samples, length, offset):
        calls.append(("halve", int(length), int(offset)))
        v = np.ctypeslib.as_array(samples, shape=(2 * length,))
        v *= np.float32(0.5)
        return 1

    @suscan.BASEBAND_FILTER
    def spy(priv, analyzer, samples, length, offset):
        v = np.ctypeslib.as_array(samples, shape=(2 * length,))
        calls.append(("spy", float(v[0]), float(v[1])))
        return 1

    assert Lb.suscan_analyzer_register_baseband_filter_with_prio(an, spy, None, 10)     # registered first, runs second
    assert Lb.suscan_analyzer_register_baseband_filter(an, halve, None)                 # prio 0
    # controls a file source cannot honour are recorded, the panoramic ones refused outside WIDE_SPECTRUM mode
    assert Lb.suscan_analyzer_set_gain(an, b"LNA", 10.0) and Lb.suscan_analyzer_set_antenna(an, b"RX2")
    assert Lb.suscan_analyzer_set_history_size(an, 1 << 20) and Lb.suscan_analyzer_replay(an, 0)
    assert not Lb.suscan_analyzer_set_hop_range(an, 1e6, 2e6) and not Lb.suscan_analyzer_set_rel_bandwidth(an, 0.5)
    assert not Lb.suscan_analyzer_set_sweep_stratrgy(an, 1) and not Lb.suscan_analyzer_set_spectrum_partitioning(an, 1)
    assert not Lb.suscan_analyzer_set_buffering_size(an, 4096)
    st = {"psd": [], "fc": [], "info": []}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            st["psd"].append(np.ctypeslib.as_array(m.psd_data, shape=(N,)).copy())
            st["fc"].append(m.fc)
            k = len(st["psd"])
            if k == 2:
                assert Lb.suscan_analyzer_set_iq_reverse(an, 1) and Lb.suscan_analyzer_set_dc_remove(an, 1)
                assert Lb.suscan_analyzer_set_freq(an, 100e6, 0.0)
                assert Lb.suscan_analyzer_set_ppm(an, 1.5) and Lb.suscan_analyzer_set_bw(an, 2e5) and Lb.suscan_analyzer_set_agc(an, 1)
        elif t == suscan.MSG_SOURCE_INFO:
            si = C.cast(ptr, C.POINTER(suscan.SourceInfo)).contents
            st["info"].append((si.frequency, bool(si.iq_reverse), bool(si.dc_remove), si.ppm, si.bandwidth, bool(si.agc)))

    _pump(Lb, an, on_msg)
    assert len(st["psd"]) == nblocks
    # filter order and arguments: halve (prio 0) runs before spy (prio 10) although it was registered later; offsets count
    # delivered samples.  (The worker starts with the analyzer: the very first blocks may precede the registrations.)
    halved = {c[2] for c in calls if c[0] == "halve"}
    assert halved and all(o % L == 0 and c[1] == L for c in calls if c[0] == "halve" for o in [c[2]])
    assert {k * L for k in range(2, nblocks)} <= halved
    for i, c in enumerate(calls):
        if c[0] == "halve" and c[2] >= 2 * L:
            k = c[2] // L
            assert calls[i + 1][0] == "spy"                                         # priority order
            assert calls[i + 1][1] == pytest.approx(0.5 * xs[k * L].real, abs=1e-7) and calls[i + 1][2] == pytest.approx(0.5 * xs[k * L].imag, abs=1e-7)
    win = sdo.window(4, N)
    # frame labels: the new frequency from some block boundary after frame 2 on
    k0 = st["fc"].index(100000000)
    assert 2 <= k0 <= 4 and all(f == 433920000 for f in st["fc"][:k0]) and all(f == 100000000 for f in st["fc"][k0:])
    assert st["info"][-1] == (100e6, True, True, 1.5, 2e5, True) and st["info"][0][0] == 433.92e6
    # samples: plain -> (reversed) -> reversed with the DC level removed; the two switches are separate calls, so a block
    # in between may see only the first.  Every frame equals the oracle's for the mode it was processed in.
    dc, first, mode, modes = np.zeros(2, np.float32), True, 0, []
    for k in range(nblocks):
        blk = (xs[k * L:(k + 1) * L] * np.float32(0.5 if k * L in halved else 1.0)).astype(np.complex64)
        for m in range(mode, 3):
            trial = dc.copy()
            cand = blk if m == 0 else sdo.source_fix(blk, True, trial if m == 2 else None, 0.1, first)
            ref = sdo.psd_frames(cand, NAVG, N, N, win, navg=NAVG, scale=1.0 / N)[0]
            if np.max(np.abs(st["psd"][k] - ref)) / np.max(ref) < 2e-5:
                mode = m
                if m == 2:
                    dc, first = trial, False
                break
        else:
            raise AssertionError(f"frame {k} matches no source mode")
        modes.append(mode)
    assert modes[:2] == [0, 0] and modes[-1] == 2 and modes.index(2) <= 5
    lvl = 0.5 * np.complex64(0.05 - 0.02j)
    assert abs(dc[0] - lvl.imag) < 3e-3 and abs(dc[1] - lvl.real) < 3e-3             # the swapped level, halved by the filter
    tv = suscan.Timeval()
    Lb.suscan_analyzer_get_source_time(an, C.byref(tv))
    assert tv.tv_sec + 1e-6 * tv.tv_usec == pytest.approx(nblocks * L / FS, abs=1e-5)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_power_inspector_class(tmp_path, sdo):
    """class "power" (RMSInspector's server-side mode): every power.integrate-samples channel samples one SAMPLES value,
    the mean channel power -- equal to the reference's own integrator on the oracle's channel samples"""
    nblocks, nint = 10, 1500
    bw, fc = 50e3, -120e3
    x = (synth.psk_carriers(L * nblocks, [2 * fc / FS], sps=16, order=4, seed=4, snr_db=15) * np.float32(0.3)).astype(np.complex64)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 3 * FS, 0)
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"power", C.byref(ch), 1, -1, 3)
    st = {"psd": 0, "cfg_at": None, "vals": []}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            st["psd"] += 1
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                assert m.class_name == b"power" and m.estimator_count == 0
                cfg = Lb.suscan_config_dup(m.config)
                assert Lb.suscan_config_set_integer(cfg, b"power.integrate-samples", nint)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, cfg, 4)
                Lb.suscan_config_destroy(cfg)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfg_at"] = st["psd"]
        elif t == suscan.MSG_SAMPLES and st["cfg_at"] is not None:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            st["vals"].append(np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,)).copy().view(np.complex64))

    _pump(Lb, an, on_msg)
    b0 = st["cfg_at"]
    assert b0 is not None and b0 < nblocks - 4
    D = 8                                                  # pow2floor(1e6 / 100e3)
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    y = sdo.chan_feed(np.zeros(254, np.complex64), x[b0 * L:], 0, sdo.chan_modulate_taps(sdo.lpf_design(255, bw / FS), dp), D, 0, dp)
    want = sdo.Power(nint).feed(y)
    got = np.concatenate(st["vals"])
    assert got.size == want.size == y.size // nint and got.size > 20
    assert np.all(got.imag == 0) and np.max(np.abs(got.real - want.real) / want.real) < 2e-7
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_seek_estimators_and_tle(tmp_path, sdo):
    """Analyzer::seek moves the file position; setInspectorEnabled switches the baud estimators of estimator_list on and
    ESTIMATOR messages carry the baud in Hz (both within a few percent of the truth, each equal to its oracle on the same
    channel samples) and -- estimator 2, "carrier" -> afc.offset -- the channel's residual carrier in Hz (the inspector is
    opened 1.5 kHz beside its carrier; the estimate is the reference's CarrierDetector centroid on the channel samples);
    Doppler correction from a TLE is refused, its removal acknowledged"""
    nblocks = 14
    baud, bw, fc = 3906.25, 30e3, 100e3                     # 16 channel samples per symbol at equiv_fs = 62.5 kS/s
    x = synth.psk_carriers(L * nblocks, [2 * fc / FS], sps=int(FS / baud), order=4, seed=9, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L)
    Lb.suscan_analyzer_set_throttle_async(an, 2 * FS, 0)
    cls = [C.cast(Lb.suscan_estimator_class_lookup(n), C.POINTER(suscan.EstimatorClass)).contents for n in (b"baud-fac", b"baud-nonlinear", b"carrier")]
    assert [c.field for c in cls] == [b"clock.baud", b"clock.baud", b"afc.offset"] and not Lb.suscan_estimator_class_lookup(b"nope")
    ch = suscan.Channel(fc=fc, f_lo=fc - bw / 2, f_hi=fc + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 7)
    df = 1500.0                                             # a second inspector, this far below the carrier: only estimator 2
    fch = fc - df
    ch2 = suscan.Channel(fc=fch, f_lo=fch - bw / 2, f_hi=fch + bw / 2, bw=bw, ft=433.92e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch2), 1, -1, 8)
    st = {"psd": 0, "ts": [], "est": {0: [], 1: [], 2: [], "off": []}, "handle2": None, "ack": [], "kinds": [], "on_at": None, "handle": None}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            st["psd"] += 1
            st["ts"].append(m.timestamp.tv_sec + 1e-6 * m.timestamp.tv_usec)
            if st["psd"] == 9:
                tv = suscan.Timeval(0, int(2 * L / FS * 1e6))            # back to the third block
                assert Lb.suscan_analyzer_seek(an, C.byref(tv))
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            st["kinds"].append(m.kind)
            if m.kind == suscan.KIND_OPEN and m.req_id == 8:
                st["handle2"] = m.handle
                assert Lb.suscan_analyzer_inspector_estimator_cmd_async(an, m.handle, 2, 1, 70)
            elif m.kind == suscan.KIND_OPEN:
                assert m.estimator_count == 3
                names = C.cast(m.estimator_list, C.POINTER(C.c_char_p))
                assert [names[0], names[1], names[2]] == [b"baud-fac", b"baud-nonlinear", b"carrier"]
                st["handle"] = m.handle
                for eid in (0, 1, 2):
                    assert Lb.suscan_analyzer_inspector_estimator_cmd_async(an, m.handle, eid, 1, 50 + eid)
                assert Lb.suscan_analyzer_inspector_estimator_cmd_async(an, m.handle, 5, 1, 59)      # no such estimator
                assert Lb.suscan_analyzer_inspector_set_tle_async(an, m.handle, None, 60)            # disable: fine
                assert Lb.suscan_analyzer_inspector_set_tle_async(an, m.handle, 1, 61)               # any orbit: refused
            elif m.kind == suscan.KIND_ESTIMATOR:
                if m.req_id in (50, 51, 52):
                    st["ack"].append((m.req_id, m.estimator_id, m.enabled))
                    st["on_at"] = st["psd"]
                elif m.req_id == 70:
                    st["on2_at"] = st["psd"]
                elif m.handle == st["handle2"]:
                    assert m.estimator_id == 2
                    st["est"]["off"].append((st["psd"], m.value))
                else:
                    st["est"][m.estimator_id].append((st["psd"], m.value))
            elif m.req_id == 59:
                assert m.kind == suscan.KIND_WRONG_OBJECT
            elif m.req_id == 60:
                assert m.kind == suscan.KIND_SET_TLE and not m.enabled
            elif m.req_id == 61:
                assert m.kind == suscan.KIND_INVALID_ARGUMENT

    _pump(Lb, an, on_msg)
    assert sorted(st["ack"]) == [(50, 0, 1), (51, 1, 1), (52, 2, 1)]
    assert {suscan.KIND_WRONG_OBJECT, suscan.KIND_SET_TLE, suscan.KIND_INVALID_ARGUMENT} <= set(st["kinds"])
    # seek: 9 blocks, then the position jumps back to block 2 and the remaining 12 blocks follow
    assert st["psd"] == 9 + (nblocks - 2) or st["psd"] == 10 + (nblocks - 2)           # the request lands one block later at most
    back = [i for i in range(1, len(st["ts"])) if st["ts"][i] < st["ts"][i - 1]]
    assert len(back) == 1 and st["ts"][back[0]] == pytest.approx(2 * L / FS, abs=1e-5)
    # estimates: a value per block once enabled
    D, efs = 16, FS / 16
    for eid, tol in ((0, 0.07), (1, 0.01)):               # the autocorrelation valley is a whole number of samples
        vals = [v for _, v in st["est"][eid]]
        assert len(vals) >= st["psd"] - st["on_at"] - 1
        assert abs(np.median(vals[2:]) - baud) / baud < tol, (eid, vals)
    # against the oracle on the channel samples of the block after the estimators were switched on
    b0 = st["on_at"]                                                                # estimators see blocks b0, b0 + 1, ...
    car, off = [v for _, v in st["est"][2]], [v for _, v in st["est"]["off"]]
    assert len(car) >= st["psd"] - st["on_at"] - 1 and abs(np.median(car[2:])) < 0.1 * df, car           # on its carrier: ~ 0 Hz
    assert len(off) >= st["psd"] - st["on2_at"] - 1 and abs(np.median(off[2:]) - df) < 0.1 * df, off      # 1.5 kHz below it
    dp = sdo.fnor_to_dphase(-2 * fc / FS)
    taps = sdo.lpf_design(255, bw / FS)
    open_at = 0
    y = sdo.chan_feed(np.zeros(254, np.complex64), x[open_at * L:], 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)
    m = L // D
    blk = y[b0 * m:(b0 + 1) * m][:4096]
    nl = sdo.baud_nonlinear(blk) * efs
    got = dict(st["est"][1])[b0 + 1]                                                # messages of block b0 arrive before PSD b0 + 1
    assert abs(got - nl) / nl < 2e-3, (got, nl)
    # the carrier estimate of that block: the reference's CarrierDetector computation (oracle pinned to the compiled
    # Tasks/CarrierDetector.cpp, tests/test_ref_pin.py) on the same channel samples, avgRelBw 1/2, no DC notch
    cref = sdo.carrier_detect(blk, 0.5, 0.0) / (2 * np.pi) * efs
    cgot = dict(st["est"][2])[b0 + 1]
    assert abs(cgot - cref) < 2e-5 * efs, (cgot, cref)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_requests_while_two_blocks_are_in_flight(tmp_path, sdo):
    """a looping source at full speed while the GUI thread opens, reconfigures, retunes and closes inspectors: requests
    drain the two-block pipeline before they touch anything, so every reply arrives, nothing crashes or hangs, and an
    inspector that was left alone throughout still delivers exactly the oracle's symbols"""
    import time
    nloop = 6
    baud, bw = 15625.0, 30e3
    fcs = [-200e3, 100e3, 260e3]
    x = synth.psk_carriers(L * nloop, [2 * f / FS for f in fcs], sps=int(FS / baud), order=4, seed=13, snr_db=25)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L, loop=True)
    Lb.suscan_analyzer_set_throttle_async(an, 0, 0)

    def chan(fc):
        return suscan.Channel(fc=float(fc), f_lo=float(fc - bw / 2), f_hi=float(fc + bw / 2), bw=bw, ft=433.92e6)

    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(chan(fcs[0])), 1, -1, 1)        # the quiet one
    st = {"psd": 0, "quiet": None, "quiet_cfg_at": None, "quiet_syms": [], "opened": [], "closed": 0, "cfgs": 0, "replies": 0,
          "looped_at": None, "next_req": 100, "live": []}

    def configure(handle, req, order=2):
        cfg_desc = Lb.suscan_inspector_config_desc(b"psk")
        cfg = Lb.suscan_config_new(cfg_desc)
        Lb.suscan_config_set_integer(cfg, b"afc.costas-order", order)
        Lb.suscan_config_set_float(cfg, b"afc.loop-bw", 40.0)
        Lb.suscan_config_set_integer(cfg, b"clock.type", 1)
        Lb.suscan_config_set_float(cfg, b"clock.baud", baud)
        assert Lb.suscan_analyzer_set_inspector_config_async(an, handle, cfg, req)
        Lb.suscan_config_destroy(cfg)

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            st["psd"] += 1
            if m.looped and st["looped_at"] is None:
                st["looped_at"] = st["psd"]
            k = st["psd"]
            if k >= 60:
                Lb.suscan_analyzer_req_halt(an)
            elif k % 3 == 0:                                   # churn: open one, poke the live ones, close the oldest
                st["next_req"] += 1
                assert Lb.suscan_analyzer_open_ex_async(an, b"psk" if k % 2 else b"fsk", C.byref(chan(fcs[1 + k % 2])), 1, -1, st["next_req"])
                for h in st["live"][-2:]:
                    assert Lb.suscan_analyzer_set_inspector_freq_overridable(an, h, fcs[1] + 1e3 * (k % 7))
                    assert Lb.suscan_analyzer_inspector_set_spectrum_async(an, h, 1 + k % 5, 0)
                if len(st["live"]) > 3:
                    assert Lb.suscan_analyzer_close_async(an, st["live"].pop(0), 7)
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            st["replies"] += 1
            if m.kind == suscan.KIND_OPEN:
                if m.req_id == 1:
                    st["quiet"] = m.handle
                    assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, 4242, 2)
                    configure(m.handle, 3)
                else:
                    st["opened"].append(m.handle)
                    st["live"].append(m.handle)
                    configure(m.handle, 50, order=1 + m.handle % 3)
            elif m.kind == suscan.KIND_SET_CONFIG:
                st["cfgs"] += 1
                if m.req_id == 3:
                    st["quiet_cfg_at"] = st["psd"]
            elif m.kind == suscan.KIND_CLOSE:
                st["closed"] += 1
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            v = np.ctypeslib.as_array(m.samples, shape=(m.sample_count * 2,))
            assert np.all(np.isfinite(v))
            if m.inspector_id == 4242 and st["quiet_cfg_at"] is not None:
                st["quiet_syms"].append(v.copy().view(np.complex64))

    t0 = time.time()
    seen = _pump(Lb, an, on_msg, limit=200000)
    assert seen[-1] == suscan.MSG_HALT and time.time() - t0 < 60
    assert st["psd"] >= 60 and len(st["opened"]) >= 15 and st["closed"] >= 10 and st["cfgs"] >= 15
    # the quiet inspector: its symbols are the oracle chain's on the looping stream from the block its configuration
    # landed at (at full speed that may be several loops in: the stream is periodic, the chain runs through the seams)
    b0 = st["quiet_cfg_at"]
    assert b0 is not None and b0 < 40
    D, efs = 16, FS / 16
    sps = efs / baud
    dp = sdo.fnor_to_dphase(-2 * fcs[0] / FS)
    xin = np.concatenate([x[(b0 % nloop) * L:], x])
    y = sdo.chan_feed(np.zeros(254, np.complex64), xin, 0, sdo.chan_modulate_taps(sdo.lpf_design(255, bw / FS), dp), D, 0, dp)
    a_ = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), y)
    z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, min(2.0 / sps, 0.95), 3, 2 * 40.0 / efs), a_)
    ref = sdo.clock_feed_bulk(sdo.clock_new(0.2, baud / efs), z)
    got = np.concatenate(st["quiet_syms"])
    k = min(len(got), len(ref))
    assert k > 1000
    assert np.array_equal(got[:k].view(np.uint32), ref[:k].view(np.uint32))
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))


def test_halt_wakes_the_reader_and_bad_source_reports_failure(tmp_path):
    x = synth.tone_noise(L * 2, seed=1)
    path = tmp_path / "iq.raw"
    x.tofile(path)
    Lb, mq, an = _start(path, L, loop=True)                  # endless source
    n = {"psd": 0}

    def on_msg(t, ptr):
        if t == suscan.MSG_PSD:
            n["psd"] += 1
            if n["psd"] == 3:
                Lb.suscan_analyzer_req_halt(an)              # Analyzer::halt (Suscan/Analyzer.cpp:318-322)

    seen = _pump(Lb, an, on_msg)
    assert seen[-1] == suscan.MSG_HALT and n["psd"] >= 3
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    # missing file: SOURCE_INIT failure with a message, then HALT (App/Application.cpp:526-538)
    Lb, mq, an = _start(tmp_path / "nope.raw", L)
    got = {}

    def on_msg2(t, ptr):
        if t == suscan.MSG_SOURCE_INIT:
            m = C.cast(ptr, C.POINTER(suscan.StatusMsg)).contents
            got["code"], got["msg"] = m.code, m.err_msg

    seen = _pump(Lb, an, on_msg2)
    assert got["code"] == -1 and b"nope.raw" in got["msg"] and seen[-1] == suscan.MSG_HALT
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
