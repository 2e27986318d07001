"""The reference's own Suscan::Analyzer class on the MI355X.

oracle/_ref/ref_live (oracle/ref_live.cpp + the reference's Suscan/*.cpp compiled unchanged, built where /root/reference is
and shipped with the snapshot) opens a file source through Suscan::Source::Config, lets Suscan::AnalyzerRequestTracker open
a "psk" inspector, pushes an inspector config through Suscan::Config, and collects PSDMessage / SamplesMessage objects
from the Qt signals.  Its output must equal what the same requests give through the ctypes binding of the same ABI
(bit for bit: one library), and its PSD frame the oracle's."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from sigdigger_amd import suscan, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIVE = os.path.join(ROOT, "oracle", "_ref", "ref_live")
FS, N = 1_000_000, 4096


def _run_ref_live(iq, out, fc, bw, cls, baud=0.0):
    r = subprocess.run([REF_LIVE, str(iq), str(FS), str(N), str(fc), str(bw), str(out), cls, str(baud)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    magic, npsd, psd_size, nb = struct.unpack_from("<4I", raw, 0)
    (ns,) = struct.unpack_from("<Q", raw, 16)
    bb, eq, bwr = struct.unpack_from("<3f", raw, 24)
    (seen_id,) = struct.unpack_from("<I", raw, 36)
    assert magic == 0x31564C52
    psd = np.frombuffer(raw, dtype=np.float32, count=psd_size, offset=40)
    sam = np.frombuffer(raw, dtype=np.complex64, count=ns, offset=40 + 4 * psd_size)
    return dict(npsd=npsd, nbatches=nb, fs=bb, equiv_fs=eq, bw=bwr, id=seen_id, psd=psd.copy(), samples=sam.copy())


def _run_ctypes(iq, fc, bw, equiv_fs, cls):
    """The same requests, in the same order, through the ctypes binding."""
    Lb = suscan.load()
    mq = suscan.MQ()
    assert Lb.suscan_mq_init(C.byref(mq))
    cfg = Lb.suscan_source_config_new(b"file", 1)
    Lb.suscan_source_config_set_samp_rate(cfg, FS)
    Lb.suscan_source_config_set_freq(cfg, 100e6)
    assert Lb.suscan_source_config_set_path(cfg, str(iq).encode())
    p = suscan.AnalyzerParams.default()
    p.detector_params.window_size = N
    p.detector_params.window = 4
    p.psd_update_int = 0.01
    an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
    assert an
    Lb.suscan_source_config_destroy(cfg)
    psd0, samples, state = [], [], {}
    while True:
        t, ptr = suscan.read_message(Lb, mq, 60.0)
        if t == suscan.MSG_HALT:
            break
        if t == suscan.MSG_SOURCE_INFO and "req" not in state:
            ch = suscan.Channel(fc=fc, f_lo=-bw / 2, f_hi=bw / 2, bw=bw, ft=0)
            assert Lb.suscan_analyzer_open_ex_async(an, cls.encode(), C.byref(ch), 0, -1, 1)
            state["req"] = True
        elif t == suscan.MSG_INSPECTOR:
            m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
            if m.kind == suscan.KIND_OPEN:
                state["handle"] = m.handle
                assert Lb.suscan_analyzer_set_inspector_id_async(an, m.handle, state.get("id", 1), 2)
            elif m.kind == suscan.KIND_SET_ID and cls == "psk":
                desc = Lb.suscan_inspector_config_desc(b"psk")
                c = Lb.suscan_config_new(desc)
                assert Lb.suscan_config_set_integer(c, b"afc.costas-order", 2)
                assert Lb.suscan_config_set_integer(c, b"afc.bits-per-symbol", 2)
                assert Lb.suscan_config_set_integer(c, b"clock.type", 1)
                assert Lb.suscan_config_set_float(c, b"clock.baud", np.float32(FS / 16))
                assert Lb.suscan_config_set_bool(c, b"clock.running", 1)
                assert Lb.suscan_analyzer_set_inspector_config_async(an, state["handle"], c, 3)
                Lb.suscan_config_destroy(c)
        elif t == suscan.MSG_PSD and not psd0:
            m = C.cast(ptr, C.POINTER(suscan.PSDMsg)).contents
            psd0.append(np.ctypeslib.as_array(m.psd_data, shape=(N,)).copy())
        elif t == suscan.MSG_SAMPLES:
            m = C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents
            samples.append(np.ctypeslib.as_array(m.samples, shape=(2 * m.sample_count,)).copy().view(np.complex64))
        Lb.suscan_analyzer_dispose_message(t, ptr)
    Lb.suscan_analyzer_destroy(an)
    Lb.suscan_mq_finalize(C.byref(mq))
    return psd0[0], np.concatenate(samples) if samples else np.zeros(0, np.complex64)


@pytest.mark.skipif(not os.path.exists(REF_LIVE), reason="oracle/_ref/ref_live was not built (it needs /root/reference)")
def test_reference_analyzer_class_drives_the_gpu_library(tmp_path, sdo):
    nblk = 12
    L = int(FS * 0.01 // N) * N or N                       # the analyzer's block: whole PSD windows per update interval
    x = synth.psk_carriers(40 * N * nblk, [0.2], sps=16, order=4, seed=21, snr_db=25)
    iq = tmp_path / "iq.f32"
    x.tofile(iq)
    fc, bw = 0.1 * FS, 0.12 * FS
    # "raw": the channel samples themselves.  A request lands on whatever block boundary the worker has reached, so the
    # two runs open the channel at different instants; the channeliser carries no state beyond its taps, hence the tails
    # of the two streams (both end at the end of the file) are bit-identical
    ref = _run_ref_live(iq, tmp_path / "o.bin", fc, bw, "raw")
    assert ref["npsd"] >= nblk and ref["nbatches"] > 0 and ref["samples"].size > 1000
    assert ref["fs"] == FS and 0 < ref["equiv_fs"] <= FS and ref["id"] != 0
    psd, samples = _run_ctypes(iq, fc, bw, ref["equiv_fs"], "raw")
    # PSDMessage's constructor has shifted the frame and taken dB in place (Suscan/Messages/PSDMessage.cpp:26-39)
    assert np.array_equal(ref["psd"], sdo.psd_shift_db(psd))
    # (a request lands on whatever block boundary the worker has reached and a channel's oscillator starts when the channel
    # is opened: the two streams may start at different samples and differ by one constant phase.
    # Align on the reference run's last kilo-sample by magnitude, then require one constant unit phasor between them.)
    a, b = ref["samples"], samples
    assert a.size > 3000 and b.size > 3000
    probe = a[-2000:-1000]
    mag_b, mag_p = np.abs(b), np.abs(probe)
    cand = np.flatnonzero(np.abs(mag_b[:b.size - probe.size + 1] - mag_p[0]) <= 1e-5 * mag_p.max())
    pos = [int(h) for h in cand if np.max(np.abs(mag_b[h:h + probe.size] - mag_p)) <= 1e-5 * mag_p.max()]
    assert len(pos) == 1, "the reference run's samples are not a stretch of the ctypes run's stream"
    n = min(a.size - 2000, pos[0], 4096)
    assert n > 500
    ra, rb = a[a.size - 2000 - n:a.size - 1000], b[pos[0] - n:pos[0] + 1000]
    rot = np.vdot(ra, rb) / np.vdot(ra, ra)
    assert abs(abs(rot) - 1) < 1e-5
    assert np.max(np.abs(rb - rot * ra)) <= 1e-5 * np.abs(ra).max()
    # "psk" with a config pushed through Suscan::Config: recovered QPSK symbols (the loops' state depends on when the
    # config took effect, so this is a constellation check, not a bit comparison)
    psk = _run_ref_live(iq, tmp_path / "p.bin", fc, bw, "psk", baud=FS / 16)
    sym = psk["samples"][-2000:]
    assert sym.size == 2000
    ang = np.angle(sym[np.abs(sym) > 0.3 * np.median(np.abs(sym))])
    off = np.abs(((ang - np.pi / 4) + np.pi / 4) % (np.pi / 2) - np.pi / 4)
    assert np.median(off) < 0.2
