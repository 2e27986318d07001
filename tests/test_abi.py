"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/sigdigger_amd.h declares, fails loudly (no CPU fallback) when no device exists,
and the product package never reaches into oracle/."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sigdigger_amd.h")
SUSCAN_HEADER = os.path.join(ROOT, "include", "suscan_amd.h")


def declared_symbols(header=HEADER, prefix="suamd_"):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"SUAMD_API[^;(]*?\b(" + prefix + r"\w+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 35
    for must in ("suamd_psd_feed", "suamd_chanbank_feed", "suamd_costas_bank_feed", "suamd_clock_bank_feed",
                 "suamd_agc_bank_feed", "suamd_pll_bank_feed", "suamd_quad_demod_batch", "suamd_xlate_bulk"):
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from sigdigger_amd import build, lib
    build.build()
    so = ctypes.CDLL(lib.SO_PATH)
    missing = [s for s in declared_symbols() if not hasattr(so, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    # the ctypes prototype table covers the header one to one
    assert sorted(lib.PROTOTYPES) == declared_symbols()
    lib.load()


def test_sigutils_headers_declare_only_what_the_library_exports():
    """include/sigutils/*.h (the per-sample su_* calls of the offline Tasks and su_specttuner_*): every declared symbol is
    exported by the product library -- no compute call here, the symbols only"""
    from sigdigger_amd import build, lib
    build.build()
    so = ctypes.CDLL(lib.SO_PATH)
    total = 0
    for h in ("ncqo.h", "pll.h", "agc.h", "clock.h", "iir.h", "taps.h", "specttuner.h"):
        syms = declared_symbols(os.path.join(ROOT, "include", "sigutils", h), "su_")
        assert syms, h
        missing = [s for s in syms if not hasattr(so, s)]
        assert not missing, f"{h}: declared but not exported: {missing}"
        total += len(syms)
    assert total >= 30


def test_live_path_abi_exports_and_headers_are_valid_c(tmp_path):
    """include/suscan_amd.h: every declared suscan_* symbol is exported, the ctypes table mirrors it,
    and both headers compile as plain C (the reference binds them from C/C++)."""
    import subprocess
    from sigdigger_amd import build, lib, suscan
    build.build()
    so = ctypes.CDLL(lib.SO_PATH)
    syms = declared_symbols(SUSCAN_HEADER, "suscan_")
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(so, s)]
    assert not missing, f"declared in suscan_amd.h but not exported: {missing}"
    assert sorted(suscan.PROTOTYPES) == syms
    src = tmp_path / "t.c"
    src.write_text('#include "suscan_amd.h"\n'
                   'int main(void) { struct suscan_analyzer_params p = suscan_analyzer_params_INITIALIZER;\n'
                   '  struct sigutils_channel c = sigutils_channel_INITIALIZER; struct suamd_agc_params a = '
                   'suamd_agc_params_INITIALIZER;\n  return (int)p.detector_params.window_size + (int)c.bw + '
                   '(int)a.hang_max == 0; }\n')
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_message_queue_and_config_without_a_gpu():
    """the queue and the inspector config vocabulary are host code (Suscan/MQ.cpp, Suscan/Config.cpp)"""
    from sigdigger_amd import suscan
    L = suscan.load()
    mq = suscan.MQ()
    assert L.suscan_mq_init(ctypes.byref(mq))
    t = ctypes.c_uint32(0)
    p = ctypes.c_void_p(0)
    assert not L.suscan_mq_poll(ctypes.byref(mq), ctypes.byref(t), ctypes.byref(p))
    L.suscan_mq_write(ctypes.byref(mq), suscan.MSG_HALT, None)
    assert L.suscan_mq_poll(ctypes.byref(mq), ctypes.byref(t), ctypes.byref(p)) and t.value == suscan.MSG_HALT
    L.suscan_mq_finalize(ctypes.byref(mq))
    desc = L.suscan_inspector_config_desc(b"psk")
    assert desc and L.suscan_inspector_config_desc(b"fsk") and L.suscan_inspector_config_desc(b"audio")
    assert not L.suscan_inspector_config_desc(b"drm")                              # an unknown class
    cfg = L.suscan_config_new(desc)
    # key vocabulary of Default/GenericInspector/InspectorCtl/{Afc,Clock,Gain}Control.cpp
    assert L.suscan_config_set_integer(cfg, b"afc.costas-order", 2)
    assert L.suscan_config_set_float(cfg, b"clock.baud", 250e3)
    assert L.suscan_config_set_bool(cfg, b"agc.enabled", 1)
    assert not L.suscan_config_set_float(cfg, b"afc.costas-order", 1.0)      # wrong type
    assert not L.suscan_config_set_integer(cfg, b"no.such.key", 1)
    dup = L.suscan_config_dup(cfg)
    assert L.suscan_config_get_value(dup, b"clock.baud")
    L.suscan_config_destroy(dup)
    L.suscan_config_destroy(cfg)
    assert L.suscan_inspector_config_desc(b"ask")                            # InspectorCtl/AskControl.cpp vocabulary
    # the registries InspectorMessage.cpp:44-61 consults for the lists of the OPEN reply

    class Cls(ctypes.Structure):
        _fields_ = [("name", ctypes.c_char_p), ("desc", ctypes.c_char_p)]

    c = L.suscan_spectsrc_class_lookup(b"cyclo")
    assert c and ctypes.cast(c, ctypes.POINTER(Cls)).contents.name == b"cyclo" and ctypes.cast(c, ctypes.POINTER(Cls)).contents.desc
    assert not L.suscan_spectsrc_class_lookup(b"nope") and not L.suscan_estimator_class_lookup(b"baud")


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the no-device path cannot be exercised")
    from sigdigger_amd import engine, lib
    with pytest.raises(lib.SigDiggerAmdError) as e:
        engine.Context(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_host_side_helpers_do_not_need_a_device(sdo):
    """frequency -> phase-step conversion and tap design are host code; they match the oracle."""
    import numpy as np
    from sigdigger_amd import lib
    L = lib.load()
    for f in (0.0, 0.25, -0.25, 0.999999, -1.0, 1e-9, 0.123456789):
        assert L.suamd_fnor_to_dphase(f) == sdo.fnor_to_dphase(f)
    for ntaps, fc in ((255, 0.8 / 64), (31, 0.5), (64, 0.1), (1, 0.3)):
        h = np.empty(ntaps, dtype=np.float32)
        L.suamd_lpf_design(h.ctypes.data_as(ctypes.c_void_p), ntaps, fc)
        assert np.array_equal(h, sdo.lpf_design(ntaps, fc))


def test_product_never_imports_the_oracle():
    """nothing under sigdigger_amd/ imports, includes, links or dlopens anything of oracle/"""
    pkg = os.path.join(ROOT, "sigdigger_amd")
    bad = re.compile(r"(from|import)\s+oracle|oracle/|oracle\.sdo|libsdo|sdo\.h|\bsdo_\w+|\bsdo\.")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                m = bad.search(txt)
                assert m is None, f"{f} references the oracle: {m.group(0)!r}"
    # and the shared library does not depend on it
    import subprocess
    from sigdigger_amd import lib
    needed = subprocess.run(["readelf", "-d", lib.SO_PATH], capture_output=True, text=True).stdout
    assert "sdo" not in needed


def test_psd_ttl_rule_literal_transcription():
    """A5, host-only: UIMediator::feedPSD's expiry rule (UIMediator/SpectrumMediator.cpp:35-85,127) against a line-by-line
    Python transcription -- frames that arrive late by more than the TTL (after the calibration phase) are dropped unless
    the source looped"""
    import ctypes as C
    import math
    import numpy as np
    from sigdigger_amd import lib as sdlib

    class TTL(C.Structure):
        _fields_ = [("rt_delta_real", C.c_double), ("rt_calibrations", C.c_uint), ("have_rt_delta", C.c_int)]

    L = sdlib.load()
    rng = np.random.default_rng(2)
    st = TTL(0.0, 0, 0)
    m_rtDeltaReal, m_rtCalibrations, m_haveRtDelta = 0.0, 0, False
    alpha = 1.0 - math.exp(-1.0 / 10)
    now, dropped = 100.0, 0
    for k in range(200):
        now += 0.04
        lag = 0.003 + (0.5 if k in (50, 51, 120) else 0.0) + 1e-4 * rng.standard_normal()
        rt = now - lag
        looped = k == 51
        delta = now - rt
        expired = False
        first = m_rtCalibrations == 0
        m_rtCalibrations += 1
        if first:
            m_rtDeltaReal = delta
        else:
            m_rtDeltaReal += alpha * (delta - m_rtDeltaReal)
        if not m_haveRtDelta:
            m_rtCalibrations += 1
            if m_rtCalibrations > 10:
                m_haveRtDelta = True
        else:
            delta -= m_rtDeltaReal
            expired = delta > 100e-3
        want = (not expired) or looped
        got = bool(L.suamd_psd_ttl_accept(C.byref(st), now, rt, 100.0, int(looped)))
        assert got == want, k
        assert st.rt_delta_real == m_rtDeltaReal and st.rt_calibrations == m_rtCalibrations
        dropped += not got
    assert dropped == 2                                    # frames 50 and 120; 51 is late too but the source looped


def test_source_config_round_trips_what_source_cpp_reads():
    """Suscan::Source::Config's getters (Suscan/Source.cpp:120-410) on a config built the way ProfileConfigTab does; no GPU
    involved (a source config is host state)."""
    import ctypes as C
    from sigdigger_amd import suscan
    L = suscan.load()
    c = L.suscan_source_config_new(b"file", 4)
    assert L.suscan_source_config_get_type(c) == b"file" and L.suscan_source_config_get_format(c) == 4
    assert L.suscan_source_config_get_path(c) is None and L.suscan_source_config_get_antenna(c) is None
    L.suscan_source_config_set_samp_rate(c, 2_400_000)
    L.suscan_source_config_set_freq(c, 433.92e6)
    L.suscan_source_config_set_lnb_freq(c, 9.75e9)
    assert L.suscan_source_config_set_average(c, 4) and not L.suscan_source_config_set_average(c, 0)
    L.suscan_source_config_set_bandwidth(c, 2e6)
    L.suscan_source_config_set_ppm(c, 1.5)
    L.suscan_source_config_set_dc_remove(c, 1)
    L.suscan_source_config_set_iq_balance(c, 1)
    L.suscan_source_config_set_loop(c, 1)
    assert L.suscan_source_config_set_label(c, b"capture") and L.suscan_source_config_set_antenna(c, b"RX2")
    assert L.suscan_source_config_set_gain(c, b"LNA", 12.5)
    assert L.suscan_source_config_set_param(c, b"k1", b"v1") and L.suscan_source_config_set_param(c, b"k2", b"v2")
    L.suscan_source_config_set_start_time(c, suscan.Timeval(1000, 250000))
    d = L.suscan_source_config_clone(c)
    L.suscan_source_config_destroy(c)
    assert L.suscan_source_config_get_samp_rate(d) == 2_400_000 and L.suscan_source_config_get_average(d) == 4
    assert L.suscan_source_config_get_freq(d) == 433.92e6 and L.suscan_source_config_get_lnb_freq(d) == 9.75e9
    assert L.suscan_source_config_get_bandwidth(d) == 2e6 and L.suscan_source_config_get_ppm(d) == 1.5
    assert L.suscan_source_config_get_dc_remove(d) and L.suscan_source_config_get_iq_balance(d) and L.suscan_source_config_get_loop(d)
    assert L.suscan_source_config_get_label(d) == b"capture" and L.suscan_source_config_get_antenna(d) == b"RX2"
    assert L.suscan_source_config_get_gain(d, b"LNA") == 12.5 and L.suscan_source_config_get_gain(d, b"nope") == 0
    assert L.suscan_source_config_get_param(d, b"k2") == b"v2" and L.suscan_source_config_get_param(d, b"zz") is None
    seen = []
    cb = suscan.WALK_PARAMS(lambda cfg, k, v, u: (seen.append((k, v)), 1)[1])
    assert L.suscan_source_config_walk_params(d, cb, None) and seen == [(b"k1", b"v1"), (b"k2", b"v2")]
    tv = suscan.Timeval()
    L.suscan_source_config_get_start_time(d, C.byref(tv))
    assert (tv.tv_sec, tv.tv_usec) == (1000, 250000)
    assert not L.suscan_source_config_is_real_time(d) and L.suscan_source_config_is_seekable(d)
    assert not L.suscan_source_config_file_is_valid(d) and not L.suscan_source_config_get_end_time(d, C.byref(tv))
    lo, hi = C.c_double(), C.c_double()
    assert L.suscan_source_config_get_freq_limits(d, C.byref(lo), C.byref(hi)) and lo.value < 0 < hi.value
    L.suscan_source_config_set_type_format(d, b"tonegen", 1)
    assert L.suscan_source_config_get_type(d) == b"tonegen" and not L.suscan_source_config_is_seekable(d)
    L.suscan_source_config_clear_params(d)
    assert L.suscan_source_config_get_param(d, b"k1") is None
    L.suscan_source_config_destroy(d)
    # source info: init / deep copy / finalize (include/Suscan/Analyzer.h:50-105)
    a, b = suscan.SourceInfo(), suscan.SourceInfo()
    L.suscan_source_info_init(C.byref(a))
    a.frequency, a.history_length = 1e9, 77
    assert L.suscan_source_info_init_copy(C.byref(b), C.byref(a)) and b.frequency == 1e9 and b.history_length == 77
    L.suscan_source_info_finalize(C.byref(b))
    L.suscan_source_info_finalize(C.byref(a))


def test_tuning_struct_through_the_abi():
    """csrc/tuning.hpp: one documented struct for every knob that changes HOW the library computes; the SUAMD_* environment
    is read once, suamd_tuning_set / _get / _describe / _reset do the rest (no GPU involved)."""
    from sigdigger_amd import engine
    fields = engine.tuning_fields()
    names = {f[0] for f in fields}
    assert len(fields) >= 25 and {"st_slots", "st_seam", "fir_stream", "psd_large_batch", "clock_mode", "analyzer_subranges"} <= names
    for name, env, dflt, lo, hi, doc in fields:
        assert env.startswith("SUAMD_") and lo <= dflt <= hi and len(doc) > 10, name
        assert engine.tuning_get(name) == engine.tuning_get(env)
    old = engine.tuning_get("clock_mode")
    try:
        engine.tuning_set("SUAMD_CLOCK_MODE", 1)
        assert engine.tuning_get("clock_mode") == 1
        with engine.tuned(clock_mode=0, st_slots=1024):
            assert engine.tuning_get("clock_mode") == 0 and engine.tuning_get("SUAMD_ST_SLOTS") == 1024
        assert engine.tuning_get("clock_mode") == 1
        import pytest
        with pytest.raises(Exception):
            engine.tuning_set("clock_mode", 7)                   # outside the field's range
        with pytest.raises(Exception):
            engine.tuning_set("no_such_field", 1)
    finally:
        engine.tuning_set("clock_mode", old)
    # no translation unit of the product reads a tuning variable behind the struct's back
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envs = {f[1] for f in fields}
    for path in glob.glob(os.path.join(root, "sigdigger_amd", "csrc", "*")):
        if path.endswith(("tuning.cpp", "tuning.hpp", ".o", ".sha256")):
            continue
        for m in re.finditer(r'getenv\("(SUAMD_[A-Z0-9_]+)"\)', open(path, errors="ignore").read()):
            assert m.group(1) not in envs, (os.path.basename(path), m.group(1))
