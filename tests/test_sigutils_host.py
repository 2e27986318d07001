"""The per-sample libsigutils calls of the offline Tasks, served by the product on the host (csrc/sigutils_host.cpp,
include/sigutils/{ncqo,pll,agc,clock,iir,taps}.h): called here one sample at a time through the C ABI, exactly as
Tasks/CostasRecoveryTask.cpp:58-61 & co. call them, and compared with the oracle's restatement of SPEC.md D - H.

BIT-EXACT: the product's host code and oracle/sdo.c are two independent implementations of the same fixed sequences of
binary32 operations (the product's are the device kernels' primitives, csrc/sd_math.hpp, compiled for the host).
CPU only -- these entry points are host code by the reference's own contract (state structs by value, one call per
sample); the GPU block forms are pinned against the same oracle in tests/test_gpu_parity.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigdigger_amd import lib as _lib  # noqa: E402


class CF(C.Structure):
    """SUCOMPLEX by value: two floats in one SSE register, the x86-64 ABI of float _Complex and std::complex<float>"""
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class Ncqo(C.Structure):
    _fields_ = [("phase", C.c_uint32), ("dphase", C.c_uint32), ("n", C.c_uint64)]


class Pll(C.Structure):
    _fields_ = [("phase", C.c_uint32), ("omega", C.c_float), ("alpha", C.c_float), ("beta", C.c_float)]


class Costas(C.Structure):
    _fields_ = [("kind", C.c_int), ("phase", C.c_uint32), ("omega", C.c_float), ("a", C.c_float), ("b", C.c_float),
                ("gain", C.c_float), ("order", C.c_int), ("fb", C.c_float * 5), ("fa", C.c_float * 5),
                ("xh", C.c_float * 10), ("yh", C.c_float * 10)]


class AgcParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("slope_factor", C.c_float), ("hang_max", C.c_uint),
                ("delay_line_size", C.c_uint), ("mag_history_size", C.c_uint),
                ("fast_rise_t", C.c_float), ("fast_fall_t", C.c_float), ("slow_rise_t", C.c_float), ("slow_fall_t", C.c_float)]


class Agc(C.Structure):
    _fields_ = [("knee", C.c_float), ("gain_slope", C.c_float), ("far", C.c_float), ("faf", C.c_float), ("sar", C.c_float),
                ("saf", C.c_float), ("hang_max", C.c_uint), ("hang_n", C.c_uint), ("delay_line_size", C.c_uint),
                ("mag_history_size", C.c_uint), ("delay_ptr", C.c_uint), ("hist_ptr", C.c_uint), ("fast_level", C.c_float),
                ("slow_level", C.c_float), ("delay_line", C.c_float * 128), ("mag_history", C.c_float * 64)]


class Clock(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("gain", C.c_float), ("phi", C.c_float), ("bnor", C.c_float),
                ("bmin", C.c_float), ("bmax", C.c_float), ("halfcycle", C.c_int), ("prev", C.c_float * 2), ("x0", C.c_float * 2),
                ("x1", C.c_float * 2), ("x2", C.c_float * 2), ("buf", C.c_void_p), ("size", C.c_uint64), ("avail", C.c_uint64)]


@pytest.fixture(scope="module")
def L():
    lib = _lib.load()
    for name, res, args in [
        ("su_ncqo_init", None, [C.POINTER(Ncqo), C.c_float]), ("su_ncqo_set_phase", None, [C.POINTER(Ncqo), C.c_float]),
        ("su_ncqo_read", CF, [C.POINTER(Ncqo)]),
        ("su_pll_init", C.c_int, [C.POINTER(Pll), C.c_float, C.c_float]), ("su_pll_track", CF, [C.POINTER(Pll), CF]),
        ("su_costas_init", C.c_int, [C.POINTER(Costas), C.c_int, C.c_float, C.c_float, C.c_uint, C.c_float]),
        ("su_costas_feed", CF, [C.POINTER(Costas), CF]),
        ("su_agc_init", C.c_int, [C.POINTER(Agc), C.POINTER(AgcParams)]), ("su_agc_feed", CF, [C.POINTER(Agc), CF]),
        ("su_clock_detector_init", C.c_int, [C.POINTER(Clock), C.c_float, C.c_float, C.c_uint64]),
        ("su_clock_detector_feed", None, [C.POINTER(Clock), CF]),
        ("su_clock_detector_read", C.c_int64, [C.POINTER(Clock), C.c_void_p, C.c_size_t]),
        ("su_clock_detector_finalize", None, [C.POINTER(Clock)]),
        ("su_taps_apply_blackmann_harris_complex", None, [C.c_void_p, C.c_uint64]),
    ]:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


def cnoise(n, seed, scale=1.0):
    r = np.random.default_rng(seed)
    return (scale * (r.standard_normal(n) + 1j * r.standard_normal(n))).astype(np.complex64)


def _per_sample(fn, state, x):
    out = np.empty(x.size, dtype=np.complex64)
    for i, v in enumerate(x):
        r = fn(C.byref(state), CF(float(v.real), float(v.imag)))
        out[i] = np.complex64(complex(r.re, r.im))
    return out


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_header_struct_sizes_match_these_mirrors():
    """by-value structs are ABI: the mirrors above must have the layout the headers declare (compiled probe)"""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <sigutils/ncqo.h>\n#include <sigutils/pll.h>\n#include <sigutils/agc.h>\n#include <sigutils/clock.h>\n'
           '#include <sigutils/iir.h>\n#include <sigutils/taps.h>\n#include <stdio.h>\n'
           'int main(void){ su_ncqo_t n = su_ncqo_INITIALIZER; su_pll_t p = su_pll_INITIALIZER; su_costas_t c = su_costas_INITIALIZER;\n'
           ' su_agc_t a = su_agc_INITIALIZER; su_clock_detector_t k = su_clock_detector_INITIALIZER; su_iir_filt_t f = su_iir_filt_INITIALIZER;\n'
           ' struct su_agc_params ap = su_agc_params_INITIALIZER; (void)n; (void)p; (void)c; (void)a; (void)k; (void)f; (void)ap;\n'
           ' printf("%zu %zu %zu %zu %zu %zu\\n", sizeof n, sizeof p, sizeof c, sizeof a, sizeof k, sizeof ap); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "p")], text=True).split()]
    assert sizes == [C.sizeof(Ncqo), C.sizeof(Pll), C.sizeof(Costas), C.sizeof(Agc), C.sizeof(Clock), C.sizeof(AgcParams)]


@pytest.mark.parametrize("fnor,phase", [(0.1234, 0.7), (-0.9, -2.0), (1e-4, 0.0)])
def test_ncqo_reads_the_closed_form_phasor(L, sdo, fnor, phase):
    n = Ncqo()
    L.su_ncqo_init(C.byref(n), fnor)
    L.su_ncqo_set_phase(C.byref(n), phase)
    got = np.array([complex(r.re, r.im) for r in (L.su_ncqo_read(C.byref(n)) for _ in range(3000))], dtype=np.complex64)
    dp = sdo.fnor_to_dphase(np.float32(fnor))
    p0 = int(round(float(np.float32(phase)) / (2 * np.pi) * 2 ** 32)) & 0xFFFFFFFF
    ref = sdo.xlate_bulk(np.ones(3000, np.complex64), p0, dp)      # 1 * phasor(p0 + n dp): exactly the phasor
    assert np.array_equal(_bits(got), _bits(ref))


@pytest.mark.parametrize("fhint,fc", [(0.0, 0.02), (0.01, 0.2)])
def test_pll_bit_exact(L, sdo, fhint, fc):
    x = (np.exp(1j * 0.011 * np.arange(5000)) + 0.1 * cnoise(5000, 3)).astype(np.complex64)
    p = Pll()
    assert L.su_pll_init(C.byref(p), fhint, fc)
    got = _per_sample(L.su_pll_track, p, x)
    ref = sdo.pll_track_bulk(sdo.pll_new(fhint, fc), x)
    assert np.array_equal(_bits(got), _bits(ref))


@pytest.mark.parametrize("kind", [1, 2, 3])
@pytest.mark.parametrize("arm_order", [1, 2, 3, 5])
def test_costas_bit_exact_every_kind_and_arm_order(L, sdo, kind, arm_order):
    n = 4000
    sym = np.exp(1j * (2 * np.pi / (2 ** kind)) * np.random.default_rng(kind).integers(0, 2 ** kind, n // 8 + 1))
    x = (np.repeat(sym, 8)[:n] * np.exp(1j * (0.3 + 0.002 * np.arange(n))) + 0.05 * cnoise(n, 5)).astype(np.complex64)
    c = Costas()
    assert L.su_costas_init(C.byref(c), kind, 0.0, 0.125, arm_order, 0.01)
    got = _per_sample(L.su_costas_feed, c, x)
    ref = sdo.costas_feed_bulk(sdo.costas_new(kind, 0.0, 0.125, arm_order, 0.01), x)
    assert np.array_equal(_bits(got), _bits(ref))


def test_costas_refuses_what_the_banks_refuse(L):
    c = Costas()
    assert not L.su_costas_init(C.byref(c), 0, 0.0, 0.1, 3, 0.01)            # SU_COSTAS_KIND_NONE
    assert not L.su_costas_init(C.byref(c), 2, 0.0, 0.1, 6, 0.01)            # arm filter order 5


@pytest.mark.parametrize("tau", [8.0, 200.0])
def test_agc_bit_exact(L, sdo, tau):
    n = 6000
    x = cnoise(n, 10) * np.concatenate([np.linspace(0.01, 3, n // 2), np.linspace(3, 1e-3, n - n // 2)]).astype(np.float32)
    op = sdo.agc_params_from_tau(tau)
    prm = AgcParams(op.threshold, op.slope_factor, op.hang_max, op.delay_line_size, op.mag_history_size,
                    op.fast_rise_t, op.fast_fall_t, op.slow_rise_t, op.slow_fall_t)
    a = Agc()
    assert L.su_agc_init(C.byref(a), C.byref(prm))
    got = _per_sample(L.su_agc_feed, a, x)
    ref = sdo.agc_feed_bulk(sdo.agc_new(op), x)
    assert np.array_equal(_bits(got), _bits(ref))
    prm.delay_line_size = 65
    assert not L.su_agc_init(C.byref(a), C.byref(prm))


@pytest.mark.parametrize("gain,bhint", [(0.5, 0.1), (0.2, 1 / 15.6), (0.0, 0.25)])
def test_clock_detector_symbols_bit_exact_and_read_in_blocks(L, sdo, gain, bhint):
    bits = np.random.default_rng(8).integers(0, 2, 900) * 2 - 1
    sps = int(round(1 / bhint))
    base = np.convolve(np.repeat(bits, sps).astype(np.float32), np.ones(max(2, sps // 2)) / max(2, sps // 2), mode="same")
    x = (base * np.exp(1j * 0.4) + 0.02 * cnoise(base.size, 2)).astype(np.complex64)[:9000]
    cd = Clock()
    assert L.su_clock_detector_init(C.byref(cd), gain, bhint, 4096) != -1
    out, buf = [], np.empty(4096, dtype=np.complex64)
    for i, v in enumerate(x):                                   # WaveSampler::sampleGardner: feed <= 4096, then read
        L.su_clock_detector_feed(C.byref(cd), CF(float(v.real), float(v.imag)))
        if i % 4096 == 4095 or i == x.size - 1:
            k = L.su_clock_detector_read(C.byref(cd), buf.ctypes.data_as(C.c_void_p), 4096)
            out.append(buf[:k].copy())
    L.su_clock_detector_finalize(C.byref(cd))
    got = np.concatenate(out)
    ref = sdo.clock_feed_bulk(sdo.clock_new(gain, bhint), x)
    assert got.size == ref.size > 100 and np.array_equal(_bits(got), _bits(ref))
    assert L.su_clock_detector_init(C.byref(cd), 0.1, 0.0, 4096) == -1      # Tasks/WaveSampler.cpp:60-65 compares with -1


def test_blackmann_harris_window_matches_the_oracle(L, sdo):
    x = cnoise(1000, 4)
    got = x.copy()
    L.su_taps_apply_blackmann_harris_complex(got.ctypes.data_as(C.c_void_p), got.size)
    ref = x * sdo.window(4, x.size)
    assert np.array_equal(_bits(got), _bits(ref.astype(np.complex64)))



class Iir(C.Structure):
    _fields_ = [("n", C.c_uint64), ("h", C.POINTER(C.c_float)), ("d", C.c_void_p)]


def test_matched_filter_and_nco_retune(L, sdo):
    """su_iir_rrc_init / su_iir_filt_feed (WaveSampler's optional matched filter): the taps are suamd_rrc_design's, the
    output the k-ascending fma chain of SPEC.md I; su_ncqo_set_freq keeps the phase running"""
    for name, res, args in [("su_iir_rrc_init", C.c_int, [C.POINTER(Iir), C.c_uint64, C.c_float, C.c_float]),
                            ("su_iir_filt_feed", CF, [C.POINTER(Iir), CF]), ("su_iir_filt_finalize", None, [C.POINTER(Iir)]),
                            ("su_ncqo_set_freq", None, [C.POINTER(Ncqo), C.c_float]),
                            ("suamd_rrc_design", None, [C.POINTER(C.c_float), C.c_uint, C.c_double, C.c_double])]:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    f = Iir()
    assert L.su_iir_rrc_init(C.byref(f), 6, 8.0, 0.35)
    n = int(f.n)
    assert n == 49                                                   # 6 symbol periods of 8 samples + 1
    taps = np.array([f.h[i] for i in range(n)], dtype=np.float32)
    want = (C.c_float * n)()
    L.suamd_rrc_design(want, n, 8.0, float(np.float32(0.35)))       # SUFLOAT beta, widened
    assert np.array_equal(taps, np.frombuffer(want, dtype=np.float32))
    assert abs(float(taps.sum()) - 1.0) < 1e-6
    x = cnoise(300, 21)
    got = _per_sample(L.su_iir_filt_feed, f, x)
    ref = np.zeros(x.size, np.complex64)
    xp = np.concatenate([np.zeros(n - 1, np.complex64), x])
    for m in range(x.size):
        yr = yi = np.float32(0)
        for k in range(n):
            v = xp[m + n - 1 - k]
            yr = np.float32(np.float64(taps[k]) * np.float64(v.real) + np.float64(yr))     # fma of binary32 operands: exact product, one rounding
            yi = np.float32(np.float64(taps[k]) * np.float64(v.imag) + np.float64(yi))
        ref[m] = yr + 1j * yi
    assert np.array_equal(_bits(got), _bits(ref))
    L.su_iir_filt_finalize(C.byref(f))
    assert not L.su_iir_rrc_init(C.byref(f), 0, 8.0, 0.35)
    # NCO: 1000 reads at one frequency, then a retune: the phase continues from where it was
    o = Ncqo()
    L.su_ncqo_init(C.byref(o), 0.01)
    a = [L.su_ncqo_read(C.byref(o)) for _ in range(1000)]
    L.su_ncqo_set_freq(C.byref(o), -0.02)
    b = L.su_ncqo_read(C.byref(o))
    dp = sdo.fnor_to_dphase(np.float32(0.01))
    want_phase = (1000 * dp) & 0xFFFFFFFF
    p = sdo.xlate_bulk(np.ones(1, np.complex64), want_phase, 0)
    assert np.complex64(complex(b.re, b.im)) == p[0] and o.n == 1 and o.dphase == sdo.fnor_to_dphase(np.float32(-0.02))
