"""ctypes binding of the CPU oracle (oracle/libsdo.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under sigdigger_amd/ imports this module.
Parity: reference-pinned (oracle/_ref, tests/test_ref_pin.py) for the rows the reference owns, "parity unpinned" vs upstream
sigutils/suscan for the rest (see oracle/sdo.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c32 = np.complex64


def build(force=False):
    so = os.path.join(_HERE, "libsdo.so")
    src = [os.path.join(_HERE, f) for f in ("sdo.c", "sdo.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsdo.so"], stdout=subprocess.DEVNULL)
    return so


def build_fast(force=False):
    """libsdo_fast.so: the same source at -O3 -march=native (SURVEY.md 8d's CPU-baseline build) -- bench.py's timed CPU
    leg only; every parity test uses libsdo.so"""
    so = os.path.join(_HERE, "libsdo_fast.so")
    src = [os.path.join(_HERE, f) for f in ("sdo.c", "sdo.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsdo_fast.so"], stdout=subprocess.DEVNULL)
    return so


def use_fast(on=True):
    """switch this module's functions to libsdo_fast.so (bench.py's cpu_baseline) and back"""
    global _LIB
    _LIB = None
    _USE["fast"] = bool(on)


_USE = {"fast": False}


class C32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class IIR(C.Structure):
    _fields_ = [("order", C.c_int), ("b", C.c_float * 5), ("a", C.c_float * 5),
                ("xh", C.c_float * 10), ("yh", C.c_float * 10)]


class Costas(C.Structure):
    _fields_ = [("kind", C.c_int), ("phase", C.c_uint32), ("omega", C.c_float),
                ("a", C.c_float), ("b", C.c_float), ("gain", C.c_float), ("af", IIR)]


class PLL(C.Structure):
    _fields_ = [("phase", C.c_uint32), ("omega", C.c_float), ("alpha", C.c_float), ("beta", C.c_float)]


class Clock(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("gain", C.c_float),
                ("phi", C.c_float), ("bnor", C.c_float), ("bmin", C.c_float), ("bmax", C.c_float),
                ("halfcycle", C.c_int), ("prev", C.c_float * 2), ("x0", C.c_float * 2),
                ("x1", C.c_float * 2), ("x2", C.c_float * 2)]


class CMA(C.Structure):
    _fields_ = [("n", C.c_int), ("mu", C.c_float), ("locked", C.c_int), ("w", C.c_float * 32), ("d", C.c_float * 32)]


class SNR(C.Structure):
    _fields_ = [("sigma", C.c_float), ("alpha", C.c_float), ("hx", C.c_float), ("delta", C.c_float), ("sqerr", C.c_float),
                ("bps", C.c_uint), ("intervals", C.c_uint), ("length", C.c_uint)]


class AGCParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("slope_factor", C.c_float), ("hang_max", C.c_uint),
                ("delay_line_size", C.c_uint), ("mag_history_size", C.c_uint),
                ("fast_rise_t", C.c_float), ("fast_fall_t", C.c_float),
                ("slow_rise_t", C.c_float), ("slow_fall_t", C.c_float)]


class AGC(C.Structure):
    _fields_ = [("knee", C.c_float), ("gain_slope", C.c_float), ("fixed_gain", C.c_float),
                ("fast_alpha_rise", C.c_float), ("fast_alpha_fall", C.c_float),
                ("slow_alpha_rise", C.c_float), ("slow_alpha_fall", C.c_float),
                ("hang_max", C.c_uint), ("hang_n", C.c_uint), ("delay_line_size", C.c_uint),
                ("mag_history_size", C.c_uint), ("delay_ptr", C.c_uint), ("hist_ptr", C.c_uint),
                ("peak", C.c_float), ("fast_level", C.c_float), ("slow_level", C.c_float),
                ("delay_line", C.c_float * 128), ("mag_history", C.c_float * 64)]


class SpecView(C.Structure):
    _fields_ = [("freqMin", C.c_double), ("freqMax", C.c_double), ("freqRange", C.c_double),
                ("spectrumSize", C.c_uint), ("fftBandwidth", C.c_double), ("fftRelBw", C.c_float),
                ("psd", C.c_void_p), ("psdAccum", C.c_void_p), ("psdCount", C.c_void_p)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build_fast() if _USE["fast"] else build())
        L = _LIB
        L.sdo_atan2f.restype = C.c_float
        L.sdo_atan2f.argtypes = [C.c_float, C.c_float]
        L.sdo_log2f.restype = C.c_float
        L.sdo_log2f.argtypes = [C.c_float]
        L.sdo_exp2f.restype = C.c_float
        L.sdo_exp2f.argtypes = [C.c_float]
        L.sdo_fnor_to_dphase.restype = C.c_uint32
        L.sdo_fnor_to_dphase.argtypes = [C.c_double]
        L.sdo_chan_feed.restype = C.c_size_t
        L.sdo_histogram_feed.restype = C.c_size_t
        L.sdo_clock_feed_bulk.restype = C.c_size_t
        L.sdo_carrier_detect.restype = C.c_float
        L.sdo_spectsrc_preproc.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C32, C.c_void_p]
        L.sdo_snr_get.restype = C.c_float
        L.sdo_snr_init.argtypes = [C.c_void_p, C.c_uint, C.c_float]
        L.sdo_decide.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_float, C.c_float, C.c_void_p]
        L.sdo_symbol_histogram.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_uint, C.c_void_p]
        L.sdo_rrc_ntaps.restype = C.c_size_t
        L.sdo_rrc_ntaps.argtypes = [C.c_double]
        L.sdo_rrc_design.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double]
        L.sdo_fir_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.sdo_scale.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
        L.sdo_cma_init.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.sdo_cma_feed_bulk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.sdo_sample_zero_crossing.restype = C.c_size_t
        L.sdo_sample_zero_crossing.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_int, C32, C32,
                                               C.c_void_p, C.c_size_t]
        L.sdo_conj_prev.argtypes = [C.c_void_p, C.c_size_t, C32, C.c_void_p]
        L.sdo_averager_feed.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(x):
    return np.ascontiguousarray(x, dtype=c32)


def _f(x):
    return np.ascontiguousarray(x, dtype=np.float32)


# ---- D ----------------------------------------------------------------------------------
def phasor_u32(p):
    p = np.ascontiguousarray(p, dtype=np.uint32)
    out = np.empty(p.shape, dtype=c32)
    lib().sdo_phasor_u32_bulk(_p(p), _p(out), C.c_size_t(p.size))
    return out


def atan2f(y, x):
    y, x = _f(y), _f(x)
    out = np.empty(y.shape, dtype=np.float32)
    lib().sdo_atan2f_bulk(_p(y), _p(x), _p(out), C.c_size_t(y.size))
    return out


def log2f(x):
    x = _f(x)
    out = np.empty(x.shape, dtype=np.float32)
    lib().sdo_log2f_bulk(_p(x), _p(out), C.c_size_t(x.size))
    return out


def exp2f(x):
    x = _f(x)
    out = np.empty(x.shape, dtype=np.float32)
    lib().sdo_exp2f_bulk(_p(x), _p(out), C.c_size_t(x.size))
    return out


def fnor_to_dphase(fnor):
    return int(lib().sdo_fnor_to_dphase(float(fnor)))


# ---- PSD --------------------------------------------------------------------------------
def psd_shift_db(psd):
    out = _f(psd).copy()
    lib().sdo_psd_shift_db(_p(out), C.c_size_t(out.size))
    return out


class Averager:
    """Misc/Averager.cpp state holder."""

    def __init__(self, alpha=1.0):
        self.alpha = alpha
        self.last = None
        self.bufsiz = C.c_size_t(0)

    def feed(self, x):
        x = _f(x)
        if self.last is None or self.last.size != x.size:
            self.last = np.zeros(x.size, dtype=np.float32)
        lib().sdo_averager_feed(_p(self.last), C.byref(self.bufsiz), _p(x), C.c_size_t(x.size),
                                C.c_float(self.alpha))
        return self.last


def inspector_spectrum_db_shift(data):
    out = _f(data).copy()
    lib().sdo_inspector_spectrum_db_shift(_p(out), C.c_size_t(out.size))
    return out


def window(kind, n):
    w = np.empty(n, dtype=np.float32)
    lib().sdo_window(C.c_int(kind), _p(w), C.c_size_t(n))
    return w


def fft_f64(x):
    re = np.ascontiguousarray(np.real(x), dtype=np.float64).copy()
    im = np.ascontiguousarray(np.imag(x), dtype=np.float64).copy()
    lib().sdo_fft_f64(_p(re), _p(im), C.c_size_t(re.size))
    return re + 1j * im


def psd_frames(x, nframes, n, hop, win, navg=1, scale=1.0):
    x = _c(x)
    win = _f(win)
    assert x.size >= (nframes - 1) * hop + n
    out = np.empty((nframes // navg, n), dtype=np.float32)
    lib().sdo_psd_frames(_p(x), C.c_size_t(nframes), C.c_size_t(n), C.c_size_t(hop), _p(win),
                         C.c_size_t(navg), C.c_float(scale), _p(out))
    return out


# ---- NCO / channelizer ------------------------------------------------------------------
def xlate_bulk(x, p0, dp, n0=0):
    x = _c(x)
    y = np.empty_like(x)
    lib().sdo_xlate_bulk(_p(x), _p(y), C.c_size_t(x.size), C.c_uint32(p0), C.c_uint32(dp), C.c_uint64(n0))
    return y


def lpf_design(ntaps, fc):
    h = np.empty(ntaps, dtype=np.float32)
    lib().sdo_lpf_design(_p(h), C.c_size_t(ntaps), C.c_double(fc))
    return h


def chan_modulate_taps(h, dp):
    h = _f(h)
    g = np.empty(h.size, dtype=c32)
    lib().sdo_chan_modulate_taps(_p(h), C.c_size_t(h.size), C.c_uint32(dp), _p(g))
    return g


def chan_feed(hist, x, n0, g, D, p0, dp):
    x = _c(x)
    g = _c(g)
    hist = _c(hist)
    assert hist.size == g.size - 1
    y = np.empty(x.size // D + 2, dtype=c32)
    n = lib().sdo_chan_feed(_p(hist), _p(x), C.c_size_t(x.size), C.c_uint64(n0), _p(g),
                            C.c_size_t(g.size), C.c_uint32(D), C.c_uint32(p0), C.c_uint32(dp), _p(y))
    return y[:n].copy()


# ---- element-wise -----------------------------------------------------------------------
def quad_demod(x, prev=0j, first=True):
    x = _c(x)
    y = np.empty_like(x)
    pv = C32(float(np.real(prev)), float(np.imag(prev)))
    lib().sdo_quad_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C32, C.c_int]
    lib().sdo_quad_demod(_p(x), _p(y), x.size, pv, int(first))
    return y


def delayed_conj(x, delay):
    x = _c(x)
    y = np.empty_like(x)
    lib().sdo_delayed_conj(_p(x), _p(y), C.c_size_t(x.size), C.c_size_t(delay))
    return y


def histogram_feed(x, space):
    x = _c(x)
    out = np.empty(x.size, dtype=np.float32)
    n = lib().sdo_histogram_feed(_p(x), C.c_size_t(x.size), C.c_int(space), _p(out))
    return out[:n].copy()


# ---- loops ------------------------------------------------------------------------------
def butter_lp(order, fc):
    b = np.zeros(5, dtype=np.float32)
    a = np.zeros(5, dtype=np.float32)
    lib().sdo_butter_lp(C.c_int(order), C.c_double(fc), _p(b), _p(a))
    return b, a


def costas_new(kind, fhint, arm_bw, arm_order, loop_bw):
    c = Costas()
    ok = lib().sdo_costas_init(C.byref(c), C.c_int(kind), C.c_float(fhint), C.c_float(arm_bw),
                               C.c_uint(arm_order), C.c_float(loop_bw))
    assert ok
    return c


def costas_feed_bulk(c, x):
    x = _c(x)
    y = np.empty_like(x)
    lib().sdo_costas_feed_bulk(C.byref(c), _p(x), _p(y), C.c_size_t(x.size))
    return y


def pll_new(fhint, fc):
    p = PLL()
    lib().sdo_pll_init(C.byref(p), C.c_float(fhint), C.c_float(fc))
    return p


def pll_track_bulk(p, x):
    x = _c(x)
    y = np.empty_like(x)
    lib().sdo_pll_track_bulk(C.byref(p), _p(x), _p(y), C.c_size_t(x.size))
    return y


def clock_new(loop_gain, bhint):
    cd = Clock()
    r = lib().sdo_clock_init(C.byref(cd), C.c_float(loop_gain), C.c_float(bhint))
    assert r != -1
    return cd


def clock_feed_bulk(cd, x):
    x = _c(x)
    out = np.empty(x.size + 1, dtype=c32)
    n = lib().sdo_clock_feed_bulk(C.byref(cd), _p(x), C.c_size_t(x.size), _p(out))
    return out[:n].copy()


def agc_params_default():
    return AGCParams.in_dll(lib(), "sdo_agc_params_default")


def agc_params_from_tau(tau):
    p = AGCParams()
    lib().sdo_agc_params_from_tau(C.byref(p), C.c_float(tau))
    return p


def agc_new(params=None):
    a = AGC()
    if params is None:
        params = agc_params_default()
    ok = lib().sdo_agc_init(C.byref(a), C.byref(params))
    assert ok
    return a


def agc_feed_bulk(a, x):
    x = _c(x)
    y = np.empty_like(x)
    lib().sdo_agc_feed_bulk(C.byref(a), _p(x), _p(y), C.c_size_t(x.size))
    return y


# ---- Tasks ------------------------------------------------------------------------------
def sample_manual(data, symbol_count, symbol_sync, space, nout=None):
    data = _c(data)
    if nout is None:
        nout = int(symbol_count)
    out = np.empty(nout, dtype=c32)
    lib().sdo_sample_manual(_p(data), C.c_size_t(data.size), C.c_double(symbol_count),
                            C.c_double(symbol_sync), C.c_int(space), _p(out), C.c_size_t(nout))
    return out


def sample_zero_crossing(data, bnor, space, amplitude=False, threshold=0j, zc_angle=1 + 0j):
    """Symbols (uint8, var > 0) of WaveSampler's ZERO_CROSSING mode over a whole capture."""
    data = _c(data)
    cap = 4096 * ((data.size + 4095) // 4096)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = lib().sdo_sample_zero_crossing(_p(data), data.size, float(bnor), int(space), int(bool(amplitude)),
                                       C32(float(np.real(threshold)), float(np.imag(threshold))),
                                       C32(float(np.real(zc_angle)), float(np.imag(zc_angle))),
                                       out.ctypes.data_as(C.c_void_p), out.size)
    return out[:n].copy()


def rrc_design(sps, rolloff, ntaps=None):
    n = int(lib().sdo_rrc_ntaps(float(sps))) if ntaps is None else int(ntaps)
    h = np.empty(n, dtype=np.float32)
    lib().sdo_rrc_design(h.ctypes.data_as(C.c_void_p), n, float(sps), float(rolloff))
    return h


def fir_feed(hist, h, x):
    """hist (ntaps-1 complex64, updated in place), real taps h, block x -> y"""
    x = _c(x)
    h = np.ascontiguousarray(h, dtype=np.float32)
    assert hist.dtype == c32 and hist.size == h.size - 1
    y = np.empty(x.size, dtype=c32)
    lib().sdo_fir_feed(_p(hist), h.ctypes.data_as(C.c_void_p), h.size, _p(x), x.size, _p(y))
    return y


def scale(x, g):
    x = _c(x)
    y = np.empty(x.size, dtype=c32)
    lib().sdo_scale(_p(x), x.size, float(g), _p(y))
    return y


def cma_new(n, mu, locked=False):
    q = CMA()
    lib().sdo_cma_init(C.byref(q), int(n), float(mu))
    q.locked = int(bool(locked))
    return q


def cma_feed_bulk(q, x):
    x = _c(x)
    y = np.empty(x.size, dtype=c32)
    lib().sdo_cma_feed_bulk(C.byref(q), _p(x), x.size, _p(y))
    return y


SPECTSRC = ("psd", "cyclo", "fmspect", "pmspect", "timediff", "abstimediff", "exp_2", "exp_4", "exp_8")


def spectsrc_preproc(kind, x, prev0=0j):
    x = _c(x)
    y = np.empty(x.size, dtype=c32)
    lib().sdo_spectsrc_preproc(int(kind), _p(x), x.size, C32(float(np.real(prev0)), float(np.imag(prev0))), _p(y))
    return y


def decision_space(x, mode):
    x = _c(x)
    out = np.empty(x.size, dtype=np.float32)
    lib().sdo_decision_space(_p(x), C.c_size_t(x.size), C.c_int(mode), out.ctypes.data_as(C.c_void_p))
    return out


def decide(x, mode, bps, vmin, vmax):
    x = _c(x)
    out = np.empty(x.size, dtype=np.uint8)
    lib().sdo_decide(_p(x), x.size, int(mode), int(bps), float(vmin), float(vmax), out.ctypes.data_as(C.c_void_p))
    return out


def symbol_histogram(x, mode, vmin, vmax, nbins, hist=None):
    x = _c(x)
    if hist is None:
        hist = np.zeros(nbins, dtype=np.uint32)
    lib().sdo_symbol_histogram(_p(x), x.size, int(mode), float(vmin), float(vmax), int(nbins), hist.ctypes.data_as(C.c_void_p))
    return hist


def snr_new(bps, alpha):
    e = SNR()
    lib().sdo_snr_init(C.byref(e), int(bps), float(alpha))
    return e


def snr_feed(e, history):
    h = np.ascontiguousarray(history, dtype=np.uint32)
    model = np.empty(h.size, dtype=np.float32)
    lib().sdo_snr_feed(C.byref(e), h.ctypes.data_as(C.c_void_p), C.c_uint(h.size), model.ctypes.data_as(C.c_void_p))
    return model


def snr_get(e):
    return float(lib().sdo_snr_get(C.byref(e)))


def ingest_iq(fmt, raw):
    raw = np.ascontiguousarray(raw)
    n = raw.nbytes // {1: 8, 2: 2, 3: 2, 4: 4}[int(fmt)]
    out = np.empty(n, dtype=c32)
    lib().sdo_ingest_iq(C.c_int(int(fmt)), raw.ctypes.data_as(C.c_void_p), C.c_size_t(n), _p(out))
    return out


def source_fix(x, iq_reverse, dc=None, alpha=0.1, first=True):
    """in place on a copy; dc: float32[2] state (updated) or None"""
    y = _c(x).copy()
    lib().sdo_source_fix(_p(y), C.c_size_t(y.size), C.c_int(int(iq_reverse)),
                         dc.ctypes.data_as(C.c_void_p) if dc is not None else None, C.c_float(alpha), C.c_int(int(first)))
    return y


def conj_prev(x, prev0=0j):
    x = _c(x)
    y = np.empty(x.size, dtype=c32)
    lib().sdo_conj_prev(_p(x), x.size, C32(float(np.real(prev0)), float(np.imag(prev0))), _p(y))
    return y


class FAC:
    """FACTab state: fac (n/2 floats), running max / min"""

    def __init__(self, n, alpha):
        self.n, self.alpha = int(n), float(alpha)
        self.fac = np.zeros(self.n // 2, dtype=np.float32)
        self.max = C.c_float(-np.inf)
        self.min = C.c_float(np.inf)

    def feed(self, buf, view_start=0, view_end=None):
        buf = _c(buf)
        assert buf.size == self.n
        ve = self.n // 2 if view_end is None else int(view_end)
        lib().sdo_fac_feed(_p(buf), C.c_size_t(self.n), C.c_float(self.alpha), C.c_long(int(view_start)), C.c_long(ve),
                           self.fac.ctypes.data_as(C.c_void_p), C.byref(self.max), C.byref(self.min))
        return self.fac


class Power(C.Structure):
    """sdo_power: the "power" class / RMSInspector raw-mode integrator"""
    _fields_ = [("acc", C.c_double), ("c", C.c_double), ("count", C.c_ulonglong), ("max_samples", C.c_ulonglong)]

    def __init__(self, n):
        super().__init__()
        lib().sdo_power_init(C.byref(self), C.c_ulonglong(int(n)))

    def feed(self, x):
        x = _c(x)
        out = np.empty((int(self.count) + x.size) // int(self.max_samples) + 1, dtype=c32)
        f = lib().sdo_power_feed
        f.restype = C.c_size_t
        k = f(C.byref(self), _p(x), C.c_size_t(x.size), _p(out))
        return out[:k]


def baud_nonlinear(x):
    x = _c(x)
    f = lib().sdo_baud_nonlinear
    f.restype = C.c_float
    return float(f(_p(x), C.c_size_t(x.size)))


def fac_first_valley(fac):
    fac = np.ascontiguousarray(fac, dtype=np.float32)
    f = lib().sdo_fac_first_valley
    f.restype = C.c_float
    return float(f(fac.ctypes.data_as(C.c_void_p), C.c_size_t(fac.size)))


def carrier_detect(data, avg_rel_bw, dc_notch_rel_bw):
    data = _c(data)
    return float(lib().sdo_carrier_detect(_p(data), C.c_size_t(data.size), C.c_float(avg_rel_bw),
                                          C.c_float(dc_notch_rel_bw)))


def doppler_calc(data, fs, f0):
    """DopplerCalculator::work: (peak velocity, sigma, max, mirrored spectrum)"""
    data = _c(data)
    alloc = 16
    while alloc < data.size:
        alloc <<= 1
    spec = np.empty(alloc, dtype=np.float32)
    res = np.empty(3, dtype=np.float32)
    lib().sdo_doppler_calc(_p(data), C.c_size_t(data.size), C.c_float(fs), C.c_double(f0), _p(spec), _p(res))
    return float(res[0]), float(res[1]), float(res[2]), spec


class SpectrumView:
    """Panoramic/Scanner.cpp SpectrumView."""
    SIZE = 65536

    def __init__(self):
        self.psd = np.zeros(self.SIZE, dtype=np.float32)
        self.accum = np.zeros(self.SIZE, dtype=np.float32)
        self.count = np.zeros(self.SIZE, dtype=np.float32)
        self.v = SpecView()
        lib().sdo_specview_init(C.byref(self.v), _p(self.psd), _p(self.accum), _p(self.count))

    def set_range(self, fmin, fmax):
        lib().sdo_specview_set_range(C.byref(self.v), C.c_double(fmin), C.c_double(fmax))

    def feed(self, psd, fmin, fmax, adjust_sides=True, count=None):
        psd = _f(psd)
        cp = _p(_f(count)) if count is not None else None
        lib().sdo_specview_feed(C.byref(self.v), _p(psd), cp, C.c_size_t(psd.size),
                                C.c_double(fmin), C.c_double(fmax), C.c_int(int(adjust_sides)))

    def interpolate(self):
        lib().sdo_specview_interpolate(C.byref(self.v))


# ---- C2: FFT channeliser (su_specttuner semantics) ------------------------------------------------
class StGeom(C.Structure):
    _fields_ = [("size", C.c_uint), ("halfsz", C.c_uint), ("width", C.c_uint), ("halfw", C.c_uint), ("decimation", C.c_uint),
                ("center", C.c_int), ("lo", C.c_double), ("dphase", C.c_uint32)]


def specttuner_geometry(W, f0, bw, guard):
    g = StGeom()
    lib().sdo_specttuner_geometry(C.c_uint(W), C.c_double(f0), C.c_double(bw), C.c_double(guard), C.byref(g))
    return g


def specttuner_response(W, size, halfw):
    hk = np.empty(size, dtype=c32)
    lib().sdo_specttuner_response(C.c_uint(W), C.c_uint(size), C.c_uint(halfw), _p(hk))
    return hk


def specttuner_run(x, W, f0, bw, guard, precise=False):
    x = _c(x)
    g = specttuner_geometry(W, f0, bw, guard)
    cap = (x.size // (W // 2) + 1) * g.halfsz
    out = np.empty(cap, dtype=c32)
    f = lib().sdo_specttuner_run
    f.restype = C.c_size_t
    n = f(_p(x), C.c_size_t(x.size), C.c_uint(W), C.c_double(f0), C.c_double(bw), C.c_double(guard), C.c_int(int(precise)),
          _p(out), C.c_size_t(cap))
    return out[:n].copy()


def st32_forward(win, narrow=True):
    """DFT_4096 of one window in the binary32 arithmetic SPEC.md C2 freezes (narrow: 64 x 64; wide: radix-16 passes)"""
    win = _c(win)
    assert win.size == 4096
    X = np.empty(4096, dtype=c32)
    (lib().sdo_st32_forward_narrow if narrow else lib().sdo_st32_forward_wide)(_p(win), _p(X))
    return X


def specttuner_bank_f32(x, f0, bw, guard, precise=None, threads=1):
    """The binary32 statement of the FFT channeliser for a bank of channels opened at the start of the stream.
    Returns a list of rows (one array per channel).  threads > 1: window ranges computed side by side (every range
    re-transforms the window before it; the result does not depend on the split)."""
    x = _c(x)
    n = len(f0)
    W, H = 4096, 2048
    nwin = 0 if x.size < W else (x.size - W) // H + 1
    geoms = [specttuner_geometry(W, f0[c], bw[c], guard[c]) for c in range(n)]
    stride = max([nwin * g.halfsz for g in geoms] + [1])
    out = np.zeros((n, stride), dtype=c32)
    F0 = (C.c_double * n)(*[float(v) for v in f0]); BW = (C.c_double * n)(*[float(v) for v in bw]); GU = (C.c_double * n)(*[float(v) for v in guard])
    PR = (C.c_int * n)(*[int(bool(v)) for v in (precise if precise is not None else [0] * n)])
    f = lib().sdo_specttuner_bank_f32
    f.restype = C.c_size_t

    def run(w0, w1):
        f(_p(x), C.c_size_t(x.size), C.c_uint(n), F0, BW, GU, PR, C.c_size_t(w0), C.c_size_t(w1), _p(out), C.c_size_t(stride))

    if threads <= 1 or nwin < 2 * threads:
        run(0, nwin)
    else:
        from concurrent.futures import ThreadPoolExecutor
        edges = [nwin * k // threads for k in range(threads + 1)]
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(lambda k: run(edges[k], edges[k + 1]), range(threads)))
    return [out[c, :nwin * geoms[c].halfsz] for c in range(n)]


def specttuner_run_f32(x, f0, bw, guard, precise=False):
    return specttuner_bank_f32(x, [f0], [bw], [guard], [precise])[0].copy()


# ---- O: channel detector ----------------------------------------------------------------------------
class _ChanDet(C.Structure):
    _fields_ = [("n", C.c_uint), ("alpha", C.c_float), ("gamma", C.c_float), ("snr", C.c_float), ("first", C.c_int),
                ("N0", C.c_float), ("S", C.c_void_p)]


class _ChanDetRecord(C.Structure):
    _fields_ = [("first", C.c_int), ("last", C.c_int), ("width", C.c_int), ("peak", C.c_float), ("sum", C.c_double), ("wsum", C.c_double)]


class ChannelDetector:
    def __init__(self, n, alpha, gamma, snr):
        self.S = np.zeros(n, dtype=np.float32)
        self.d = _ChanDet(n, alpha, gamma, snr, 1, 0.0, self.S.ctypes.data)

    def feed(self, P):
        lib().sdo_chandet_feed(C.byref(self.d), _p(_f(P)))

    def find(self, cap=1024):
        rec = (_ChanDetRecord * cap)()
        f = lib().sdo_chandet_find
        f.restype = C.c_uint
        n = f(C.byref(self.d), rec, C.c_uint(cap))
        return [(r.first, r.last, r.width, r.peak, r.sum, r.wsum) for r in rec[:n]]

    @property
    def N0(self):
        return float(self.d.N0)


def chandet_track(prev, now, beta):
    """SPEC.md section O, beta: `now` = [(fc, f_lo, f_hi, S0_dB), ...] of this update, `prev` = [(fc, S0_dB, age), ...] as
    reported by the previous one.  A channel that contains the centre of a previous channel continues it (the first such, in
    list order): S0 <- S0_prev + beta (S0_now - S0_prev) in binary32, age + 1; any other starts at S0_now, age 0.
    Returns [(fc, S0, age), ...].  (A handful of values per update: plain Python.)"""
    out = []
    b = np.float32(beta)
    for fc, lo, hi, s0 in now:
        s0, age = np.float32(s0), 0
        if 0.0 < beta < 1.0:
            for pfc, ps0, page in prev:
                if lo <= pfc <= hi:
                    s0 = np.float32(np.float32(ps0) + np.float32(b * np.float32(s0 - np.float32(ps0))))
                    age = page + 1
                    break
        out.append((fc, s0, age))
    return out


# ---- Q: audio inspector ---------------------------------------------------------------------------------
def audio_run(x, mode, efs, bw, fa, cutoff, volume=1.0):
    x = _c(x)
    cap = int(x.size * fa / efs) + 8
    out = np.empty(cap, dtype=c32)
    f = lib().sdo_audio_run
    f.restype = C.c_size_t
    n = f(_p(x), C.c_size_t(x.size), C.c_int(mode), C.c_double(efs), C.c_double(bw), C.c_double(fa), C.c_double(cutoff),
          C.c_float(volume), _p(out), C.c_size_t(cap))
    return out[:n].copy()
