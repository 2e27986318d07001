"""ctypes binding of oracle/_ref/libsdref.so: the REFERENCE'S OWN loops (compiled from /root/reference by
oracle/Makefile.ref, driven by oracle/ref_glue.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/test_ref_pin.py to pin oracle/sdo.c against the reference.  Nothing under
sigdigger_amd/ imports it.  The library exists only where /root/reference is (this container) or where a prebuilt
oracle/_ref/ has travelled; `available()` tells.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libsdref.so")
_LIB = None
c32 = np.complex64
SPECTRUM_SIZE = 65536


def build():
    """Builds oracle/_ref/ from /root/reference (needs the product library: it is what the reference's wrappers link to)."""
    if not os.path.isdir("/root/reference"):
        return os.path.exists(_SO)
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsdo.so"])
    subprocess.check_call(["make", "-s", "-C", _HERE, "-f", "Makefile.ref", "-j8", "all"])
    return True


def available():
    return os.path.exists(_SO)


def lib():
    global _LIB
    if _LIB is None:
        # the system libstdc++ first: libQt5Core comes from /opt/conda/lib, whose own (older) libstdc++ must not win
        for cand in ("/usr/lib/x86_64-linux-gnu/libstdc++.so.6", "libstdc++.so.6"):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
        _LIB = C.CDLL(_SO)
        _LIB.ref_carrier_detector.restype = C.c_float
        for f in ("ref_averager", "ref_histogram_feeder", "ref_wave_sampler", "ref_doppler", "ref_snr_estimator"):
            getattr(_LIB, f).restype = C.c_size_t
        _LIB.ref_specview_new.restype = C.c_void_p
    return _LIB


def _c(x):
    return np.ascontiguousarray(x, dtype=c32)


def _f(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def psd_message(frame):
    frame = _f(frame)
    out = np.empty_like(frame)
    lib().ref_psd_message(_p(frame), C.c_size_t(frame.size), _p(out))
    return out


def averager(frames, alpha):
    """frames: list of 1-D linear frames (sizes may change).  Returns Averager::get() after the last feed."""
    frames = [_f(f) for f in frames]
    sizes = np.array([f.size for f in frames], dtype=np.uint64)
    flat = np.concatenate(frames)
    out = np.empty(int(sizes[-1]), dtype=np.float32)
    n = lib().ref_averager(_p(flat), _p(sizes), C.c_size_t(len(frames)), C.c_float(alpha), _p(out))
    assert n == out.size
    return out


def _task2(fn, x, *args):
    x = _c(x)
    y = np.zeros_like(x)
    fn(_p(x), _p(y), C.c_size_t(x.size), *args)
    return y


def quad_demod(x):
    return _task2(lib().ref_quad_demod, x)


def delayed_conj(x, delay):
    return _task2(lib().ref_delayed_conj, x, C.c_size_t(delay))


def carrier_xlate(x, rel_freq, phase):
    return _task2(lib().ref_carrier_xlate, x, C.c_float(rel_freq), C.c_float(phase))


def agc_task(x, tau):
    return _task2(lib().ref_agc_task, x, C.c_float(tau))


def costas_task(x, kind, tau, loop_bw):
    return _task2(lib().ref_costas_task, x, C.c_int(kind), C.c_float(tau), C.c_float(loop_bw))


def pll_task(x, cutoff):
    return _task2(lib().ref_pll_task, x, C.c_float(cutoff))


def lpf_task(x, bw):
    """The reference's LPFTask on the product's GPU-backed su_specttuner_* (GPU only).  None if the task threw."""
    x = _c(x)
    y = np.zeros_like(x)
    ok = lib().ref_lpf_task(_p(x), _p(y), C.c_size_t(x.size), C.c_float(bw))
    return y if ok else None


def histogram_feeder(x, space):
    x = _c(x)
    out = np.empty(x.size, dtype=np.float32)
    n = lib().ref_histogram_feeder(_p(x), C.c_size_t(x.size), C.c_int(space), _p(out))
    return out[:n].copy()


def wave_sampler(x, sync, space, fs=1.0, rate=0.0, loop_gain=0.0, symbol_count=1.0, symbol_sync=0, amplitude=False,
                 threshold=0j, zc_angle=1 + 0j, dec_mode=1, dec_bps=1, dec_min=-np.pi, dec_max=np.pi):
    x = _c(x)
    cap = x.size + 8192
    out = np.empty(cap, dtype=c32)
    sym = np.empty(cap, dtype=np.uint8)
    n = lib().ref_wave_sampler(_p(x), C.c_size_t(x.size), C.c_int(sync), C.c_int(space), C.c_double(fs), C.c_double(rate),
                               C.c_double(loop_gain), C.c_double(symbol_count), C.c_size_t(symbol_sync),
                               C.c_int(int(amplitude)), C.c_float(np.real(threshold)), C.c_float(np.imag(threshold)),
                               C.c_float(np.real(zc_angle)), C.c_float(np.imag(zc_angle)), C.c_int(dec_mode),
                               C.c_uint(dec_bps), C.c_float(dec_min), C.c_float(dec_max), _p(out), _p(sym), C.c_size_t(cap))
    return out[:n].copy(), sym[:n].copy()


def carrier_detector(x, avg_rel_bw, dc_notch_rel_bw):
    x = _c(x)
    return float(lib().ref_carrier_detector(_p(x), C.c_size_t(x.size), C.c_double(avg_rel_bw), C.c_double(dc_notch_rel_bw)))


def doppler(x, fs, f0):
    x = _c(x)
    alloc = 1
    while alloc < x.size:
        alloc <<= 1
    spec = np.empty(alloc, dtype=c32)
    res = np.empty(2, dtype=np.float32)
    n = lib().ref_doppler(_p(x), C.c_size_t(x.size), C.c_float(fs), C.c_double(f0), _p(spec), C.c_size_t(alloc), _p(res))
    return float(res[0]), float(res[1]), spec[:n].copy()


class SpectrumView:
    def __init__(self):
        self.h = C.c_void_p(lib().ref_specview_new())

    def __del__(self):
        if self.h:
            lib().ref_specview_destroy(self.h)
            self.h = None

    def set_range(self, fmin, fmax):
        lib().ref_specview_set_range(self.h, C.c_double(fmin), C.c_double(fmax))

    def set_fft(self, fft_bandwidth, rel_bw):
        lib().ref_specview_set_fft(self.h, C.c_double(fft_bandwidth), C.c_float(rel_bw))

    def feed(self, psd, fmin, fmax, adjust_sides=True, count=None):
        psd = _f(psd)
        cp = _p(_f(count)) if count is not None else None
        lib().ref_specview_feed(self.h, _p(psd), cp, C.c_size_t(psd.size), C.c_double(fmin), C.c_double(fmax),
                                C.c_int(int(adjust_sides)))

    def interpolate(self):
        lib().ref_specview_interpolate(self.h)

    def get(self):
        psd, accum, count = (np.empty(SPECTRUM_SIZE, dtype=np.float32) for _ in range(3))
        lib().ref_specview_get(self.h, _p(psd), _p(accum), _p(count))
        return psd, accum, count


def snr_estimator(history, bps, alpha, nfeeds):
    h = np.ascontiguousarray(history, dtype=np.uint32)
    out = np.empty(3, dtype=np.float32)
    model = np.empty(h.size + 16, dtype=np.float32)
    n = lib().ref_snr_estimator(_p(h), C.c_uint(h.size), C.c_uint(bps), C.c_float(alpha), C.c_uint(nfeeds), _p(out), _p(model))
    return float(out[0]), float(out[1]), float(out[2]), model[:n].copy()
