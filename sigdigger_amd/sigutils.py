"""ctypes binding of include/sigutils/specttuner.h (the su_specttuner_* names Tasks/LPFTask.cpp is written against),
served by libsigdigger_amd.so.  Plumbing for tests and examples only."""
import ctypes as C

import numpy as np

from . import lib as _l

# SUBOOL on_data(const struct sigutils_specttuner_channel *, void *privdata, const SUCOMPLEX *data, SUSCOUNT size)
ON_DATA = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint64)


class SpecttunerParams(C.Structure):
    _fields_ = [("window_size", C.c_uint64), ("early_windowing", C.c_int)]


class ChannelParams(C.Structure):
    _fields_ = [("f0", C.c_float), ("delta_f", C.c_float), ("bw", C.c_float), ("guard", C.c_float), ("precise", C.c_int),
                ("privdata", C.c_void_p), ("on_data", ON_DATA)]


PROTOTYPES = {
    "su_specttuner_new": (C.c_void_p, [C.POINTER(SpecttunerParams)]),
    "su_specttuner_destroy": (None, [C.c_void_p]),
    "su_specttuner_open_channel": (C.c_void_p, [C.c_void_p, C.POINTER(ChannelParams)]),
    "su_specttuner_close_channel": (C.c_int, [C.c_void_p, C.c_void_p]),
    "su_specttuner_feed_bulk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "su_specttuner_channel_get_decimation": (C.c_float, [C.c_void_p]),
    "su_specttuner_channel_get_bw": (C.c_float, [C.c_void_p]),
    "su_specttuner_channel_get_f0": (C.c_float, [C.c_void_p]),
    "su_specttuner_channel_get_size": (C.c_uint, [C.c_void_p]),
}

_bound = None


def load():
    global _bound
    if _bound is None:
        L = _l.load()
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = L
    return _bound


class SpectTuner:
    """su_specttuner_t with numpy-collecting callbacks (what LPFTask::onData does with its destination buffer)."""

    def __init__(self, window_size=4096):
        self.L = load()
        p = SpecttunerParams(window_size, 1)
        self.h = self.L.su_specttuner_new(C.byref(p))
        if not self.h:
            raise RuntimeError("su_specttuner_new: " + _l.last_error())
        self._keep = []
        self.out = {}

    def open_channel(self, f0, bw, guard=1.0, precise=False):
        key = len(self._keep)
        self.out[key] = []

        def on_data(chan, priv, data, size, key=key):
            self.out[key].append(np.ctypeslib.as_array(data, shape=(2 * size,)).copy().view(np.complex64))
            return 1

        cb = ON_DATA(on_data)
        p = ChannelParams(f0, 0.0, bw, guard, int(precise), None, cb)
        ch = self.L.su_specttuner_open_channel(self.h, C.byref(p))
        if not ch:
            raise RuntimeError("su_specttuner_open_channel: " + _l.last_error())
        self._keep.append((cb, p, ch))
        return key

    def channel(self, key):
        return self._keep[key][2]

    def close_channel(self, key):
        if not self.L.su_specttuner_close_channel(self.h, self._keep[key][2]):
            raise RuntimeError("su_specttuner_close_channel: " + _l.last_error())

    def feed(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if not self.L.su_specttuner_feed_bulk(self.h, x.ctypes.data_as(C.c_void_p), x.size):
            raise RuntimeError("su_specttuner_feed_bulk: " + _l.last_error())

    def samples(self, key):
        return np.concatenate(self.out[key]) if self.out[key] else np.zeros(0, np.complex64)

    def close(self):
        if self.h:
            self.L.su_specttuner_destroy(self.h)
            self.h = None
