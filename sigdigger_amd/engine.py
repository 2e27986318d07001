"""Host-side plumbing over the C ABI (include/sigdigger_amd.h) for Python callers.

PyTorch is used only as the owner of device memory and streams (and torch.distributed for
the multi-GPU IQ broadcast in pipeline.py); every computation happens inside
libsigdigger_amd.so.  Tensors are torch.complex64 / float32 CUDA tensors, whose memory layout
is exactly SUCOMPLEX / SUFLOAT.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _l
from .lib import SigDiggerAmdError, check

WINDOW_NONE, WINDOW_HAMMING, WINDOW_HANN, WINDOW_FLAT_TOP, WINDOW_BLACKMANN_HARRIS = range(5)
PSD_LINEAR, PSD_DB_SHIFTED = 0, 1
FORMAT_F32, FORMAT_U8, FORMAT_S8, FORMAT_S16 = 1, 2, 3, 4
COSTAS_BPSK, COSTAS_QPSK, COSTAS_8PSK = 1, 2, 3


def _stream(stream=None):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _chk_c64(t, name):
    if not (t.is_cuda and t.dtype == torch.complex64 and t.is_contiguous()):
        raise SigDiggerAmdError(f"{name} must be a contiguous CUDA complex64 tensor")


def _chk_rows(t, name):
    if not (t.is_cuda and t.dtype == torch.complex64 and t.dim() == 2):
        raise SigDiggerAmdError(f"{name} must be a CUDA complex64 [channels, time] tensor")


def _view(t):
    """suamd_view of a [channels, time] tensor: any strides (channel-major, or the transpose of a
    [time, channels] buffer = time-major)."""
    return _l.View(t.stride(0), t.stride(1))


def kernel_timing(enable):
    """Switches the library's kernel timer (suamd_kernel_timing): launches of the channeliser / PSD kernels carry an
    event pair bound to the dispatch itself."""
    _l.load().suamd_kernel_timing(1 if enable else 0)


def kernel_timing_read(kernel=None):
    """{sum_ms, min_ms, max_ms, launches} of `kernel`'s launches since the last read (waits for them)."""
    s, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint()
    check(_l.load().suamd_kernel_timing_read(kernel.encode() if kernel else None, C.byref(s), C.byref(lo), C.byref(hi), C.byref(n)),
               "suamd_kernel_timing_read")
    return {"sum_ms": s.value, "min_ms": lo.value, "max_ms": hi.value, "launches": n.value}


def tuning_set(name, value):
    """suamd_tuning_set: one field of the library's tuning struct (csrc/tuning.hpp) by its name or its SUAMD_* environment
    name -- how the A / B tests pick a kernel or a plan inside one process (the environment itself is read once)."""
    check(_l.load().suamd_tuning_set(name.encode(), int(value)), "suamd_tuning_set")


def tuning_get(name):
    v = C.c_longlong()
    check(_l.load().suamd_tuning_get(name.encode(), C.byref(v)), "suamd_tuning_get")
    return v.value


def tuning_fields():
    """[(name, env, default, lo, hi, doc)] of every tuning field"""
    out, i = [], 0
    n, e, d = C.c_char_p(), C.c_char_p(), C.c_char_p()
    df, lo, hi = C.c_longlong(), C.c_longlong(), C.c_longlong()
    while _l.load().suamd_tuning_describe(i, C.byref(n), C.byref(e), C.byref(df), C.byref(lo), C.byref(hi), C.byref(d)):
        out.append((n.value.decode(), e.value.decode(), df.value, lo.value, hi.value, d.value.decode()))
        i += 1
    return out


class tuned:
    """with engine.tuned(SUAMD_ST_SEAM=0, clock_mode=1): ... -- fields set for the block, restored after"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: tuning_get(k) for k in self.kw}
        for k, v in self.kw.items():
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def time_major(nchan, length, device, dtype=torch.complex64):
    """[channels, time] tensor stored time-major ([time][channel] in memory): the layout the
    one-lane-per-channel kernels stream with one contiguous access per wavefront."""
    return torch.empty((length, nchan), dtype=dtype, device=device).t()


class Context:
    """suamd_ctx_t: binds one GPU (replaces suscan_sigutils_init for this path)."""

    def __init__(self, device=0):
        self.lib = _l.load()
        self.device = int(device)
        self.h = self.lib.suamd_ctx_new(self.device)
        if not self.h:
            raise SigDiggerAmdError("suamd_ctx_new: " + _l.last_error())

    def close(self):
        if self.h:
            self.lib.suamd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- streams confined to a set of compute units ---------------------------------------
    def cu_count(self):
        return int(self.lib.suamd_ctx_cu_count(self.h))

    def masked_stream(self, cus):
        """A HIP stream whose kernels only run on the compute units in `cus` (driver numbering: consecutive indices sit on
        consecutive XCDs), wrapped for torch (an ExternalStream does not own it: destroy_stream).  Default stream flags
        (hipExtStreamCreateWithCUMask takes none): it synchronises with the null stream -- torch's default stream on ROCm --
        so work that is to run beside it must be on a stream of its own."""
        n = max(self.cu_count(), max(cus) + 1)
        words = (C.c_uint32 * ((n + 31) // 32))()
        for c in cus:
            words[c >> 5] |= 1 << (c & 31)
        h = self.lib.suamd_stream_new_cu_mask(self.h, words, len(words))
        if not h:
            raise SigDiggerAmdError("suamd_stream_new_cu_mask: " + _l.last_error())
        return torch.cuda.ExternalStream(h, device=torch.device("cuda", self.device))

    def destroy_stream(self, stream):
        check(self.lib.suamd_stream_destroy(self.h, C.c_void_p(stream.cuda_stream)), "suamd_stream_destroy")

    def probe_placement(self, stream, nblocks=2048, spin_ticks=20000):
        """(xcc, se, cu) of every workgroup of a probe launch on `stream` (suamd_probe_placement)."""
        out = (C.c_uint32 * nblocks)()
        check(self.lib.suamd_probe_placement(self.h, _stream(stream), nblocks, spin_ticks, out), "suamd_probe_placement")
        return [((w >> 16) & 0xf, (w >> 8) & 0xf, w & 0xf) for w in out]

    # ---- element-wise entry points -------------------------------------------------------
    def psd_shift_db(self, psd, stream=None):
        """PSDMessage ctor loop, in place on [frames, n] (or [n])."""
        n = psd.shape[-1]
        nfr = psd.numel() // n
        check(self.lib.suamd_psd_shift_db(self.h, _ptr(psd), n, nfr, _stream(stream)), "suamd_psd_shift_db")
        return psd

    def averager_feed(self, last, x, alpha, blend=True, stream=None):
        check(self.lib.suamd_averager_feed(self.h, _ptr(last), _ptr(x), x.numel(), float(alpha), int(blend),
                                           _stream(stream)), "suamd_averager_feed")
        return last

    def inspector_spectrum_db_shift(self, data, stream=None):
        n = data.shape[-1]
        check(self.lib.suamd_inspector_spectrum_db_shift(self.h, _ptr(data), n, data.numel() // n,
                                                         _stream(stream)), "suamd_inspector_spectrum_db_shift")
        return data

    def fnor_to_dphase(self, fnor):
        return int(self.lib.suamd_fnor_to_dphase(float(fnor)))

    def xlate(self, x, phase0, dphase, n0=0, out=None, stream=None):
        _chk_c64(x, "x")
        if out is None:
            out = torch.empty_like(x)
        check(self.lib.suamd_xlate_bulk(self.h, _ptr(x), _ptr(out), x.numel(), phase0 & 0xFFFFFFFF,
                                        dphase & 0xFFFFFFFF, int(n0), _stream(stream)), "suamd_xlate_bulk")
        return out

    def quad_demod(self, x, prev=None, first=True, out=None, prev_out=None, stream=None):
        """x: [channels, len] (or [len]).  Tasks/QuadDemodTask.cpp loop per row."""
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        _chk_rows(x2, "x")
        if out is None:
            out = torch.empty_like(x2)
        o2 = out if out.dim() == 2 else out.unsqueeze(0)
        check(self.lib.suamd_quad_demod_batch(
            self.h, _ptr(x2), _view(x2), _ptr(o2), _view(o2), x2.shape[0], x2.shape[1],
            _ptr(prev) if prev is not None else None, int(first),
            _ptr(prev_out) if prev_out is not None else None, _stream(stream)), "suamd_quad_demod_batch")
        return out if x.dim() == 2 else o2[0]

    def delayed_conj(self, x, delay, out=None, stream=None):
        _chk_c64(x, "x")
        if out is None:
            out = torch.empty_like(x)
        check(self.lib.suamd_delayed_conj_bulk(self.h, _ptr(x), _ptr(out), x.numel(), int(delay),
                                               _stream(stream)), "suamd_delayed_conj_bulk")
        return out

    def histogram_feed(self, x, space, stream=None):
        _chk_c64(x, "x")
        n = x.numel() - (1 if space == 2 else 0)
        out = torch.empty(max(n, 0), dtype=torch.float32, device=x.device)
        check(self.lib.suamd_histogram_feed_bulk(self.h, _ptr(x), x.numel(), int(space), _ptr(out),
                                                 _stream(stream)), "suamd_histogram_feed_bulk")
        return out

    def sample_manual(self, data, symbol_count, symbol_sync, space, nout=None, stream=None):
        """WaveSampler::sampleManual over a whole capture (Tasks/WaveSampler.cpp:96-175)."""
        _chk_c64(data, "data")
        if nout is None:
            nout = int(symbol_count)
        out = torch.empty(nout, dtype=torch.complex64, device=data.device)
        check(self.lib.suamd_sample_manual_bulk(self.h, _ptr(data), data.numel(), float(symbol_count),
                                                int(symbol_sync), int(space), _ptr(out), nout, _stream(stream)),
              "suamd_sample_manual_bulk")
        return out

    def ingest(self, raw, fmt, out=None, stream=None):
        """raw interleaved I/Q (torch uint8 / int8 / int16 / float32 tensor on the GPU) -> complex64."""
        bps = int(self.lib.suamd_format_bytes_per_sample(int(fmt)))
        if bps == 0:
            raise SigDiggerAmdError(f"unknown sample format {fmt}")
        n = raw.numel() * raw.element_size() // bps
        if out is None:
            out = torch.empty(n, dtype=torch.complex64, device=raw.device)
        check(self.lib.suamd_ingest_iq(self.h, int(fmt), _ptr(raw), n, _ptr(out), _stream(stream)), "suamd_ingest_iq")
        return out

    def decision_space(self, x, mode, stream=None):
        _chk_c64(x, "x")
        out = torch.empty(x.numel(), dtype=torch.float32, device=x.device)
        check(self.lib.suamd_decision_space(self.h, _ptr(x), x.numel(), int(mode), _ptr(out), _stream(stream)), "suamd_decision_space")
        return out

    def decide(self, x, mode, bps, vmin, vmax, stream=None):
        _chk_c64(x, "x")
        out = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
        check(self.lib.suamd_decide(self.h, _ptr(x), x.numel(), int(mode), int(bps), float(vmin), float(vmax), _ptr(out),
                                    _stream(stream)), "suamd_decide")
        return out

    def symbol_histogram(self, x, mode, vmin, vmax, nbins, hist=None, stream=None):
        _chk_c64(x, "x")
        if hist is None:
            hist = torch.zeros(nbins, dtype=torch.int32, device=x.device)
        check(self.lib.suamd_symbol_histogram(self.h, _ptr(x), x.numel(), int(mode), float(vmin), float(vmax), int(nbins),
                                              _ptr(hist), _stream(stream)), "suamd_symbol_histogram")
        return hist

    def spectsrc_preproc(self, kind, x, prev0=0j, out=None, stream=None):
        """per-sample transform of an inspector spectrum source (1-based id, suamd_spectsrc_name)"""
        _chk_c64(x, "x")
        if out is None:
            out = torch.empty_like(x)
        check(self.lib.suamd_spectsrc_preproc(self.h, int(kind), _ptr(x), x.numel(), float(prev0.real), float(prev0.imag),
                                              _ptr(out), _stream(stream)), "suamd_spectsrc_preproc")
        return out

    def sample_zero_crossing(self, data, bnor, space, amplitude=False, threshold=0j, zc_angle=1 + 0j, stream=None):
        """WaveSampler::sampleZeroCrossing over a whole capture (Tasks/WaveSampler.cpp:215-292) -> uint8 symbols."""
        _chk_c64(data, "data")
        cap = 4096 * ((data.numel() + 4095) // 4096)
        out = torch.empty(max(cap, 1), dtype=torch.uint8, device=data.device)
        n = self.lib.suamd_sample_zero_crossing_bulk(self.h, _ptr(data), data.numel(), float(bnor), int(space),
                                                     int(bool(amplitude)), float(threshold.real), float(threshold.imag),
                                                     float(zc_angle.real), float(zc_angle.imag), _ptr(out), out.numel(),
                                                     _stream(stream))
        if n < 0:
            raise SigDiggerAmdError("suamd_sample_zero_crossing_bulk: " + _l.last_error())
        return out[:n]

    def conj_prev(self, x, prev0=0j, out=None, stream=None):
        """x[p] * conj(x[p-1]) (Gardner sampler in FREQUENCY space, Tasks/WaveSampler.cpp:188-196)."""
        _chk_c64(x, "x")
        if out is None:
            out = torch.empty_like(x)
        check(self.lib.suamd_conj_prev_bulk(self.h, _ptr(x), _ptr(out), x.numel(), float(prev0.real), float(prev0.imag),
                                            _stream(stream)), "suamd_conj_prev_bulk")
        return out

    def fft_forward(self, x, stream=None):
        """forward FFT of a power-of-two length capture (16..2^24 points)"""
        _chk_c64(x, "x")
        n = x.numel()
        log2n = n.bit_length() - 1
        if (1 << log2n) != n:
            raise SigDiggerAmdError("length must be a power of two")
        out, work = torch.empty_like(x), torch.empty_like(x)
        check(self.lib.suamd_fft_forward_bulk(self.h, _ptr(x), _ptr(out), _ptr(work), log2n, _stream(stream)),
              "suamd_fft_forward_bulk")
        return out

    def carrier_detect(self, data, avg_rel_bw, dc_notch_rel_bw, stream=None):
        """CarrierDetector::work (Tasks/CarrierDetector.cpp): carrier in rad/sample."""
        _chk_c64(data, "data")
        pk = C.c_float(0)
        check(self.lib.suamd_carrier_detect(self.h, _ptr(data), data.numel(), float(avg_rel_bw),
                                            float(dc_notch_rel_bw), C.byref(pk), _stream(stream)),
              "suamd_carrier_detect")
        return pk.value

    def doppler_calc(self, data, fs, f0, want_spectrum=True, stream=None):
        """DopplerCalculator::work: (peak velocity m/s, sigma, max, mirrored spectrum or None)."""
        _chk_c64(data, "data")
        alloc = int(self.lib.suamd_doppler_alloc_size(data.numel()))
        spec = torch.empty(alloc, dtype=torch.float32, device=data.device) if want_spectrum else None
        pk, sg, mx = C.c_float(0), C.c_float(0), C.c_float(0)
        check(self.lib.suamd_doppler_calc(self.h, _ptr(data), data.numel(), float(fs), float(f0),
                                          _ptr(spec) if spec is not None else None, C.byref(pk), C.byref(sg),
                                          C.byref(mx), _stream(stream)), "suamd_doppler_calc")
        return pk.value, sg.value, mx.value, spec

    def rrc_design(self, sps, rolloff, ntaps=None):
        n = int(self.lib.suamd_rrc_ntaps(float(sps))) if ntaps is None else int(ntaps)
        h = np.empty(n, dtype=np.float32)
        self.lib.suamd_rrc_design(h.ctypes.data_as(C.c_void_p), n, float(sps), float(rolloff))
        return h

    def rows_scale(self, x, gain, out=None, stream=None):
        _chk_rows(x, "x")
        if out is None:
            out = torch.empty_like(x)
        check(self.lib.suamd_rows_scale(self.h, _ptr(x), _view(x), _ptr(out), _view(out), x.shape[0], x.shape[1],
                                        float(gain), _stream(stream)), "suamd_rows_scale")
        return out

    def lpf_design(self, ntaps, fc):
        h = np.empty(ntaps, dtype=np.float32)
        self.lib.suamd_lpf_design(h.ctypes.data_as(C.c_void_p), ntaps, float(fc))
        return h


class PSD:
    """suamd_psd_t: plan for detector_params.window_size / .window."""

    def __init__(self, ctx, window_size, window=WINDOW_BLACKMANN_HARRIS):
        self.ctx = ctx
        self.n = int(window_size)
        self.h = ctx.lib.suamd_psd_new(ctx.h, self.n, int(window))
        if not self.h:
            raise SigDiggerAmdError("suamd_psd_new: " + _l.last_error())

    def close(self):
        if self.h:
            self.ctx.lib.suamd_psd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_split_target(self, workgroups):
        """workgroups a launch with few outputs is split into (suamd_psd_set_split_target; 0: the default)"""
        check(self.ctx.lib.suamd_psd_set_split_target(self.h, int(workgroups)), "suamd_psd_set_split_target")

    def feed(self, x, nframes=None, hop=None, navg=1, scale=1.0, mode=PSD_LINEAR, out=None, stream=None):
        _chk_c64(x, "x")
        hop = self.n if hop is None else int(hop)
        if nframes is None:
            nframes = (x.numel() - self.n) // hop + 1 if x.numel() >= self.n else 0
        if nframes and (nframes - 1) * hop + self.n > x.numel():
            raise SigDiggerAmdError("x too short for nframes")
        nout = nframes // navg
        if out is None:
            out = torch.empty((nout, self.n), dtype=torch.float32, device=x.device)
        check(self.ctx.lib.suamd_psd_feed(self.h, _ptr(x), nframes, hop, navg, float(scale), int(mode),
                                          _ptr(out), _stream(stream)), "suamd_psd_feed")
        return out


class ChannelBank:
    """suamd_chanbank_t: translate + low-pass + decimate for a bank of inspector channels."""

    def __init__(self, ctx, fnor, decimation, taps):
        self.ctx = ctx
        fn = np.ascontiguousarray(fnor, dtype=np.float64)
        tp = np.ascontiguousarray(taps, dtype=np.float32)
        self.nchan, self.D, self.ntaps = fn.size, int(decimation), tp.size
        self.h = ctx.lib.suamd_chanbank_new(ctx.h, fn.size, fn.ctypes.data_as(C.c_void_p), self.D,
                                            tp.ctypes.data_as(C.c_void_p), tp.size)
        if not self.h:
            raise SigDiggerAmdError("suamd_chanbank_new: " + _l.last_error())

    def close(self):
        if self.h:
            self.ctx.lib.suamd_chanbank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exclusive(self, on=True):
        """launch plan for a bank that has the device to itself while it is fed (suamd_chanbank_set_exclusive)"""
        check(self.ctx.lib.suamd_chanbank_set_exclusive(self.h, 1 if on else 0), "suamd_chanbank_set_exclusive")

    def output_count(self, length):
        return int(self.ctx.lib.suamd_chanbank_output_count(self.h, int(length)))

    def feed(self, x, out=None, stream=None):
        """x: [len] complex64.  Returns out[:, :n_out] (channel-major)."""
        _chk_c64(x, "x")
        n = self.output_count(x.numel())
        if out is None:
            out = torch.empty((self.nchan, max(n, 1)), dtype=torch.complex64, device=x.device)
        _chk_rows(out, "out")
        nout = C.c_uint64(0)
        if out.shape[0] != self.nchan or out.shape[1] < n:
            raise SigDiggerAmdError(f"out must be [{self.nchan}, >= {n}]")
        check(self.ctx.lib.suamd_chanbank_feed(self.h, _ptr(x), x.numel(), _ptr(out), _view(out),
                                               C.byref(nout), _stream(stream)), "suamd_chanbank_feed")
        return out[:, :nout.value]

    def reset(self, stream=None):
        check(self.ctx.lib.suamd_chanbank_reset(self.h, _stream(stream)), "suamd_chanbank_reset")


class Audio:
    """suamd_audio_t: the "audio" inspector's demodulator + resampler (SPEC.md section Q)."""

    def __init__(self, ctx, equiv_fs, bandwidth):
        self.ctx = ctx
        self.h = ctx.lib.suamd_audio_new(ctx.h, float(equiv_fs), float(bandwidth))
        if not self.h:
            raise SigDiggerAmdError("suamd_audio_new: " + _l.last_error())

    def close(self):
        if self.h:
            self.ctx.lib.suamd_audio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, demodulator, sample_rate, cutoff, volume=1.0, squelch=False, squelch_level=0.0):
        check(self.ctx.lib.suamd_audio_configure(self.h, int(demodulator), float(sample_rate), float(cutoff), float(volume),
                                                 int(bool(squelch)), float(squelch_level)), "suamd_audio_configure")

    def feed(self, x, stream=None):
        _chk_c64(x, "x")
        n = int(self.ctx.lib.suamd_audio_output_count(self.h, x.numel()))
        out = torch.empty(max(n, 1), dtype=torch.complex64, device=x.device)
        got = C.c_uint64(0)
        check(self.ctx.lib.suamd_audio_feed(self.h, _ptr(x), x.numel(), _ptr(out), C.byref(got), _stream(stream)), "suamd_audio_feed")
        return out[:got.value]


class ChannelDetector:
    """suamd_chandet_t: su_channel_detector on the device (SPEC.md section O)."""

    def __init__(self, ctx, n, alpha=1e-2, beta=1e-3, gamma=0.5, snr=2.0):
        self.ctx, self.n = ctx, int(n)
        self.h = ctx.lib.suamd_chandet_new(ctx.h, self.n, float(alpha), float(beta), float(gamma), float(snr))
        if not self.h:
            raise SigDiggerAmdError("suamd_chandet_new: " + _l.last_error())

    def close(self):
        if self.h:
            self.ctx.lib.suamd_chandet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, psd, stream=None):
        assert psd.dtype == torch.float32 and psd.is_contiguous() and psd.numel() == self.n
        check(self.ctx.lib.suamd_chandet_feed(self.h, _ptr(psd), _stream(stream)), "suamd_chandet_feed")

    def channels(self, samp_rate, cap=1024, stream=None):
        out = (_l.Channel * cap)()
        n = self.ctx.lib.suamd_chandet_channels(self.h, float(samp_rate), out, cap, _stream(stream))
        if n < 0:
            raise SigDiggerAmdError("suamd_chandet_channels: " + _l.last_error())
        return [dict(fc=c.fc, f_lo=c.f_lo, f_hi=c.f_hi, bw=c.bw, snr=c.snr, S0=c.S0, N0=c.N0, age=c.age) for c in out[:n]]

    def noise_floor(self, stream=None):
        return float(self.ctx.lib.suamd_chandet_noise_floor(self.h, _stream(stream)))


class SpectTuner:
    """suamd_specttuner_t: the FFT channeliser (su_specttuner semantics, SPEC.md C2)."""

    def __init__(self, ctx, window_size=4096):
        self.ctx = ctx
        self.W = int(window_size)
        self.h = ctx.lib.suamd_specttuner_new(ctx.h, self.W)
        if not self.h:
            raise SigDiggerAmdError("suamd_specttuner_new: " + _l.last_error())
        self.nchan = 0

    def close(self):
        if self.h:
            self.ctx.lib.suamd_specttuner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def open_channel(self, f0, bw, guard=1.0, precise=False):
        """f0, bw in angular units (rad / sample).  Returns the channel index."""
        c = self.ctx.lib.suamd_specttuner_open_channel(self.h, float(f0), float(bw), float(guard), int(bool(precise)))
        if c < 0:
            raise SigDiggerAmdError("suamd_specttuner_open_channel: " + _l.last_error())
        self.nchan = max(self.nchan, c + 1)
        return c

    def close_channel(self, c):
        check(self.ctx.lib.suamd_specttuner_close_channel(self.h, int(c)), "suamd_specttuner_close_channel")

    def channel_size(self, c):
        return int(self.ctx.lib.suamd_specttuner_channel_size(self.h, int(c)))

    def decimation(self, c):
        return int(self.ctx.lib.suamd_specttuner_channel_decimation(self.h, int(c)))

    def set_run(self, run):
        check(self.ctx.lib.suamd_specttuner_set_run(self.h, int(run)), "suamd_specttuner_set_run")

    def set_slots(self, slots):
        """window slots a launch may plan for (768; 1024 when nothing else runs on the device)"""
        check(self.ctx.lib.suamd_specttuner_set_slots(self.h, int(slots)), "suamd_specttuner_set_slots")

    def capacity(self):
        """entries a counts array must hold (suamd_specttuner_channel_capacity)"""
        return int(self.ctx.lib.suamd_specttuner_channel_capacity(self.h))

    def reset(self, stream=None):
        """forget the stream position (seek / gap): history and cross-fade partners"""
        check(self.ctx.lib.suamd_specttuner_reset(self.h, _stream(stream)), "suamd_specttuner_reset")

    def feed(self, x, out=None, stream=None):
        """x: [len] complex64, len a multiple of W/2.  Returns (out, counts): out [nchan, cap] channel-major (or the
        tensor passed in: any 2-D view), counts[c] samples valid in row c."""
        _chk_c64(x, "x")
        if out is None:
            out = torch.empty((max(self.nchan, 1), x.numel() + 16), dtype=torch.complex64, device=x.device)   # decimation 1 at worst
        counts = (C.c_uint64 * max(self.capacity(), 1))()
        check(self.ctx.lib.suamd_specttuner_feed(self.h, _ptr(x), x.numel(), _ptr(out), _view(out), counts, _stream(stream)),
              "suamd_specttuner_feed")
        return out, [int(v) for v in counts]

    def feed_rows(self, x, rows, near=False, stream=None):
        """The same with one row tensor per channel index (suamd_specttuner_feed_rows): rows[c] receives channel c's
        samples.  near=True: suamd_specttuner_feed_rows_near -- the caller vouches that the rows start within 2 GiB of the
        lowest one (rows carved from one arena): 32-bit output addressing.  Returns counts."""
        _chk_c64(x, "x")
        ptrs = [0 if r is None else int(r.data_ptr()) for r in rows]
        # the kernel reads the table after this call returns: it lives on the tuner, not in a temporary the caching
        # allocator may hand to someone else while the launch is still queued on `stream`; a table of another
        # length replaces the old one only after the device has finished with it
        table = getattr(self, "_row_table", None)
        if table is None or table.numel() != len(ptrs) or table.device != x.device:
            if table is not None:
                torch.cuda.synchronize(table.device)
            table = self._row_table = torch.empty(len(ptrs), dtype=torch.int64, device=x.device)
            self._row_table_host = None
        if self._row_table_host != ptrs:
            torch.cuda.synchronize(x.device)                    # an earlier feed may still read the old contents
            table.copy_(torch.tensor(ptrs, dtype=torch.int64))
            self._row_table_host = list(ptrs)
        counts = (C.c_uint64 * max(self.capacity(), 1))()
        if near:
            live = [p for p in ptrs if p]
            check(self.ctx.lib.suamd_specttuner_feed_rows_near(self.h, _ptr(x), x.numel(), table.data_ptr(), min(live), max(live) - min(live) + 8,
                                                               counts, _stream(stream)), "suamd_specttuner_feed_rows_near")
        else:
            check(self.ctx.lib.suamd_specttuner_feed_rows(self.h, _ptr(x), x.numel(), table.data_ptr(), counts, _stream(stream)),
                  "suamd_specttuner_feed_rows")
        return [int(v) for v in counts]

    def feed_mixed(self, x, slab, view_max_size, rows, near=True, stream=None):
        """suamd_specttuner_feed_mixed: channels of at most view_max_size bins become columns of `slab` (2-D complex64
        [time][pitch], column = channel index), wider ones go to rows[c] as in feed_rows.  Returns counts."""
        _chk_c64(x, "x")
        ptrs = [0 if r is None else int(r.data_ptr()) for r in rows]
        table = getattr(self, "_row_table", None)
        if table is None or table.numel() != len(ptrs) or table.device != x.device:
            if table is not None:
                torch.cuda.synchronize(table.device)
            table = self._row_table = torch.empty(max(len(ptrs), 1), dtype=torch.int64, device=x.device)
            self._row_table_host = None
        if self._row_table_host != ptrs:
            torch.cuda.synchronize(x.device)
            if ptrs:
                table[:len(ptrs)].copy_(torch.tensor(ptrs, dtype=torch.int64))
            self._row_table_host = list(ptrs)
        counts = (C.c_uint64 * max(self.capacity(), 1))()
        live = [p for p in ptrs if p]
        base, span = (min(live), max(live) - min(live) + 8) if (near and live) else (None, 0)
        check(self.ctx.lib.suamd_specttuner_feed_mixed(self.h, _ptr(x), x.numel(), _ptr(slab), _l.View(1, slab.shape[1]), int(view_max_size),
                                                       table.data_ptr(), base, span, counts, _stream(stream)), "suamd_specttuner_feed_mixed")
        return [int(v) for v in counts]


class _LoopBank:
    _destroy = None

    def close(self):
        if getattr(self, "h", None):
            getattr(self.ctx.lib, self._destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rows(self, x, out):
        _chk_rows(x, "x")
        if x.shape[0] != self.nchan:
            raise SigDiggerAmdError(f"expected {self.nchan} rows, got {x.shape[0]}")
        if out is None:
            out = torch.empty_like(x)
        _chk_rows(out, "out")
        return out


class CostasBank(_LoopBank):
    """nchan x su_costas_t (Tasks/CostasRecoveryTask.cpp)."""
    _destroy = "suamd_costas_bank_destroy"

    def __init__(self, ctx, nchan, kind, fhint, arm_bw, arm_order, loop_bw):
        self.ctx, self.nchan = ctx, int(nchan)
        self.h = ctx.lib.suamd_costas_bank_new(ctx.h, self.nchan, int(kind), float(fhint), float(arm_bw),
                                               int(arm_order), float(loop_bw))
        if not self.h:
            raise SigDiggerAmdError("suamd_costas_bank_new: " + _l.last_error())

    def feed(self, x, out=None, stream=None):
        out = self._rows(x, out)
        check(self.ctx.lib.suamd_costas_bank_feed(self.h, _ptr(x), _view(x), _ptr(out), _view(out),
                                                  x.shape[1], _stream(stream)), "suamd_costas_bank_feed")
        return out

    def state(self, stream=None):
        om = np.empty(self.nchan, dtype=np.float32)
        ph = np.empty(self.nchan, dtype=np.uint32)
        check(self.ctx.lib.suamd_costas_bank_get_state(self.h, om.ctypes.data_as(C.c_void_p),
                                                       ph.ctypes.data_as(C.c_void_p), _stream(stream)),
              "suamd_costas_bank_get_state")
        return om, ph


class PLLBank(_LoopBank):
    """nchan x su_pll_t (Tasks/PLLSyncTask.cpp)."""
    _destroy = "suamd_pll_bank_destroy"

    def __init__(self, ctx, nchan, fhint, fc):
        self.ctx, self.nchan = ctx, int(nchan)
        self.h = ctx.lib.suamd_pll_bank_new(ctx.h, self.nchan, float(fhint), float(fc))
        if not self.h:
            raise SigDiggerAmdError("suamd_pll_bank_new: " + _l.last_error())

    def feed(self, x, out=None, stream=None):
        out = self._rows(x, out)
        check(self.ctx.lib.suamd_pll_bank_feed(self.h, _ptr(x), _view(x), _ptr(out), _view(out),
                                               x.shape[1], _stream(stream)), "suamd_pll_bank_feed")
        return out

    def state(self, stream=None):
        om = np.empty(self.nchan, dtype=np.float32)
        ph = np.empty(self.nchan, dtype=np.uint32)
        check(self.ctx.lib.suamd_pll_bank_get_state(self.h, om.ctypes.data_as(C.c_void_p),
                                                    ph.ctypes.data_as(C.c_void_p), _stream(stream)),
              "suamd_pll_bank_get_state")
        return om, ph


class FAC(_LoopBank):
    """FACTab::feed (Default/GenericInspector/FACTab.cpp:181-246): fast autocorrelation with EMA."""
    _destroy = "suamd_fac_destroy"

    def __init__(self, ctx, size, alpha):
        self.ctx, self.size = ctx, int(size)
        self.h = ctx.lib.suamd_fac_new(ctx.h, self.size, float(alpha))
        if not self.h:
            raise SigDiggerAmdError("suamd_fac_new: " + _l.last_error())

    def feed(self, x, view_start=0, view_end=None, stream=None):
        _chk_c64(x, "x")
        nb = x.numel() // self.size
        ve = self.size // 2 if view_end is None else int(view_end)
        check(self.ctx.lib.suamd_fac_feed(self.h, _ptr(x), nb, int(view_start), ve, _stream(stream)), "suamd_fac_feed")

    def array(self):
        torch.cuda.synchronize()
        t = torch.empty(self.size // 2, dtype=torch.float32, device="cuda")
        _memcpy_d2d(t, self.ctx.lib.suamd_fac_array(self.h), self.size * 2)
        return t.cpu().numpy()

    def range(self):
        mn, mx = C.c_float(), C.c_float()
        check(self.ctx.lib.suamd_fac_get_range(self.h, C.byref(mn), C.byref(mx), None), "suamd_fac_get_range")
        return mn.value, mx.value


class PowerBank(_LoopBank):
    """suamd_power_bank_t: the "power" inspector class -- mean |x|^2 over windows of N channel samples."""
    _destroy = "suamd_power_bank_destroy"

    def __init__(self, ctx, integrate_samples):
        self.ctx = ctx
        self.h = ctx.lib.suamd_power_bank_new(ctx.h, int(integrate_samples))
        if not self.h:
            raise SigDiggerAmdError("suamd_power_bank_new: " + _l.last_error())

    def set_integrate(self, n, stream=None):
        check(self.ctx.lib.suamd_power_bank_set_integrate(self.h, int(n), _stream(stream)), "suamd_power_bank_set_integrate")

    def feed(self, x, stream=None):
        _chk_c64(x, "x")
        k = int(self.ctx.lib.suamd_power_bank_output_count(self.h, x.numel()))
        out = torch.empty(max(k, 1), dtype=torch.complex64, device=x.device)
        n = C.c_uint64(0)
        check(self.ctx.lib.suamd_power_bank_feed(self.h, _ptr(x), x.numel(), _ptr(out), C.byref(n), _stream(stream)), "suamd_power_bank_feed")
        return out[:n.value]


class BaudEstimator(_LoopBank):
    """suamd_baud_estimator_t: kind 0 = fast autocorrelation valley, 1 = nonlinear (|dx|^2 line): normalised baud; kind 2 =
    carrier (the reference's CarrierDetector centroid, avgRelBw 1/2): residual carrier in cycles per sample."""
    _destroy = "suamd_baud_estimator_destroy"
    FAC, NONLINEAR, CARRIER = 0, 1, 2

    def __init__(self, ctx, kind, size):
        self.ctx = ctx
        self.h = ctx.lib.suamd_baud_estimator_new(ctx.h, int(kind), int(size))
        if not self.h:
            raise SigDiggerAmdError("suamd_baud_estimator_new: " + _l.last_error())

    def feed(self, x, stream=None):
        _chk_c64(x, "x")
        check(self.ctx.lib.suamd_baud_estimator_feed(self.h, _ptr(x), x.numel(), _stream(stream)), "suamd_baud_estimator_feed")

    def get(self):
        torch.cuda.synchronize()
        return float(self.ctx.lib.suamd_baud_estimator_get(self.h))


def _ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


def export_capture(ctx, path, fmt, x, fs, stream=None):
    """suamd_export_capture: a device capture to disk as "raw" / "wav" / "m" / "mat" (ExportSamplesTask's formats)."""
    _chk_c64(x, "x")
    check(ctx.lib.suamd_export_capture(ctx.h, str(path).encode(), fmt.encode(), _ptr(x), x.numel(), float(fs), _stream(stream)),
          "suamd_export_capture")


def source_fix(ctx, x, iq_reverse, dc=None, alpha=0.1, first=True, stream=None):
    """suamd_source_fix, in place on x: I/Q swap and / or removal of the tracked DC level dc (float32[2] device tensor)."""
    _chk_c64(x, "x")
    check(ctx.lib.suamd_source_fix(ctx.h, _ptr(x), x.numel(), int(bool(iq_reverse)), _ptr(dc) if dc is not None else None,
                                   float(alpha), int(bool(first)), _stream(stream)), "suamd_source_fix")
    return x


def gang_chan(ctx, banks, x, outs, stream=None):
    """suamd_chanbank_gang_feed: 1-channel ChannelBanks (own carrier / decimation / taps / stream position) fed the
    same block x in one launch; outs[i]: contiguous 1-D complex64 row.  Returns the rows trimmed to their output counts."""
    _chk_c64(x, "x")
    n = len(banks)
    nout = (C.c_uint64 * n)()
    for b, o in zip(banks, outs):
        if o.numel() < b.output_count(x.numel()):
            raise SigDiggerAmdError("output row too short")
    check(ctx.lib.suamd_chanbank_gang_feed(ctx.h, _ptr_array([b.h for b in banks]), n, _ptr(x), x.numel(),
                                           _ptr_array([_ptr(o) for o in outs]), nout, _stream(stream)), "suamd_chanbank_gang_feed")
    return [o[:int(k)] for o, k in zip(outs, nout)]


def gang_costas(ctx, banks, xs, ys, stream=None):
    """banks: 1-channel CostasBanks; xs / ys: contiguous 1-D complex64 rows (own length each)."""
    n = len(banks)
    lens = (C.c_uint64 * n)(*[x.numel() for x in xs])
    check(ctx.lib.suamd_costas_gang_feed(ctx.h, _ptr_array([b.h for b in banks]), n, _ptr_array([_ptr(x) for x in xs]),
                                         _ptr_array([_ptr(y) for y in ys]), lens, _stream(stream)), "suamd_costas_gang_feed")


def gang_agc(ctx, banks, xs, ys, stream=None):
    n = len(banks)
    lens = (C.c_uint64 * n)(*[x.numel() for x in xs])
    check(ctx.lib.suamd_agc_gang_feed(ctx.h, _ptr_array([b.h for b in banks]), n, _ptr_array([_ptr(x) for x in xs]),
                                      _ptr_array([_ptr(y) for y in ys]), lens, _stream(stream)), "suamd_agc_gang_feed")


def gang_agc_split(ctx, banks, xs, ys, parts=4, stream=None):
    """suamd_agc_gang_feed in its four steps: pre over the whole block, {level, apply} over `parts` consecutive
    sub-ranges of every row, finish.  Same result as gang_agc (the analyzer pipelines its stages this way)."""
    n = len(banks)
    lens_l = [x.numel() for x in xs]
    lens = (C.c_uint64 * n)(*lens_l)
    b = _ptr_array([bk.h for bk in banks])
    px, py = _ptr_array([_ptr(x) for x in xs]), _ptr_array([_ptr(y) for y in ys])
    st = _stream(stream)
    check(ctx.lib.suamd_agc_gang_pre(ctx.h, b, n, px, lens, st), "suamd_agc_gang_pre")
    for j in range(parts):
        m0 = (C.c_uint64 * n)(*[L * j // parts for L in lens_l])
        m1 = (C.c_uint64 * n)(*[L * (j + 1) // parts for L in lens_l])
        check(ctx.lib.suamd_agc_gang_level(ctx.h, b, n, lens, m0, m1, st), "suamd_agc_gang_level")
        check(ctx.lib.suamd_agc_gang_apply(ctx.h, b, n, px, py, lens, m0, m1, st), "suamd_agc_gang_apply")
    check(ctx.lib.suamd_agc_gang_finish(ctx.h, b, n, px, lens, st), "suamd_agc_gang_finish")


def rows_deliver(ctx, srcs, counts, dsts, count_outs, stream=None):
    """suamd_rows_deliver: row i hands counts[i] samples (an int, or a 1-element int32 device counter that is
    cleared afterwards) of srcs[i] to dsts[i] (device, or pinned host tensors) and the number to count_outs[i]."""
    n = len(srcs)
    dc = _ptr_array([None if isinstance(c, int) else _ptr(c) for c in counts])
    fx = (C.c_uint64 * n)(*[c if isinstance(c, int) else 0 for c in counts])
    check(ctx.lib.suamd_rows_deliver(ctx.h, n, _ptr_array([_ptr(x) for x in srcs]), dc, fx, _ptr_array([_ptr(d) for d in dsts]),
                                     _ptr_array([_ptr(c) for c in count_outs]), _stream(stream)), "suamd_rows_deliver")


def gang_pll(ctx, banks, xs, ys, stream=None):
    n = len(banks)
    lens = (C.c_uint64 * n)(*[x.numel() for x in xs])
    check(ctx.lib.suamd_pll_gang_feed(ctx.h, _ptr_array([b.h for b in banks]), n, _ptr_array([_ptr(x) for x in xs]),
                                      _ptr_array([_ptr(y) for y in ys]), lens, _stream(stream)), "suamd_pll_gang_feed")


def gang_cma(ctx, banks, syms, counts, outs, stream=None):
    """syms / outs: 1-D complex64 symbol rows; counts: 1-element int32 device tensors (symbols present per row)."""
    n = len(banks)
    check(ctx.lib.suamd_cma_gang_feed(ctx.h, _ptr_array([b.h for b in banks]), n, _ptr_array([_ptr(s) for s in syms]),
                                      _ptr_array([_ptr(c) for c in counts]), None, _ptr_array([_ptr(o) for o in outs]),
                                      _stream(stream)), "suamd_cma_gang_feed")


def gang_clock(ctx, banks, xs, syms, counts, stream=None):
    n = len(banks)
    lens = (C.c_uint64 * n)(*[x.numel() for x in xs])
    check(ctx.lib.suamd_clock_gang_feed(ctx.h, _ptr_array([b.h for b in banks]), n, _ptr_array([_ptr(x) for x in xs]), lens,
                                        _ptr_array([_ptr(s) for s in syms]), _ptr_array([_ptr(c) for c in counts]),
                                        _stream(stream)), "suamd_clock_gang_feed")


# ---- gangs on time-major slabs: slab = 2-D complex64 tensor [rows][pitch], item i = column cols[i], its first sample in
# row r0[i] (default 0), lens[i] samples
def _slab_cols(slab, cols, r0=None):
    if slab.dim() != 2 or not slab.is_contiguous():
        raise SigDiggerAmdError("a slab is a contiguous 2-D tensor [rows][pitch]")
    pitch = slab.shape[1]
    es = slab.element_size()
    return _ptr_array([int(slab.data_ptr()) + (int(r0[i] if r0 is not None else 0) * pitch + int(c)) * es for i, c in enumerate(cols)])


def gang_costas_slab(ctx, banks, xslab, xcols, yslab, ycols, lens, r0=None, stream=None):
    n = len(banks)
    check(ctx.lib.suamd_costas_gang_feed_slab(ctx.h, _ptr_array([b.h for b in banks]), n, _slab_cols(xslab, xcols, r0), xslab.shape[1],
                                              _slab_cols(yslab, ycols, r0), yslab.shape[1], (C.c_uint64 * n)(*[int(v) for v in lens]), _stream(stream)),
          "suamd_costas_gang_feed_slab")


def gang_pll_slab(ctx, banks, xslab, xcols, yslab, ycols, lens, r0=None, stream=None):
    n = len(banks)
    check(ctx.lib.suamd_pll_gang_feed_slab(ctx.h, _ptr_array([b.h for b in banks]), n, _slab_cols(xslab, xcols, r0), xslab.shape[1],
                                           _slab_cols(yslab, ycols, r0), yslab.shape[1], (C.c_uint64 * n)(*[int(v) for v in lens]), _stream(stream)),
          "suamd_pll_gang_feed_slab")


def gang_clock_slab(ctx, banks, xslab, xcols, lens, syms, counts, r0=None, stream=None):
    n = len(banks)
    check(ctx.lib.suamd_clock_gang_feed_slab(ctx.h, _ptr_array([b.h for b in banks]), n, _slab_cols(xslab, xcols, r0), xslab.shape[1],
                                             (C.c_uint64 * n)(*[int(v) for v in lens]), _ptr_array([_ptr(s) for s in syms]),
                                             _ptr_array([_ptr(c) for c in counts]), _stream(stream)), "suamd_clock_gang_feed_slab")


def gang_agc_slab(ctx, banks, xslab, xcols, yslab, ycols, lens, work, parts=4, stream=None):
    """The AGC's four steps on slabs (suamd_agc_gang_*_slab): work = float32 tensor of 2 * rows * pitch elements."""
    n = len(banks)
    lens_l = [int(v) for v in lens]
    lens_c = (C.c_uint64 * n)(*lens_l)
    b = _ptr_array([bk.h for bk in banks])
    px, py = _slab_cols(xslab, xcols), _slab_cols(yslab, ycols)
    pitch = xslab.shape[1]
    rows = work.numel() // (2 * pitch)
    st = _stream(stream)
    check(ctx.lib.suamd_agc_gang_pre_slab(ctx.h, b, n, _ptr(xslab), pitch, px, lens_c, _ptr(work), rows, st), "suamd_agc_gang_pre_slab")
    for j in range(parts):
        m0 = (C.c_uint64 * n)(*[L * j // parts for L in lens_l])
        m1 = (C.c_uint64 * n)(*[L * (j + 1) // parts for L in lens_l])
        check(ctx.lib.suamd_agc_gang_level_slab(ctx.h, b, n, _ptr(xslab), pitch, px, lens_c, m0, m1, _ptr(work), rows, st), "suamd_agc_gang_level_slab")
        check(ctx.lib.suamd_agc_gang_apply_slab(ctx.h, b, n, _ptr(xslab), pitch, px, _ptr(yslab), yslab.shape[1], py, lens_c, m0, m1,
                                                _ptr(work), rows, st), "suamd_agc_gang_apply_slab")
    check(ctx.lib.suamd_agc_gang_finish_slab(ctx.h, b, n, _ptr(xslab), pitch, px, lens_c, _ptr(work), rows, st), "suamd_agc_gang_finish_slab")


def rows_deliver_strided(ctx, src_ptrs, strides, counts, dsts, count_outs, stream=None):
    """suamd_rows_deliver_strided: src_ptrs[i] = address of the first sample, strides[i] = its element stride."""
    n = len(src_ptrs)
    dc = _ptr_array([None if isinstance(c, int) else _ptr(c) for c in counts])
    fx = (C.c_uint64 * n)(*[c if isinstance(c, int) else 0 for c in counts])
    check(ctx.lib.suamd_rows_deliver_strided(ctx.h, n, _ptr_array(list(src_ptrs)), (C.c_uint64 * n)(*strides), dc, fx,
                                             _ptr_array([_ptr(d) for d in dsts]), _ptr_array([_ptr(c) for c in count_outs]),
                                             _stream(stream)), "suamd_rows_deliver_strided")


class SNREstimator(_LoopBank):
    """SigDigger::SNREstimator (Misc/SNREstimator.cpp) on the device."""
    _destroy = "suamd_snr_estimator_destroy"

    def __init__(self, ctx, bps, alpha):
        self.ctx = ctx
        self.h = ctx.lib.suamd_snr_estimator_new(ctx.h, int(bps), float(alpha))
        if not self.h:
            raise SigDiggerAmdError("suamd_snr_estimator_new: " + _l.last_error())

    def feed(self, hist, stream=None):
        self.length = hist.numel()
        check(self.ctx.lib.suamd_snr_estimator_feed(self.h, _ptr(hist), self.length, _stream(stream)), "suamd_snr_estimator_feed")

    def get(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        check(self.ctx.lib.suamd_snr_estimator_get(self.h, C.byref(a), C.byref(b), C.byref(c), None), "suamd_snr_estimator_get")
        return a.value, b.value, c.value

    def model(self):
        t = torch.empty(self.length, dtype=torch.float32, device="cuda")
        _memcpy_d2d(t, self.ctx.lib.suamd_snr_estimator_model(self.h), self.length * 4)
        return t.cpu().numpy()


class NCOBank(_LoopBank):
    """free-running su_ncqo per channel (afc.offset, InspectorCtl/AfcControl.cpp:54-83)."""
    _destroy = "suamd_nco_bank_destroy"

    def __init__(self, ctx, fnor):
        fn = np.ascontiguousarray(fnor, dtype=np.float64)
        self.ctx, self.nchan = ctx, int(fn.size)
        self.h = ctx.lib.suamd_nco_bank_new(ctx.h, self.nchan, fn.ctypes.data_as(C.c_void_p))
        if not self.h:
            raise SigDiggerAmdError("suamd_nco_bank_new: " + _l.last_error())

    def feed(self, x, out=None, stream=None):
        out = self._rows(x, out)
        check(self.ctx.lib.suamd_nco_bank_feed(self.h, _ptr(x), _view(x), _ptr(out), _view(out), x.shape[1],
                                               _stream(stream)), "suamd_nco_bank_feed")
        return out


class FIRBank(_LoopBank):
    """real-tap FIR at the channel rate for a bank of rows (matched filter, InspectorCtl/MfControl.cpp:56-78)."""
    _destroy = "suamd_fir_bank_destroy"

    def __init__(self, ctx, nchan, taps):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self.ctx, self.nchan, self.ntaps = ctx, int(nchan), int(t.size)
        self.h = ctx.lib.suamd_fir_bank_new(ctx.h, self.nchan, t.ctypes.data_as(C.c_void_p), self.ntaps)
        if not self.h:
            raise SigDiggerAmdError("suamd_fir_bank_new: " + _l.last_error())

    def feed(self, x, out=None, stream=None):
        out = self._rows(x, out)
        check(self.ctx.lib.suamd_fir_bank_feed(self.h, _ptr(x), _view(x), _ptr(out), _view(out), x.shape[1],
                                               _stream(stream)), "suamd_fir_bank_feed")
        return out


class CMABank(_LoopBank):
    """constant-modulus equalizer per channel at the symbol rate (InspectorCtl/EqualizerControl.cpp:56-75)."""
    _destroy = "suamd_cma_bank_destroy"

    def __init__(self, ctx, nchan, ntaps, rate):
        self.ctx, self.nchan, self.ntaps = ctx, int(nchan), int(ntaps)
        self.h = ctx.lib.suamd_cma_bank_new(ctx.h, self.nchan, self.ntaps, float(rate))
        if not self.h:
            raise SigDiggerAmdError("suamd_cma_bank_new: " + _l.last_error())

    def set_locked(self, locked):
        self.ctx.lib.suamd_cma_bank_set_locked(self.h, int(bool(locked)))

    def feed(self, sym, count=None, out=None, stream=None):
        """sym: [nchan, M] channel-major symbol rows; count: int32 [nchan] symbols per row (None = all M)."""
        _chk_rows(sym, "sym")
        if sym.stride(1) != 1:
            raise SigDiggerAmdError("sym must be channel-major")
        if out is None:
            out = torch.empty_like(sym)
        check(self.ctx.lib.suamd_cma_bank_feed(self.h, _ptr(sym), sym.stride(0), _ptr(count) if count is not None else None,
                                               sym.shape[1], _ptr(out), out.stride(0), _stream(stream)),
              "suamd_cma_bank_feed")
        return out

    def weights(self, stream=None):
        w = np.empty((self.ntaps, self.nchan), dtype=np.complex64)
        check(self.ctx.lib.suamd_cma_bank_get_weights(self.h, w.ctypes.data_as(C.c_void_p), _stream(stream)),
              "suamd_cma_bank_get_weights")
        return w


class ClockBank(_LoopBank):
    """nchan x su_clock_detector_t, Gardner (Tasks/WaveSampler.cpp:177-213)."""
    _destroy = "suamd_clock_bank_destroy"

    def __init__(self, ctx, nchan, loop_gain, bhint):
        self.ctx, self.nchan = ctx, int(nchan)
        self.h = ctx.lib.suamd_clock_bank_new(ctx.h, self.nchan, float(loop_gain), float(bhint))
        if not self.h:
            raise SigDiggerAmdError("suamd_clock_bank_new: " + _l.last_error())

    def feed(self, x, sym, count, stream=None):
        """Appends recovered symbols of row c at sym[c, count[c]...]; count is uint32-as-int32 [nchan]."""
        _chk_rows(x, "x")
        _chk_rows(sym, "sym")
        if sym.stride(1) != 1:
            raise SigDiggerAmdError("sym must be channel-major (unit time stride)")
        check(self.ctx.lib.suamd_clock_bank_feed(self.h, _ptr(x), _view(x), x.shape[1], _ptr(sym),
                                                 sym.stride(0), _ptr(count), _stream(stream)),
              "suamd_clock_bank_feed")
        return sym, count

    def set_phase(self, phi, stream=None):
        check(self.ctx.lib.suamd_clock_bank_set_phase(self.h, float(phi), _stream(stream)), "suamd_clock_bank_set_phase")

    def state(self, stream=None):
        bn = np.empty(self.nchan, dtype=np.float32)
        ph = np.empty(self.nchan, dtype=np.float32)
        check(self.ctx.lib.suamd_clock_bank_get_state(self.h, bn.ctypes.data_as(C.c_void_p),
                                                      ph.ctypes.data_as(C.c_void_p), _stream(stream)),
              "suamd_clock_bank_get_state")
        return bn, ph


class AGCBank(_LoopBank):
    """nchan x su_agc_t (Tasks/AGCTask.cpp)."""
    _destroy = "suamd_agc_bank_destroy"

    def __init__(self, ctx, nchan, params=None, tau=None):
        self.ctx, self.nchan = ctx, int(nchan)
        p = _l.AgcParams(-100.0, 6.0, 100, 20, 20, 2.0, 4.0, 20.0, 40.0)
        if tau is not None:
            ctx.lib.suamd_agc_params_from_tau(C.byref(p), float(tau))
        if params is not None:
            p = params
        self.params = p
        self.h = ctx.lib.suamd_agc_bank_new(ctx.h, self.nchan, C.byref(p))
        if not self.h:
            raise SigDiggerAmdError("suamd_agc_bank_new: " + _l.last_error())

    def feed(self, x, out=None, stream=None, wide=None):
        """`wide`: a second stream for the AGC's feed-forward kernels (suamd_agc_bank_feed_split); `stream` then only
        carries the level trackers, and x is read / out complete on `wide`."""
        out = self._rows(x, out)
        if wide is not None:
            check(self.ctx.lib.suamd_agc_bank_feed_split(self.h, _ptr(x), _view(x), _ptr(out), _view(out),
                                                         x.shape[1], _stream(stream), _stream(wide)), "suamd_agc_bank_feed_split")
            return out
        check(self.ctx.lib.suamd_agc_bank_feed(self.h, _ptr(x), _view(x), _ptr(out), _view(out),
                                               x.shape[1], _stream(stream)), "suamd_agc_bank_feed")
        return out


class SpectrumView:
    """suamd_specview_t: the panoramic scanner's SpectrumView (Panoramic/Scanner.cpp) on the GPU."""
    SIZE = 65536

    def __init__(self, ctx):
        self.ctx = ctx
        self.h = ctx.lib.suamd_specview_new(ctx.h)
        if not self.h:
            raise SigDiggerAmdError("suamd_specview_new: " + _l.last_error())

    def close(self):
        if self.h:
            self.ctx.lib.suamd_specview_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_range(self, fmin, fmax, stream=None):
        check(self.ctx.lib.suamd_specview_set_range(self.h, float(fmin), float(fmax), _stream(stream)),
              "suamd_specview_set_range")

    def set_fft(self, bandwidth, rel_bw=0.5):
        self.ctx.lib.suamd_specview_set_fft(self.h, float(bandwidth), float(rel_bw))

    @property
    def spectrum_size(self):
        return int(self.ctx.lib.suamd_specview_spectrum_size(self.h))

    def feed(self, psd, fmin, fmax, adjust_sides=True, count=None, stream=None):
        check(self.ctx.lib.suamd_specview_feed(self.h, _ptr(psd), _ptr(count) if count is not None else None,
                                               psd.numel(), float(fmin), float(fmax), int(adjust_sides),
                                               _stream(stream)), "suamd_specview_feed")

    def feed_sweep(self, frames, centers, adjust_sides=True, stream=None):
        """frames: [F, N] float32 (shifted dB PSD frames); centers: F centre frequencies (Hz)."""
        cen = np.ascontiguousarray(centers, dtype=np.float64)
        check(self.ctx.lib.suamd_specview_feed_sweep(self.h, _ptr(frames), frames.shape[1], frames.shape[0],
                                                     cen.ctypes.data_as(C.c_void_p), int(adjust_sides),
                                                     _stream(stream)), "suamd_specview_feed_sweep")

    def arrays(self):
        """(psd, accum, count) as host numpy arrays (synchronises)."""
        torch.cuda.synchronize()
        out = []
        for which in range(3):
            p = self.ctx.lib.suamd_specview_array(self.h, which)
            t = torch.empty(self.SIZE, dtype=torch.float32, device="cuda")
            _memcpy_d2d(t, p, self.SIZE * 4)
            out.append(t.cpu().numpy())
        return out


def _memcpy_d2d(dst_tensor, src_ptr, nbytes):
    """hipMemcpy DtoD from a raw device pointer into a tensor (uses the HIP runtime torch loaded)."""
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    rc = hip.hipMemcpy(C.c_void_p(dst_tensor.data_ptr()), C.c_void_p(src_ptr), nbytes, 3)   # hipMemcpyDeviceToDevice
    if rc != 0:
        raise SigDiggerAmdError(f"hipMemcpy failed: {rc}")
