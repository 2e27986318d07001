"""Throughput of the live path: the suscan_analyzer_* C ABI (include/suscan_amd.h) with N PSK inspectors that share
nothing -- different carriers, bandwidths, bauds, Costas orders and loop bandwidths -- on a looping capture in the
page cache, unthrottled.  Used by bench.py ("live64" under other_workloads) and tools/analyzer_bench.py."""
import ctypes as C
import os
import shutil
import tempfile
import time

import numpy as np

from . import suscan


LIVE_KERNELS = ("stp_kernel", "stw_kernel", "st_kernel", "chan_fir_gang_kernel", "psd_kernel", "psd_reduce_kernel")


def live_rate(n_inspectors=64, nblocks=40, fs=50_000_000, nfft=8192, block=1 << 21, timeout_s=60.0, cls=b"psk", uniform=None,
              ktimer=False, capture=None):
    """uniform: None = 64 inspectors that share nothing (below); dict(spacing=Hz, bw=Hz, baud=Hz, costas_order=n) = every
    inspector the same kind on its own carrier -- BASELINE.json configs[3]'s per-GPU slice through the boundary (bw 300 kHz at
    50 MS/s: decimation 64, 64-bin channels of the FFT filter bank).  ktimer: the library's kernel timer
    (suamd_kernel_timing, process-global) runs over the timed blocks and the result carries `kernels` = {name: {launches,
    avg_ms, min_ms, max_ms}} for the channeliser and PSD kernels the analyzer's worker launched.  capture: the IQ the file
    source loops over (default: four blocks of noise)."""
    Lb = suscan.load()
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(d, "cap.raw")
    try:
        rng = np.random.default_rng(1)
        if capture is not None:                                   # the caller's IQ (complex64, a whole number of blocks), looped
            assert capture.dtype == np.complex64 and capture.size % block == 0
            capture.tofile(path)
        else:
            (0.1 * rng.standard_normal(2 * block * 4).astype(np.float32)).tofile(path)      # 4 blocks of noise, looped
        mq = suscan.MQ()
        assert Lb.suscan_mq_init(C.byref(mq))
        cfg = Lb.suscan_source_config_new(b"file", 1)
        Lb.suscan_source_config_set_samp_rate(cfg, fs)
        Lb.suscan_source_config_set_path(cfg, path.encode())
        Lb.suscan_source_config_set_loop(cfg, 1)
        p = suscan.AnalyzerParams.default()
        p.detector_params.window_size = nfft
        p.detector_params.window = 4
        p.psd_update_int = block / fs
        an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
        assert an
        Lb.suscan_source_config_destroy(cfg)
        Lb.suscan_analyzer_set_throttle_async(an, 0, 0)
        spacing = min(300e3, 0.9 * fs / max(n_inspectors, 1))                               # every channel inside +-fs/2
        if uniform:
            spacing = float(uniform["spacing"])
        for k in range(n_inspectors):
            fc = (k - n_inspectors / 2 + 0.5) * spacing
            bw = (100e3 + 10e3 * (k % 7)) * spacing / 300e3
            if uniform:
                bw = float(uniform["bw"])
            ch = suscan.Channel(fc=float(fc), f_lo=float(fc - bw / 2), f_hi=float(fc + bw / 2), bw=float(bw), ft=100e6)
            assert Lb.suscan_analyzer_open_ex_async(an, cls, C.byref(ch), 1, -1, 1000 + k)
        st = {"psd": 0, "sym": 0, "t0": None, "cfg": 0, "result": None}
        deadline = time.time() + timeout_s
        while True:
            try:                                                  # a deadline on every read: a benchmark must not hang
                tv, ptr = suscan.read_message(Lb, mq, min(timeout_s, 30.0))
            except suscan.AnalyzerStalled as e:
                return {"error": str(e)}                          # the analyzer is left alone (destroying it would join its worker)
            t = C.c_uint32(tv)
            if t.value == suscan.MSG_HALT:
                break
            if t.value == suscan.MSG_INSPECTOR:
                m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
                if m.kind == suscan.KIND_OPEN:
                    k = m.req_id - 1000
                    c2 = Lb.suscan_config_dup(m.config)
                    if uniform:
                        Lb.suscan_config_set_integer(c2, b"afc.costas-order", int(uniform.get("costas_order", 2)))
                        Lb.suscan_config_set_float(c2, b"afc.loop-bw", float(uniform.get("loop_bw", 100.0)))
                        Lb.suscan_config_set_integer(c2, b"clock.type", 1)
                        Lb.suscan_config_set_float(c2, b"clock.baud", float(uniform["baud"]))
                    else:
                        Lb.suscan_config_set_integer(c2, b"afc.costas-order", 1 + k % 3)
                        Lb.suscan_config_set_float(c2, b"afc.loop-bw", 50.0 + 5 * (k % 11))
                        Lb.suscan_config_set_integer(c2, b"clock.type", 1)
                        Lb.suscan_config_set_float(c2, b"clock.baud", (20e3 + 1e3 * (k % 13)) * spacing / 300e3)
                    Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, c2, 2000 + k)
                    Lb.suscan_config_destroy(c2)
                elif m.kind == suscan.KIND_SET_CONFIG:
                    st["cfg"] += 1
            elif t.value == suscan.MSG_PSD:
                if st["cfg"] == n_inspectors and st["t0"] is None:
                    st["warm"] = st.get("warm", 0) + 1
                    if st["warm"] < 4:                            # the first blocks behind the last SET_CONFIG build chains
                        Lb.suscan_analyzer_dispose_message(t.value, ptr)
                        continue
                    st["t0"], st["psd"] = time.time(), 0
                    if ktimer:
                        _ktimer_read(None)
                        suscan._l.load().suamd_kernel_timing(1)
                st["psd"] += 1
                if st["t0"] is not None and st["psd"] == nblocks and st["result"] is None:
                    dt = time.time() - st["t0"]
                    worker = float(Lb.suscan_analyzer_get_measured_samp_rate(an))   # the worker's own rate (EMA over blocks): this
                    kern = None
                    if ktimer:
                        suscan._l.load().suamd_kernel_timing(0)
                        kern = {}
                        for kn in LIVE_KERNELS:
                            r = _ktimer_read(kn)
                            if r["launches"]:
                                kern[kn] = {"launches": r["launches"], "avg_ms": r["sum_ms"] / r["launches"], "min_ms": r["min_ms"], "max_ms": r["max_ms"]}
                        _ktimer_read(None)
                    st["result"] = {"workload": f"live analyzer through the suscan ABI: {nfft}-pt PSD + {n_inspectors} heterogeneous PSK "
                                                f"inspectors (own carrier / bandwidth / baud / Costas order / loop bandwidth), file source, "
                                                f"{block}-sample blocks at {fs / 1e6:g} MS/s",
                                    "value_MSps": round(nblocks * block / dt / 1e6, 3), "ms_per_block": round(dt / nblocks * 1e3, 4),
                                    "symbols_Msps": round(st["sym"] / dt / 1e6, 3), "inspectors": n_inspectors, "blocks": nblocks,
                                    "worker_MSps": round(worker / 1e6, 3), "psd_frames_per_s": round(nblocks / dt, 1),
                                    "block_samples": block, "kernels": kern,
                                    "note": "value_MSps is timed at this Python consumer (one message per inspector and block: it "
                                            "becomes the limit beyond ~100 inspectors); worker_MSps is suscan_analyzer_get_measured_samp_rate"}
                    Lb.suscan_analyzer_req_halt(an)
                elif time.time() > deadline and st["result"] is None:
                    st["result"] = {"error": f"only {st['cfg']} of {n_inspectors} inspectors configured within {timeout_s} s"}
                    Lb.suscan_analyzer_req_halt(an)
            elif t.value == suscan.MSG_SAMPLES and st["t0"] is not None:
                st["sym"] += C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents.sample_count
            elif t.value == suscan.MSG_EOS:
                Lb.suscan_analyzer_dispose_message(t.value, ptr)
                break
            Lb.suscan_analyzer_dispose_message(t.value, ptr)
        Lb.suscan_analyzer_destroy(an)
        Lb.suscan_mq_finalize(C.byref(mq))
        return st["result"] or {"error": "analyzer halted before the measurement finished"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _ktimer_read(kernel):
    s, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint()
    suscan._l.load().suamd_kernel_timing_read(kernel.encode() if kernel else None, C.byref(s), C.byref(lo), C.byref(hi), C.byref(n))
    return {"sum_ms": s.value, "min_ms": lo.value, "max_ms": hi.value, "launches": n.value}


def live_psd_only(fs=2_400_000, nfft=8192, interval_s=0.04, nblocks=400, timeout_s=60.0):
    """BASELINE.json configs[0] through the drop-in boundary: file source, PSD only, no inspector, unthrottled.  The block is
    what the analyzer derives from psd_update_int (csrc/analyzer.cpp setup_psd: nfft x round(fs x interval / nfft) samples ->
    one PSD message per block), i.e. the reference's operating point (include/AppConfig.h:35-38: 25 fps) when interval_s = 0.04."""
    frames = max(1, int(fs * interval_s / nfft + 0.5))
    block = nfft * frames
    # the capture: a whole number of blocks, at least 4 (looped), in the page cache
    d = live_rate(0, nblocks, fs=fs, nfft=nfft, block=block, timeout_s=timeout_s)
    if "error" not in d:
        d["workload"] = (f"live analyzer through the suscan ABI: file source (f32 IQ, page cache, looped), {nfft}-pt PSD only, no inspector, "
                         f"unthrottled; {frames} frames averaged per PSD message ({block}-sample blocks = {interval_s * 1e3:g} ms at {fs / 1e6:g} MS/s)")
        d["realtime_factor"] = round(d["value_MSps"] * 1e6 / fs, 1)
        d.pop("symbols_Msps", None)
        d.pop("note", None)
    return d
