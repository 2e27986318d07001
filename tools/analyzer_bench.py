"""Live path throughput: the suscan_analyzer_* ABI with N PSK inspectors (different carriers, bauds and loop
bandwidths -- nothing a bank could share) on a looping capture, unthrottled.  Usage: analyzer_bench.py [N] [blocks]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from sigdigger_amd import suscan

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NBLK = int(sys.argv[2]) if len(sys.argv) > 2 else 40
FS, NFFT, L = 50_000_000, 8192, 1 << 21
Lb = suscan.load()
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "cap.raw")
rng = np.random.default_rng(1)
(0.1 * (rng.standard_normal(2 * L * 4).astype(np.float32))).tofile(path)          # 4 blocks, looped
mq = suscan.MQ()
assert Lb.suscan_mq_init(C.byref(mq))
cfg = Lb.suscan_source_config_new(b"file", 1)
Lb.suscan_source_config_set_samp_rate(cfg, FS)
Lb.suscan_source_config_set_path(cfg, path.encode())
Lb.suscan_source_config_set_loop(cfg, 1)
p = suscan.AnalyzerParams.default()
p.detector_params.window_size = NFFT
p.detector_params.window = 4
p.psd_update_int = L / FS
an = Lb.suscan_analyzer_new(C.byref(p), cfg, C.byref(mq))
assert an
Lb.suscan_analyzer_set_throttle_async(an, 0, 0)
for k in range(N):
    fc = (k - N / 2 + 0.5) * 300e3
    bw = 100e3 + 10e3 * (k % 7)
    ch = suscan.Channel(fc=float(fc), f_lo=float(fc - bw / 2), f_hi=float(fc + bw / 2), bw=float(bw), ft=100e6)
    assert Lb.suscan_analyzer_open_ex_async(an, b"psk", C.byref(ch), 1, -1, 1000 + k)
state = {"psd": 0, "sym": 0, "t0": None, "cfg": 0}
while True:
    t = C.c_uint32(0)
    ptr = Lb.suscan_analyzer_read(an, C.byref(t))
    if t.value == suscan.MSG_HALT:
        break
    if t.value == suscan.MSG_INSPECTOR:
        m = C.cast(ptr, C.POINTER(suscan.InspectorMsg)).contents
        if m.kind == suscan.KIND_OPEN:
            k = m.req_id - 1000
            c2 = Lb.suscan_config_dup(m.config)
            Lb.suscan_config_set_integer(c2, b"afc.costas-order", 1 + k % 3)
            Lb.suscan_config_set_float(c2, b"afc.loop-bw", 50.0 + 5 * (k % 11))
            Lb.suscan_config_set_integer(c2, b"clock.type", 1)
            Lb.suscan_config_set_float(c2, b"clock.baud", 20e3 + 1e3 * (k % 13))
            Lb.suscan_analyzer_set_inspector_config_async(an, m.handle, c2, 2000 + k)
            Lb.suscan_config_destroy(c2)
        elif m.kind == suscan.KIND_SET_CONFIG:
            state["cfg"] += 1
    elif t.value == suscan.MSG_PSD:
        if state["cfg"] == N and state["t0"] is None:
            state["t0"], state["psd"] = time.time(), 0
        state["psd"] += 1
        if state["t0"] is not None and state["psd"] == NBLK:
            dt = time.time() - state["t0"]
            print(f"{N} inspectors: {NBLK * L / dt / 1e6:.1f} MS/s sustained ({dt / NBLK * 1e3:.2f} ms per {L}-sample block, "
                  f"{state['sym'] / dt / 1e6:.2f} Msym/s delivered)")
            Lb.suscan_analyzer_req_halt(an)
    elif t.value == suscan.MSG_SAMPLES and state["t0"] is not None:
        state["sym"] += C.cast(ptr, C.POINTER(suscan.SampleBatchMsg)).contents.sample_count
    Lb.suscan_analyzer_dispose_message(t.value, ptr)
Lb.suscan_analyzer_destroy(an)
os.remove(path)
