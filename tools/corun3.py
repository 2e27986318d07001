"""Which property of a resident one-wavefront kernel slows the FIR bank?  (see corun.py)"""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.getcwd())
from sigdigger_amd import engine, synth
ctx = engine.Context(0)
spin = ctypes.CDLL(os.path.join(os.getcwd(), "tools/libspin.so"))
spin.launch_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L = 1 << 22
x = torch.randn(L, dtype=torch.complex64, device="cuda")
bank = engine.ChannelBank(ctx, synth.raster(64, 0.0036), 64, ctx.lpf_design(255, 0.75 / 64))
out = engine.time_major(64, L // 64 + 4, "cuda")
a = torch.zeros(1 << 22, dtype=torch.complex64, device="cuda"); b = torch.zeros_like(a)
side = torch.cuda.Stream()
y = engine.time_major(64, L // 64, "cuda"); z = engine.time_major(64, L // 64, "cuda")
cos = engine.CostasBank(ctx, 64, 2, 0.0, 0.125, 3, 0.005)
clk = engine.ClockBank(ctx, 64, 0.1, 1 / 15.6)
agc = engine.AGCBank(ctx, 64)
sym = torch.zeros((64, L // 64), dtype=torch.complex64, device="cuda"); cnt = torch.zeros(64, dtype=torch.int32, device="cuda")
def corun(kind):
    if kind == "spin": spin.launch_spin(side.cuda_stream, 0, 300000, None, b.data_ptr(), 0)
    elif kind == "spin_mem": spin.launch_spin(side.cuda_stream, 1, 300000, a.data_ptr(), b.data_ptr(), 65536)
    elif kind == "spin_smem": spin.launch_spin(side.cuda_stream, 2, 300000, a.data_ptr(), b.data_ptr(), 1 << 20)
    elif kind == "costas": cos.feed(y, out=z, stream=side)
    elif kind == "clock":
        cnt.zero_(); clk.feed(y, sym, cnt, stream=side)
    elif kind == "agc": agc.feed(y, out=z, stream=side)
bank.feed(x, out=out); torch.cuda.synchronize()
for kind in ("none", "spin", "spin_mem", "spin_smem", "costas", "clock", "agc"):
    ts = []
    for k in range(6):
        corun(kind); time.sleep(0.001)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); bank.feed(x, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"FIR C=64 D=64 next to {kind:10s}: {sum(ts[1:])/5*1e3:6.0f} us")
