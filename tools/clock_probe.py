"""Shader clock and power while a kernel runs back to back (sysfs samples during a ~1.5 s stream of launches)."""
import glob, os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, ".")
from sigdigger_amd import engine, synth


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sample():
    """(MHz, W) of every card sysfs shows (the box has eight; the busy one is ours)"""
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        f = p = None
        for h in glob.glob(d + "/hwmon/hwmon*"):
            f = f or read(h + "/freq1_input")
            p = p or read(h + "/power1_input") or read(h + "/power1_average")
        if f or p:
            out.append((int(f) // 1000000 if f else None, int(p) // 1000000 if p else None))
    return out


ctx = engine.Context(0)


def run(name, enqueue, seconds=1.5):
    enqueue(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); enqueue(); e1.record(); torch.cuda.synchronize()
    one = e0.elapsed_time(e1)
    n = max(8, int(seconds * 1e3 / one))
    t0 = time.time()
    samples, stop = [], threading.Event()

    def watch():
        while not stop.is_set():
            samples.append(sample()); time.sleep(0.1)
    th = threading.Thread(target=watch); th.start()
    e0.record()
    for _ in range(n):
        enqueue()
    e1.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch over {n} launches ({time.time() - t0:.2f} s)")
    for s in samples[3:8]:
        print("   ", s)


print("idle", sample())
L = 1 << 22
x = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(x).normal_()
for r in (2, 3, 4):
    st = engine.SpectTuner(ctx, 4096); st.set_run(r)
    for f in synth.raster(64, 1.8 / 64):
        st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / 64)
    out = engine.time_major(64, L // 64 + 64, "cuda")
    def go(st=st, out=out):
        for _ in range(50):
            st.feed(x, out=out)
    run(f"stw run={r} (x50)", go)
    st.close()
L = 1 << 28
xb = torch.empty(L, dtype=torch.complex64, device="cuda"); torch.view_as_real(xb).normal_()
for n in (8192, 1024):
    psd = engine.PSD(ctx, n)
    o = psd.feed(xb, nframes=L // n, navg=256)
    run(f"psd {n} 2^28", lambda psd=psd, o=o, n=n: psd.feed(xb, nframes=L // n, navg=256, out=o))
# a pure copy for comparison
y = torch.empty_like(xb)
run("copy 2 GiB", lambda: y.copy_(xb))
