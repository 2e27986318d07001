"""Distribution of the in-pipeline channeliser time over the timed steps of the bench workload (diagnostic)."""
import sys, os, argparse
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from sigdigger_amd import engine
args = argparse.Namespace(block=22, channeliser="fft", warmup=3, steps=100)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
ctx = engine.Context(0)
cfg, L, dt, stages, fn, pipe = bench.run_workload("c4", args, 0, 1, dev, ctx, None)
ev = pipe.ev
for a, b in (("fir0", "fir1"), ("psd0", "psd1")):
    t = np.array([s.elapsed_time(e) for s, e in zip(ev[a], ev[b])]) * 1e3
    srt = np.sort(t)
    print(a[:3], f"mean {t.mean():.1f} median {np.median(t):.1f} min {srt[0]:.1f} p90 {srt[89]:.1f} max3 {srt[-3:]} argmax {int(np.argmax(t))} first5 {np.round(t[:5],1)}")
