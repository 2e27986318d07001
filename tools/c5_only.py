"""C5 (panoramic sweep over a capture) alone, for profiling."""
import sys, os, argparse, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from sigdigger_amd import engine
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
ctx = engine.Context(0)
print(json.dumps(bench.run_c5(argparse.Namespace(steps=40), dev, ctx)))
