"""BASELINE configs[3]'s per-GPU slice through the boundary (suscan_analyzer_* ABI): 8192-pt PSD + 64 QPSK inspectors (D = 64) on a
capture of staggered carriers, unthrottled.  LIVE_LOG2=21|22 (block), LIVE_BLOCKS (timed blocks)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

lg = int(os.environ.get("LIVE_LOG2", "21"))
torch.cuda.init()
out = bench.run_live_roofline(torch.device("cuda", 0), log2_blocks=(lg,), nblocks=int(os.environ.get("LIVE_BLOCKS", "24")))
print(json.dumps(out))
