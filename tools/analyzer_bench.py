"""Live path throughput (sigdigger_amd/livebench.py).  Usage: analyzer_bench.py [N inspectors] [blocks] [class: psk|fsk|ask|raw|audio]
SUAMD_ANALYZER_TRACE=1 adds the host / device timeline of a block; SUAMD_ANALYZER_SUBRANGES sets the stage pipelining."""
import os
import sys

sys.path.insert(0, os.getcwd())
from sigdigger_amd.livebench import live_rate

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NBLK = int(sys.argv[2]) if len(sys.argv) > 2 else 40
CLS = (sys.argv[3] if len(sys.argv) > 3 else "psk").encode()
r = live_rate(N, NBLK, cls=CLS)
if "error" in r:
    sys.exit(r["error"])
print(f"{N} inspectors: {r['value_MSps']:.1f} MS/s sustained ({r['ms_per_block']:.2f} ms per 2097152-sample block, "
      f"{r['symbols_Msps']:.2f} Msym/s delivered); the worker's own measured rate {r['worker_MSps']:.1f} MS/s")
