"""Does the live analyzer's rate depend on how many HIP streams the process already has (hardware-queue sharing)?"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, json, torch
sys.path.insert(0, %r)
k = int(sys.argv[1])
ss = [torch.cuda.Stream() for _ in range(k)]
x = torch.zeros(1024, device='cuda')
for s in ss:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
from sigdigger_amd.livebench import live_rate
r = live_rate(64, 40)
print('LIVE', k, r.get('value_MSps'), r.get('ms_per_block'), r.get('worker_MSps'), flush=True)
""" % ROOT
for k in (0, 1, 2, 3, 4, 5, 6, 7, 8):
    r = subprocess.run([sys.executable, "-c", code, str(k)], capture_output=True, text=True, cwd=ROOT, timeout=300)
    print([ln for ln in r.stdout.splitlines() if ln.startswith("LIVE")] or r.stderr[-300:], flush=True)
