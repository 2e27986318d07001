"""Multi-GPU rehearsal of the sharded live analyzer on ONE GPU: SUAMD_DEVICES = 0 x G, the block reaching the
shards through the analyzer's RCCL branch served by tests/rccl_standin.cpp (single-process librccl semantics on one device), 64
heterogeneous PSK inspectors per shard -- C4's real shape at G = 8: 512 inspectors behind one suscan_analyzer handle.  Every shard
shares the one GPU here, so the rates say where the HOST side saturates (source thread, message queue, consumer), not what 8 GPUs
compute.  Two consumers: the Python one of sigdigger_amd/livebench.py and the C one of examples/analyzer_live_bench.c.

    python tools/live_sharded.py [G ...]        (default 1 2 4 8)
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    gs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    d = tempfile.mkdtemp(dir="/tmp")                             # (/dev/shm is mounted noexec on the GPU box)
    try:
        so = os.path.join(d, "librccl_standin.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                               os.path.join(ROOT, "tests", "rccl_standin.cpp"), "-lpthread"])
        exe = os.path.join(d, "analyzer_live_bench")
        subprocess.check_call(["gcc", "-O2", "-std=c99", os.path.join(ROOT, "examples", "analyzer_live_bench.c"), "-I" + os.path.join(ROOT, "include"),
                               "-L" + os.path.join(ROOT, "sigdigger_amd"), "-lsigdigger_amd", "-Wl,-rpath," + os.path.join(ROOT, "sigdigger_amd"),
                               "-lm", "-o", exe])
        cap = os.path.join(d, "cap.raw")
        rng = np.random.default_rng(1)
        (0.1 * rng.standard_normal(2 * (1 << 21) * 4).astype(np.float32)).tofile(cap)
        for g in gs:
            env = dict(os.environ, SUAMD_DEVICES=",".join(["0"] * g))
            if g > 1:
                env.update(SUAMD_ANALYZER_BCAST="rccl", SUAMD_RCCL_LIB=so, SUAMD_RCCL_ALLOW_SAME_DEVICE="1")
            n = 64 * g
            blocks = max(12, 60 // g)
            r = subprocess.run([exe, cap, str(n), str(blocks)], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            c = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
            code = ("import json, sys; sys.path.insert(0, %r); from sigdigger_amd.livebench import live_rate; "
                    "print('LIVE ' + json.dumps(live_rate(%d, %d, timeout_s=240.0)))" % (ROOT, n, blocks))
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("LIVE ")]
            p = json.loads(line[-1][5:]) if line else {"error": (r.stderr or r.stdout)[-300:]}
            r = subprocess.run([exe, cap, str(n), str(blocks * 4), "raw"], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            raw = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
            print(f"shards {g} ({n:3d} inspectors, block exchange: {'ncclBroadcast (stand-in, one device)' if g > 1 else 'none'}):")
            for name, x in (("C consumer     ", c), ("Python consumer", p), ("C consumer, \"raw\" inspectors (channel samples only: the host side's own ceiling)", raw)):
                if "error" in x:
                    print(f"    {name}: {x['error']}")
                else:
                    print(f"    {name}: {x['value_MSps']:8.1f} MS/s at the consumer ({x['ms_per_block']:.2f} ms per 2 Mi-sample block), "
                          f"worker {x['worker_MSps']:8.1f} MS/s, {x['symbols_Msps']:.1f} Msym/s delivered"
                          + (f", {x['sample_messages_per_s']:.0f} messages/s" if "sample_messages_per_s" in x else ""))
            sys.stdout.flush()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
