"""Builds libsigdigger_amd.so (gfx950 code objects + C ABI) in-tree with hipcc.

    python -m sigdigger_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsigdigger_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function", "-Wno-unused-value"] + os.environ.get("SUAMD_BUILD_DEFS", "").split()
# SPEC.md section D: the inspector chain is a fixed sequence of binary32 operations; only the
# fma calls written in the source may fuse.  The FFT PSD is not bit-pinned and may contract.
SOURCES = {
    "psd.hip":   ["-ffp-contract=fast"],
    "chan.hip":  ["-ffp-contract=off"],
    "chan_stream.hip": ["-ffp-contract=off"],
    "loops.hip": ["-ffp-contract=off"],
    "gangs.hip": ["-ffp-contract=off"],
    "specview.hip": ["-ffp-contract=off"],
    "fft.hip": ["-ffp-contract=fast"],
    "psd_large.hip": ["-ffp-contract=fast"],
    "specttuner.hip": ["-ffp-contract=off"],
    "specttuner_wave.hip": ["-ffp-contract=off"],
    "specttuner_pair.hip": ["-ffp-contract=off"],
    "specttuner_host.cpp": ["-ffp-contract=off"],
    "chandet.hip": ["-ffp-contract=off"],
    "audio.hip": ["-ffp-contract=off"],
    "chandet_host.cpp": ["-ffp-contract=off"],
    "ingest.hip": ["-ffp-contract=off"],
    "stages.hip": ["-ffp-contract=off"],
    "capi.hip":  ["-ffp-contract=off"],
    "capi_gangs.hip":  ["-ffp-contract=off"],
    "analyzer.cpp": ["-ffp-contract=off"],
    "analyzer_config.cpp": ["-ffp-contract=off"],
    "export.cpp": ["-ffp-contract=off"],
    "tuning.cpp": ["-ffp-contract=off"],
    "sigutils_host.cpp": ["-ffp-contract=off", "-Wno-return-type-c-linkage"],   # std::complex<float> == float _Complex in the x86-64 ABI
}
HEADERS = ["kernels.hpp", "sd_math.hpp", "loops_dev.hpp", "capi_internal.hpp", "analyzer_internal.hpp", "design.hpp", "fft_core.hpp", "fft_reg.hpp", "tuning.hpp", os.path.join("..", "..", "include", "sigdigger_amd.h"),
           os.path.join("..", "..", "include", "suscan_amd.h")]


def _digest(cmd, deps):
    """sha256 over the compile command and the bytes of every input: what an object or the library was built FROM.
    (mtimes do not survive a checkout or the gpurun snapshot; a stale object shipped beside a newer source must rebuild.)"""
    h = hashlib.sha256()
    h.update("\0".join(cmd).encode())
    for d in deps:
        h.update(b"\0" + os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, stamp):
    """True unless `target` exists and its side file `<target>.sha256` holds `stamp`."""
    if not os.path.exists(target):
        return True
    try:
        with open(target + ".sha256") as f:
            return f.read().strip() != stamp
    except OSError:
        return True


def _mark(target, stamp):
    with open(target + ".sha256", "w") as f:
        f.write(stamp + "\n")


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src, flags in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        cmd = [HIPCC] + COMMON + flags + ["-c", s, "-o", o]
        stamp = _digest(cmd, [s] + hdrs)
        if force or _stale(o, stamp):
            jobs.append((cmd, o, stamp))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    def compile_one(job):
        cmd, o, stamp = job
        run(cmd)
        _mark(o, stamp)

    if jobs:
        with ThreadPoolExecutor(max_workers=int(os.environ.get("SUAMD_BUILD_JOBS", "6"))) as ex:
            list(ex.map(compile_one, jobs))
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-ldl"]
    # the library's stamp covers the objects' stamps, i.e. every source byte and flag
    lstamp = hashlib.sha256(("\0".join(link) + "".join(open(o + ".sha256").read() for o in objs)).encode()).hexdigest()
    if force or jobs or _stale(OUT, lstamp):
        run(link)
        _mark(OUT, lstamp)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
