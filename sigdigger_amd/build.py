"""Builds libsigdigger_amd.so (gfx950 code objects + C ABI) in-tree with hipcc.

    python -m sigdigger_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
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
    "loops.hip": ["-ffp-contract=off"],
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
    "analyzer.cpp": ["-ffp-contract=off"],
    "export.cpp": ["-ffp-contract=off"],
    "sigutils_host.cpp": ["-ffp-contract=off", "-Wno-return-type-c-linkage"],   # std::complex<float> == float _Complex in the x86-64 ABI
}
HEADERS = ["kernels.hpp", "sd_math.hpp", "design.hpp", "fft_core.hpp", "fft_reg.hpp", os.path.join("..", "..", "include", "sigdigger_amd.h"),
           os.path.join("..", "..", "include", "suscan_amd.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src, flags in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + COMMON + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs +
            ["-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-ldl"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
