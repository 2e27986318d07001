"""The analyzer's RCCL branch (csrc/analyzer.cpp: setup_rccl, one ncclBroadcast per block) against the REAL librccl, as far as
a box without a second GPU allows: the library loads under the names the analyzer tries and exports every symbol it looks
up (CPU); one rank on one GPU runs the analyzer's exact call shapes -- ncclCommInitAll from a device list, the block in
place as bytes, ncclCommDestroy / ncclCommAbort (GPU).  tests/rccl_one_rank.cpp is the program; the two-GPU case proper is
"0,1:rccl" in test_gpu_analyzer_fft.py (skipped on a one-GPU box), the control flow around the calls runs on
tests/rccl_standin.cpp there."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build(tmp_path):
    exe = str(tmp_path / "rccl_one_rank")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "rccl_one_rank.cpp"), "-ldl"])
    return exe


def test_the_program_looks_up_what_the_analyzer_looks_up():
    """the symbol names and library names of the test program are the analyzer's (a rename on either side fails here)"""
    src = open(os.path.join(ROOT, "sigdigger_amd", "csrc", "analyzer.cpp")).read()
    prog = open(os.path.join(HERE, "rccl_one_rank.cpp")).read()
    names = set(re.findall(r'dlsym\(lib, "(nccl\w+)"\)', src))
    assert names == {"ncclCommInitAll", "ncclBroadcast", "ncclCommDestroy", "ncclCommAbort"}
    assert names == set(re.findall(r'dlsym\(lib, "(nccl\w+)"\)', prog))
    libs = re.findall(r'dlopen\("(librccl[\w.]*)"', src)
    assert libs == ["librccl.so.1", "librccl.so"] == re.findall(r'dlopen\("(librccl[\w.]*)"', prog)
    # the function-pointer types: the same three `using` lines on both sides
    for decl in ("using InitAll = int (*)(void **, int, const int *);",
                 "using Bcast = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);"):
        assert decl in src and decl in prog, decl


def test_the_real_librccl_exports_every_symbol_the_analyzer_resolves(tmp_path):
    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no librccl on this machine")
    r = subprocess.run([_build(tmp_path), "--symbols-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK symbols"), (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_one_rank_of_the_real_librccl_takes_the_analyzers_calls(tmp_path):
    """ncclCommInitAll([0]) -> three in-place byte broadcasts of a 16 MiB block on a non-blocking stream -> ncclCommDestroy;
    again ending in ncclCommAbort: every call ncclSuccess, the block intact"""
    env = dict(os.environ, NCCL_DEBUG="WARN")
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "OK one rank" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
