"""The C++ consumer under examples/ (CostasRecoveryTask on top of the C ABI, no Python in the loop) builds against
include/ + libsigdigger_amd.so and produces the oracle's output bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from sigdigger_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _build(tmp_path):
    exe = str(tmp_path / "costas_task")
    libdir = os.path.join(ROOT, "sigdigger_amd")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", "costas_task.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lsigdigger_amd", "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_example_builds_against_the_headers(tmp_path):
    """CPU: the example compiles and links (cross-compiled for gfx950) -- the boundary is a plain C ABI."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,order", [(1, 2), (2, 4), (3, 8)])
def test_example_matches_the_oracle(tmp_path, sdo, kind, order):
    exe = _build(tmp_path)
    x = synth.psk_carriers(200000, [0.003], sps=8, order=order, seed=60 + kind)      # > 3 slices of 65536
    fin, fout = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    x.tofile(fin)
    r = subprocess.run([exe, fin, fout, str(kind), "8", "0.01"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(fout, dtype=np.complex64)
    ref = sdo.costas_feed_bulk(sdo.costas_new(kind, 0.0, np.float32(1.0) / np.float32(8.0), 3, 0.01), x)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _build_pump(tmp_path):
    exe = str(tmp_path / "analyzer_pump")
    libdir = os.path.join(ROOT, "sigdigger_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "examples", "analyzer_pump.c"), "-I" + os.path.join(ROOT, "include"),
           "-L" + libdir, "-lsigdigger_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_plain_c_pump_builds_with_gcc(tmp_path):
    """CPU: the live-path header is plain C99 -- a C program links the analyzer without hipcc."""
    assert os.path.exists(_build_pump(tmp_path))


@pytest.mark.gpu
def test_plain_c_pump_reads_spectra_until_eos(tmp_path):
    exe = _build_pump(tmp_path)
    fs, n, nblk = 1_000_000, 4096, 6
    x = synth.tone_noise(16 * n * nblk + 100, f_rel=0.1, sigma2=1e-3, seed=3)       # tone at +100 kHz
    cap = str(tmp_path / "cap.raw")
    x.tofile(cap)
    r = subprocess.run([exe, cap, str(fs), str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("psd ")]
    assert len(lines) == nblk and f"end of stream after {nblk} spectra" in r.stdout
    for ln in lines:
        hz = float(ln.split(" at ")[1].split(" Hz")[0])
        assert abs(hz - 100e3) <= fs / n
