"""The drop-in boundary, proven by the compiler and the linker.

oracle/Makefile.ref compiles the reference's own live-path wrappers -- Suscan/Analyzer.cpp, MQ.cpp, Message.cpp,
AnalyzerParams.cpp, AnalyzerRequestTracker.cpp, Config.cpp, Source.cpp, Messages/{PSD,Samples,Inspector,Status,
SourceInfo,Channel,Generic}Message.cpp -- UNCHANGED, from where they lie under /root/reference, against the product's
public headers (include/analyzer/analyzer.h, include/sigutils/types.h ...), and links them against
sigdigger_amd/libsigdigger_amd.so with -Wl,--no-undefined: every suscan_analyzer_* / suscan_mq_* / suscan_config_* /
suscan_source_config_* / suscan_source_info_* symbol those files call resolves to the product.  (The control plane they
mention next to it -- XML objects, device discovery, config database -- is stubbed in oracle/ref_stubs.cpp.)

CPU only: here the reference's Suscan::Analyzer must FAIL LOUDLY (no GPU, no CPU path); tests/test_gpu_ref_live.py runs
the same binary on the MI355X.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sdref  # noqa: E402

HAVE_REF = os.path.isdir("/root/reference")
pytestmark = pytest.mark.skipif(not HAVE_REF, reason="/root/reference is not on this machine")

LIVE_TUS = ["Analyzer", "MQ", "Message", "AnalyzerParams", "AnalyzerRequestTracker", "Config", "Source", "Exception"]
MSG_TUS = ["PSDMessage", "SamplesMessage", "InspectorMessage", "StatusMessage", "SourceInfoMessage", "ChannelMessage",
           "GenericMessage"]


@pytest.fixture(scope="module")
def built():
    from sigdigger_amd import build as b
    b.build()
    assert sdref.build()
    return os.path.join(ROOT, "oracle", "_ref")


def test_reference_wrappers_compile_unchanged_against_public_headers(built):
    for tu in LIVE_TUS:
        assert os.path.exists(os.path.join(built, "obj", "Suscan", tu + ".o")), tu
    for tu in MSG_TUS:
        assert os.path.exists(os.path.join(built, "obj", "Suscan", "Messages", tu + ".o")), tu
    # ... and the Tasks whose work() loops the oracle restates
    for tu in ["QuadDemodTask", "DelayedConjTask", "HistogramFeeder", "WaveSampler", "CarrierDetector", "DopplerCalculator",
               "CarrierXlator", "AGCTask", "CostasRecoveryTask", "PLLSyncTask", "LPFTask"]:
        assert os.path.exists(os.path.join(built, "obj", "Tasks", tu + ".o")), tu


def test_every_live_path_symbol_resolves_to_the_product_library(built):
    """nm -u of the reference objects: each undefined suscan_* / su_specttuner_* symbol is exported by libsigdigger_amd.so
    (or is control plane, listed in oracle/ref_stubs.cpp)."""
    so = os.path.join(ROOT, "sigdigger_amd", "libsigdigger_amd.so")
    exported = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", so], text=True).splitlines() if l}
    stubs = open(os.path.join(ROOT, "oracle", "ref_stubs.cpp")).read()
    wanted = set()
    for tu in LIVE_TUS:
        wanted |= _undefined(os.path.join(built, "obj", "Suscan", tu + ".o"))
    for tu in MSG_TUS:
        wanted |= _undefined(os.path.join(built, "obj", "Suscan", "Messages", tu + ".o"))
    wanted |= _undefined(os.path.join(built, "obj", "Tasks", "LPFTask.o"))     # su_specttuner_*: the GPU channeliser
    wanted = {s for s in wanted if s.startswith(("suscan_", "su_", "sigutils_"))}
    assert len(wanted) > 80
    served = {s for s in wanted if s in exported}
    control_plane = {s for s in wanted - served if s in stubs}
    assert wanted == served | control_plane, sorted(wanted - served - control_plane)
    # the whole live ABI is served, none of it stubbed
    for s in wanted:
        if s.startswith(("suscan_analyzer_", "suscan_mq_", "suscan_source_info_", "su_specttuner_")) or \
                (s.startswith("suscan_config_") and not s.startswith("suscan_config_context_")):   # context = config database
            assert s in served, s


TASK_TUS = ["CarrierXlator", "LPFTask", "CostasRecoveryTask", "PLLSyncTask", "QuadDemodTask", "WaveSampler", "AGCTask",
            "CarrierDetector", "DopplerCalculator", "DelayedConjTask", "HistogramFeeder"]


def test_the_offline_tasks_link_unchanged_against_the_product(built):
    """north_star: "Tasks/ (CarrierXlator, LPFTask, CostasRecoveryTask, PLLSyncTask, QuadDemodTask, WaveSampler) link
    unchanged".  Every su_* symbol the reference's Task objects leave undefined -- the per-sample su_ncqo_* / su_pll_* /
    su_costas_* / su_agc_* / su_clock_detector_* calls, su_taps_apply_blackmann_harris_complex, su_specttuner_* -- is
    exported by libsigdigger_amd.so (csrc/sigutils_host.cpp, csrc/specttuner_host.cpp); none of them is defined by the
    test glue or the oracle, and libsdref.so (linked --no-undefined) leaves them undefined for the product to resolve."""
    so = os.path.join(ROOT, "sigdigger_amd", "libsigdigger_amd.so")
    exported = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", so], text=True).splitlines() if l}
    wanted = set()
    for tu in TASK_TUS:
        wanted |= {s for s in _undefined(os.path.join(built, "obj", "Tasks", tu + ".o")) if s.startswith("su_")}
    for s in ("su_ncqo_init", "su_ncqo_set_phase", "su_ncqo_read", "su_pll_init", "su_pll_track", "su_pll_finalize",
              "su_costas_init", "su_costas_feed", "su_costas_finalize", "su_agc_init", "su_agc_feed", "su_agc_finalize",
              "su_clock_detector_init", "su_clock_detector_feed", "su_clock_detector_read", "su_clock_detector_finalize",
              "su_taps_apply_blackmann_harris_complex", "su_specttuner_new", "su_specttuner_open_channel",
              "su_specttuner_feed_bulk", "su_specttuner_destroy"):
        assert s in wanted, s                                  # the call sites VERDICT r2 listed
    assert wanted <= exported, sorted(wanted - exported)
    # ... and nothing else serves them: not the glue, not the oracle
    ref = os.path.join(built, "libsdref.so")
    ref_defined = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", ref], text=True).splitlines() if l}
    sdo_so = os.path.join(ROOT, "oracle", "libsdo.so")
    sdo_defined = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", sdo_so], text=True).splitlines() if l}
    assert not (wanted & ref_defined) and not (wanted & sdo_defined)
    assert wanted <= _undefined_dyn(ref)


def _undefined_dyn(so):
    out = subprocess.check_output(["nm", "-D", "-u", so], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def _undefined(obj):
    out = subprocess.check_output(["nm", "-u", obj], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_reference_analyzer_fails_loudly_without_a_gpu(built, tmp_path):
    import numpy as np
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: tests/test_gpu_ref_live.py covers this binary")
    except ImportError:
        pass
    iq = tmp_path / "iq.f32"
    np.zeros(1 << 16, dtype=np.complex64).tofile(iq)
    r = subprocess.run([os.path.join(built, "ref_live"), str(iq), "1000000", "4096", "0", "50000", str(tmp_path / "o.bin")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert "no CPU fallback" in r.stderr
