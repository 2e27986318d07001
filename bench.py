#!/usr/bin/env python
"""bench.py -- MS/s of complex IQ sustained through (PSD + N inspectors) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c4|c2|c3] [--block LOG2]

A "step" is one pass of the hot path over one resident block of synthetic IQ: 8192-pt windowed FFT
PSD over every window of the block + the bank of inspector chains (channeliser -> AGC -> Costas -> Gardner; the
channeliser is the FFT filter bank with su_specttuner's semantics by default, --channeliser fir = translate + 255-tap
polyphase decimating low-pass).  With N>1 GPUs (one process per GPU, RCCL)
the inspector channels are sharded 64 per GPU, rank 0's IQ block is broadcast every step
(double-buffered, overlapped with compute), and there is no other collective.

Prints ONE JSON line (rank 0) with the driver's contract fields plus "roofline" and "cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

# one HIP hardware queue per pipeline stream (default is 4; the analyzer uses 4 + RCCL's)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sigdigger_amd import engine, pipeline, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290
FP32_PEAK_TFLOPS = 157.3     # vector FP32 spec peak

WORKLOADS = {
    # C4 / north-star target line: 50 MS/s-class IQ, 8192-pt PSD + 64 QPSK inspectors per GPU
    "c4": dict(what="C4 slice: 8192-pt PSD + 64 QPSK inspectors per GPU",
               rate="D=64, 50 kBd @ 50 MS/s (15.6 sps)",
               psd=8192, per_gpu=64, D=64, T=255, sps_in=1000, kind="psk", spacing=2 * 90e3 / 50e6),
    # C2: 20 MS/s, 8192-pt PSD + 1 PSK inspector
    "c2": dict(what="C2: 8192-pt PSD + 1 QPSK inspector", rate="D=16, 250 kBd @ 20 MS/s (5 sps)",
               psd=8192, per_gpu=1, D=16, T=255, sps_in=80, kind="psk", spacing=0.25),
    # C3: 50 MS/s, 16384-pt PSD + 64 FSK inspectors (quad-demod path)
    "c3": dict(what="C3: 16384-pt PSD + 64 2-FSK inspectors", rate="D=64, 100 kBd @ 50 MS/s (7.8 sps)",
               psd=16384, per_gpu=64, D=64, T=255, sps_in=500, kind="fsk", spacing=2 * 700e3 / 50e6),
}


def describe(cfg, channeliser):
    """what ran, channeliser included, in a few words (the long form goes to the END of the line, `notes`: the driver's
    parsed copy of the line is cut after a few kB)"""
    ch = f"FFT filter bank, {4096 // cfg['D']}-bin channels" if channeliser == "fft" else f"translate + {cfg['T']}-tap polyphase FIR"
    return f"{cfg['what']}; {ch}; {cfg['rate']}"


def describe_long(cfg, channeliser):
    if channeliser == "fft":
        ch = (f"behind the FFT filter bank (su_specttuner semantics: one 4096-pt forward FFT per half window for all channels, "
              f"{4096 // cfg['D']}-bin channels, inverse FFT + cross-fade per channel)")
    else:
        ch = f"behind translate + {cfg['T']}-tap polyphase decimating LPF per channel"
    return f"{cfg['what']} {ch}, {cfg['rate']}"


def make_block(n, fnor, sps_in, kind, device, seed=1234, stagger=True, ppm=100.0):
    """Synthetic IQ block generated on the device: sum of rect-pulse QPSK / 2-FSK carriers + noise.

    stagger (default): the band as it is -- every carrier has its own symbol timing (offset uniform in one symbol) and its own
    baud (+- `ppm` parts per million around the nominal one), so the 64 Gardner detectors of a wavefront cross their half
    cycles at unrelated instants.  stagger=False is rounds 1-5's block: every carrier's symbol boundaries at
    the same samples -- the best case of the lock-step clock kernel -- kept as `value_aligned_clocks` beside the headline."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n, device=device, dtype=torch.float64)
    x = torch.zeros(n, dtype=torch.complex64, device=device)
    nsym = n // sps_in + 4
    rng = np.random.default_rng(seed + 77)
    offs = rng.random(len(fnor)) * sps_in if stagger else np.zeros(len(fnor))
    spss = sps_in * (1.0 + (2.0 * rng.random(len(fnor)) - 1.0) * ppm * 1e-6) if stagger else np.full(len(fnor), float(sps_in))
    for c, f in enumerate(fnor):
        idx = torch.floor((t + float(offs[c])) / float(spss[c])).to(torch.int64)
        if kind == "psk":
            sym = torch.randint(0, 4, (nsym,), generator=g, device=device)
            ph = (np.pi / 2) * sym[idx].to(torch.float64) + np.pi / 4
        else:
            bits = torch.randint(0, 2, (nsym,), generator=g, device=device).to(torch.float64) * 2 - 1
            ph = torch.cumsum(bits[idx] * (np.pi / float(spss[c])), 0)
        ph = ph + (np.pi * float(f)) * t + 0.37 * c
        x += torch.polar(torch.ones_like(ph), ph).to(torch.complex64)
    noise = torch.randn(n, 2, generator=g, device=device, dtype=torch.float32) * 0.05
    x += torch.view_as_complex(noise)
    x *= 1.0 / max(len(fnor), 1) ** 0.5
    return x


_FAST_BUILT = False


def _build_fast_once(sdo):
    global _FAST_BUILT
    if not _FAST_BUILT:
        sdo.build_fast(force=True)
        _FAST_BUILT = True


def cpu_psd_baseline(N, nsamples, ncores):
    """PSD only (BASELINE.json configs[0]'s CPU leg): the oracle's windowed FFT power + averaging over every window of a
    bounded sample, -O3 -march=native build, one thread and all of them (frame ranges)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import sdo
    _build_fast_once(sdo)
    sdo.use_fast(True)
    try:
        rng = np.random.default_rng(9)
        x = (rng.standard_normal(nsamples) + 1j * rng.standard_normal(nsamples)).astype(np.complex64) * 0.3
        win = sdo.window(4, N)
        nfr = nsamples // N

        def rng_(fr):
            return sdo.psd_frames(x[fr[0] * N:fr[1] * N], fr[1] - fr[0], N, N, win, navg=fr[1] - fr[0], scale=1.0 / N)
        t0 = time.perf_counter()
        rng_((0, nfr))
        t1 = time.perf_counter() - t0
        tn = t1
        if ncores > 1:
            edges = [nfr * k // ncores for k in range(ncores + 1)]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(ncores) as ex:
                list(ex.map(rng_, [(edges[k], edges[k + 1]) for k in range(ncores) if edges[k + 1] > edges[k]]))
            tn = time.perf_counter() - t0
        return {"value": round(nsamples / tn / 1e6, 3), "unit": "MS/s", "cores": ncores, "kind": "port",
                "value_1thread": round(nsamples / t1 / 1e6, 3), "frames_per_s_1thread": round(nfr / t1, 1),
                "sample": f"{nsamples} complex samples: {N}-pt Blackman-Harris windowed FFT power over every window, averaged "
                          f"(oracle/sdo.c sdo_psd_frames, -O3 -march=native; a restatement, not upstream sigutils)",
                "cpu_model": _cpu_model()}
    finally:
        sdo.use_fast(False)


def cpu_baseline(cfg, nsamples, fnor_rank, ncores, channeliser):
    """Times the CPU oracle (oracle/sdo.c, a restatement -- NOT upstream sigutils) on a bounded sample of the SAME
    pipeline the GPU leg ran -- same channeliser algorithm, same chains -- on this box's host cores, from the
    -O3 -march=native build of the oracle (oracle/libsdo_fast.so).  Parallel form: the PSD over frame
    ranges, the FFT filter bank over window ranges (every channel per window, the forward transform shared as on the
    GPU), then the serial chains one channel per task."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import sdo
    _build_fast_once(sdo)                                     # -march=native: always built on the box that times it
    sdo.use_fast(True)
    try:
        rng = np.random.default_rng(7)
        x = (rng.standard_normal(nsamples) + 1j * rng.standard_normal(nsamples)).astype(np.complex64) * 0.3
        N, D, T = cfg["psd"], cfg["D"], cfg["T"]
        nch = len(fnor_rank)
        sps = cfg["sps_in"] / D
        win = sdo.window(4, N)
        f0 = [(np.pi * f) % (2 * np.pi) for f in fnor_rank]
        bw = 2 * np.pi * 0.75 / D
        taps = sdo.lpf_design(T, 0.75 / D)

        def fir_row(c):
            dp = sdo.fnor_to_dphase(-fnor_rank[c])
            return sdo.chan_feed(np.zeros(T - 1, np.complex64), x, 0, sdo.chan_modulate_taps(taps, dp), D, 0, dp)

        def chain(y):
            if cfg["kind"] == "psk":
                a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(sps)), y)
                z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, 2.0 / sps, 3, 0.005), a)
            else:
                z = sdo.quad_demod(y)
            return len(sdo.clock_feed_bulk(sdo.clock_new(0.2, 1.0 / sps), z))

        def psd_range(fr):
            return sdo.psd_frames(x[fr[0] * N:fr[1] * N], fr[1] - fr[0], N, N, win, navg=fr[1] - fr[0], scale=1.0 / N)

        def run(threads):
            nfr = nsamples // N
            t0 = time.perf_counter()
            if threads == 1:
                psd_range((0, nfr))
                rows = sdo.specttuner_bank_f32(x, f0, [bw] * nch, [1.0] * nch) if channeliser == "fft" else [fir_row(c) for c in range(nch)]
                for y in rows:
                    chain(y)
            else:
                with ThreadPoolExecutor(threads) as ex:
                    edges = [nfr * k // threads for k in range(threads + 1)]
                    psd_job = ex.map(psd_range, [(edges[k], edges[k + 1]) for k in range(threads) if edges[k + 1] > edges[k]])
                    if channeliser == "fft":
                        list(psd_job)
                        rows = sdo.specttuner_bank_f32(x, f0, [bw] * nch, [1.0] * nch, threads=threads)
                    else:
                        rows = list(ex.map(fir_row, range(nch)))
                        list(psd_job)
                    list(ex.map(chain, rows))
            return time.perf_counter() - t0

        t1 = run(1)
        tn = run(ncores) if ncores > 1 else t1
        alg = ("FFT filter bank in the oracle's binary32 statement (sdo_specttuner_bank_f32: 64 x 64 forward transform shared "
               "by all channels, per-channel inverse transform + cross-fade)") if channeliser == "fft" else \
              f"translate + {T}-tap direct-form FIR per channel (sdo_chan_feed)"
        return {
            "value": round(nsamples / tn / 1e6, 4), "unit": "MS/s", "cores": ncores, "kind": "port",
            "value_1thread": round(nsamples / t1 / 1e6, 4),
            "threads_busy": {"psd": ncores, "channeliser": ncores if channeliser == "fft" else min(ncores, nch), "chains": min(ncores, nch)},
            "channeliser": channeliser,
            "sample": f"{nsamples} complex samples of the same pipeline as the GPU leg: PSD {N}-pt over every window + {alg} + "
                      f"{nch} chains ({'AGC -> Costas -> Gardner' if cfg['kind'] == 'psk' else 'quad demod -> Gardner'}) through "
                      f"oracle/sdo.c built -O3 -march=native -ffp-contract=off (libsdo_fast.so; scalar code, a restatement, not "
                      f"upstream sigutils); `cores` = threads of the pool ({os.cpu_count()} logical CPUs): PSD over frame ranges, "
                      f"channeliser over window ranges, chains one channel per task (at most {nch} busy)",
            "cpu_model": _cpu_model(),
        }
    finally:
        sdo.use_fast(False)


def reference_loops(cfg_c3_D=64):
    """The reference's OWN compiled loops (oracle/_ref/libsdref.so: translation units of /root/reference built by
    oracle/Makefile.ref, g++ -O2 as SigDigger.pro builds them), timed on this box's host cores, one thread -- as the
    reference runs them (GUI thread / one task thread).  These are the literal "reference CPU path" for rows A3 + A4
    (PSDMessage ctor + Averager::feed per PSD frame), P2 / P3 (SpectrumView::feed over C5's 512 dwells) and T5
    (QuadDemodTask::work, C3's demodulator).  None if the library did not travel."""
    try:
        from oracle import sdref
        if not sdref.available():
            return None
        sdref.lib()
    except Exception as e:                                    # a reported extra must not take the bench line down
        return {"error": repr(e)}
    out = {"kind": "reference", "cores": 1,
           "built_from": "Suscan/Messages/PSDMessage.cpp, Misc/Averager.cpp, Panoramic/Scanner.cpp, Tasks/QuadDemodTask.cpp "
                         "(/root/reference, compiled unchanged: oracle/Makefile.ref)"}
    rng = np.random.default_rng(11)
    # A3 + A4: 8192-bin frames through PSDMessage's constructor (fftshift + dB) and Averager::feed
    N, F = 8192, 2000
    frames = [rng.random(N, dtype=np.float32) + 1e-3 for _ in range(8)] * (F // 8)
    t0 = time.perf_counter()
    sdref.averager(frames, 0.1)
    dt = time.perf_counter() - t0
    out["psd_message_plus_averager"] = {"frames_per_s": round(F / dt, 1), "frame_bins": N,
                                        "equivalent_MSps": round(F * N / dt / 1e6, 2),
                                        "what": "PSDMessage ctor + Averager::feed per frame (PSDMessage.cpp:26-39, Averager.cpp:25-50); "
                                                "MS/s if every 8192-sample window became a frame"}
    # P2 / P3: C5's sweep -- 512 dwells of 8192 bins into the 65536-bin view, interpolate after each (Scanner.cpp:239-256)
    fs, rel, dw = 20e6, 0.5, 512
    v = sdref.SpectrumView()
    f_lo = 100e6
    v.set_range(f_lo, f_lo + dw * fs * rel)
    v.set_fft(fs, rel)
    fr = (-100 + 30 * rng.random(N)).astype(np.float32)
    t0 = time.perf_counter()
    for k in range(dw):
        fc = f_lo + (k + 0.5) * fs * rel
        v.feed(fr, fc - fs / 2, fc + fs / 2)
        v.interpolate()
    dt = time.perf_counter() - t0
    out["spectrum_view_sweep"] = {"dwells": dw, "ms_per_sweep": round(dt * 1e3, 2), "dwells_per_s": round(dw / dt, 1),
                                  "what": "SpectrumView::feed + interpolate per dwell, 8192-bin frames (Scanner.cpp:56-256)"}
    # T5: QuadDemodTask::work over one channel's samples (C3: 64 channels at 1/64 of the input rate each)
    m = 1 << 20
    y = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    t0 = time.perf_counter()
    sdref.quad_demod(y)
    dt = time.perf_counter() - t0
    out["quad_demod_task"] = {"channel_MSps": round(m / dt / 1e6, 2),
                              "what": f"QuadDemodTask::work (QuadDemodTask.cpp:39-76) on one channel's stream; C3 has 64 channels at "
                                      f"1/{cfg_c3_D} of the input rate each, i.e. this many input MS/s on one thread"}
    return out


def pmc_traffic(kernel, workload, block):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json:
    2*FETCH_SIZE + WRITE_SIZE, the gfx950 half-count correction calibrated there); None when the
    profile was taken on a different workload / block size."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if w.get("name") == workload and w.get("block_samples") == block and kernel in d.get("kernels", {}):
            import hashlib
            raw = open(f, "rb").read()
            blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()[:12]
            best = (d["kernels"][kernel].get("hbm_bytes_per_launch"),
                    f"profiles/{os.path.basename(f)} (git blob {blob}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER run "
                    f"of this workload (2 x FETCH + WRITE), not measured in this run")
    return best if best else (None, None)


# The benched block: 16 Mi samples (0.34 s of a 50 MS/s stream) since round 4.  The sustained rate does not depend on it
# (the step is the Costas recurrence: 809 / 813 / 815 MS/s at 4 / 8 / 16 Mi), the channeliser's and the PSD's launches
# do: a 4 Mi block gives a workgroup three windows, so a launch's start (every workgroup's first 32 KiB at once), seam
# hand-off and drain are a third of it; measured INSIDE the pipeline, same box, back to back (tools/block_sweep.sh,
# profiles/r04_block_sweep.txt): channeliser 0.277 / 0.307 / 0.347 of the HBM peak, PSD 0.203 / 0.235 / 0.284.
# `--block 22` is round 1-3's line; `roofline.fir_stage_alone` reports both sizes with the kernel alone.
DEFAULT_LOG2_BLOCK = 24

CHANNELISER_KERNELS = ("stp_kernel", "stw_kernel", "st_kernel", "chan_pair_kernel", "chan_fir_kernel")
KERNELS = {
    "stp_kernel": "stp_kernel (FFT channeliser, two wavefronts per window: 4096-pt forward FFT as two 64-pt DFTs on registers "
                  "around an LDS transposition, each DFT split between the wavefronts with one swap through LDS; lane = "
                  "channel: bin pick x response, 64-pt inverse FFT, cross-fade)",
    "stw_kernel": "stw_kernel (FFT channeliser, one wavefront per window: 4096-pt forward FFT as two register DFT64 "
                  "around an LDS transposition, shared by all channels; lane = channel: bin pick x response, 64-pt "
                  "inverse FFT, cross-fade)",
    "st_kernel": "st_kernel (FFT channeliser, one workgroup per run of windows: radix-16 passes through LDS)",
    "chan_pair_kernel": "chan_pair_kernel (translate + polyphase decimating FIR for one or two channels: tiles of 256 outputs, "
                        "lane = two adjacent outputs sharing their samples in LDS, taps from scalar loads, the SPEC's fma chain "
                        "per output in asm; SUAMD_FIR_PAIR_NW=8: persistent workgroups streaming runs of 1024-output tiles)",
    "chan_fir_kernel": "chan_fir_kernel (translate + 255-tap polyphase decimating FIR bank)"}


def fir_stage_roofline(C, D, L, kernel_ms, kname, workload, timing, traffic=None):
    """roofline object of the north star's "FIR stage" (the channeliser): algorithmic (compulsory) bytes per launch
    (the shared input once + every channel's decimated output) over the kernel's own duration.
    `traffic`: (bytes per launch, source) measured by this run's own PMC passes; else the committed profile's figure."""
    nbytes = 8.0 * L + 8.0 * C * (L // D)
    tr, tsrc = traffic if traffic and traffic[0] else pmc_traffic(kname, workload, L)
    return {"kernel": kname, "bound": "hbm",
            "achieved": round(nbytes / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kernel_ms else None,
            "traffic": tr, "traffic_over_algorithmic": round(tr / nbytes, 4) if tr else None, "traffic_source": tsrc,
            "algorithmic_bytes_per_launch": nbytes, "kernel_ms": round(kernel_ms, 4) if kernel_ms else None, "timing": timing}


def pmc_traffic_in_run(args, kernels):
    """HBM bytes per launch of `kernels`, COUNTED IN THIS RUN: two child runs of this script's default workload under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, nothing else traced -- MI355X_MICROARCH.md's
    recipe), 4 steps each; bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950 counts half of the streamed reads: calibrated in
    profiles/r01_pmc_traffic.json with a copy kernel).  {} when rocprofv3 is not on the box or a pass fails."""
    import csv, glob, shutil, subprocess, tempfile
    if os.environ.get("SUAMD_BENCH_CHILD") or not shutil.which("rocprofv3"):
        return {}
    acc = {}
    tmp = tempfile.mkdtemp(prefix="suamd_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "4", "--warmup", "1", "--lean", "--block", str(args.block),
                   "--workload", args.workload, "--channeliser", args.channeliser] + (["--aligned"] if args.aligned else [])
            env = dict(os.environ, SUAMD_BENCH_CHILD="1", TMPDIR="/tmp")
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != ctr:
                        continue
                    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
                    if k in kernels:
                        acc.setdefault(k, {}).setdefault(ctr, []).append(float(row["Counter_Value"]))
    except Exception:
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k, v in acc.items():
        if v.get("FETCH_SIZE") and v.get("WRITE_SIZE"):
            f, w = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            res[k] = {"hbm_bytes_per_launch": int(2048 * f + 1024 * w), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                      "launches": len(v["FETCH_SIZE"])}
    return res


def live_kernel_trace_in_run(log2_blocks=(21, 22), nblocks=16):
    """The channeliser's and the PSD kernels' OWN durations inside the C++ live analyzer: child runs of the live leg
    (tools/live_c4.py = run_live_roofline at one block length) under `rocprofv3 --kernel-trace --stats`.  The analyzer's
    worker keeps four streams busy, and a dispatch-bound event pair there starts at a marker IN FRONT of the kernel's packet:
    it also counts the time the dispatch waits for compute units that other streams' workgroups hold (58 - 62 us read for a
    kernel rocprofv3 times at 30 - 40 us).  The profiler reads the dispatch's own timestamps.  {} when rocprofv3 is not on
    the box or a pass fails; keys as run_live_roofline's."""
    import csv, glob, shutil, subprocess, tempfile
    if os.environ.get("SUAMD_BENCH_CHILD") or not shutil.which("rocprofv3"):
        return {}
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    tmp = tempfile.mkdtemp(prefix="suamd_live_", dir="/tmp")
    try:
        for lg in log2_blocks:
            out = os.path.join(tmp, str(lg))
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
                   os.path.join(here, "tools", "live_c4.py")]
            env = dict(os.environ, SUAMD_BENCH_CHILD="1", TMPDIR="/tmp", LIVE_LOG2=str(lg), LIVE_BLOCKS=str(nblocks), PYTHONPATH=here)
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=180)
            files = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
            if r.returncode != 0 or not files:
                continue
            L = 1 << lg
            chan, psd = None, 0.0
            for row in csv.DictReader(open(files[0])):
                k = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
                avg_us, calls = float(row["AverageNs"]) / 1e3, int(row["Calls"])
                if k in CHANNELISER_KERNELS and (chan is None or avg_us * calls > chan[1] * chan[2]):
                    chan = (k, avg_us, calls, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3)
                if k in ("psd_kernel", "psd_reduce_kernel"):
                    psd += avg_us
            if chan:
                nbytes = 8.0 * L + 8.0 * 64 * (L // 64)
                e = {"kernel": chan[0], "kernel_us": round(chan[1], 2), "launches": chan[2], "min_max_us": [round(chan[3], 2), round(chan[4], 2)],
                     "frac": round(nbytes / (chan[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                if psd:
                    e["psd_kernel_us"] = round(psd, 2)
                    e["psd_frac"] = round((8.0 * L + 4.0 * 8192) / (psd * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                res[f"{L >> 20}Mi"] = e
    except Exception:
        pass
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def channeliser_alone(ctx, dev, fn, D, T, channeliser, log2_block, slots=None, reps=10):
    """the channeliser kernel alone on a resident block (dispatch-bound event pairs): ms per launch and the kernel's name"""
    Lb = 1 << log2_block
    x = torch.empty(Lb, dtype=torch.complex64, device=dev)
    torch.view_as_real(x).normal_()
    C = len(fn)
    if channeliser == "fft":
        st = engine.SpectTuner(ctx, 4096)
        if slots:
            st.set_slots(slots)
        for f in fn:
            st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
        out = engine.time_major(C, Lb // D + 64, dev)
        feed = lambda: st.feed(x, out=out)
    else:
        bank = engine.ChannelBank(ctx, fn, D, ctx.lpf_design(T, 0.75 / D))
        out = engine.time_major(C, Lb // D + 8, dev)
        feed = lambda: bank.feed(x, out=out)
    feed()
    torch.cuda.synchronize(dev)
    engine.kernel_timing_read()
    engine.kernel_timing(True)
    for _ in range(reps):
        feed()
    torch.cuda.synchronize(dev)
    engine.kernel_timing(False)
    best = None
    for k in CHANNELISER_KERNELS:
        r = engine.kernel_timing_read(k)
        if r["launches"] and (best is None or r["sum_ms"] > best[1]["sum_ms"]):
            best = (k, r)
    engine.kernel_timing_read()
    if channeliser == "fft":
        st.close()
    if best is None:
        return None, None
    return best[1]["sum_ms"] / reps, best[0]        # (a bank of several sizes launches once per size: the sum per feed)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_workload(name, args, rank, world, dev, ctx, dist, stagger=True):
    cfg = WORKLOADS[name]
    L = 1 << args.block
    nch_total = cfg["per_gpu"] * world
    fn_all = synth.raster(nch_total, cfg["spacing"])
    fn_rank = pipeline.shard_channels(fn_all, rank, world)
    bank = pipeline.InspectorBankConfig(kind=cfg["kind"], fnor=fn_rank, decimation=cfg["D"], ntaps=cfg["T"],
                                        sps=cfg["sps_in"] / cfg["D"], channeliser=args.channeliser)
    # PSD runs on rank 0 only; frames averaged to ~25 fps at 50 MS/s
    navg = min(256, L // cfg["psd"])
    pipe = pipeline.AnalyzerPipeline(ctx, L, psd_size=cfg["psd"], psd_navg=navg, bank=bank, do_psd=(rank == 0))
    pipe.enable_delivery()       # the recovered symbols reach (pinned) host memory inside the timed region

    bufs = [make_block(L, fn_all, cfg["sps_in"], cfg["kind"], dev, seed=1234, stagger=stagger),
            torch.empty(L, dtype=torch.complex64, device=dev)]
    bufs[1].copy_(bufs[0])
    torch.cuda.synchronize(dev)

    def one_step(k, timed):
        cur = bufs[k & 1]
        # rank 0's next block -> every GPU over xGMI: started right BEHIND this block's PSD + channeliser (it must not take CUs
        # from their one round of workgroups) and overlapped with the serial stages
        work = []
        pipe.step(cur, timed=timed, after_transforms=lambda: work.append(pipeline.broadcast_block(bufs[(k + 1) & 1], dist)))
        pipe.deliver()
        if work and work[0] is not None:
            work[0].wait()

    def fence():
        pipe.flush()                  # (a transform window holds the last block's tail back until the next block comes)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the steps are enqueued with the pipeline's transform stream current (CU-partitioned device: the masked stream the PSD
    # and the channeliser run on), so the block broadcast orders itself against that stream and a step costs no extra events
    with torch.cuda.stream(pipe.main_stream()):
        if dist is not None:
            pipeline.broadcast_block(bufs[0], dist, async_op=False)
        for k in range(args.warmup):
            one_step(k, False)
        fence()
        pipe.reset_events()
        # the roofline leg: the channeliser's and the PSD's launches carry an event pair bound to the dispatch itself
        # (suamd_kernel_timing) -- the kernel's own duration, what rocprofv3 --kernel-trace reports for it
        engine.kernel_timing_read()
        engine.kernel_timing(os.environ.get("SUAMD_BENCH_NO_KTIMER") != "1")     # (diagnosis: what the event pairs themselves cost)
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(k, True)
        fence()
        dt = time.perf_counter() - t0
        engine.kernel_timing(False)
    pipe.kernel_ms = {}
    for kname in CHANNELISER_KERNELS + ("psd_kernel", "psd_reduce_kernel"):
        r = engine.kernel_timing_read(kname)
        if r["launches"]:
            n, tot = r["launches"], r["sum_ms"]
            # `avg` and `per_step` -- what roofline.frac is computed from -- are over ALL launches of the timed region.  Now
            # and then an event pair of a 50-100 us kernel reads 1-2 ms (the queue stalls between the two events; rocprofv3's
            # trace of the same runs never shows such a kernel): the mean without the single worst sample is reported BESIDE
            # it (`avg_without_worst`) when that sample is more than five times the shortest, never instead of it.
            entry = {"avg": tot / n, "min": r["min_ms"], "max": r["max_ms"], "launches": n}
            if n >= 8 and r["max_ms"] > 5.0 * r["min_ms"]:
                entry.update({"avg_without_worst": (tot - r["max_ms"]) / (n - 1), "worst_sample_ms": r["max_ms"]})
            entry["per_step"] = entry["avg"] * n / args.steps
            pipe.kernel_ms[kname] = entry
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stages = pipe.stage_times_ms()
    return cfg, L, dt, stages, fn_rank, pipe


def run_host_fed(name, args, dev, ctx):
    """PCIe-inclusive rate of the default workload (never `value`): every block starts in pinned host memory and
    crosses PCIe on a copy stream into one of two device buffers while the previous block is processed."""
    cfg = WORKLOADS[name]
    L = 1 << args.block
    fn = synth.raster(cfg["per_gpu"], cfg["spacing"])
    bank = pipeline.InspectorBankConfig(kind=cfg["kind"], fnor=fn, decimation=cfg["D"], ntaps=cfg["T"], sps=cfg["sps_in"] / cfg["D"],
                                        channeliser=args.channeliser)
    pipe = pipeline.AnalyzerPipeline(ctx, L, psd_size=cfg["psd"], psd_navg=min(256, L // cfg["psd"]), bank=bank, do_psd=True)
    pipe.enable_delivery()
    host = make_block(L, fn, cfg["sps_in"], cfg["kind"], dev, seed=99).cpu().pin_memory()
    bufs = [torch.empty(L, dtype=torch.complex64, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(k):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[k & 1])           # the FIR / PSD of the block that used this buffer are done
            bufs[k & 1].copy_(host, non_blocking=True)
            ready[k & 1].record(copy_stream)

    for e in consumed:
        e.record(main)
    steps, warm = max(6, min(50, args.steps // 2)), 2
    upload(0)
    t0 = None
    for k in range(warm + steps):
        if k == warm:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        upload(k + 1)
        main.wait_event(ready[k & 1])
        pipe.step(bufs[k & 1], timed=False)
        pipe.deliver()
        consumed[k & 1].record(main)                          # step() enqueued its readers of the block on `main`
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"value_MSps": round(L * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "what": "same workload with every block crossing PCIe (pinned host -> HBM on a copy stream, double-buffered)",
            "pcie_GBps": round(8.0 * L * steps / dt / 1e9, 2)}


def run_fir_stage_alone(cfg, dev, ctx, fn_rank, log2_block):
    """The FFT channeliser of the default workload alone (same 64 channels, same kernel, one resident block re-fed) on a
    block of 2^log2_block samples.  A 4 Mi block gives a workgroup three windows, so its start (the first 32 KiB of every
    workgroup at once) and the 2-or-3-windows quantisation weigh a third of the launch; on a longer block they amortise.
    Reported beside the headline's `roofline` (the kernel inside the pipeline), never instead of it."""
    Lb, D = 1 << log2_block, cfg["D"]
    x = torch.empty(Lb, dtype=torch.complex64, device=dev)
    torch.view_as_real(x).normal_()
    res = {}
    # two launch plans: the default budget of 768 window slots (what a tuner inside the pipeline gets: the recurrence
    # kernels hold a few slots), and all 1024 (suamd_specttuner_set_slots: a tuner that has the device to itself, as here)
    for slots in (768, 1024):
        st = engine.SpectTuner(ctx, 4096)
        st.set_slots(slots)
        for f in fn_rank:
            st.open_channel(np.pi * f % (2 * np.pi), 2 * np.pi * 0.75 / D)
        out = engine.time_major(len(fn_rank), Lb // D + 64, dev)
        st.feed(x, out=out)
        torch.cuda.synchronize(dev)
        engine.kernel_timing_read()
        engine.kernel_timing(True)
        for _ in range(10):
            st.feed(x, out=out)
        torch.cuda.synchronize(dev)
        engine.kernel_timing(False)
        r = engine.kernel_timing_read()
        ms = r["sum_ms"] / max(r["launches"], 1)
        st.close()
        nbytes = 8.0 * Lb + 8.0 * len(fn_rank) * Lb / D
        res[f"slots_{slots}"] = {"kernel_ms": round(ms, 4), "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                 "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    best = res["slots_768"]
    return {"block_samples": Lb, "kernel_ms": best["kernel_ms"], "algorithmic_bytes_per_launch": nbytes,
            "achieved": best["achieved"], "unit": "GB/s", "frac": best["frac"], "plans": res,
            "what": "the channeliser kernel alone, one resident block re-fed (dispatch-bound event pairs); headline fields: the "
                    "default 768-slot plan (what the pipeline runs); `plans` also has the 1024-slot plan of a tuner that has "
                    "the device to itself (suamd_specttuner_set_slots)"}


def run_c5(args, dev, ctx, log2_file=30, N=8192, tile=256, rank=0, world=1, dist=None, steps=None, warm=1):
    """C5: panoramic-scanner sweep over a captured file resident in HBM -- every dwell is `tile` frames of
    N points averaged into one shifted-dB PSD message (PSDMessage.cpp:26-39 fused), fed to the SpectrumView
    at the dwell's centre frequency (Panoramic/Scanner.cpp:239-293).  One step = one pass over the file.

    world > 1: the dwells (frames of the sweep) are independent (Panoramic/Scanner.cpp:503-523 feeds them one by one into
    bins that only they cover), so rank r takes the dwells d = r (mod world) -- its share of the capture resident in its
    own HBM, its own SpectrumView over the whole range -- and NOTHING is exchanged ("C5 shards by frame").
    The capture is fixed (2^log2_file samples over all ranks): strong scaling."""
    total = 1 << log2_file
    dwells_all = total // (N * tile)
    mine = pipeline.shard_frames(dwells_all, rank, world)
    dwells = len(mine)
    share = dwells * N * tile
    x = torch.empty(share, dtype=torch.complex64, device=dev)
    xr = torch.view_as_real(x)
    chunk = 1 << 26
    g = torch.Generator(device=dev); g.manual_seed(5 + rank)
    for o in range(0, share, chunk):
        xr[o:o + chunk].normal_(generator=g)
    psd = engine.PSD(ctx, N)                     # Blackman-Harris (the analyzer default)
    frames = torch.empty((dwells, N), dtype=torch.float32, device=dev)
    fs, rel = 20e6, 0.5
    view = engine.SpectrumView(ctx)
    f0 = 100e6
    view.set_range(f0, f0 + dwells_all * fs * rel)
    view.set_fft(fs, rel)
    centers = f0 + (mine + 0.5) * fs * rel

    def step(ev=None):
        if ev: ev[0].record()
        psd.feed(x, nframes=dwells * tile, hop=N, navg=tile, scale=1.0 / N, mode=engine.PSD_DB_SHIFTED, out=frames)
        if ev: ev[1].record()
        view.feed_sweep(frames, centers)
        if ev: ev[2].record()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    steps = steps or max(3, min(10, args.steps // 4))
    for _ in range(warm):
        step()
    fence()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    engine.kernel_timing_read()
    engine.kernel_timing(True)
    t0 = time.perf_counter()
    for k in range(steps):
        step(evs[k])
    fence()
    dt = time.perf_counter() - t0
    engine.kernel_timing(False)
    kt = engine.kernel_timing_read("psd_kernel")
    engine.kernel_timing_read()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    psd_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
    view_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / steps
    k_ms = kt["sum_ms"] / kt["launches"] if kt["launches"] else psd_ms
    psd_bytes = 8.0 * share + 4.0 * N * dwells
    del x
    return {"workload": f"C5: panoramic sweep over a {total / 1e9:.2f} GS capture in HBM, {dwells_all} dwells x {tile} x "
                        f"{N}-pt frames (Blackman-Harris, averaged, shift+dB fused) -> SpectrumView"
                        + (f"; dwells sharded d mod {world}, no exchange" if world > 1 else ""),
            "value_MSps": round(total * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "dwells_per_rank": dwells, "capture_samples": total,
            "stage_ms": {"psd": round(psd_ms, 4), "specview": round(view_ms, 4)},
            "psd_roofline": {"kernel": "psd_kernel", "bound": "hbm", "achieved": round(psd_bytes / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(psd_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms": round(k_ms, 4),
                             "algorithmic_bytes_per_launch": psd_bytes, "traffic": None,
                             "timing": "kernel's own duration (dispatch-bound event pairs) on this rank"}}


def run_c1(args):
    """BASELINE.json configs[0]: "CPU reference: suscan file source, 2.4 MS/s synthetic IQ, 8192-pt PSD only, 1 inspector off
    (Averager + FFT plumbing)" -- through the drop-in boundary (the live analyzer, unthrottled) at three operating points, with
    the CPU leg (the oracle's PSD on this box's cores; the reference's own PSDMessage + Averager loop is
    cpu_baseline.reference_loops.psd_message_plus_averager) beside it."""
    from sigdigger_amd.livebench import live_psd_only
    out = {"workload": "C1 (BASELINE.json configs[0]): file source, 2.4 MS/s, 8192-pt PSD only, no inspector, unthrottled"}
    for key, a in (("live_25fps", (2_400_000, 8192, 0.04, 500)),                       # 12 frames per message: the GUI's refresh rate
                   ("live_reference_defaults", (3_000_000, 4096, 0.04, 500)),           # include/AppConfig.h:35-38: 4096 bins, 25 fps
                   ("live_2Mi_blocks", (2_400_000, 8192, (1 << 21) / 2.4e6, 60))):     # 256 frames per message: the stream rate
        try:
            out[key] = live_psd_only(*a)
        except Exception as e:                                # a secondary figure must not take the bench line down
            out[key] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_psd_baseline(8192, 1 << 25, os.cpu_count() or 1)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def run_capacity(args, dev, ctx):
    """How many PSK inspectors (D = 64: C4's kind) one MI355X carries: the default workload with 64 ... 4096 inspectors on one
    GPU -- the stream rate, the slowest stage and the channeliser's time per step for each.  Same pipeline, same kernels, the
    headline's block (64 QPSK carriers); inspector c sits on carrier c mod 64 (independent opens of the same channels, as
    Suscan/Analyzer.cpp:411-432 allows), so every chain locks as in the headline -- a Gardner wavefront's time depends on how
    its lanes' symbol clocks relate (design/recurrences.md), noise-only channels would run it twice as long.  16 Mi samples up to
    1024 inspectors, 4 Mi beyond (the inter-stage rings of 4096 inspectors at 16 Mi samples would take 77 GB)."""
    cfg = WORKLOADS["c4"]
    rows = []
    blocks = {}
    for n, lg in ((64, args.block), (128, args.block), (256, args.block), (512, args.block), (1024, args.block), (2048, 22), (4096, 22)):
        L = 1 << lg
        try:
            if lg not in blocks:
                blocks[lg] = make_block(L, synth.raster(64, cfg["spacing"]), cfg["sps_in"], "psk", dev, seed=4321)
            x = blocks[lg]
            fn = np.tile(synth.raster(64, cfg["spacing"]), n // 64)
            bank = pipeline.InspectorBankConfig(kind="psk", fnor=fn, decimation=cfg["D"], ntaps=cfg["T"], sps=cfg["sps_in"] / cfg["D"],
                                                channeliser="fft")
            pipe = pipeline.AnalyzerPipeline(ctx, L, psd_size=cfg["psd"], psd_navg=min(256, L // cfg["psd"]), bank=bank, do_psd=True)
            pipe.enable_delivery()
            steps = 6 if lg >= 24 else 12
            with torch.cuda.stream(pipe.main_stream()):
                for _ in range(2):
                    pipe.step(x)
                    pipe.deliver()
                pipe.sync()
                pipe.reset_events()
                engine.kernel_timing_read()
                engine.kernel_timing(True)
                t0 = time.perf_counter()
                for _ in range(steps):
                    pipe.step(x, timed=True)
                    pipe.deliver()
                pipe.sync()
                dt = time.perf_counter() - t0
                engine.kernel_timing(False)
            kt = {k: engine.kernel_timing_read(k) for k in CHANNELISER_KERNELS}
            ck = max(kt, key=lambda k: kt[k]["sum_ms"])
            engine.kernel_timing_read()
            st = pipe.stage_times_ms()
            slow = max(st, key=st.get)
            rows.append({"inspectors": n, "block_samples": L, "value_MSps": round(L * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3),
                         "slowest_stage": slow, "stage_ms": {k: round(v, 3) for k, v in st.items()},
                         "channeliser_kernel": ck, "channeliser_ms_per_step": round(kt[ck]["sum_ms"] / steps, 4),
                         "channeliser_launches_per_step": round(kt[ck]["launches"] / steps, 2),
                         "realtime_factor_at_50MSps": round(L * steps / dt / 50e6, 1)})
            del pipe
            torch.cuda.empty_cache()
        except Exception as e:
            rows.append({"inspectors": n, "error": repr(e)[:300]})
            break
    ok = [r for r in rows if "value_MSps" in r]
    best = max((r["inspectors"] for r in ok if r["value_MSps"] >= 50.0), default=None)
    out = {"what": run_capacity.__doc__.split("\n\n")[0].replace("\n    ", " "), "rows": rows,
           "max_inspectors_measured_at_ge_50MSps": best}
    if ok:
        last = ok[-1]
        # beyond the last measured row the step grows in proportion to the inspectors (its slowest stage is throughput-bound
        # by then), so the count at which the rate falls to 50 MS/s follows from that row
        out["extrapolated_inspectors_at_50MSps"] = int(last["inspectors"] * last["value_MSps"] / 50.0)
        out["note"] = (f"every measured count sustains >= 50 MS/s; the last row ({last['inspectors']} inspectors) runs at "
                       f"{last['value_MSps']} MS/s with `{last['slowest_stage']}` as its step, and from there the step grows with the count: "
                       f"~{out['extrapolated_inspectors_at_50MSps']} inspectors at 50 MS/s (extrapolated, not measured)") if best == last["inspectors"] else None
    return out


def multi_gpu_diagnostics(dist, rank, world, local_rank, dev_index, dev, L, share):
    """Before the timed region of an N > 1 run: who is here (one rank per GPU, distinct devices) and what one block costs to
    broadcast.  Fails loudly instead of timing a job that is not the one asked for."""
    props = torch.cuda.get_device_properties(dev_index)
    me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "pid": os.getpid(), "name": props.name,
          "uuid": str(getattr(props, "uuid", "")), "pci": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}"}
    seen = [None] * world
    dist.all_gather_object(seen, me)
    ids = {(g["uuid"], g["pci"]) for g in seen}
    if len(seen) != world or any(g is None for g in seen):
        raise SystemExit(f"bench.py: {sum(g is not None for g in seen)} ranks answered, {world} expected")
    if not share and len(ids) != world:
        raise SystemExit(f"bench.py: {world} ranks on {len(ids)} distinct GPU(s): {sorted(ids)}")
    buf = torch.empty(2 * L, dtype=torch.float32, device=dev)
    buf.normal_()
    dist.broadcast(buf, src=0)                                # (communicator set-up and the first transfer's channel set-up)
    torch.cuda.synchronize(dev)
    dist.barrier()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.broadcast(buf, src=0)
    torch.cuda.synchronize(dev)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    nbytes = buf.numel() * 4
    del buf
    return {"rccl_ranks_seen": len(seen), "backend": dist.get_backend(), "distinct_devices": len(ids), "ranks": seen,
            "bcast_block_bytes": nbytes, "bcast_ms": round(float(dt.item()) / reps * 1e3, 4),
            "bcast_GBps": round(nbytes * reps / float(dt.item()) / 1e9, 2),
            "what": "one IQ block broadcast from rank 0 to every rank (max over ranks), back to back, before the timed region"}


def run_live_sharded(n_gpus, inspectors_per_gpu=64, nblocks=40, timeout_s=150):
    """The drop-in itself on N GPUs: ONE process, the suscan_analyzer_* ABI with SUAMD_DEVICES=0..N-1 (csrc/analyzer.cpp:
    one worker thread per GPU, inspector handle h on GPU h mod N, the block to every shard by ncclBroadcast over xGMI,
    one message queue) and 64 heterogeneous PSK inspectors per GPU.  Run in a child process with a deadline: a secondary
    figure must not be able to hang the bench."""
    import subprocess
    env = dict(os.environ, SUAMD_DEVICES=",".join(str(i) for i in range(n_gpus)))
    if n_gpus > 1:
        env["SUAMD_ANALYZER_BCAST"] = "rccl"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    code = ("import json, sys; sys.path.insert(0, %r); from sigdigger_amd.livebench import live_rate; "
            "print('LIVE ' + json.dumps(live_rate(%d, %d)))" % (ROOT, inspectors_per_gpu * n_gpus, nblocks))
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for ln in r.stdout.splitlines():
            if ln.startswith("LIVE "):
                d = json.loads(ln[5:])
                d["devices"] = env["SUAMD_DEVICES"]
                d["block_exchange"] = "ncclBroadcast (RCCL over xGMI), root GPU 0" if n_gpus > 1 else "none (one GPU)"
                return d
        return {"error": (r.stderr or r.stdout)[-400:]}
    except Exception as e:
        return {"error": repr(e)}


def run_extra(args, dev, ctx, cfg, fn_rank, TIMING):
    """The secondary workloads (`--extra`; detail file only): C2 both ways, C3, the default workload at rounds 1-3's 4 Mi block,
    C1 through the live analyzer, the one-GPU capacity sweep, C5, the 64-heterogeneous-inspector live rate, the FIR stage alone."""
    extra = {}
    for w in ("c2", "c3"):
        if w == args.workload:
            continue
        # C2 as BASELINE.json states it -- "1 PSK inspector, 255-tap LPF": the translate + 255-tap polyphase FIR --
        # AND behind the FFT filter bank (what the live analyzer puts in front of an inspector); C3 on the default
        variants = (("fir", "255-tap LPF (BASELINE.json configs[1])"), ("fft", "FFT filter bank")) if w == "c2" else ((args.channeliser, None),)
        for chn, label in variants:
            a2 = argparse.Namespace(**vars(args))
            a2.steps, a2.warmup, a2.channeliser = max(5, min(20, args.steps // 4)), 3, chn
            c2, L2, dt2, st2, fn2, pipe2 = run_workload(w, a2, 0, 1, dev, ctx, None)
            k2 = getattr(pipe2, "kernel_ms", {})
            ck = next((k for k in CHANNELISER_KERNELS if k in k2), None)
            entry = {"workload": describe(c2, chn), "value_MSps": round(L2 * a2.steps / dt2 / 1e6, 3),
                     "ms_per_step": round(dt2 / a2.steps * 1e3, 4),
                     "stage_ms": {k: round(v, 4) for k, v in st2.items()},
                     "config": {"channeliser": chn, "taps": c2["T"] if chn == "fir" else None,
                                "channel_bins": 4096 // c2["D"] if chn == "fft" else None, "decimation": c2["D"],
                                "inspectors": len(fn2), "block_samples": L2, "psd_size": c2["psd"]}}
            if ck:
                entry["roofline"] = fir_stage_roofline(len(fn2), c2["D"], L2, k2[ck]["per_step"], ck, w, TIMING + " (in the pipeline)")
                entry["roofline"]["kernel_launches_ms"] = {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in k2.items()}
                if "psd_kernel" in k2:
                    pms = sum(k2[k]["per_step"] for k in ("psd_kernel", "psd_reduce_kernel") if k in k2)
                    pb = 8.0 * L2 + 4.0 * c2["psd"] * (L2 // c2["psd"] // pipe2.navg)
                    entry["psd_roofline"] = {"kernel_ms": round(pms, 4), "frac": round(pb / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": pb,
                                             "psd_size": c2["psd"]}
                try:                                  # the same kernel alone: this block size, and a 16 Mi-sample block
                    lgs = sorted({int(np.log2(L2)), 22, 24})
                    entry["roofline"]["fir_stage_alone"] = {}
                    for lg in lgs:
                        msa, ka = channeliser_alone(ctx, dev, fn2, c2["D"], c2["T"], chn, lg)
                        if msa:
                            entry["roofline"]["fir_stage_alone"][f"{(1 << lg) >> 20}Mi"] = fir_stage_roofline(
                                len(fn2), c2["D"], 1 << lg, msa, ka, w, TIMING + " (the kernel alone, re-feeding one resident block)")
                except Exception as e:
                    entry["roofline"]["fir_stage_alone"] = {"error": repr(e)}
            if not args.no_cpu_baseline:              # the same workload on this box's host cores (bounded sample)
                try:
                    entry["cpu_baseline"] = cpu_baseline(c2, (1 << 23) if w == "c2" else (1 << 25), fn2, os.cpu_count() or 1, chn)
                except Exception as e:
                    entry["cpu_baseline"] = {"error": repr(e)}
            if label:
                extra.setdefault(w, {})[chn] = dict(entry, variant=label)
            else:
                extra[w] = entry
            del pipe2
    if args.block != 22 and args.workload == "c4":
        # rounds 1-3 benched 4 Mi-sample blocks: the same workload at that block, so that the rounds stay comparable
        try:
            a4 = argparse.Namespace(**vars(args))
            a4.block, a4.steps, a4.warmup = 22, max(20, min(80, args.steps)), 5
            c4_, L4, dt4, st4, fn4, pipe4 = run_workload("c4", a4, 0, 1, dev, ctx, None)
            k4 = getattr(pipe4, "kernel_ms", {})
            ck4 = next((k for k in CHANNELISER_KERNELS if k in k4), None)
            extra["c4_block_4Mi"] = {"block_samples": L4, "value_MSps": round(L4 * a4.steps / dt4 / 1e6, 3), "ms_per_step": round(dt4 / a4.steps * 1e3, 4),
                                     "stage_ms": {k: round(v, 4) for k, v in st4.items()},
                                     "roofline": fir_stage_roofline(len(fn4), c4_["D"], L4, k4[ck4]["per_step"], ck4, "c4", TIMING + " (in the pipeline)") if ck4 else None}
            del pipe4
        except Exception as e:
            extra["c4_block_4Mi"] = {"error": repr(e)}
    extra["c1"] = run_c1(args)
    extra["capacity"] = run_capacity(args, dev, ctx)
    extra["c5"] = run_c5(args, dev, ctx)
    try:                                              # the drop-in boundary itself, end to end (host thread, file source)
        from sigdigger_amd.livebench import live_rate
        extra["live64"] = live_rate(64, 60)
    except Exception as e:                            # a secondary figure must not take the bench line down
        extra["live64"] = {"error": repr(e)}
    if args.channeliser == "fft" and cfg["kind"] == "psk":
        try:
            extra["fir_stage_alone"] = {f"{(1 << lg) >> 20}Mi": run_fir_stage_alone(cfg, dev, ctx, fn_rank, lg) for lg in sorted({args.block, 22, 24})}
        except Exception as e:
            extra["fir_stage_alone"] = {"error": repr(e)}
    return extra


def run_live_roofline(dev, log2_blocks=(21, 22), nblocks=24):
    """The roofline of the FIR stage and of the PSD ON THE BOUNDARY: BASELINE.json configs[3]'s per-GPU slice -- 8192-pt PSD + 64
    QPSK inspectors (Costas + Gardner) at 50 MS/s, decimation 64 -- through the suscan_analyzer_* C ABI (csrc/analyzer.cpp's own
    worker thread, file source looping over a capture of staggered carriers), at the blocks a GUI lives with (25 fps x 50 MS/s =
    2 Mi samples, include/AppConfig.h:35-36) and twice that.  Kernel durations: the library's dispatch-bound event pairs
    (suamd_kernel_timing is process-global, so the worker thread's launches are timed like the harness's)."""
    from sigdigger_amd.livebench import live_rate
    cfg = WORKLOADS["c4"]
    fs = 50e6
    spacing = 700e3
    out = {}
    for lg in log2_blocks:
        L = 1 << lg
        try:
            fn = (np.arange(64) - 32 + 0.5) * spacing * 2.0 / fs
            cap = make_block(4 * L, fn, cfg["sps_in"], "psk", dev, seed=2468).cpu().numpy()
            d = live_rate(64, nblocks, fs=int(fs), nfft=8192, block=L, uniform=dict(spacing=spacing, bw=300e3, baud=fs / cfg["sps_in"], costas_order=2),
                          ktimer=True, capture=cap)
            del cap
            if "error" in d:
                out[f"{L >> 20}Mi"] = d
                continue
            kern = d.get("kernels") or {}
            ck = max((k for k in kern if k in CHANNELISER_KERNELS or k == "chan_fir_gang_kernel"), key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"], default=None)
            e = {"block_samples": L, "value_MSps": d["value_MSps"], "worker_MSps": d["worker_MSps"], "ms_per_block": d["ms_per_block"]}
            if ck:
                per_block = kern[ck]["avg_ms"] * kern[ck]["launches"] / nblocks
                nbytes = 8.0 * L + 8.0 * 64 * (L // 64)
                e.update({"kernel": ck, "kernel_ms": round(per_block, 5), "launches_per_block": round(kern[ck]["launches"] / nblocks, 2),
                          "algorithmic_bytes_per_launch": nbytes, "achieved": round(nbytes / (per_block * 1e-3) / 1e9, 1),
                          "frac": round(nbytes / (per_block * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            if "psd_kernel" in kern:
                pms = sum(kern[k]["avg_ms"] * kern[k]["launches"] for k in ("psd_kernel", "psd_reduce_kernel") if k in kern) / nblocks
                pb = 8.0 * L + 4.0 * 8192
                e["psd"] = {"kernel_ms": round(pms, 5), "frac": round(pb / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": pb}
            out[f"{L >> 20}Mi"] = e
        except Exception as ex:                               # a secondary figure must not take the bench line down
            out[f"{L >> 20}Mi"] = {"error": repr(ex)[:200]}
    return out


def _r(v, n=4):
    return round(v, n) if isinstance(v, float) else v


LINE_LIMIT = 4096


def compact_line(out, limit=LINE_LIMIT):
    """The ONE line the driver parses (the last line that starts with "{"; round 5's 23 KB line was not parsed): under
    `limit` bytes whatever the run added.  A line that would be longer loses its secondary objects one at a time, least
    important first (they are all in the detail file) -- never the contract fields, `roofline`'s core or `cpu_baseline`."""
    out = json.loads(json.dumps(out))                         # (a deep copy: the detail file keeps the long form)
    drop = [("multi_gpu", "ranks"), ("other_workloads",), ("live_sharded_analyzer",), ("roofline", "live_analyzer"), ("roofline", "psd"),
            ("multi_gpu",), ("roofline", "traffic_source"), ("roofline", "timing"), ("stage_ms",), ("cpu_baseline", "sample"),
            ("config", "symbol_clocks"), ("config", "schedule"), ("roofline", "fp32_vector")]
    line = json.dumps(out, separators=(",", ":"))
    dropped = []
    for path in drop:
        if len(line) < limit:
            break
        o = out
        for k in path[:-1]:
            o = o.get(k) if isinstance(o, dict) else None
        if isinstance(o, dict) and path[-1] in o:
            del o[path[-1]]
            dropped.append(".".join(path))
            out["dropped_for_length"] = dropped
            line = json.dumps(out, separators=(",", ":"))
    if len(line) >= limit:                                    # (cannot happen with the fields above gone: long strings cut as a last resort)
        def cut(o):
            return {k: cut(v) for k, v in o.items()} if isinstance(o, dict) else (o[:60] if isinstance(o, str) else o)
        line = json.dumps(cut(out), separators=(",", ":"))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 150 steps (3 s timed): the three serial stages run as a pipeline over consecutive blocks, and the timed region pays
    # its fill and drain once (~1.3 steps); and the chip needs seconds of load to settle its clocks (4 Mi blocks: the
    # slowest kernel alone goes 5.30 -> 5.19 -> 5.16 -> 5.14 ms per block over 0.5 / 1.5 / 3 / 5 s) -- a sustained-rate
    # metric wants both amortised
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS) + ["c5"])
    ap.add_argument("--block", type=int, default=DEFAULT_LOG2_BLOCK, help="log2 of the IQ block length (samples)")
    ap.add_argument("--channeliser", default="fft", choices=("fft", "fir"),
                    help="fft: the FFT filter bank with su_specttuner's semantics (what the reference runs behind its "
                         "channels); fir: translate + 255-tap direct-form low-pass + decimate")
    ap.add_argument("--aligned", action="store_true", help="rounds 1-5's input: every carrier's symbol boundaries at the same samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", action="store_true",
                    help="also run the secondary workloads (c1, c2, c3, c5, capacity sweep, host-fed, FIR stage alone): their results go "
                         "to the detail file only")
    ap.add_argument("--no-extra", action="store_true", help="(the default since round 6; accepted for old command lines)")
    ap.add_argument("--no-pmc", action="store_true", help="do not count HBM traffic in child runs under rocprofv3 --pmc")
    ap.add_argument("--no-live", action="store_true", help="skip the live-analyzer roofline leg (the C++ boundary at 2 Mi / 4 Mi blocks)")
    ap.add_argument("--lean", action="store_true", help="the headline workload only: no PMC children, live leg, aligned run, CPU baseline")
    ap.add_argument("--detail", default=os.environ.get("SUAMD_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where the long form goes (everything that is not in the one compact line)")
    # 32 Mi samples: ~2.5 s of one core + ~0.3 s of all of them on the box's EPYC (the bounded sample of the CPU leg)
    ap.add_argument("--cpu-samples", type=int, default=1 << 25)
    ap.add_argument("--live", action="store_true",
                    help="time the sharded live analyzer instead (one process, SUAMD_DEVICES=0..N-1; run WITHOUT torchrun): "
                         "the drop-in boundary itself on N GPUs")
    ap.add_argument("--isolated", action="store_true",
                    help="after the timed region also time the FIR and PSD kernels alone on an idle GPU")
    args = ap.parse_args()
    if args.lean:
        args.no_pmc = args.no_live = args.no_cpu_baseline = True

    if args.live:
        d = run_live_sharded(args.gpus, nblocks=max(20, min(200, args.steps)))
        val = d.get("value_MSps")
        print(json.dumps({"metric": "MS/s complex IQ sustained (PSD + N inspectors)", "value": val, "unit": "MS/s", "n_gpus": args.gpus,
                          "steps": d.get("blocks"), "warmup": 0, "ms_per_step": d.get("ms_per_block"), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": d.get("workload"), "parallelism": f"live analyzer, inspectors sharded over {args.gpus} GPU(s) "
                                     f"behind one suscan_analyzer handle (SUAMD_DEVICES)", "inspectors_total": d.get("inspectors"),
                                     "block_exchange": d.get("block_exchange")},
                          "live": d}), flush=True)
        return
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ       # under torch.distributed.run
    # test hook: SUAMD_BENCH_SHARE_GPU=1 puts every rank on GPU 0 with the gloo backend, so that the
    # N > 1 control flow (sharding, block broadcast, barriers, max-over-ranks) can be exercised on a
    # one-GPU box; RCCL itself refuses two ranks on one device
    share = os.environ.get("SUAMD_BENCH_SHARE_GPU") == "1"
    if args.gpus > 1 and not launched:
        # a bare `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU under torch.distributed.run,
        # exactly the command the contract names) instead of timing one GPU and calling it N
        ndev = torch.cuda.device_count()
        if not share and ndev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not share and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    dist = None
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)
    ctx = engine.Context(dev_index)
    mg = None
    if dist is not None:
        try:
            mg = multi_gpu_diagnostics(dist, rank, world, local_rank, dev_index, dev, 1 << args.block, share)
        except SystemExit:
            raise                                             # fewer ranks / GPUs than asked for: no line at all
        except Exception as e:                                # a diagnostic that fails must not take the timed run with it
            mg = {"error": repr(e)}

    t_start = time.perf_counter()
    if args.workload == "c5":
        # BASELINE.json configs[4], sharded by frame (dwell d on rank d mod N; no exchange): a line of its own
        d = run_c5(args, dev, ctx, rank=rank, world=world, dist=dist, steps=args.steps, warm=max(1, args.warmup))
        if rank == 0:
            out = {"metric": "MS/s complex IQ sustained (PSD + N inspectors)", "value": d["value_MSps"], "unit": "MS/s", "n_gpus": world,
                   "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": d["ms_per_step"], "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": d["workload"], "psd_size": 8192, "tile_frames": 256, "capture_samples": d["capture_samples"],
                              "dwells_per_rank": d["dwells_per_rank"], "parallelism": f"dwells sharded d mod {world}, no exchange" if world > 1 else "single GPU"},
                   "stage_ms": d["stage_ms"], "roofline": d["psd_roofline"]}
            if mg is not None and "error" not in mg:
                out["multi_gpu"] = {k: mg[k] for k in ("rccl_ranks_seen", "backend", "distinct_devices")}
            print(json.dumps(out, separators=(",", ":")), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    cfg, L, dt, stages, fn_rank, pipe = run_workload(args.workload, args, rank, world, dev, ctx, dist, stagger=not args.aligned)

    if rank == 0:
        K = args.steps
        # `value` is the rate of the IQ STREAM: every rank consumes the same broadcast block and runs its own shard of
        # the inspectors on it, so N GPUs carry N x the inspectors at (ideally) the same stream rate -- weak scaling in
        # inspectors.  The aggregate channel rate (stream rate x inspectors) is reported beside it.
        value = float(L) * K / dt / 1e6
        C, D, T = len(fn_rank), cfg["D"], cfg["T"]
        m_out = L // D
        fft_bank = args.channeliser == "fft"
        # the north star's "FIR stage": the channeliser.  Algorithmic (compulsory) bytes per launch:
        # the shared input once + every channel's decimated output
        fir_bytes = 8.0 * L + 8.0 * C * m_out
        fir_flops_built = float(C) * m_out * T * 8.0 + 14.0 * C * m_out      # direct form as built: 4 fma / tap + de-rotation
        fir_flops_survey = float(C) * L * (6.0 + 4.0 * T / D)                # translate + real taps
        # per-launch durations: the dispatch-bound event pairs (pipe.kernel_ms) where the library provides them; the
        # stream-event pairs around each stage (stage_ms: they include the queue's gaps) are reported beside them
        kms = getattr(pipe, "kernel_ms", {})
        chan_k = next((k for k in CHANNELISER_KERNELS if k in kms), None)
        fir_ms = kms[chan_k]["per_step"] if chan_k else stages.get("fir")
        psd_ms = (sum(kms[k]["per_step"] for k in ("psd_kernel", "psd_reduce_kernel") if k in kms)
                  if "psd_kernel" in kms else stages.get("psd"))
        psd_bytes = 8.0 * L + 4.0 * cfg["psd"] * (L // cfg["psd"] // pipe.navg)
        kname = chan_k or ("stp_kernel" if fft_bank else "chan_fir_kernel")
        TIMING = ("dispatch-bound event pairs (hipExtLaunchKernelGGL start/stop events through suamd_kernel_timing): the "
                  "kernel's own duration, averaged over every launch of the timed region")
        detail = {"argv": sys.argv[1:], "kernel_launches_ms": {k: {kk: _r(vv, 5) for kk, vv in v.items()} for k, v in kms.items()},
                  "stage_ms": {k: round(v, 4) for k, v in stages.items()},
                  "stalled_samples_dropped": dict(getattr(pipe, "stalled_samples", {})),
                  "stage_ms_unfiltered": {k: {kk: _r(vv) for kk, vv in v.items()} for k, v in getattr(pipe, "stage_raw", {}).items()},
                  "timing": TIMING}

        # ---- the legs beside the headline (N = 1 only) ----
        aligned = None
        if world == 1 and not args.lean and not args.aligned:
            # rounds 1-5's input (every symbol clock aligned): the lock-step clock kernel's best case, beside the headline
            try:
                a2 = argparse.Namespace(**vars(args))
                a2.steps, a2.warmup = max(4, min(10, args.steps)), 3
                _, L2, dt2, st2, _, pipe2 = run_workload(args.workload, a2, 0, 1, dev, ctx, None, stagger=False)
                aligned = {"value_MSps": round(L2 * a2.steps / dt2 / 1e6, 1), "steps": a2.steps, "stage_ms": {k: round(v, 3) for k, v in st2.items()}}
                del pipe2
            except Exception as e:
                aligned = {"error": repr(e)[:200]}
        live = None
        if world == 1 and not args.no_live:
            live = run_live_roofline(dev)
            detail["live_roofline"] = live
        # HBM traffic of the two transform kernels, counted by this run's own PMC passes (child runs of this command line;
        # they run while this process times the CPU leg -- its GPU is idle then)
        pmc = {}
        pmc_thread = None
        if world == 1 and not args.no_pmc:
            import threading
            box = {}
            def children():
                box.update(r=pmc_traffic_in_run(args, (kname, "psd_kernel", "psd_reduce_kernel")))
                if live:                                      # (the live leg's kernels by the profiler's clock: see live_kernel_trace_in_run)
                    box.update(live=live_kernel_trace_in_run())
            pmc_thread = threading.Thread(target=children)
            pmc_thread.start()
        cpu = None
        if not args.no_cpu_baseline and world == 1:         # reported at N = 1 only (rank 0's host cores)
            try:
                cpu = cpu_baseline(cfg, args.cpu_samples, fn_rank, os.cpu_count() or 1, args.channeliser)
                detail["cpu_baseline"] = dict(cpu)
                ref = reference_loops()
                if ref is not None:
                    detail["cpu_baseline"]["reference_loops"] = ref
            except Exception as e:                            # (the line must come out even if the CPU leg cannot run)
                cpu = None
                detail["cpu_baseline"] = {"error": repr(e)[:300]}
                print(f"# bench.py: cpu_baseline failed: {e!r}", file=sys.stderr)
        if pmc_thread is not None:
            pmc_thread.join()
            pmc = box.get("r") or {}
            if live and box.get("live"):
                detail["live_roofline_rocprofv3"] = box["live"]
                for k, v in box["live"].items():
                    if k in live and "error" not in live[k]:
                        live[k]["rocprofv3"] = v
        detail["pmc_counters"] = pmc or None
        src_now = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (4 steps), 2 x FETCH + WRITE"
        roof = fir_stage_roofline(C, D, L, fir_ms, kname, args.workload, TIMING if chan_k else "stream events around the stage",
                                  traffic=(pmc[kname]["hbm_bytes_per_launch"], src_now) if kname in pmc else None)
        detail["roofline_long"] = dict(roof, kernel_description=KERNELS.get(kname, kname))
        psd_tr = (pmc["psd_kernel"]["hbm_bytes_per_launch"] + pmc.get("psd_reduce_kernel", {}).get("hbm_bytes_per_launch", 0)) if "psd_kernel" in pmc else None
        roof_c = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "kernel_ms", "algorithmic_bytes_per_launch",
                                        "traffic", "traffic_over_algorithmic")}
        roof_c["traffic_source"] = (roof.get("traffic_source") or "")[:110] or None
        roof_c["timing"] = "kernel's own duration (dispatch-bound event pairs), mean over every launch of the timed region"
        if chan_k:
            roof_c["kernel_ms_min_max"] = [round(kms[chan_k]["min"], 4), round(kms[chan_k]["max"], 4)]
        roof_c["psd"] = {"kernel": "psd_kernel" + ("+psd_reduce_kernel" if "psd_reduce_kernel" in kms else ""),
                         "frac": round(psd_bytes / (psd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if psd_ms else None,
                         "kernel_ms": round(psd_ms, 4) if psd_ms else None, "algorithmic_bytes_per_launch": psd_bytes,
                         "traffic_over_algorithmic": round(psd_tr / psd_bytes, 4) if psd_tr else None}
        if live:
            # the same two kernels inside the C++ live analyzer (the drop-in boundary), at a GUI's block lengths
            # (frac / kernel_ms: rocprofv3's kernel-trace of a child run of the same leg when the box has the profiler -- `timing`
            # says which --; frac_event_pairs: the dispatch-bound event pairs, which inside the analyzer's four busy streams also
            # count the dispatch's wait for compute units)
            def live_entry(v):
                if "error" in v:
                    return {"error": v["error"][:80]}
                e = {kk: v.get(kk) for kk in ("kernel", "frac", "kernel_ms", "value_MSps")}
                if "psd" in v:
                    e["psd_frac"] = v["psd"]["frac"]
                e["timing"] = "event pairs"
                rp = v.get("rocprofv3")
                if rp:
                    e.update({"frac_event_pairs": e["frac"], "frac": rp["frac"], "kernel_ms": round(rp["kernel_us"] / 1e3, 5), "timing": "rocprofv3 child"})
                    if "psd_frac" in rp:
                        e["psd_frac"] = rp["psd_frac"]
                return e
            roof_c["live_analyzer"] = {k: live_entry(v) for k, v in live.items()}
        if not fft_bank and fir_ms:
            roof_c["fp32_vector"] = {"peak_tflops": FP32_PEAK_TFLOPS, "frac_as_built": round(fir_flops_built / (fir_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                     "frac_survey_8d": round(fir_flops_survey / (fir_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}
        out = {
            "metric": "MS/s complex IQ sustained (PSD + N inspectors)", "value": round(value, 3), "unit": "MS/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "block_log2": args.block,
            "config": {"workload": describe(cfg, args.channeliser), "block_samples": L, "psd_size": cfg["psd"],
                       "inspectors_per_gpu": cfg["per_gpu"], "inspectors_total": cfg["per_gpu"] * world,
                       "decimation": D, "taps": None if fft_bank else T,
                       "channel_bins": 4096 // D if fft_bank else None,
                       "parallelism": f"channel-sharded x{world}, RCCL broadcast of the IQ block" if world > 1
                       else "single GPU",
                       "channeliser": args.channeliser,
                       "schedule": "transform window" if pipe.window else "free-running streams",
                       "symbol_clocks": "aligned (rounds 1-5's input)" if args.aligned else "staggered: per-carrier timing offset, +-100 ppm baud"},
            "aggregate_inspector_MSps": round(value * cfg["per_gpu"] * world, 1),
            "stage_ms": {k: round(v, 3) for k, v in stages.items()},
            "roofline": roof_c,
        }
        if aligned is not None:
            out["value_aligned_clocks"] = aligned.get("value_MSps")
            detail["aligned_clocks"] = aligned
        if args.isolated:
            # the same launches on an otherwise idle chip (in the pipeline they share the GPU with the
            # AGC / Costas / Gardner kernels of neighbouring blocks)
            iso = {}
            x = torch.randn(L, dtype=torch.complex64, device=dev)
            for name, fn in (("fir", lambda: pipe._channelise(x, pipe.y[0], torch.cuda.current_stream(dev))),
                             ("psd", (lambda: pipe.psd.feed(x, nframes=pipe.nframes, navg=pipe.navg,
                                                            scale=1.0 / pipe.psd_size, out=pipe.psd_out))
                              if pipe.do_psd else None)):
                if fn is None:
                    continue
                fn()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize(dev)
                iso[name + "_ms"] = round(e0.elapsed_time(e1) / 10, 4)
            iso["fir_hbm_frac"] = round(fir_bytes / (iso["fir_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if "psd_ms" in iso:
                iso["psd_hbm_frac"] = round(psd_bytes / (iso["psd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            detail["isolated"] = iso
        if world == 1 and args.extra:
            detail["other_workloads"] = run_extra(args, dev, ctx, cfg, fn_rank, TIMING)
            detail["host_fed"] = run_host_fed(args.workload, args, dev, ctx)
            ow = detail["other_workloads"]
            out["other_workloads"] = {k: (v.get("value_MSps") if "value_MSps" in v else {kk: vv.get("value_MSps") for kk, vv in v.items() if isinstance(vv, dict) and "value_MSps" in vv})
                                      for k, v in ow.items() if isinstance(v, dict) and k in ("c2", "c3", "c5", "c4_block_4Mi")}
        if mg is not None:
            detail["multi_gpu"] = mg
            out["multi_gpu"] = mg if "error" in mg else {
                "rccl_ranks_seen": mg["rccl_ranks_seen"], "backend": mg["backend"], "distinct_devices": mg["distinct_devices"],
                "bcast_block_bytes": mg["bcast_block_bytes"], "bcast_ms": mg["bcast_ms"], "bcast_GBps": mg["bcast_GBps"],
                "ranks": [{"rank": g["rank"], "device": g["device"], "pci": g["pci"]} for g in mg["ranks"]]}
        if world > 1 and os.environ.get("SUAMD_BENCH_LIVE_SHARDED", "1") != "0" and not share:
            # the curve of the drop-in itself: the C++ analyzer sharded over the same N GPUs (the other ranks idle at the
            # barrier below; their GPUs are free)
            ls = run_live_sharded(world)
            detail["live_sharded_analyzer"] = ls
            out["live_sharded_analyzer"] = {k: ls.get(k) for k in ("value_MSps", "worker_MSps", "inspectors", "devices", "block_exchange", "error") if k in ls}
        if cpu is not None:
            out["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                   "value_1thread": cpu["value_1thread"], "cpu_model": cpu["cpu_model"],
                                   "sample": f"{args.cpu_samples} complex samples of the same pipeline (PSD + {'FFT filter bank' if fft_bank else 'FIR bank'} + "
                                             f"{C} chains) through oracle/sdo.c -O3 -march=native (a restatement, not upstream sigutils)"}
        detail["notes"] = {
            "workload": describe_long(cfg, args.channeliser), "kernel": KERNELS.get(kname, kname),
            "value_definition": "rate of the IQ stream: every rank consumes the same broadcast block and runs its shard of the "
                                "inspectors on it; symbols are copied to pinned host memory inside the timed region.  (Round 1 "
                                "multiplied by the GPU count; since round 2 it does not -- N GPUs carry N x the inspectors at the "
                                "same stream rate; aggregate_inspector_MSps is the product.)",
            "schedule": "transform window (round 5): once per block the three recurrence streams pause at a skewed cut (AGC done, "
                        "Costas one sub-range from its end, Gardner two) and the block's PSD + channeliser run on the idle chip, on "
                        "the slowest stage's stream; SUAMD_PIPELINE_WINDOW=0 restores round 4's free-running streams" if pipe.window else
                        "free-running streams (round 4)"}
        out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
        try:
            with open(args.detail, "w") as f:
                json.dump(dict(out, detail=detail), f, indent=1)
            out["detail_file"] = os.path.relpath(args.detail, ROOT)
        except OSError as e:
            out["detail_file"] = None
            print(f"# bench.py: detail file not written: {e}", file=sys.stderr)
        assert out["n_gpus"] == args.gpus == world, (out["n_gpus"], args.gpus, world)
        print(compact_line(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
