"""bench.py contract: one JSON line with the required keys at N = 1, and the N > 1 control flow (torchrun launch,
channel sharding, block broadcast, barriers, max over ranks, rank-0 print) exercised with two ranks sharing the
one GPU of the test box over gloo (SUAMD_BENCH_SHARE_GPU=1 -- RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--no-pmc", "--no-live", "--cpu-samples", "65536"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "MS/s" and d["value"] > 50.0 and d["dtype"] == "f32" and d["data"] == "synthetic"
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert d["config"]["channeliser"] == "fft" and roof["kernel"].startswith("stp_kernel") and roof["frac"] > 0.08
    assert abs(d["value"] - d["config"]["block_samples"] * 4 / (d["ms_per_step"] * 4e-3) / 1e6) < 0.01 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_the_drivers_exact_command_prints_one_compact_parseable_line(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the command the driver runs at round end, nothing added: exactly
    one line starting with "{", under 4 KB (round 5's 23 KB line was not parsed: BENCH_r05.json parsed == null), carrying the
    contract's keys plus `roofline` (with PMC traffic counted in the run and the live analyzer's own figures) and `cpu_baseline`;
    the long form goes to the detail file the line names."""
    env = dict(os.environ, SUAMD_BENCH_DETAIL=str(tmp_path / "detail.json"))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0]), r.stdout[-1000:]
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    roof = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "traffic_over_algorithmic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 0.01 * roof["achieved"]
    # the kernel's duration fits the step it is part of, and the step fits the run
    assert roof["kernel_ms"] < d["ms_per_step"] and d["ms_per_step"] * 20e-3 < d["bench_wall_s"]
    assert "psd" in roof and roof["psd"]["frac"] > 0.05
    live = roof["live_analyzer"]                                 # the same kernels inside the C++ analyzer, at a GUI's blocks
    assert set(live) == {"2Mi", "4Mi"} and all("error" not in v and v["frac"] > 0.02 and v["value_MSps"] > 50.0 for v in live.values()), live
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["value_1thread"] > 0 and "sample" in cb
    assert d["config"]["symbol_clocks"].startswith("staggered") and d["value_aligned_clocks"] > 50.0 and d["value"] > 50.0
    det = json.load(open(tmp_path / "detail.json"))
    assert det["value"] == d["value"] and "kernel_launches_ms" in det["detail"] and "notes" in det["detail"]
    import shutil
    if shutil.which("rocprofv3"):
        assert roof["traffic"] and 0.9 < roof["traffic_over_algorithmic"] < 2.0, roof


def test_c5_shards_by_frame_over_two_ranks_without_an_exchange():
    """BASELINE.json configs[4] as a workload of its own: dwell d on rank d mod N (Panoramic/Scanner.cpp:503-523: independent
    frames), each rank's share of the capture resident in its own HBM, nothing exchanged; two ranks sharing the test GPU."""
    env = dict(os.environ, SUAMD_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, "bench.py", "--workload", "c5", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    two = subprocess.run([sys.executable, "bench.py", "--workload", "c5", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr + two.stderr)[-3000:]
    d1, d2 = _last_json(one.stdout), _last_json(two.stdout)
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d1["scaling"] == d2["scaling"] == "strong"
    assert d1["config"]["dwells_per_rank"] == 2 * d2["config"]["dwells_per_rank"] == 512
    assert d1["config"]["capture_samples"] == d2["config"]["capture_samples"] == 1 << 30
    for d in (d1, d2):
        assert d["value"] > 1000.0 and d["roofline"]["bound"] == "hbm" and 0.05 < d["roofline"]["frac"] < 1.0


def test_two_ranks_share_the_gpu_over_gloo():
    env = dict(os.environ, SUAMD_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["inspectors_total"] == 2 * d["config"]["inspectors_per_gpu"]
    assert "cpu_baseline" not in d and "other_workloads" not in d                  # N = 1 only
    # the run says who took part and what a block costs to broadcast, before it times anything
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 2 and [g["rank"] for g in mg["ranks"]] == [0, 1] and mg["bcast_GBps"] > 0
    assert mg["bcast_block_bytes"] == 8 * d["config"]["block_samples"]
    # `value` is the rate of the IQ stream: both ranks consume the same broadcast block (each runs its own shard of the
    # inspectors on it); the aggregate channel rate is reported beside it
    assert abs(d["value"] - d["config"]["block_samples"] / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    assert abs(d["aggregate_inspector_MSps"] - d["value"] * d["config"]["inspectors_total"]) < 0.01 * d["aggregate_inspector_MSps"]


def test_bare_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run around it must not time one GPU and print n_gpus 1
    (VERDICT r3 #2): bench.py re-executes itself under torch.distributed.run with one rank per GPU.  Here both ranks share
    the test box's GPU over gloo; on a real node the same path initialises RCCL with N ranks."""
    env = dict(os.environ, SUAMD_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["inspectors_total"] == 2 * d["config"]["inspectors_per_gpu"]
    assert "RCCL broadcast" in d["config"]["parallelism"]
    # without the test hook a one-GPU box must refuse, loudly, rather than time one GPU
    env.pop("SUAMD_BENCH_SHARE_GPU")
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_n1_line_of_the_self_launching_bench_reproduces_the_drivers():
    """--gpus 1 takes no launcher: same steps and warm-up give the same kind of line whether RANK / WORLD_SIZE are set by
    torch.distributed.run (world 1) or absent (the driver's N = 1 command)."""
    args = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    a = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port())] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr + b.stderr)[-3000:]
    da, db = _last_json(a.stdout), _last_json(b.stdout)
    assert da["n_gpus"] == db["n_gpus"] == 1 and da["steps"] == db["steps"] == 20
    assert abs(da["value"] - db["value"]) < 0.1 * da["value"], (da["value"], db["value"])


def test_live_mode_times_the_sharded_analyzer_itself():
    """`bench.py --live --gpus N` (no torchrun): one process, the suscan_analyzer_* ABI with SUAMD_DEVICES = 0..N-1 -- the
    C++ sharded analyzer that IS the drop-in (csrc/analyzer.cpp), not the Python harness.  N = 1 on the test box."""
    r = subprocess.run([sys.executable, "bench.py", "--live", "--gpus", "1", "--steps", "20"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["unit"] == "MS/s" and d["live"]["devices"] == "0" and "error" not in d["live"]
    assert d["value"] > 50.0 and d["config"]["inspectors_total"] == 64


def test_kernel_timer_reports_the_kernels_own_duration():
    """suamd_kernel_timing: dispatch-bound event pairs around the channeliser / PSD launches -- what bench.py's roofline
    leg divides the algorithmic bytes by.  The pair's time is never longer than two stream events around the launch."""
    import numpy as np
    import torch
    from sigdigger_amd import engine
    ctx = engine.Context(0)
    L = 1 << 20
    x = torch.empty(L, dtype=torch.complex64, device="cuda")
    torch.view_as_real(x).normal_()
    st = engine.SpectTuner(ctx, 4096)
    for k in range(8):
        st.open_channel(0.3 + 0.5 * k, 2 * np.pi * 0.75 / 64)
    out = engine.time_major(8, L // 64 + 64, "cuda")
    psd = engine.PSD(ctx, 8192)
    st.feed(x, out=out)
    po = psd.feed(x, nframes=L // 8192, navg=16)
    torch.cuda.synchronize()
    assert engine.kernel_timing_read()["launches"] == 0          # off: nothing is recorded
    engine.kernel_timing(True)
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            st.feed(x, out=out)
        e1.record()
        for _ in range(3):
            psd.feed(x, nframes=L // 8192, navg=16, out=po)
        torch.cuda.synchronize()
    finally:
        engine.kernel_timing(False)
    r = engine.kernel_timing_read("stw_kernel")              # 8 channels of 64 bins, runs of one window on a 1 Mi block: the one-wavefront kernel
    assert r["launches"] == 5
    assert 1e-3 < r["min_ms"] <= r["sum_ms"] / 5 <= r["max_ms"] < 1.0
    assert r["sum_ms"] <= e0.elapsed_time(e1) * 1.02
    p = engine.kernel_timing_read("psd_kernel")
    assert p["launches"] == 3 and p["min_ms"] > 1e-3
    assert engine.kernel_timing_read()["launches"] in (0, 3)     # the rest: psd_reduce_kernel if the plan split the frames
    assert engine.kernel_timing_read()["launches"] == 0
    st.close()
