"""bench.py's final line stays under the length the driver parses whatever a run adds to it (round 5's 23 KB line was not
parsed; an assert that fires instead of printing would lose the line just the same)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _line():
    return {"metric": "MS/s complex IQ sustained (PSD + N inspectors)", "value": 801.4, "unit": "MS/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 20.9, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4 slice", "schedule": "transform window", "symbol_clocks": "staggered"},
            "stage_ms": {"psd": 0.08, "fir": 0.1, "agc": 13.4, "costas": 20.2, "clock": 15.1},
            "roofline": {"kernel": "stp_kernel", "bound": "hbm", "achieved": 3343.7, "peak": 8000.0, "unit": "GB/s", "frac": 0.418, "traffic": 320144761,
                         "traffic_source": "x" * 110, "timing": "y" * 100, "psd": {"frac": 0.35}, "live_analyzer": {"2Mi": {"frac": 0.16}, "4Mi": {"frac": 0.22}}},
            "cpu_baseline": {"value": 174.4, "unit": "MS/s", "cores": 256, "kind": "port", "sample": "z" * 200}}


def test_a_short_line_is_printed_as_it_is():
    import bench
    o = _line()
    assert bench.compact_line(o) == json.dumps(o, separators=(",", ":"))


def test_a_long_line_loses_secondary_objects_not_the_contract():
    import bench
    o = _line()
    o["multi_gpu"] = {"rccl_ranks_seen": 8, "backend": "nccl", "distinct_devices": 8,
                      "ranks": [{"rank": i, "device": i, "pci": "0000:%02x:00" % i, "name": "n" * 400} for i in range(8)]}
    o["live_sharded_analyzer"] = {"error": "e" * 3000}
    o["other_workloads"] = {"c2": {"fir": 183.3, "fft": 184.1, "note": "w" * 2000}}
    line = bench.compact_line(o)
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert d[k] == o[k] or k == "config"
    assert d["roofline"]["frac"] == 0.418 and d["roofline"]["bound"] == "hbm" and d["cpu_baseline"]["value"] == 174.4
    assert d["multi_gpu"]["rccl_ranks_seen"] == 8                 # the summary stays when dropping the per-rank list is enough ...
    assert "multi_gpu.ranks" in d["dropped_for_length"] and "live_sharded_analyzer" in d["dropped_for_length"]
    assert "live_sharded_analyzer" in o and "ranks" in o["multi_gpu"]   # ... and the caller's object (the detail file's) is untouched


def test_even_absurd_strings_cannot_push_the_line_over():
    import bench
    o = _line()
    o["config"]["workload"] = "q" * 6000
    line = bench.compact_line(o)
    assert len(line) < bench.LINE_LIMIT and json.loads(line)["value"] == 801.4
