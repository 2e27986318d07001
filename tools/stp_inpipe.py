"""Where the channeliser's 3-4 us between 'alone' and 'inside the pipeline' go: the same kernel timed (kernel timer)
(a) re-fed one resident block, (b) in the serial pipeline (nothing runs beside it, buffers rotate), (c) in the overlapped
pipeline (recurrence kernels of earlier blocks resident)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from sigdigger_amd import engine, pipeline, synth

dev = torch.device("cuda", 0)
ctx = engine.Context(0)
cfg = bench.WORKLOADS["c4"]
L = 1 << 22
fn = synth.raster(cfg["per_gpu"], cfg["spacing"])
bank = pipeline.InspectorBankConfig(kind=cfg["kind"], fnor=fn, decimation=cfg["D"], ntaps=cfg["T"], sps=cfg["sps_in"] / cfg["D"], channeliser="fft")
bufs = [bench.make_block(L, fn, cfg["sps_in"], cfg["kind"], dev, seed=1234), torch.empty(L, dtype=torch.complex64, device=dev)]
bufs[1].copy_(bufs[0])
for overlap, do_psd in ((False, True), (False, False), (True, True), (True, False)):
    pipe = pipeline.AnalyzerPipeline(ctx, L, psd_size=cfg["psd"], psd_navg=256, bank=bank, do_psd=do_psd, overlap=overlap)
    if overlap:
        pipe.enable_delivery()
    for k in range(5):
        pipe.step(bufs[k & 1])
    torch.cuda.synchronize()
    engine.kernel_timing_read(); engine.kernel_timing(True)
    n = 100 if overlap else 30
    for k in range(n):
        pipe.step(bufs[k & 1])
        if overlap:
            pipe.deliver()
    torch.cuda.synchronize(); engine.kernel_timing(False)
    r = engine.kernel_timing_read("stp_kernel")
    engine.kernel_timing_read()
    print(f"overlap={overlap} psd={do_psd}: stp_kernel avg {r['sum_ms'] / r['launches'] * 1e3:.1f} us (min {r['min_ms'] * 1e3:.1f}, max {r['max_ms'] * 1e3:.1f}, {r['launches']} launches)")
    del pipe
