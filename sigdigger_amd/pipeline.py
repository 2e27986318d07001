"""The analyzer worker step for one GPU: main-spectrum PSD + a bank of inspector chains.

Mirrors what libsuscan's source worker does per block of IQ: feed the PSD
(-> psd_message) and every open inspector (-> samples_message).  All arithmetic happens in
libsigdigger_amd.so; this module only sequences the C-ABI calls on HIP streams and, with
several GPUs, shards the inspector channels across ranks (channel c -> rank c mod G) and
broadcasts the IQ block with RCCL (torch.distributed backend "nccl").
"""
import os

import numpy as np
import torch

from . import engine


_STAGE_STREAMS = {}


class InspectorBankConfig:
    """Parameters shared by the inspectors of one bank (one decimation)."""

    def __init__(self, kind="psk", fnor=(), decimation=64, ntaps=255, bw_rel=0.75, sps=16.0,
                 costas_kind=engine.COSTAS_QPSK, loop_bw=0.005, agc=True, clock_gain=0.2, channeliser="fir"):
        self.kind = kind                      # "psk": AGC -> Costas -> Gardner ; "fsk": quad demod -> Gardner
        self.fnor = np.asarray(fnor, dtype=np.float64)
        self.decimation = int(decimation)
        self.ntaps = int(ntaps)
        self.bw_rel = float(bw_rel)           # low-pass cut-off relative to the decimated Nyquist
        self.sps = float(sps)                 # samples per symbol AFTER decimation
        self.costas_kind = costas_kind
        self.loop_bw = float(loop_bw)
        self.agc = bool(agc)
        self.clock_gain = float(clock_gain)
        # "fir": translate + `ntaps`-tap low-pass + decimate per channel (SPEC.md C, bit-exact against the oracle);
        # "fft": the FFT filter bank with su_specttuner's semantics (SPEC.md C2) -- what the reference itself runs
        # behind its channels (Tasks/LPFTask.cpp:52-69); the decimation is the power of two W / size
        self.channeliser = channeliser


class AnalyzerPipeline:
    """PSD + inspector bank on one device, block at a time, state carried between blocks.

    The inspector chain has three serial (one-lane-per-channel) stages -- AGC level tracking,
    Costas, Gardner -- each a single wavefront per 64 channels.  With overlap=True every stage
    runs on its own HIP stream and block k's stage s only waits for block k's stage s-1 (and,
    for buffer reuse, for the consumer of the buffer it overwrites), so consecutive blocks flow
    through the stages like a pipeline: steady-state cost per block = the slowest stage, not the
    sum.  The rest of the chip stays free for the PSD and FIR kernels of the next blocks.
    """

    NBUF = 3                                   # ring depth of every inter-stage buffer
    SUB = max(1, int(os.environ.get("SUAMD_PIPELINE_SUB", "4")))   # sub-ranges a block takes through the serial stages of a PSK chain

    def __init__(self, ctx, block_len, psd_size=8192, psd_window=engine.WINDOW_BLACKMANN_HARRIS,
                 psd_navg=None, bank=None, do_psd=True, overlap=True):
        self.ctx = ctx
        self.block_len = int(block_len)
        self.dev = torch.device("cuda", ctx.device)
        self.do_psd = do_psd
        self.psd_size = int(psd_size)
        self.overlap = overlap
        self.k = 0                             # blocks fed so far
        if do_psd:
            assert self.block_len % self.psd_size == 0
            self.nframes = self.block_len // self.psd_size
            self.navg = int(psd_navg or self.nframes)
            self.psd = engine.PSD(ctx, self.psd_size, psd_window)
            self.psd_out = torch.empty((self.nframes // self.navg, self.psd_size), dtype=torch.float32,
                                       device=self.dev)
        self.bank_cfg = bank
        self.nchan = 0
        if bank is not None and len(bank.fnor):
            self.nchan = len(bank.fnor)
            D = bank.decimation
            if bank.channeliser == "fft":
                self.chan = None
                self.st = engine.SpectTuner(ctx, 4096)
                for f in bank.fnor:
                    c = self.st.open_channel((np.pi * f) % (2 * np.pi), 2 * np.pi * bank.bw_rel / D, 1.0)
                    assert self.st.decimation(c) == D, "the FFT channeliser decimates by powers of two"
                assert self.block_len % 2048 == 0
            else:
                taps = ctx.lpf_design(bank.ntaps, bank.bw_rel / D)
                self.chan = engine.ChannelBank(ctx, bank.fnor, D, taps)
            self.m_max = self.block_len // D + 2
            nb = self.NBUF if overlap else 1
            # intermediates between stages are time-major ([time][channel] in memory): each
            # time step of the 64-lane recurrences is one contiguous 512-byte access
            self.y = [engine.time_major(self.nchan, self.m_max, self.dev) for _ in range(nb)]
            self.a = [engine.time_major(self.nchan, self.m_max, self.dev) for _ in range(nb)]
            self.z = [engine.time_major(self.nchan, self.m_max, self.dev) for _ in range(nb)]
            # recovered symbols leave the device per inspector: channel-major rows
            self.sym = [torch.empty((self.nchan, self.m_max), dtype=torch.complex64, device=self.dev)
                        for _ in range(nb)]
            self.count = [torch.zeros(self.nchan, dtype=torch.int32, device=self.dev) for _ in range(nb)]
            self.agc = None
            if bank.kind == "psk":
                self.agc = engine.AGCBank(ctx, self.nchan, tau=bank.sps) if bank.agc else None
                self.costas = engine.CostasBank(ctx, self.nchan, bank.costas_kind, 0.0, 2.0 / bank.sps, 3,
                                                bank.loop_bw)
            else:
                # last sample of the previous block per channel; ping-pong so that a launch never
                # reads and writes the same buffer
                self.qprev = [torch.zeros(self.nchan, dtype=torch.complex64, device=self.dev) for _ in range(2)]
                self.first = True
            self.clock = engine.ClockBank(ctx, self.nchan, bank.clock_gain, 1.0 / bank.sps)
        self.host_sym = None
        if overlap:
            # one set of stage streams per device, shared by every pipeline of the process: HIP spreads its streams over
            # four hardware queues, and a second pipeline with streams of its own would have two of its stages behind
            # one queue (bench.py builds C2 / C3 after the default workload: their kernels ran 2-3 x slower for it)
            key = str(self.dev)
            if key not in _STAGE_STREAMS:
                # ... and of a priority of their own (SUAMD_PIPELINE_STAGE_PRIORITY, default -1 = high): streams of
                # another priority come out of another set of hardware queues, so the three never share one with each
                # other or with whatever streams the process made before (csrc/analyzer.cpp init_device: same reason)
                prio = int(os.environ.get("SUAMD_PIPELINE_STAGE_PRIORITY", "-1"))
                _STAGE_STREAMS[key] = tuple(torch.cuda.Stream(self.dev, priority=prio) for _ in range(3))
            self.s_agc, self.s_dem, self.s_clk = _STAGE_STREAMS[key]      # AGC; Costas / quad demod; clock recovery
        self.done = {}                         # (stage, block index) -> event
        self.marks_per_step = 1                # timing marks per step of a serial stage (step() with sub-ranges: SUB)
        self.ev = {}                           # per-stage timing events

    # ---- helpers ---------------------------------------------------------------------------
    def _mark(self, name, stream, timed):
        if not timed:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        self.ev.setdefault(name, []).append(e)

    def _signal(self, stage, k, stream):
        e = torch.cuda.Event()
        e.record(stream)
        self.done[(stage, k)] = e
        self.done.pop((stage, k - 2 * self.NBUF), None)

    def _wait(self, stream, stage, k):
        e = self.done.get((stage, k))
        if e is not None:
            stream.wait_event(e)

    def _channelise(self, x, out, st):
        if self.chan is not None:
            return self.chan.feed(x, out=out, stream=st)
        y, counts = self.st.feed(x, out=out, stream=st)           # the first block is one half window short
        return y[:, :counts[0]]

    def step(self, x, timed=False, stream=None):
        """One pass of the hot path over one resident IQ block x (complex64 [block_len])."""
        st = stream or torch.cuda.current_stream(self.dev)
        k = self.k
        self.k += 1
        if not self.overlap:
            return self._step_serial(x, timed, st)
        nb = self.NBUF
        i = k % nb
        # ---- main spectrum on its own stream (reads x only) ----
        # (on the caller's stream: four streams = the four HIP hardware queues, so the three
        #  serial stages never share a queue and really run concurrently)
        if self.do_psd:
            self._mark("psd0", st, timed)
            self.psd.feed(x, nframes=self.nframes, navg=self.navg, scale=1.0 / self.psd_size,
                          out=self.psd_out, stream=st)
            self._mark("psd1", st, timed)
        if not self.nchan:
            return self.psd_out if self.do_psd else None
        cfg = self.bank_cfg
        psk = cfg.kind == "psk"
        # ---- channel bank: many workgroups, on the caller's stream ----
        self._wait(st, "agc" if (psk and self.agc is not None) else "dem", k - nb)   # y[i] free again
        self._mark("fir0", st, timed)
        y = self._channelise(x, self.y[i], st)
        self._mark("fir1", st, timed)
        self._signal("fir", k, st)
        m = y.shape[1]
        src = y
        # The three serial stages take the block in SUB sub-ranges (PSK chains): sub-range j of the Costas stage only waits for
        # sub-range j of the AGC, the Gardner stage for sub-range j of Costas -- a block's way through the stages (and with it
        # the pipeline's fill and drain: 1.3 steps per timed region with whole blocks) shrinks to the slowest stage plus a
        # quarter of the others.  The banks are stream processors (state carried from call to call), so the sub-ranges give
        # the same samples as the whole block, bit for bit (csrc/analyzer.cpp pushes its inspectors through the same way).
        nsub = self.SUB if psk else 1
        cuts = [m * j // nsub for j in range(nsub + 1)]
        # ---- AGC ----
        if psk and self.agc is not None:
            self._wait(self.s_agc, "fir", k)
            self._wait(self.s_agc, "dem", k - nb)                                    # a[i] free again
            for j in range(nsub):
                lo, hi = cuts[j], cuts[j + 1]
                self._mark("agc0", self.s_agc, timed)
                if hi > lo:
                    self.agc.feed(y[:, lo:hi], out=self.a[i][:, lo:hi], stream=self.s_agc)
                self._mark("agc1", self.s_agc, timed)
                self._signal(("agc", j), k, self.s_agc)
            src = self.a[i][:, :m]
            self._signal("agc", k, self.s_agc)
        # ---- carrier recovery / quadrature demod ----
        self._wait(self.s_dem, "clk", k - nb)                                        # z[i] free again
        if psk:
            if self.agc is None:
                self._wait(self.s_dem, "fir", k)
            for j in range(nsub):
                lo, hi = cuts[j], cuts[j + 1]
                if self.agc is not None:
                    self._wait(self.s_dem, ("agc", j), k)
                self._mark("dem0", self.s_dem, timed)              # (behind the wait: the stage's own time, per sub-range)
                if hi > lo:
                    self.costas.feed(src[:, lo:hi], out=self.z[i][:, lo:hi], stream=self.s_dem)
                self._mark("dem1", self.s_dem, timed)
                self._signal(("dem", j), k, self.s_dem)
            z = self.z[i][:, :m]
        else:
            self._wait(self.s_dem, "fir", k)
            self._mark("dem0", self.s_dem, timed)
            z = self.ctx.quad_demod(y, prev=self.qprev[k & 1], first=self.first, out=self.z[i][:, :m],
                                    prev_out=self.qprev[(k + 1) & 1], stream=self.s_dem)
            self.first = False
            self._mark("dem1", self.s_dem, timed)
            self._signal(("dem", 0), k, self.s_dem)
        self._signal("dem", k, self.s_dem)
        # ---- clock recovery (appends to the block's symbol rows: one count per channel, cleared once per block) ----
        self._wait(self.s_clk, ("dem", 0), k)
        with torch.cuda.stream(self.s_clk):
            self.count[i].zero_()
        for j in range(nsub):
            lo, hi = cuts[j], cuts[j + 1]
            self._wait(self.s_clk, ("dem", j), k)
            self._mark("clk0", self.s_clk, timed)
            if hi > lo:
                self.clock.feed(z[:, lo:hi], self.sym[i], self.count[i], stream=self.s_clk)
            self._mark("clk1", self.s_clk, timed)
        self._signal("clk", k, self.s_clk)
        self.marks_per_step = nsub
        return self.psd_out if self.do_psd else None

    def _step_serial(self, x, timed, st):
        self.marks_per_step = 1
        if self.do_psd:
            self._mark("psd0", st, timed)
            self.psd.feed(x, nframes=self.nframes, navg=self.navg, scale=1.0 / self.psd_size,
                          out=self.psd_out, stream=st)
            self._mark("psd1", st, timed)
        if self.nchan:
            cfg = self.bank_cfg
            self._mark("fir0", st, timed)
            y = self._channelise(x, self.y[0], st)
            self._mark("fir1", st, timed)
            m = y.shape[1]
            with torch.cuda.stream(st):
                self.count[0].zero_()
            src = y
            if cfg.kind == "psk":
                if self.agc is not None:
                    self._mark("agc0", st, timed)
                    src = self.agc.feed(y, out=self.a[0][:, :m], stream=st)
                    self._mark("agc1", st, timed)
                self._mark("dem0", st, timed)
                z = self.costas.feed(src, out=self.z[0][:, :m], stream=st)
                self._mark("dem1", st, timed)
            else:
                self._mark("dem0", st, timed)
                kk = self.k - 1
                z = self.ctx.quad_demod(y, prev=self.qprev[kk & 1], first=self.first, out=self.z[0][:, :m],
                                        prev_out=self.qprev[(kk + 1) & 1], stream=st)
                self.first = False
                self._mark("dem1", st, timed)
            self._mark("clk0", st, timed)
            self.clock.feed(z, self.sym[0], self.count[0], stream=st)
            self._mark("clk1", st, timed)
        return self.psd_out if self.do_psd else None

    def enable_delivery(self):
        """Hand-off to the host, as the analyzer's SAMPLES messages need it: after every step the recovered symbols and
        their counts are copied to pinned host memory on the clock stage's stream (overlap=True only)."""
        assert self.overlap and self.nchan
        nb = self.NBUF
        self.sym_cap = min(self.m_max, int(self.m_max / max(1.0, 0.5 * self.bank_cfg.sps)) + 64)
        self.host_sym = [torch.empty((self.nchan, self.sym_cap), dtype=torch.complex64).pin_memory() for _ in range(nb)]
        self.host_count = [torch.zeros(self.nchan, dtype=torch.int32).pin_memory() for _ in range(nb)]

    def deliver(self):
        """Enqueues the device -> host copy of the block just fed; returns (host symbols, host counts) of that slot --
        valid once the clock stream has passed (latest_symbols() / sync())."""
        i = (self.k - 1) % self.NBUF
        with torch.cuda.stream(self.s_clk):
            self.host_sym[i].copy_(self.sym[i][:, :self.sym_cap], non_blocking=True)
            self.host_count[i].copy_(self.count[i], non_blocking=True)
        return self.host_sym[i], self.host_count[i]

    def latest_symbols(self):
        """(sym, count) of the most recently fed block (synchronises the clock stage)."""
        i = (self.k - 1) % (self.NBUF if self.overlap else 1)
        if self.overlap:
            self.s_clk.synchronize()
        return self.sym[i], self.count[i]

    def sync(self):
        torch.cuda.synchronize(self.dev)

    def stage_times_ms(self):
        """Average kernel time of each stage over the timed steps (HIP events recorded on the
        stream each stage runs on; call after a device sync)."""
        ev = self.ev
        self.stalled_samples = {}
        self.stage_raw = {}                   # per stage: mean and median over ALL timed steps, nothing dropped

        def avg(a, b):
            # Mean over the timed steps.  A sample more than 5x the median is not a launch: roughly every other run has
            # ONE step in which the stream sits 1-1.6 ms between the two events of a 25-40 us kernel (queue
            # housekeeping on the host side; tools/fir_dist.py shows the distribution) -- such samples are dropped and
            # counted in self.stalled_samples, so the figure is the kernel's launch duration, as rocprofv3 reports it.
            if a in ev and b in ev:
                t = np.array([s.elapsed_time(e) for s, e in zip(ev[a], ev[b])])
                n = getattr(self, "marks_per_step", 1)          # the serial stages of a PSK chain mark every sub-range: per step = their sum
                if a[:3] in ("agc", "dem", "clk") and n > 1 and t.size % n == 0:
                    t = t.reshape(-1, n).sum(axis=1)
                keep = t <= 5.0 * np.median(t)
                self.stage_raw[a[:-1]] = {"mean": float(t.mean()), "median": float(np.median(t)), "max": float(t.max()), "n": int(t.size)}
                if not keep.all():
                    self.stalled_samples[a[:-1]] = int((~keep).sum())
                return float(t[keep].mean())
            return None
        demod = "costas" if (self.bank_cfg is not None and self.bank_cfg.kind == "psk") else "quad"
        out = {"psd": avg("psd0", "psd1"), "fir": avg("fir0", "fir1"), "agc": avg("agc0", "agc1"),
               demod: avg("dem0", "dem1"), "clock": avg("clk0", "clk1")}
        return {k: v for k, v in out.items() if v is not None}

    def reset_events(self):
        self.ev = {}


def shard_channels(fnor, rank, world):
    """channel c -> rank c mod G: independent chains, no data-path collective."""
    fnor = np.asarray(fnor, dtype=np.float64)
    return fnor[rank::world]


def channel_owner(c, world):
    """rank that owns inspector channel c, and its index inside that rank's bank"""
    return c % world, c // world


def broadcast_block(buf, dist, src=0, async_op=True):
    """The one exchange step of the multi-GPU path: rank `src` holds the IQ block, every rank
    needs it (RCCL broadcast over xGMI on GPUs; gloo in the CPU tests).  Returns the work handle
    (None when not distributed)."""
    if dist is None or not dist.is_initialized():
        return None
    if buf.is_complex():
        buf = torch.view_as_real(buf)          # same memory; the collective only needs the bytes
    return dist.broadcast(buf, src=src, async_op=async_op)
