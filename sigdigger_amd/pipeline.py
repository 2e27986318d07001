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


def cu_partition(ncus, reserved_per_xcd, nxcd=8):
    """(reserved, transform) compute-unit index lists.  The driver deals consecutive mask bits to consecutive XCDs (bit i
    -> XCD i mod 8; suamd_probe_placement / tests/test_gpu_cu_mask.py check it), and the dispatcher deals the workgroups of
    a launch to the XCDs in turn whatever their enabled CUs -- so the reserved set takes the SAME number of CUs from every
    XCD: a transform launch planned for one round then has the same slots on each."""
    r = max(0, min(int(reserved_per_xcd), ncus // nxcd - 1))
    reserved = list(range(r * nxcd))
    return reserved, list(range(r * nxcd, ncus))


def _make_stage_streams(ctx, dev):
    """The three recurrence stage streams, and the stream + CU count the transform kernels get.

    SUAMD_PIPELINE_CU_PARTITION=1 (the number = reserved CUs per XCD; default 0 = off, the transform window of
    AnalyzerPipeline does better): the recurrences -- one wavefront per 64
    inspectors holding ~200 registers of a SIMD (the clock kernel also 32 KB of LDS) for milliseconds -- run on streams
    confined to 8 reserved CUs (hipExtStreamCreateWithCUMask), the transform kernels on a stream confined to the other
    248.  Before (round 4) a recurrence wavefront sat wherever the dispatcher put it and took a slot from a transform
    launch planned for exactly one round of workgroups (profiles/r04_inpipe_penalty.txt: stp_kernel 85 -> 96 us, and a launch
    could not be planned for more than 3/4 of the chip).  Measured (round 5, 16 Mi block): stp_kernel 96 -> 89-90 us, no launch
    above 112 us any more -- but 248 CUs hold 992 window slots, so a 16 Mi block (8192 windows) takes 9 windows per
    workgroup where the whole chip takes 8: the transform window (idle chip, 1024 slots) reaches 76 us.
    SUAMD_PIPELINE_CU_PARTITION=0: streams of a priority of their own, as in round 4."""
    part = int(os.environ.get("SUAMD_PIPELINE_CU_PARTITION", "0"))
    ncus = ctx.cu_count()
    if part > 0 and ncus >= 16:
        reserved, transform = cu_partition(ncus, part)
        # every stage gets the whole reserved set; loops.hip's serial_xcd() puts AGC / Costas / clock on three XCDs of it
        stages = tuple(ctx.masked_stream(reserved) for _ in range(3))
        # two streams on the other CUs: the transforms' (PSD, channeliser) and one for the feed-forward kernels of the
        # serial stages (the AGC's magnitudes / sliding maximum / gain: many workgroups, microseconds each)
        return stages + (ctx.masked_stream(transform), len(transform), ctx.masked_stream(transform))
    # streams of a priority of their own (SUAMD_PIPELINE_STAGE_PRIORITY, default -1 = high): streams of another priority
    # come out of another set of hardware queues, so the three never share one with each other or with whatever streams
    # the process made before (csrc/analyzer.cpp init_device: same reason)
    prio = int(os.environ.get("SUAMD_PIPELINE_STAGE_PRIORITY", "-1"))
    return tuple(torch.cuda.Stream(dev, priority=prio) for _ in range(3)) + (None, ncus, None)


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

    Transform window (round 5, default): the TAIL of a block's serial stages -- the last carrier sub-range, the last two
    clock sub-ranges, the hand-off copy of deliver() -- is only enqueued by the NEXT step(), behind that block's PSD and
    channeliser, which get the idle chip.  A caller that wants the results of the last block it fed calls flush() (or
    latest_symbols() / sync(), which do) before it waits for them.
    """

    NBUF = 3                                   # ring depth of every inter-stage buffer
    SUB = max(1, int(os.environ.get("SUAMD_PIPELINE_SUB", "4")))   # sub-ranges a block takes through the serial stages of a PSK chain

    def __init__(self, ctx, block_len, psd_size=8192, psd_window=engine.WINDOW_BLACKMANN_HARRIS,
                 psd_navg=None, bank=None, do_psd=True, overlap=True, window=None):
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
        self._pending = []                     # transform window: closures of the last block's held-back tail
        # Transform window (default on; SUAMD_PIPELINE_WINDOW=0 or window=False: round 4's free-running streams).  Once per
        # block the recurrence streams pause at a skewed cut (AGC done, Costas one sub-range from its end, Gardner two) and
        # the block's PSD + channeliser run on an otherwise idle chip: the channeliser plans all 1024 window slots (8 windows
        # per workgroup on a 16 Mi block instead of 11 on 3/4 of them) and nothing displaces a workgroup of its one round.
        # Measured (16 Mi block, 64 inspectors, same box, alternating): stp_kernel 96.4 us (max 130) -> 76.1 (max 77),
        # psd_kernel + reduce 50 -> 44 us; the slowest stream pauses ~0.15 ms per 21 ms step (795 -> 789 MS/s).
        if window is None:
            window = os.environ.get("SUAMD_PIPELINE_WINDOW", "1") != "0"
        # (a window exists only where a tail can be held back: at least three sub-ranges -- with fewer, nothing pauses and the
        # plans below, which assume the transforms have the chip to themselves, would be applied beside running recurrences)
        self.window = bool(overlap and window and self.SUB >= 3)
        if self.window and do_psd and self.psd_size == 8192 and os.environ.get("SUAMD_PSD_SPLIT_TARGET") is None:
            self.psd.set_split_target(512)     # two 8192-point workgroups per CU, and the chip is the PSD's own in the window
        if overlap:
            # one set of stage streams per device, shared by every pipeline of the process: HIP spreads its streams over
            # four hardware queues, and a second pipeline with streams of its own would have two of its stages behind
            # one queue (bench.py builds C2 / C3 after the default workload: their kernels ran 2-3 x slower for it)
            key = str(self.dev)
            if key not in _STAGE_STREAMS:
                _STAGE_STREAMS[key] = _make_stage_streams(ctx, self.dev)
            self.s_agc, self.s_dem, self.s_clk, self.s_main, self.transform_cus, self.s_wide = _STAGE_STREAMS[key]      # AGC; Costas / quad demod; clock recovery
            self.s_crit = self.s_dem if (bank is None or bank.kind == "psk") else self.s_clk
            if (self.s_main is not None or self.window) and self.nchan and self.chan is None and not os.environ.get("SUAMD_PIPE_KEEP_SLOTS"):
                # the transform kernels have `transform_cus` compute units to themselves (a CU partition, or the whole chip
                # inside a transform window): the channeliser plans its ONE round of workgroups for all their window
                # slots (4 per CU) instead of 3/4 of the chip's
                self.st.set_slots(4 * self.transform_cus)
            if self.window and self.nchan and self.chan is not None:
                self.chan.set_exclusive(True)      # inside the window the FIR bank's feed is alone on the chip too
        self.done = {}                         # (stage, block index) -> event
        self.marks_per_step = 1                # timing marks per step of a serial stage (step() with sub-ranges: SUB)
        self.ev = {}                           # per-stage timing events

    # ---- helpers ---------------------------------------------------------------------------
    def main_stream(self):
        """the stream the PSD and the channeliser run on: the slowest stage's stream inside a transform window, the CU-masked
        transform stream when the device is partitioned, else the caller's current stream.  Callers that order other work
        against a step -- the block broadcast, uploads -- make it current: a step then costs no cross-stream hop to and from
        the caller's stream."""
        if self.overlap and self.window and self.nchan:
            return self.s_crit                 # transform window: the slowest stage's stream carries the transforms
        s = getattr(self, "s_main", None) if self.overlap else None
        return s if s is not None else torch.cuda.current_stream(self.dev)

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

    def step(self, x, timed=False, stream=None, after_transforms=None):
        """One pass of the hot path over one resident IQ block x (complex64 [block_len]).

        after_transforms: called (no arguments) right behind the enqueue of the block's PSD + channeliser, with their stream
        current -- the place to start work that must not run beside them and may run beside the recurrences (the broadcast of
        the NEXT block to the other GPUs: ordered behind this block's transforms, it overlaps the serial stages)."""
        if not self.overlap:
            self.k += 1
            out = self._step_serial(x, timed, stream or torch.cuda.current_stream(self.dev))
            if after_transforms is not None:
                after_transforms()
            return out
        self._after_transforms = after_transforms
        caller = torch.cuda.current_stream(self.dev)
        # Transform window: the PSD and the channeliser go onto the slowest stage's stream (carrier recovery; the clock stage
        # of an FSK chain) -- the one whose pause is the cost of the window: between its launches there is then no cross-queue hop (measured with the
        # transforms on a stream of their own: 37 us from the last Costas launch to the PSD, 43 us from the channeliser to the
        # next Costas launch, on top of the kernels' 120 us)
        st = stream or (self.s_crit if self.window else self.main_stream())
        if stream is None and st != caller:
            # ... on a stream of the pipeline's, but with the caller's stream semantics: the transforms start behind what
            # the caller has enqueued (its x) and the caller's stream continues behind them (psd_out, its reuse of x)
            st.wait_stream(caller)
            try:
                return self._step_overlapped(x, timed, st)
            finally:
                caller.wait_event(self.done[("fir", self.k - 1)] if self.nchan else self._psd_done)
        return self._step_overlapped(x, timed, st)

    def _step_overlapped(self, x, timed, st):
        k = self.k
        self.k += 1
        nb = self.NBUF
        i = k % nb
        stage_streams = [q for q in (self.s_agc, self.s_wide, self.s_dem, self.s_clk) if q is not None]
        pend, self._pending = self._pending, []
        if pend:
            # transform window (see _make_stage_streams): the PSD and the channeliser of this block start when the recurrence
            # launches enqueued so far -- block k-1 up to its skewed cut -- have drained ...
            for q in stage_streams:
                if q != st:
                    st.wait_stream(q)
        # ---- main spectrum (reads x only) ----
        if self.do_psd:
            self._mark("psd0", st, timed)
            self.psd.feed(x, nframes=self.nframes, navg=self.navg, scale=1.0 / self.psd_size,
                          out=self.psd_out, stream=st)
            self._mark("psd1", st, timed)
        if not self.nchan:
            self._psd_done = torch.cuda.Event()
            self._psd_done.record(st)
            hook, self._after_transforms = getattr(self, "_after_transforms", None), None
            if hook is not None:
                with torch.cuda.stream(st):
                    hook()
            return self.psd_out if self.do_psd else None
        cfg = self.bank_cfg
        psk = cfg.kind == "psk"
        # ---- channel bank: many workgroups, on the transform stream ----
        self._wait(st, "agc" if (psk and self.agc is not None) else "dem", k - nb)   # y[i] free again
        self._mark("fir0", st, timed)
        y = self._channelise(x, self.y[i], st)
        self._mark("fir1", st, timed)
        self._signal("fir", k, st)
        hook, self._after_transforms = getattr(self, "_after_transforms", None), None
        if hook is not None:
            with torch.cuda.stream(st):
                hook()
        if pend:
            # ... and the rest of block k-1 (and everything after it) runs behind them: the transforms have the chip to
            # themselves for their ~130 us, the slowest recurrence stream pauses for exactly that long
            for q in stage_streams:
                if q != st:
                    q.wait_event(self.done[("fir", k)])
            for f in pend:
                f()
        m = y.shape[1]
        src = y
        # The three serial stages take the block in SUB sub-ranges (PSK chains): sub-range j of the Costas stage only waits for
        # sub-range j of the AGC, the Gardner stage for sub-range j of Costas -- a block's way through the stages (and with it
        # the pipeline's fill and drain: 1.3 steps per timed region with whole blocks) shrinks to the slowest stage plus a
        # quarter of the others.  The banks are stream processors (state carried from call to call), so the sub-ranges give
        # the same samples as the whole block, bit for bit (csrc/analyzer.cpp pushes its inspectors through the same way).
        nsub = self.SUB if (psk or self.window) else 1       # (FSK chains: only the clock stage, and only for the window's cut)
        cuts = [m * j // nsub for j in range(nsub + 1)]
        ops = {"agc": [], "dem": [], "clk": []}            # per stage: one closure per sub-range, enqueued in order
        # ---- AGC ----
        if psk and self.agc is not None:
            # CU-partitioned device: only the level trackers run on the (confined) AGC stream; the feed-forward kernels,
            # the stage's dependencies and its completion live on the wide stream
            sa = self.s_wide if self.s_wide is not None else self.s_agc

            def agc_op(j):
                lo, hi = cuts[j], cuts[j + 1]
                if j == 0:
                    self._wait(sa, "fir", k)
                    self._wait(sa, "dem", k - nb)                                    # a[i] free again
                self._mark("agc0", sa, timed)
                if hi > lo:
                    self.agc.feed(y[:, lo:hi], out=self.a[i][:, lo:hi], stream=self.s_agc, wide=self.s_wide)
                self._mark("agc1", sa, timed)
                self._signal(("agc", j), k, sa)
                if j == nsub - 1:
                    self._signal("agc", k, sa)
            ops["agc"] = [(lambda j=j: agc_op(j)) for j in range(nsub)]
            src = self.a[i][:, :m]
        # ---- carrier recovery / quadrature demod ----
        z = self.z[i][:, :m]
        if psk:
            def dem_op(j):
                lo, hi = cuts[j], cuts[j + 1]
                if j == 0:
                    self._wait(self.s_dem, "clk", k - nb)                            # z[i] free again
                    if self.agc is None:
                        self._wait(self.s_dem, "fir", k)
                if self.agc is not None:
                    self._wait(self.s_dem, ("agc", j), k)
                self._mark("dem0", self.s_dem, timed)              # (behind the wait: the stage's own time, per sub-range)
                if hi > lo:
                    self.costas.feed(src[:, lo:hi], out=self.z[i][:, lo:hi], stream=self.s_dem)
                self._mark("dem1", self.s_dem, timed)
                self._signal(("dem", j), k, self.s_dem)
                if j == nsub - 1:
                    self._signal("dem", k, self.s_dem)
            ops["dem"] = [(lambda j=j: dem_op(j)) for j in range(nsub)]
        else:
            def quad_op():
                self._wait(self.s_dem, "clk", k - nb)
                self._wait(self.s_dem, "fir", k)
                self._mark("dem0", self.s_dem, timed)
                self.ctx.quad_demod(y, prev=self.qprev[k & 1], first=self.first, out=self.z[i][:, :m],
                                    prev_out=self.qprev[(k + 1) & 1], stream=self.s_dem)
                self.first = False
                self._mark("dem1", self.s_dem, timed)
                self._signal(("dem", 0), k, self.s_dem)
                self._signal("dem", k, self.s_dem)
            ops["dem"] = [quad_op]

        # ---- clock recovery (appends to the block's symbol rows: one count per channel, cleared once per block) ----
        def clk_op(j):
            lo, hi = cuts[j], cuts[j + 1]
            if j == 0:
                self._wait(self.s_clk, ("dem", 0), k)
                with torch.cuda.stream(self.s_clk):
                    self.count[i].zero_()
            self._wait(self.s_clk, ("dem", j), k)
            self._mark("clk0", self.s_clk, timed)
            if hi > lo:
                self.clock.feed(z[:, lo:hi], self.sym[i], self.count[i], stream=self.s_clk)
            self._mark("clk1", self.s_clk, timed)
            if j == nsub - 1:
                self._signal("clk", k, self.s_clk)
        ops["clk"] = [(lambda j=j: clk_op(j)) for j in range(nsub)]
        # Transform window: the tail of the block -- the last sub-range of the carrier stage, the last two of the clock stage --
        # is held back until the NEXT block's transforms are enqueued.  The stages run skewed by one sub-range (AGC ahead of
        # Costas ahead of Gardner), so AGC j+2 / Costas j+1 / clock j end together: cutting there pauses the slowest stream
        # for the transforms' own duration and the others inside their slack.
        hold = {"agc": 0, "dem": 0, "clk": 0}
        if self.window and nsub >= 3:
            hold = {"agc": 0, "dem": 1, "clk": 2} if psk else {"agc": 0, "dem": 0, "clk": 1}
        for name in ("agc", "dem", "clk"):
            n = len(ops[name]) - hold[name]
            for f in ops[name][:n]:
                f()
            self._pending.extend(ops[name][n:])
        self.marks_per_step = nsub
        self.marks_per_stage = {name: len(ops[name]) for name in ops}
        return self.psd_out if self.do_psd else None

    def flush(self):
        """enqueues what a transform window holds back (the last block's tail): call before waiting for results"""
        pend, self._pending = self._pending, []
        for f in pend:
            f()

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

        def copy():
            with torch.cuda.stream(self.s_clk):
                self.host_sym[i].copy_(self.sym[i][:, :self.sym_cap], non_blocking=True)
                self.host_count[i].copy_(self.count[i], non_blocking=True)
        if self._pending:
            self._pending.append(copy)         # behind the held-back clock sub-ranges (transform window)
        else:
            copy()
        return self.host_sym[i], self.host_count[i]

    def latest_symbols(self):
        """(sym, count) of the most recently fed block (synchronises the clock stage)."""
        i = (self.k - 1) % (self.NBUF if self.overlap else 1)
        if self.overlap:
            self.flush()
            self.s_clk.synchronize()
        return self.sym[i], self.count[i]

    def sync(self):
        self.flush()
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
                n = getattr(self, "marks_per_stage", {}).get(a[:3], 1) if self.overlap else 1   # a serial stage marks every sub-range: per step = their sum
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


def shard_frames(nframes, rank, world):
    """frame (dwell) f of a panoramic sweep -> rank f mod G ("C5 shards by frame"; the frames are
    independent -- Panoramic/Scanner.cpp:503-523 feeds them one by one into bins that only they cover): the indices this rank
    transforms.  No exchange: every rank keeps its own SpectrumView over the whole range."""
    return np.arange(int(rank), int(nframes), int(world))


def broadcast_block(buf, dist, src=0, async_op=True):
    """The one exchange step of the multi-GPU path: rank `src` holds the IQ block, every rank
    needs it (RCCL broadcast over xGMI on GPUs; gloo in the CPU tests).  Returns the work handle
    (None when not distributed)."""
    if dist is None or not dist.is_initialized():
        return None
    if buf.is_complex():
        buf = torch.view_as_real(buf)          # same memory; the collective only needs the bytes
    return dist.broadcast(buf, src=src, async_op=async_op)
