"""The analyzer worker step for one GPU: main-spectrum PSD + a bank of inspector chains.

Mirrors what libsuscan's source worker does per block of IQ: feed the PSD
(-> psd_message) and every open inspector (-> samples_message).  All arithmetic happens in
libsigdigger_amd.so; this module only sequences the C-ABI calls on HIP streams and, with
several GPUs, shards the inspector channels across ranks (channel c -> rank c mod G) and
broadcasts the IQ block with RCCL (torch.distributed backend "nccl").
"""
import numpy as np
import torch

from . import engine


class InspectorBankConfig:
    """Parameters shared by the inspectors of one bank (one decimation)."""

    def __init__(self, kind="psk", fnor=(), decimation=64, ntaps=255, bw_rel=0.75, sps=16.0,
                 costas_kind=engine.COSTAS_QPSK, loop_bw=0.005, agc=True, clock_gain=0.2):
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


class AnalyzerPipeline:
    """PSD + inspector bank on one device, block at a time, state carried between blocks."""

    def __init__(self, ctx, block_len, psd_size=8192, psd_window=engine.WINDOW_BLACKMANN_HARRIS,
                 psd_navg=None, bank=None, do_psd=True):
        self.ctx = ctx
        self.block_len = int(block_len)
        self.dev = torch.device("cuda", ctx.device)
        self.do_psd = do_psd
        self.psd_size = int(psd_size)
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
            taps = ctx.lpf_design(bank.ntaps, bank.bw_rel / D)
            self.chan = engine.ChannelBank(ctx, bank.fnor, D, taps)
            self.m_max = self.block_len // D + 2
            stride = (self.m_max + 7) // 8 * 8
            self.y = torch.empty((self.nchan, stride), dtype=torch.complex64, device=self.dev)
            self.a = torch.empty_like(self.y)
            self.z = torch.empty_like(self.y)
            self.sym = torch.empty_like(self.y)
            self.count = torch.zeros(self.nchan, dtype=torch.int32, device=self.dev)
            if bank.kind == "psk":
                self.agc = engine.AGCBank(ctx, self.nchan, tau=bank.sps) if bank.agc else None
                self.costas = engine.CostasBank(ctx, self.nchan, bank.costas_kind, 0.0, 2.0 / bank.sps, 3,
                                                bank.loop_bw)
            else:
                self.qprev = torch.zeros(self.nchan, dtype=torch.complex64, device=self.dev)
                self.first = True
            self.clock = engine.ClockBank(ctx, self.nchan, bank.clock_gain, 1.0 / bank.sps)
        # per-stage HIP events (recorded on the stream the kernels are launched on)
        self.ev = {}

    def _mark(self, name, stream, timed):
        if not timed:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        self.ev.setdefault(name, []).append(e)

    def step(self, x, timed=False, stream=None):
        """One pass of the hot path over one resident IQ block x (complex64 [block_len])."""
        st = stream or torch.cuda.current_stream(self.dev)
        if self.do_psd:
            self._mark("psd0", st, timed)
            self.psd.feed(x, nframes=self.nframes, navg=self.navg, scale=1.0 / self.psd_size,
                          out=self.psd_out, stream=st)
            self._mark("psd1", st, timed)
        if self.nchan:
            cfg = self.bank_cfg
            self._mark("fir0", st, timed)
            y = self.chan.feed(x, out=self.y, stream=st)
            self._mark("fir1", st, timed)
            m = y.shape[1]
            self.count.zero_()
            if cfg.kind == "psk":
                src = y
                if self.agc is not None:
                    src = self.agc.feed(y, out=self.a[:, :m], stream=st)
                    self._mark("agc1", st, timed)
                z = self.costas.feed(src, out=self.z[:, :m], stream=st)
                self._mark("costas1", st, timed)
            else:
                z = self.ctx.quad_demod(y, prev=self.qprev, first=self.first, out=self.z[:, :m],
                                        prev_out=self.qprev, stream=st)
                self.first = False
                self._mark("quad1", st, timed)
            self.clock.feed(z, self.sym, self.count, stream=st)
            self._mark("clock1", st, timed)
        return self.psd_out if self.do_psd else None

    def stage_times_ms(self):
        """Average duration of each stage over the timed steps (HIP events; call after a sync)."""
        out = {}
        ev = self.ev

        def avg(a, b):
            if a in ev and b in ev:
                return float(np.mean([s.elapsed_time(e) for s, e in zip(ev[a], ev[b])]))
            return None
        out["psd"] = avg("psd0", "psd1")
        out["fir"] = avg("fir0", "fir1")
        if "agc1" in ev:
            out["agc"] = avg("fir1", "agc1")
            out["costas"] = avg("agc1", "costas1")
        elif "costas1" in ev:
            out["costas"] = avg("fir1", "costas1")
        if "quad1" in ev:
            out["quad"] = avg("fir1", "quad1")
            out["clock"] = avg("quad1", "clock1")
        elif "costas1" in ev:
            out["clock"] = avg("costas1", "clock1")
        return {k: v for k, v in out.items() if v is not None}

    def reset_events(self):
        self.ev = {}


def shard_channels(fnor, rank, world):
    """channel c -> rank c mod G: independent chains, no data-path collective."""
    fnor = np.asarray(fnor, dtype=np.float64)
    return fnor[rank::world]
