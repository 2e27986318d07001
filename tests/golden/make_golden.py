"""Generates tests/golden/oracle_v1.npz, oracle_v2.npz and oracle_v3.npz from the CPU oracle.

The reference (SigDigger) ships NO golden vectors and its DSP libraries are absent
(SURVEY.md section 8c), so these fixtures are produced by oracle/sdo.c itself on seeded inputs: they
pin the oracle (and through the GPU parity tests, the HIP path) against regressions; they are
not reference-generated vectors.

    python -m tests.golden.make_golden
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def compute(sdo):
    from sigdigger_amd import synth
    out = {}
    x = synth.psk_carriers(4096, [0.2, -0.35], sps=16, order=4, seed=123, snr_db=20)
    out["input_iq"] = x
    p = np.arange(0, 2 ** 32, 2 ** 32 // 257, dtype=np.uint64).astype(np.uint32)
    out["phasor"] = sdo.phasor_u32(p)
    out["psd_bh_1024"] = sdo.psd_frames(x, 4, 1024, 1024, sdo.window(4, 1024), navg=2, scale=1.0 / 1024)
    out["psd_shift_db"] = sdo.psd_shift_db(out["psd_bh_1024"][0])
    dp = sdo.fnor_to_dphase(-0.2)
    out["xlate"] = sdo.xlate_bulk(x, 7, dp, 1000)
    taps = sdo.lpf_design(63, 0.1)
    out["taps"] = taps
    g = sdo.chan_modulate_taps(taps, dp)
    y = sdo.chan_feed(np.zeros(62, np.complex64), x, 0, g, 8, 0, dp)
    out["chan_D8"] = y
    out["quad"] = sdo.quad_demod(y)
    out["delayed_conj"] = sdo.delayed_conj(y, 3)
    a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(2.0)), y)
    out["agc"] = a
    st = sdo.costas_new(2, 0.0, 1.0, 3, 0.02)
    out["costas_qpsk"] = sdo.costas_feed_bulk(st, a)
    out["costas_state"] = np.array([st.phase, np.float32(st.omega).view(np.uint32)], dtype=np.uint32)
    out["pll"] = sdo.pll_track_bulk(sdo.pll_new(0.0, 0.05), y)
    out["gardner"] = sdo.clock_feed_bulk(sdo.clock_new(0.5, 0.5), out["costas_qpsk"])
    return out


def compute_v2(sdo):
    """second fixture set: the stages added after v1 (samplers, A7 stages, ingest, spectrum sources, FAC)"""
    from sigdigger_amd import synth
    out = {}
    x = synth.psk_carriers(3 * 4096 + 500, [0.0], sps=8, order=2, seed=77, snr_db=25)
    out["input_iq"] = x
    out["zc_amplitude"] = sdo.sample_zero_crossing(x, 1.0 / 8, 0, False, 0.05 + 0j, 1 + 0j)
    out["zc_phase"] = sdo.sample_zero_crossing(x, 1.0 / 8, 1, False, 0j, -1j)
    out["zc_frequency"] = sdo.sample_zero_crossing(x, 1.0 / 8, 2)
    out["conj_prev"] = sdo.conj_prev(x[:2048], 0.5 - 0.25j)
    out["manual_amp"] = sdo.sample_manual(x[:4096], 500.0, 3, 0)
    out["manual_freq"] = sdo.sample_manual(x[:4096], 500.0, 3, 2)
    h = sdo.rrc_design(8.0, 0.35)
    out["rrc_taps"] = h
    out["matched"] = sdo.fir_feed(np.zeros(h.size - 1, np.complex64), h, x[:3000])
    cd = sdo.clock_new(0.0, 1.0 / 8)
    cd.phi = np.float32(0.5) * np.float32(0.3)
    sym = sdo.clock_feed_bulk(cd, out["matched"])
    out["manual_clock"] = sym
    q = sdo.cma_new(8, 2e-3)
    out["cma"] = sdo.cma_feed_bulk(q, sym)
    out["cma_weights"] = np.array(q.w[:16], dtype=np.float32)
    out["scale"] = sdo.scale(x[:1000], 0.37)
    rng = np.random.default_rng(5)
    raw8 = rng.integers(0, 256, 2 * 999).astype(np.uint8)
    raw16 = rng.integers(-32768, 32768, 2 * 999).astype(np.int16)
    out["raw_u8"], out["raw_s16"] = raw8, raw16
    out["ingest_u8"] = sdo.ingest_iq(2, raw8)
    out["ingest_s8"] = sdo.ingest_iq(3, raw8.view(np.int8))
    out["ingest_s16"] = sdo.ingest_iq(4, raw16)
    for k, name in enumerate(sdo.SPECTSRC, start=1):
        out["spectsrc_" + name] = sdo.spectsrc_preproc(k, x[:2048], 0.1 + 0.2j)
    f = sdo.FAC(1024, 0.5)
    f.feed(x[:1024], 2, 500)
    f.feed(x[1024:2048] * np.complex64(2), 2, 500)
    out["fac_1024"] = f.fac.copy()
    out["fac_range"] = np.array([f.min.value, f.max.value], dtype=np.float32)
    return out


V3_CHANNELS = [  # (f0, bw, guard, precise): sizes 64, 64, 16, 8 (narrow form), 256 and 2 (wide form)
    (0.83, 2 * np.pi * 0.75 / 64, 1.0, False), (4.91, 2 * np.pi * 0.5 / 64, 1.0, True), (2.2, 2 * np.pi * 0.8 / 256, 1.0, False),
    (5.9, 2 * np.pi * 0.7 / 512, 1.0, True), (1.4, 2 * np.pi * 0.75 / 16, 1.0, True), (3.3, 2 * np.pi * 0.9 / 2048, 1.0, False)]


def compute_v3(sdo):
    """third fixture set (round 3): the FFT channeliser in its binary32 statement (SPEC.md C2, "binary32 arithmetic") -- what
    the device kernels equal bit for bit -- with the default chain behind one of its channels, and the channel designs"""
    from sigdigger_amd import synth
    out = {}
    x = synth.psk_carriers(2048 * 9, [0.83 / np.pi, (4.91 - 2 * np.pi) / np.pi], sps=256, order=4, seed=321, snr_db=25)
    out["input_iq"] = x
    rows = sdo.specttuner_bank_f32(x, [c[0] for c in V3_CHANNELS], [c[1] for c in V3_CHANNELS], [c[2] for c in V3_CHANNELS],
                                   [c[3] for c in V3_CHANNELS])
    for k, r in enumerate(rows):
        out[f"st32_row{k}"] = r
    geom, hk = [], []
    for f0, bw, guard, _ in V3_CHANNELS:
        g = sdo.specttuner_geometry(4096, f0, bw, guard)
        geom.append([g.size, g.halfsz, g.halfw, g.decimation, g.center, g.dphase])
        hk.append(sdo.specttuner_response(4096, g.size, g.halfw))
    out["st_geometry"] = np.array(geom, dtype=np.uint32)
    out["st_response_64"] = hk[0]
    out["st_response_256"] = hk[2].astype(np.complex64)
    a = sdo.agc_feed_bulk(sdo.agc_new(sdo.agc_params_from_tau(4.0)), rows[0])
    z = sdo.costas_feed_bulk(sdo.costas_new(2, 0.0, 0.5, 3, 0.01), a)
    out["st32_chain_symbols"] = sdo.clock_feed_bulk(sdo.clock_new(0.2, 0.25), z)
    return out


def main():
    from oracle import sdo
    np.savez_compressed(os.path.join(HERE, "oracle_v3.npz"), **compute_v3(sdo))
    print("wrote", os.path.join(HERE, "oracle_v3.npz"))
    np.savez_compressed(os.path.join(HERE, "oracle_v1.npz"), **compute(sdo))
    print("wrote", os.path.join(HERE, "oracle_v1.npz"))
    np.savez_compressed(os.path.join(HERE, "oracle_v2.npz"), **compute_v2(sdo))
    print("wrote", os.path.join(HERE, "oracle_v2.npz"))


if __name__ == "__main__":
    main()
