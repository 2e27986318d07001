"""Generates tests/golden/oracle_v1.npz from the CPU oracle.

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


def main():
    from oracle import sdo
    np.savez_compressed(os.path.join(HERE, "oracle_v1.npz"), **compute(sdo))
    print("wrote", os.path.join(HERE, "oracle_v1.npz"))


if __name__ == "__main__":
    main()
