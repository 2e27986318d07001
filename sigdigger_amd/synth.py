"""Deterministic synthetic IQ generators for tests and bench.py.

Harness input only (the reference's analogue is suscan's tonegen / file source,
Default/SourceConfig/ToneGenSourcePage.cpp:81-90); numpy, host side, float32 output.
"""
import numpy as np


def tone_noise(n, f_rel=0.1, sigma2=1e-3, seed=1):
    """S1: unit-power complex tone at f_rel*fs plus AWGN (sigma2 per component)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    x = np.exp(2j * np.pi * f_rel * t)
    x = x + np.sqrt(sigma2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def _rrc(beta, sps, span):
    """root-raised-cosine taps, unit energy"""
    n = np.arange(-span * sps, span * sps + 1, dtype=np.float64)
    t = n / sps
    h = np.empty_like(t)
    for i, ti in enumerate(t):
        if abs(ti) < 1e-12:
            h[i] = 1.0 - beta + 4 * beta / np.pi
        elif beta > 0 and abs(abs(ti) - 1 / (4 * beta)) < 1e-9:
            h[i] = beta / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * beta)) +
                                        (1 - 2 / np.pi) * np.cos(np.pi / (4 * beta)))
        else:
            h[i] = (np.sin(np.pi * ti * (1 - beta)) + 4 * beta * ti * np.cos(np.pi * ti * (1 + beta))) / \
                   (np.pi * ti * (1 - (4 * beta * ti) ** 2))
    return h / np.sqrt(np.sum(h * h))


def psk_carriers(n, fnor, sps, order=4, beta=0.35, snr_db=20.0, seed=2, amp=None):
    """S2/S4: sum of RRC-shaped M-PSK carriers.

    fnor: iterable of carrier centres (normalised, 2f/fs); sps: samples per symbol at the
    input rate (int); each carrier gets its own seeded symbol stream.
    """
    fnor = np.atleast_1d(np.asarray(fnor, dtype=np.float64))
    h = _rrc(beta, sps, 6) * np.sqrt(sps)
    nsym = n // sps + 14
    t = np.arange(n, dtype=np.float64)
    x = np.zeros(n, dtype=np.complex128)
    for c, f in enumerate(fnor):
        rng = np.random.default_rng(seed * 1000 + c)
        sym = rng.integers(0, order, nsym)
        ph = np.exp(1j * (2 * np.pi * sym / order + (np.pi / 4 if order == 4 else 0.0)))
        up = np.zeros(nsym * sps, dtype=np.complex128)
        up[::sps] = ph
        bb = np.convolve(up, h)[6 * sps: 6 * sps + n]
        a = 1.0 if amp is None else amp[c]
        x += a * bb * np.exp(1j * (np.pi * f * t + 0.37 * (c + 1)))
    rng = np.random.default_rng(seed)
    npow = 10 ** (-snr_db / 10)
    x += np.sqrt(npow / 2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def fsk_carriers(n, fnor, sps, h_index=1.0, snr_db=20.0, seed=3):
    """S3: sum of continuous-phase 2-FSK carriers (modulation index h_index)."""
    fnor = np.atleast_1d(np.asarray(fnor, dtype=np.float64))
    t = np.arange(n, dtype=np.float64)
    x = np.zeros(n, dtype=np.complex128)
    nsym = n // sps + 2
    for c, f in enumerate(fnor):
        rng = np.random.default_rng(seed * 1000 + 100 + c)
        bits = rng.integers(0, 2, nsym) * 2 - 1
        dev = np.repeat(bits, sps)[:n] * (np.pi * h_index / sps)
        ph = np.cumsum(dev)
        x += np.exp(1j * (ph + np.pi * f * t + 0.11 * c))
    rng = np.random.default_rng(seed)
    npow = 10 ** (-snr_db / 10)
    x += np.sqrt(npow / 2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def raster(nchan, spacing_nor):
    """uniform channel raster centred on 0 (normalised frequencies)"""
    return (np.arange(nchan) - (nchan - 1) / 2.0) * spacing_nor
