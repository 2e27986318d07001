"""ctypes loader for libsigdigger_amd.so -- the C ABI declared in include/sigdigger_amd.h.

The library is the product; this module only declares prototypes.  There is no Python or
CPU implementation of any operation behind it: if the .so is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libsigdigger_amd.so")

_lib = None

VP = C.c_void_p
U32 = C.c_uint32
U64 = C.c_uint64
F32 = C.c_float
F64 = C.c_double
INT = C.c_int
UINT = C.c_uint


class View(C.Structure):
    """suamd_view: element (c, m) at base[c*chan_stride + m*time_stride]"""
    _fields_ = [("chan_stride", U64), ("time_stride", U64)]


class Channel(C.Structure):
    """struct suamd_channel"""
    _fields_ = [("fc", F64), ("f_lo", F64), ("f_hi", F64), ("bw", F32), ("snr", F32), ("S0", F32), ("N0", F32), ("age", UINT)]


class AgcParams(C.Structure):
    """struct suamd_agc_params"""
    _fields_ = [("threshold", F32), ("slope_factor", F32), ("hang_max", UINT),
                ("delay_line_size", UINT), ("mag_history_size", UINT),
                ("fast_rise_t", F32), ("fast_fall_t", F32), ("slow_rise_t", F32), ("slow_fall_t", F32)]


# name -> (restype, argtypes); mirrors include/sigdigger_amd.h one to one
PROTOTYPES = {
    "suamd_last_error": (C.c_char_p, []),
    "suamd_version": (C.c_char_p, []),
    "suamd_kernel_timing": (None, [C.c_int]),
    "suamd_tuning_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "suamd_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "suamd_tuning_describe": (C.c_int, [C.c_uint, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                        C.POINTER(C.c_longlong), C.POINTER(C.c_char_p)]),
    "suamd_tuning_reset": (None, []),
    "suamd_kernel_timing_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint)]),
    "suamd_ctx_new": (VP, [INT]),
    "suamd_ctx_destroy": (None, [VP]),
    "suamd_ctx_device": (INT, [VP]),
    "suamd_ctx_cu_count": (UINT, [VP]),
    "suamd_stream_new_cu_mask": (VP, [VP, C.POINTER(U32), UINT]),
    "suamd_stream_destroy": (INT, [VP, VP]),
    "suamd_probe_placement": (INT, [VP, VP, UINT, UINT, C.POINTER(U32)]),
    "suamd_psd_new": (VP, [VP, UINT, INT]),
    "suamd_psd_destroy": (None, [VP]),
    "suamd_psd_set_split_target": (INT, [VP, UINT]),
    "suamd_psd_feed": (INT, [VP, VP, U64, U64, UINT, F32, INT, VP, VP]),
    "suamd_psd_shift_db": (INT, [VP, VP, U64, U64, VP]),
    "suamd_averager_feed": (INT, [VP, VP, VP, U64, F32, INT, VP]),
    "suamd_inspector_spectrum_db_shift": (INT, [VP, VP, U64, U64, VP]),
    "suamd_audio_new": (VP, [VP, F32, F32]),
    "suamd_audio_destroy": (None, [VP]),
    "suamd_audio_configure": (INT, [VP, INT, F32, F32, F32, INT, F32]),
    "suamd_audio_output_count": (U64, [VP, U64]),
    "suamd_audio_feed": (INT, [VP, VP, U64, VP, C.POINTER(U64), VP]),
    "suamd_chandet_new": (VP, [VP, UINT, F32, F32, F32, F32]),
    "suamd_chandet_destroy": (None, [VP]),
    "suamd_chandet_feed": (INT, [VP, VP, VP]),
    "suamd_chandet_channels": (INT, [VP, F32, VP, UINT, VP]),
    "suamd_chandet_find": (INT, [VP, INT, VP]),
    "suamd_chandet_collect": (INT, [VP, INT, F32, VP, UINT]),
    "suamd_chandet_noise_floor": (F32, [VP, VP]),
    "suamd_specttuner_new": (VP, [VP, UINT]),
    "suamd_specttuner_destroy": (None, [VP]),
    "suamd_specttuner_open_channel": (INT, [VP, F64, F64, F64, INT]),
    "suamd_specttuner_close_channel": (INT, [VP, INT]),
    "suamd_specttuner_channel_size": (UINT, [VP, INT]),
    "suamd_specttuner_channel_decimation": (UINT, [VP, INT]),
    "suamd_specttuner_feed": (INT, [VP, VP, U64, VP, View, C.POINTER(U64), VP]),
    "suamd_specttuner_feed_rows": (INT, [VP, VP, U64, VP, C.POINTER(U64), VP]),
    "suamd_specttuner_feed_rows_near": (INT, [VP, VP, U64, VP, VP, C.c_size_t, C.POINTER(U64), VP]),
    "suamd_specttuner_feed_mixed": (INT, [VP, VP, U64, VP, View, UINT, VP, VP, C.c_size_t, C.POINTER(U64), VP]),
    "suamd_specttuner_set_run": (INT, [VP, UINT]),
    "suamd_specttuner_set_slots": (INT, [VP, UINT]),
    "suamd_specttuner_channel_capacity": (UINT, [VP]),
    "suamd_specttuner_reset": (INT, [VP, VP]),
    "suamd_specttuner_design": (INT, [UINT, F64, F64, F64, C.POINTER(U32), VP]),
    "suamd_fnor_to_dphase": (U32, [F64]),
    "suamd_xlate_bulk": (INT, [VP, VP, VP, U64, U32, U32, U64, VP]),
    "suamd_lpf_design": (None, [VP, UINT, F64]),
    "suamd_chanbank_new": (VP, [VP, UINT, VP, UINT, VP, UINT]),
    "suamd_chanbank_destroy": (None, [VP]),
    "suamd_chanbank_output_count": (U64, [VP, U64]),
    "suamd_chanbank_feed": (INT, [VP, VP, U64, VP, View, C.POINTER(U64), VP]),
    "suamd_chanbank_reset": (INT, [VP, VP]),
    "suamd_chanbank_set_exclusive": (INT, [VP, INT]),
    "suamd_quad_demod_batch": (INT, [VP, VP, View, VP, View, UINT, U64, VP, INT, VP, VP]),
    "suamd_delayed_conj_bulk": (INT, [VP, VP, VP, U64, U64, VP]),
    "suamd_histogram_feed_bulk": (INT, [VP, VP, U64, INT, VP, VP]),
    "suamd_sample_manual_bulk": (INT, [VP, VP, U64, F64, U64, INT, VP, U64, VP]),
    "suamd_sample_zero_crossing_bulk": (C.c_int64, [VP, VP, U64, F32, INT, INT, F32, F32, F32, F32, VP, U64, VP]),
    "suamd_conj_prev_bulk": (INT, [VP, VP, VP, U64, F32, F32, VP]),
    "suamd_ingest_iq": (INT, [VP, INT, VP, U64, VP, VP]),
    "suamd_decision_space": (INT, [VP, VP, U64, INT, VP, VP]),
    "suamd_decide": (INT, [VP, VP, U64, INT, UINT, F32, F32, VP, VP]),
    "suamd_symbol_histogram": (INT, [VP, VP, U64, INT, F32, F32, UINT, VP, VP]),
    "suamd_snr_estimator_new": (VP, [VP, UINT, F32]),
    "suamd_snr_estimator_destroy": (None, [VP]),
    "suamd_snr_estimator_feed": (INT, [VP, VP, UINT, VP]),
    "suamd_snr_estimator_get": (INT, [VP, VP, VP, VP, VP]),
    "suamd_snr_estimator_model": (VP, [VP]),
    "suamd_spectsrc_count": (UINT, []),
    "suamd_spectsrc_name": (C.c_char_p, [UINT]),
    "suamd_spectsrc_preproc": (INT, [VP, UINT, VP, U64, F32, F32, VP, VP]),
    "suamd_fac_new": (VP, [VP, UINT, F32]),
    "suamd_fac_destroy": (None, [VP]),
    "suamd_fac_set_alpha": (None, [VP, F32]),
    "suamd_fac_reset": (INT, [VP, VP]),
    "suamd_fac_feed": (INT, [VP, VP, U64, C.c_int64, C.c_int64, VP]),
    "suamd_fac_array": (VP, [VP]),
    "suamd_fac_get_range": (INT, [VP, VP, VP, VP]),
    "suamd_costas_gang_feed": (INT, [VP, VP, UINT, VP, VP, VP, VP]),
    "suamd_agc_gang_feed": (INT, [VP, VP, UINT, VP, VP, VP, VP]),
    "suamd_pll_gang_feed": (INT, [VP, VP, UINT, VP, VP, VP, VP]),
    "suamd_cma_gang_feed": (INT, [VP, VP, UINT, VP, VP, VP, VP, VP]),
    "suamd_agc_gang_pre": (INT, [VP, VP, UINT, VP, VP, VP]),
    "suamd_agc_gang_level": (INT, [VP, VP, UINT, VP, VP, VP, VP]),
    "suamd_agc_gang_apply": (INT, [VP, VP, UINT, VP, VP, VP, VP, VP, VP]),
    "suamd_agc_gang_finish": (INT, [VP, VP, UINT, VP, VP, VP]),
    "suamd_baud_estimator_new": (VP, [VP, INT, UINT]),
    "suamd_baud_estimator_destroy": (None, [VP]),
    "suamd_baud_estimator_size": (UINT, [VP]),
    "suamd_baud_estimator_feed": (INT, [VP, VP, U64, VP]),
    "suamd_baud_estimator_get": (C.c_float, [VP]),
    "suamd_baud_estimator_feed_to": (INT, [VP, VP, U64, VP, VP]),
    "suamd_spectsrc_preproc_from": (INT, [VP, UINT, VP, U64, VP, VP, VP]),
    "suamd_power_bank_new": (VP, [VP, U64]),
    "suamd_power_bank_destroy": (None, [VP]),
    "suamd_power_bank_set_integrate": (INT, [VP, U64, VP]),
    "suamd_power_bank_output_count": (U64, [VP, U64]),
    "suamd_power_bank_feed": (INT, [VP, VP, U64, VP, VP, VP]),
    "suamd_psd_ttl_accept": (INT, [VP, C.c_double, C.c_double, C.c_double, INT]),
    "suamd_export_capture": (INT, [VP, C.c_char_p, C.c_char_p, VP, U64, C.c_float, VP]),
    "suamd_source_fix": (INT, [VP, VP, U64, INT, VP, C.c_float, INT, VP]),
    "suamd_chanbank_gang_feed": (INT, [VP, VP, UINT, VP, U64, VP, VP, VP]),
    "suamd_rows_deliver": (INT, [VP, UINT, VP, VP, VP, VP, VP, VP]),
    "suamd_clock_gang_feed": (INT, [VP, VP, UINT, VP, VP, VP, VP, VP]),
    "suamd_costas_gang_feed_slab": (INT, [VP, VP, UINT, VP, U64, VP, U64, VP, VP]),
    "suamd_pll_gang_feed_slab": (INT, [VP, VP, UINT, VP, U64, VP, U64, VP, VP]),
    "suamd_clock_gang_feed_slab": (INT, [VP, VP, UINT, VP, U64, VP, VP, VP, VP]),
    "suamd_agc_gang_pre_slab": (INT, [VP, VP, UINT, VP, U64, VP, VP, VP, U64, VP]),
    "suamd_agc_gang_level_slab": (INT, [VP, VP, UINT, VP, U64, VP, VP, VP, VP, VP, U64, VP]),
    "suamd_agc_gang_apply_slab": (INT, [VP, VP, UINT, VP, U64, VP, VP, U64, VP, VP, VP, VP, VP, U64, VP]),
    "suamd_agc_gang_finish_slab": (INT, [VP, VP, UINT, VP, U64, VP, VP, VP, U64, VP]),
    "suamd_rows_deliver_strided": (INT, [VP, UINT, VP, VP, VP, VP, VP, VP, VP]),
    "suamd_rows_scale": (INT, [VP, VP, View, VP, View, UINT, U64, F32, VP]),
    "suamd_nco_bank_new": (VP, [VP, UINT, VP]),
    "suamd_nco_bank_destroy": (None, [VP]),
    "suamd_nco_bank_feed": (INT, [VP, VP, View, VP, View, U64, VP]),
    "suamd_rrc_ntaps": (UINT, [F64]),
    "suamd_rrc_design": (None, [VP, UINT, F64, F64]),
    "suamd_fir_bank_new": (VP, [VP, UINT, VP, UINT]),
    "suamd_fir_bank_destroy": (None, [VP]),
    "suamd_fir_bank_feed": (INT, [VP, VP, View, VP, View, U64, VP]),
    "suamd_cma_bank_new": (VP, [VP, UINT, UINT, F32]),
    "suamd_cma_bank_destroy": (None, [VP]),
    "suamd_cma_bank_set_locked": (None, [VP, INT]),
    "suamd_cma_bank_set_rate": (None, [VP, F32]),
    "suamd_cma_bank_feed": (INT, [VP, VP, U64, VP, U64, VP, U64, VP]),
    "suamd_cma_bank_get_weights": (INT, [VP, VP, VP]),
    "suamd_clock_bank_set_phase": (INT, [VP, F32, VP]),
    "suamd_format_bytes_per_sample": (UINT, [INT]),
    "suamd_costas_bank_new": (VP, [VP, UINT, INT, F32, F32, UINT, F32]),
    "suamd_costas_bank_destroy": (None, [VP]),
    "suamd_costas_bank_feed": (INT, [VP, VP, View, VP, View, U64, VP]),
    "suamd_costas_bank_get_state": (INT, [VP, VP, VP, VP]),
    "suamd_pll_bank_new": (VP, [VP, UINT, F32, F32]),
    "suamd_pll_bank_destroy": (None, [VP]),
    "suamd_pll_bank_feed": (INT, [VP, VP, View, VP, View, U64, VP]),
    "suamd_pll_bank_get_state": (INT, [VP, VP, VP, VP]),
    "suamd_clock_bank_new": (VP, [VP, UINT, F32, F32]),
    "suamd_clock_bank_destroy": (None, [VP]),
    "suamd_clock_bank_feed": (INT, [VP, VP, View, U64, VP, U64, VP, VP]),
    "suamd_clock_bank_get_state": (INT, [VP, VP, VP, VP]),
    "suamd_agc_params_from_tau": (None, [C.POINTER(AgcParams), F32]),
    "suamd_agc_bank_new": (VP, [VP, UINT, C.POINTER(AgcParams)]),
    "suamd_agc_bank_destroy": (None, [VP]),
    "suamd_agc_bank_feed": (INT, [VP, VP, View, VP, View, U64, VP]),
    "suamd_agc_bank_feed_split": (INT, [VP, VP, View, VP, View, U64, VP, VP]),
    "suamd_fft_forward_bulk": (INT, [VP, VP, VP, VP, UINT, VP]),
    "suamd_carrier_detect": (INT, [VP, VP, U64, F32, F32, C.POINTER(F32), VP]),
    "suamd_doppler_alloc_size": (U64, [U64]),
    "suamd_doppler_calc": (INT, [VP, VP, U64, F32, F64, VP, C.POINTER(F32), C.POINTER(F32), C.POINTER(F32), VP]),
    "suamd_specview_new": (VP, [VP]),
    "suamd_specview_destroy": (None, [VP]),
    "suamd_specview_set_range": (INT, [VP, F64, F64, VP]),
    "suamd_specview_set_fft": (None, [VP, F64, F32]),
    "suamd_specview_spectrum_size": (UINT, [VP]),
    "suamd_specview_feed": (INT, [VP, VP, VP, U64, F64, F64, INT, VP]),
    "suamd_specview_feed_sweep": (INT, [VP, VP, U64, U64, VP, INT, VP]),
    "suamd_specview_array": (VP, [VP, INT]),
}


class SigDiggerAmdError(RuntimeError):
    pass


def load():
    """Loads the shared library (once) and installs the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise SigDiggerAmdError(
            f"{SO_PATH} is missing: build it with `python -m sigdigger_amd.build` "
            "(there is no CPU fallback for the SigDigger hot path)")
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().suamd_last_error().decode("utf-8", "replace")


def check(ok, what):
    if not ok:
        raise SigDiggerAmdError(f"{what}: {last_error()}")
    return ok
