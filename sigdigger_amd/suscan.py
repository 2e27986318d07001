"""ctypes binding of the live-path ABI (include/suscan_amd.h): what SigDigger's Suscan::Analyzer
wrapper (Suscan/Analyzer.cpp) does, for Python callers and the tests.  Plumbing only."""
import ctypes as C

from . import lib as _l

MSG_SOURCE_INFO, MSG_SOURCE_INIT, MSG_CHANNEL, MSG_EOS, MSG_READ_ERROR, MSG_INTERNAL = 0, 1, 2, 3, 4, 5
MSG_INSPECTOR, MSG_PSD, MSG_SAMPLES, MSG_PARAMS = 7, 8, 9, 0xB
MSG_HALT = 0xFFFFFFFF
KIND_OPEN, KIND_SET_ID, KIND_GET_CONFIG, KIND_SET_CONFIG = 0, 1, 2, 3
KIND_ESTIMATOR, KIND_SPECTRUM, KIND_INVALID_ARGUMENT = 4, 5, 13
KIND_CLOSE, KIND_SET_WATERMARK, KIND_WRONG_HANDLE, KIND_WRONG_KIND, KIND_INVALID_CHANNEL = 7, 10, 11, 14, 15


class Timeval(C.Structure):
    _fields_ = [("tv_sec", C.c_long), ("tv_usec", C.c_long)]


class MQ(C.Structure):
    _fields_ = [("impl", C.c_void_p)]


class DetectorParams(C.Structure):
    _fields_ = [("alpha", C.c_float), ("beta", C.c_float), ("gamma", C.c_float), ("snr", C.c_float),
                ("window_size", C.c_uint64), ("window", C.c_int)]


class AnalyzerParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("detector_params", DetectorParams), ("channel_update_int", C.c_float),
                ("psd_update_int", C.c_float), ("min_freq", C.c_double), ("max_freq", C.c_double)]

    @classmethod
    def default(cls):
        return cls(0, DetectorParams(1e-2, 1e-3, .5, 2., 4096, 4), .1, .04, -1, -1)


class Channel(C.Structure):
    """struct sigutils_channel"""
    _fields_ = [("fc", C.c_double), ("f_lo", C.c_double), ("f_hi", C.c_double), ("bw", C.c_float),
                ("snr", C.c_float), ("S0", C.c_float), ("N0", C.c_float), ("ft", C.c_double),
                ("age", C.c_uint32), ("present", C.c_uint32)]


class PSDMsg(C.Structure):
    _fields_ = [("fc", C.c_int64), ("inspector_id", C.c_uint32), ("timestamp", Timeval), ("rt_time", Timeval),
                ("looped", C.c_int), ("history_size", C.c_uint64), ("samp_rate", C.c_float),
                ("measured_samp_rate", C.c_float), ("N0", C.c_float), ("psd_size", C.c_uint64),
                ("psd_data", C.POINTER(C.c_float))]


class SampleBatchMsg(C.Structure):
    _fields_ = [("inspector_id", C.c_uint32), ("samples", C.POINTER(C.c_float)), ("sample_count", C.c_uint64)]


class OrbitReport(C.Structure):
    """struct suscan_orbit_report"""
    _fields_ = [("rx_time", Timeval), ("satpos", C.c_double * 3), ("freq_corr", C.c_float), ("vlos_vel", C.c_double)]


class ChannelMsg(C.Structure):
    """struct suscan_analyzer_channel_msg"""
    _fields_ = [("channel_list", C.POINTER(C.POINTER(Channel))), ("channel_count", C.c_uint)]


class InspectorMsg(C.Structure):
    _fields_ = [("kind", C.c_int), ("inspector_id", C.c_uint32), ("req_id", C.c_uint32), ("handle", C.c_int32),
                ("status", C.c_int), ("class_name", C.c_char_p), ("channel", Channel), ("config", C.c_void_p),
                ("precise", C.c_int), ("fs", C.c_uint32), ("equiv_fs", C.c_float), ("bandwidth", C.c_float),
                ("lo", C.c_float), ("spectsrc_count", C.c_uint), ("spectsrc_list", C.c_void_p),
                ("estimator_count", C.c_uint), ("estimator_list", C.c_void_p), ("spectsrc_id", C.c_uint32),
                ("estimator_id", C.c_uint32), ("spectrum_data", C.c_void_p), ("spectrum_size", C.c_uint64),
                ("samp_rate", C.c_uint64), ("watermark", C.c_uint64), ("enabled", C.c_int), ("value", C.c_float),
                ("orbit_report", OrbitReport), ("tle_enable", C.c_int), ("signal_name", C.c_char_p),
                ("signal_value", C.c_double)]


class SourceInfo(C.Structure):
    """struct suscan_source_info"""
    _fields_ = [("permissions", C.c_uint64), ("source_samp_rate", C.c_uint64), ("effective_samp_rate", C.c_uint64),
                ("measured_samp_rate", C.c_float), ("frequency", C.c_double), ("freq_min", C.c_double), ("freq_max", C.c_double),
                ("lnb", C.c_double), ("bandwidth", C.c_float), ("ppm", C.c_float), ("antenna", C.c_char_p),
                ("dc_remove", C.c_int), ("iq_reverse", C.c_int), ("agc", C.c_int), ("seekable", C.c_int),
                ("replay", C.c_int), ("history_length", C.c_uint64), ("source_start", Timeval), ("source_end", Timeval),
                ("gain_list", C.c_void_p), ("gain_count", C.c_uint), ("antenna_list", C.c_void_p), ("antenna_count", C.c_uint)]


class EstimatorClass(C.Structure):
    _fields_ = [("name", C.c_char_p), ("desc", C.c_char_p), ("field", C.c_char_p)]


# SUBOOL f(void *privdata, suscan_analyzer_t *, SUCOMPLEX *samples, SUSCOUNT length, SUSCOUNT offset)
BASEBAND_FILTER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.c_uint64)
KIND_WRONG_OBJECT, KIND_SET_TLE = 12, 17


class StatusMsg(C.Structure):
    _fields_ = [("code", C.c_int), ("err_msg", C.c_char_p)]


VP, U32, U64, INT = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
# SUBOOL f(const suscan_source_config_t *, const char *key, const char *value, void *userdata)
WALK_PARAMS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p)
PROTOTYPES = {
    "suscan_mq_init": (INT, [C.POINTER(MQ)]),
    "suscan_mq_finalize": (None, [C.POINTER(MQ)]),
    "suscan_mq_read": (VP, [C.POINTER(MQ), C.POINTER(U32)]),
    "suscan_mq_poll": (INT, [C.POINTER(MQ), C.POINTER(U32), C.POINTER(VP)]),
    "suscan_mq_write": (INT, [C.POINTER(MQ), U32, VP]),
    "suscan_source_config_new": (VP, [C.c_char_p, INT]),
    "suscan_source_config_destroy": (None, [VP]),
    "suscan_source_config_set_samp_rate": (None, [VP, C.c_uint]),
    "suscan_source_config_set_freq": (None, [VP, C.c_double]),
    "suscan_source_config_set_path": (INT, [VP, C.c_char_p]),
    "suscan_source_config_set_loop": (None, [VP, INT]),
    "suscan_source_config_set_param": (INT, [VP, C.c_char_p, C.c_char_p]),
    "suscan_source_config_clone": (VP, [VP]),
    "suscan_source_config_get_label": (C.c_char_p, [VP]),
    "suscan_source_config_set_label": (INT, [VP, C.c_char_p]),
    "suscan_source_config_get_type": (C.c_char_p, [VP]),
    "suscan_source_config_get_format": (INT, [VP]),
    "suscan_source_config_set_type_format": (None, [VP, C.c_char_p, INT]),
    "suscan_source_config_get_path": (C.c_char_p, [VP]),
    "suscan_source_config_get_freq": (C.c_double, [VP]),
    "suscan_source_config_get_lnb_freq": (C.c_double, [VP]),
    "suscan_source_config_set_lnb_freq": (None, [VP, C.c_double]),
    "suscan_source_config_get_samp_rate": (C.c_uint, [VP]),
    "suscan_source_config_get_average": (C.c_uint, [VP]),
    "suscan_source_config_set_average": (INT, [VP, C.c_uint]),
    "suscan_source_config_get_bandwidth": (C.c_float, [VP]),
    "suscan_source_config_set_bandwidth": (None, [VP, C.c_float]),
    "suscan_source_config_get_ppm": (C.c_float, [VP]),
    "suscan_source_config_set_ppm": (None, [VP, C.c_float]),
    "suscan_source_config_get_loop": (INT, [VP]),
    "suscan_source_config_get_dc_remove": (INT, [VP]),
    "suscan_source_config_set_dc_remove": (None, [VP, INT]),
    "suscan_source_config_get_iq_balance": (INT, [VP]),
    "suscan_source_config_set_iq_balance": (None, [VP, INT]),
    "suscan_source_config_get_start_time": (None, [VP, C.POINTER(Timeval)]),
    "suscan_source_config_set_start_time": (None, [VP, Timeval]),
    "suscan_source_config_get_end_time": (INT, [VP, C.POINTER(Timeval)]),
    "suscan_source_config_file_is_valid": (INT, [VP]),
    "suscan_source_config_is_real_time": (INT, [VP]),
    "suscan_source_config_is_seekable": (INT, [VP]),
    "suscan_source_config_get_freq_limits": (INT, [VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "suscan_source_config_get_antenna": (C.c_char_p, [VP]),
    "suscan_source_config_set_antenna": (INT, [VP, C.c_char_p]),
    "suscan_source_config_get_gain": (C.c_float, [VP, C.c_char_p]),
    "suscan_source_config_set_gain": (INT, [VP, C.c_char_p, C.c_float]),
    "suscan_source_config_get_param": (C.c_char_p, [VP, C.c_char_p]),
    "suscan_source_config_clear_params": (None, [VP]),
    "suscan_source_config_walk_params": (INT, [VP, WALK_PARAMS, VP]),
    "suscan_inspector_config_desc": (VP, [C.c_char_p]),
    "suscan_config_new": (VP, [VP]),
    "suscan_config_dup": (VP, [VP]),
    "suscan_config_destroy": (None, [VP]),
    "suscan_config_get_value": (VP, [VP, C.c_char_p]),
    "suscan_config_set_integer": (INT, [VP, C.c_char_p, U64]),
    "suscan_config_set_float": (INT, [VP, C.c_char_p, C.c_float]),
    "suscan_config_set_bool": (INT, [VP, C.c_char_p, INT]),
    "suscan_config_set_string": (INT, [VP, C.c_char_p, C.c_char_p]),
    "suscan_config_desc_has_prefix": (INT, [VP, C.c_char_p]),
    "suscan_source_info_init": (None, [C.POINTER(SourceInfo)]),
    "suscan_source_info_init_copy": (INT, [C.POINTER(SourceInfo), C.POINTER(SourceInfo)]),
    "suscan_source_info_finalize": (None, [C.POINTER(SourceInfo)]),
    "suscan_analyzer_new": (VP, [C.POINTER(AnalyzerParams), VP, C.POINTER(MQ)]),
    "suscan_analyzer_destroy": (None, [VP]),
    "suscan_analyzer_read": (VP, [VP, C.POINTER(U32)]),
    "suscan_analyzer_dispose_message": (None, [U32, VP]),
    "suscan_analyzer_req_halt": (None, [VP]),
    "suscan_analyzer_set_params_async": (INT, [VP, C.POINTER(AnalyzerParams), U32]),
    "suscan_analyzer_set_throttle_async": (INT, [VP, U64, U32]),
    "suscan_analyzer_get_samp_rate": (C.c_uint, [VP]),
    "suscan_analyzer_get_measured_samp_rate": (C.c_float, [VP]),
    "suscan_analyzer_get_source_info": (VP, [VP]),
    "suscan_analyzer_open_async": (INT, [VP, C.c_char_p, C.POINTER(Channel), U32]),
    "suscan_analyzer_open_ex_async": (INT, [VP, C.c_char_p, C.POINTER(Channel), INT, C.c_int32, U32]),
    "suscan_analyzer_close_async": (INT, [VP, C.c_int32, U32]),
    "suscan_analyzer_set_inspector_id_async": (INT, [VP, C.c_int32, U32, U32]),
    "suscan_analyzer_set_inspector_config_async": (INT, [VP, C.c_int32, VP, U32]),
    "suscan_analyzer_set_inspector_watermark_async": (INT, [VP, C.c_int32, U64, U32]),
    "suscan_analyzer_inspector_set_spectrum_async": (INT, [VP, C.c_int32, U32, U32]),
    "suscan_spectsrc_class_lookup": (VP, [C.c_char_p]),
    "suscan_estimator_class_lookup": (VP, [C.c_char_p]),
    "suscan_analyzer_set_inspector_freq_overridable": (INT, [VP, C.c_int32, C.c_double]),
    "suscan_analyzer_set_inspector_bandwidth_overridable": (INT, [VP, C.c_int32, C.c_double]),
    "suscan_analyzer_inspector_estimator_cmd_async": (INT, [VP, C.c_int32, U32, INT, U32]),
    "suscan_analyzer_inspector_set_tle_async": (INT, [VP, C.c_int32, VP, U32]),
    "suscan_analyzer_register_baseband_filter": (INT, [VP, BASEBAND_FILTER, VP]),
    "suscan_analyzer_register_baseband_filter_with_prio": (INT, [VP, BASEBAND_FILTER, VP, C.c_int64]),
    "suscan_analyzer_set_freq": (INT, [VP, C.c_double, C.c_double]),
    "suscan_analyzer_set_gain": (INT, [VP, C.c_char_p, C.c_float]),
    "suscan_analyzer_set_antenna": (INT, [VP, C.c_char_p]),
    "suscan_analyzer_set_bw": (INT, [VP, C.c_float]),
    "suscan_analyzer_set_ppm": (INT, [VP, C.c_float]),
    "suscan_analyzer_set_agc": (INT, [VP, INT]),
    "suscan_analyzer_set_dc_remove": (INT, [VP, INT]),
    "suscan_analyzer_set_iq_reverse": (INT, [VP, INT]),
    "suscan_analyzer_seek": (INT, [VP, C.POINTER(Timeval)]),
    "suscan_analyzer_set_history_size": (INT, [VP, U64]),
    "suscan_analyzer_replay": (INT, [VP, INT]),
    "suscan_analyzer_get_source_time": (None, [VP, C.POINTER(Timeval)]),
    "suscan_analyzer_set_sweep_stratrgy": (INT, [VP, INT]),
    "suscan_analyzer_set_spectrum_partitioning": (INT, [VP, INT]),
    "suscan_analyzer_set_hop_range": (INT, [VP, C.c_double, C.c_double]),
    "suscan_analyzer_set_rel_bandwidth": (INT, [VP, C.c_float]),
    "suscan_analyzer_set_buffering_size": (INT, [VP, U64]),
}

_bound = None


class AnalyzerStalled(RuntimeError):
    pass


def read_message(lib, mq, timeout_s=60.0):
    """suscan_analyzer_read with a deadline (Analyzer::read blocks for ever; tests and benchmarks must not): polls the
    analyzer's queue and raises AnalyzerStalled if nothing arrives in time.  Returns (type, pointer)."""
    import time
    t, ptr = C.c_uint32(0), C.c_void_p()
    deadline = time.time() + timeout_s
    pause = 0.0
    while not lib.suscan_mq_poll(C.byref(mq), C.byref(t), C.byref(ptr)):
        if time.time() > deadline:
            raise AnalyzerStalled(f"no message from the analyzer for {timeout_s} s")
        time.sleep(pause)
        pause = min(0.002, pause + 0.0001)
    return t.value, ptr.value


def load():
    global _bound
    if _bound is None:
        L = _l.load()
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = L
    return _bound
