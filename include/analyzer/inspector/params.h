/* include/analyzer/inspector/params.h -- <analyzer/inspector/params.h>: the values of the integer config fields the
 * inspector controls write (Default/GenericInspector/InspectorCtl/MfControl.cpp:56-72, EqualizerControl.cpp:56-70,
 * ClockRecovery.cpp:66-86, AfcControl.cpp:54-83).  csrc/analyzer.cpp reads mf.type / clock.type / equalizer.type /
 * afc.costas-order with these meanings. */
#ifndef SIGDIGGER_AMD_ANALYZER_INSPECTOR_PARAMS_H
#define SIGDIGGER_AMD_ANALYZER_INSPECTOR_PARAMS_H
enum suscan_inspector_gain_control    { SUSCAN_INSPECTOR_GAIN_CONTROL_MANUAL = 0, SUSCAN_INSPECTOR_GAIN_CONTROL_AUTOMATIC = 1 };
enum suscan_inspector_carrier_control { SUSCAN_INSPECTOR_CARRIER_CONTROL_MANUAL = 0, SUSCAN_INSPECTOR_CARRIER_CONTROL_COSTAS_2 = 1,
                                        SUSCAN_INSPECTOR_CARRIER_CONTROL_COSTAS_4 = 2, SUSCAN_INSPECTOR_CARRIER_CONTROL_COSTAS_8 = 3 };
enum suscan_inspector_matched_filter  { SUSCAN_INSPECTOR_MATCHED_FILTER_BYPASS = 0, SUSCAN_INSPECTOR_MATCHED_FILTER_MANUAL = 1 };
enum suscan_inspector_equalizer       { SUSCAN_INSPECTOR_EQUALIZER_BYPASS = 0, SUSCAN_INSPECTOR_EQUALIZER_CMA = 1 };
enum suscan_inspector_baudrate_control { SUSCAN_INSPECTOR_BAUDRATE_CONTROL_MANUAL = 0, SUSCAN_INSPECTOR_BAUDRATE_CONTROL_GARDNER = 1 };
#endif
