/* include/analyzer/source/info.h -- <analyzer/source/info.h> (include/Suscan/Messages/SourceInfoMessage.h:26,
 * include/Suscan/Device.h:30): struct suscan_source_info, struct suscan_source_gain_info and
 * suscan_source_info_init / _init_copy / _finalize (include/Suscan/Analyzer.h:50-105).  Declared in suscan_amd.h. */
#ifndef SIGDIGGER_AMD_ANALYZER_SOURCE_INFO_H
#define SIGDIGGER_AMD_ANALYZER_SOURCE_INFO_H
#include <sigutils/types.h>
#include "../../suscan_amd.h"
#endif
