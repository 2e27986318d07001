/*
 * include/analyzer/analyzer.h -- the <analyzer/analyzer.h> SigDigger includes (include/Suscan/Analyzer.h:42,
 * include/Suscan/Messages/PSDMessage.h:25, include/Suscan/AnalyzerParams.h:24 ...), served by libsigdigger_amd.so.
 *
 * A thin wrapper: everything is declared in suscan_amd.h (functions, message structs, enums, INITIALIZER macros,
 * each citing the reference call site); including <sigutils/types.h> first makes every SUCOMPLEX* in those
 * signatures the reference's own sample type.  tests/test_ref_compile.py compiles and links the reference's
 * Suscan/{Analyzer,MQ,Message,AnalyzerParams}.cpp and Suscan/Messages/*.cpp against this header unchanged and runs
 * the resulting Suscan::Analyzer against the GPU library.
 */
#ifndef SIGDIGGER_AMD_ANALYZER_ANALYZER_H
#define SIGDIGGER_AMD_ANALYZER_ANALYZER_H

#include <sigutils/types.h>
#include <sigutils/softtune.h>
#include "../suscan_amd.h"
#include "mq.h"
#include "msg.h"
#include "source/info.h"
#include "inspector/params.h"
#include "spectsrc.h"
#include "estimator.h"

#endif
