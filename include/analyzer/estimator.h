/* include/analyzer/estimator.h -- <analyzer/estimator.h> (include/Suscan/Library.h:34): struct suscan_estimator_class
 * and suscan_estimator_class_lookup (Suscan/Messages/InspectorMessage.cpp:55).  Declared in suscan_amd.h. */
#ifndef SIGDIGGER_AMD_ANALYZER_ESTIMATOR_H
#define SIGDIGGER_AMD_ANALYZER_ESTIMATOR_H
#include <sigutils/types.h>
#include "../suscan_amd.h"
#endif
