/* include/analyzer/spectsrc.h -- <analyzer/spectsrc.h> (include/Suscan/Library.h:35): struct suscan_spectsrc_class and
 * suscan_spectsrc_class_lookup (Suscan/Messages/InspectorMessage.cpp:46).  Declared in suscan_amd.h. */
#ifndef SIGDIGGER_AMD_ANALYZER_SPECTSRC_H
#define SIGDIGGER_AMD_ANALYZER_SPECTSRC_H
#include <sigutils/types.h>
#include "../suscan_amd.h"
#endif
