/* include/sigutils/softtune.h -- <sigutils/softtune.h> (include/Suscan/Channel.h:22): struct sigutils_channel and its
 * INITIALIZER, as filled in by Analyzer::open* (Suscan/Analyzer.cpp:411-484).  Declared in suscan_amd.h. */
#ifndef SIGDIGGER_AMD_SIGUTILS_SOFTTUNE_H
#define SIGDIGGER_AMD_SIGUTILS_SOFTTUNE_H
#include "types.h"
#include "../suscan_amd.h"
#endif
