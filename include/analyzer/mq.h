/* include/analyzer/mq.h -- <analyzer/mq.h> (include/Suscan/MQ.h:26): struct suscan_mq, suscan_mq_init / _finalize /
 * _read (Suscan/MQ.cpp:28-43).  Declared in suscan_amd.h. */
#ifndef SIGDIGGER_AMD_ANALYZER_MQ_H
#define SIGDIGGER_AMD_ANALYZER_MQ_H
#include <sigutils/types.h>
#include "../suscan_amd.h"
#endif
