/* include/analyzer/msg.h -- <analyzer/msg.h> (include/Suscan/Message.h:27): the message type tags, payload structs and
 * suscan_analyzer_dispose_message (Suscan/Message.cpp:46).  Declared in suscan_amd.h. */
#ifndef SIGDIGGER_AMD_ANALYZER_MSG_H
#define SIGDIGGER_AMD_ANALYZER_MSG_H
#include <sigutils/types.h>
#include "../suscan_amd.h"
#endif
