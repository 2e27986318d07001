/* refshim: <suscan/util/cfg.h> / <cfg.h>: suscan_config_t and friends are product types (suscan_amd.h) */
#ifndef REFSHIM_SUSCAN_CFG_H
#define REFSHIM_SUSCAN_CFG_H
#include <sigutils/types.h>
#include <suscan_amd.h>
#endif
