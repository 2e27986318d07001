/* refshim: <suscan.h> */
#ifndef REFSHIM_SUSCAN_H
#define REFSHIM_SUSCAN_H
#include <analyzer/analyzer.h>
#include <analyzer/source.h>
#include <analyzer/version.h>
#endif
