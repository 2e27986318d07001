/* refshim: <analyzer/version.h> */
#ifndef REFSHIM_ANALYZER_VERSION_H
#define REFSHIM_ANALYZER_VERSION_H
#define SUSCAN_VERSION_STRING "0.3.0-amd"
#ifdef __cplusplus
extern "C" {
#endif
unsigned int suscan_api_version(void);
const char *suscan_pkgversion(void);
#ifdef __cplusplus
}
#endif
#endif
