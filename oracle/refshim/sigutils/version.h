/* refshim: <sigutils/version.h> */
#ifndef REFSHIM_SIGUTILS_VERSION_H
#define REFSHIM_SIGUTILS_VERSION_H
#define SU_VER(a, b, c) (((a) << 16) | ((b) << 8) | (c))
#define SIGUTILS_VERSION_STRING "0.3.0-amd"
#ifdef __cplusplus
extern "C" {
#endif
unsigned int sigutils_api_version(void);
const char *sigutils_pkgversion(void);
#ifdef __cplusplus
}
#endif
#endif
