/* refshim: <analyzer/inspector/inspector.h> (absent): the interface registry listed by Suscan/Library.cpp:923-926 */
#ifndef REFSHIM_INSPECTOR_H
#define REFSHIM_INSPECTOR_H
#include <sigutils/types.h>
#include <suscan_amd.h>
#ifdef __cplusplus
extern "C" {
#endif
struct suscan_inspector_interface { const char *name; const char *desc; };
void suscan_inspector_interface_get_list(const struct suscan_inspector_interface ***list, unsigned int *count);
#ifdef __cplusplus
}
#endif
#endif
