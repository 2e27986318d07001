/* refshim: <suscan/plugin.h> (absent; control plane) */
#ifndef REFSHIM_SUSCAN_PLUGIN_H
#define REFSHIM_SUSCAN_PLUGIN_H
#include <sigutils/types.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct suscan_plugin suscan_plugin_t;
struct suscan_plugin_service_desc { const char *name; void *(*ctor)(suscan_plugin_t *); SUBOOL (*post_load)(void *); void (*dtor)(void *); };
const char *suscan_plugin_get_name(const suscan_plugin_t *);
const char *suscan_plugin_get_path(const suscan_plugin_t *);
const char *suscan_plugin_get_description(const suscan_plugin_t *);
uint32_t suscan_plugin_get_version(const suscan_plugin_t *);
uint32_t suscan_plugin_get_api_version(const suscan_plugin_t *);
void *suscan_plugin_get_service(const suscan_plugin_t *, const char *);
SUBOOL suscan_plugin_register_service(const struct suscan_plugin_service_desc *);
SUBOOL suscan_plugin_load_all(void);
#ifdef __cplusplus
}
#endif
#endif
