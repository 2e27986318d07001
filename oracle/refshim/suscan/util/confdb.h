/* refshim: <suscan/util/confdb.h> (config database, absent; control plane) */
#ifndef REFSHIM_SUSCAN_CONFDB_H
#define REFSHIM_SUSCAN_CONFDB_H
#include <suscan/util/object.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct suscan_config_context suscan_config_context_t;
suscan_config_context_t *suscan_config_context_lookup(const char *name);
suscan_config_context_t *suscan_config_context_assert(const char *name);
const suscan_object_t *suscan_config_context_get_list(const suscan_config_context_t *);
SUBOOL suscan_config_context_put(suscan_config_context_t *, suscan_object_t *);
SUBOOL suscan_config_context_remove(suscan_config_context_t *, suscan_object_t *);
void   suscan_config_context_flush(suscan_config_context_t *);
void   suscan_config_context_set_save(suscan_config_context_t *, SUBOOL);
SUBOOL suscan_confdb_use(const char *name);
SUBOOL suscan_confdb_save_all(void);
const char *suscan_confdb_get_local_tle_path(void);
#ifdef __cplusplus
}
#endif
#endif
