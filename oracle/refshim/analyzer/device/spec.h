/* refshim: <analyzer/device/spec.h> (device discovery, absent; control plane) */
#ifndef REFSHIM_DEVICE_SPEC_H
#define REFSHIM_DEVICE_SPEC_H
#include <sigutils/types.h>
#include <suscan/util/object.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct suscan_device_spec suscan_device_spec_t;
typedef struct suscan_device_properties suscan_device_properties_t;
typedef struct strmap strmap_t;
#include <assert.h>
#define SU_DISPOSE(cls, ptr) cls##_destroy(ptr)
strmap_t *strmap_new(void);
SUBOOL strmap_set(strmap_t *, const char *, const char *);
void strmap_destroy(strmap_t *);
suscan_device_spec_t *suscan_device_spec_new(void);
suscan_device_spec_t *suscan_device_spec_copy(const suscan_device_spec_t *);
suscan_device_spec_t *suscan_device_spec_from_uri(const char *);
suscan_device_spec_t *suscan_device_spec_from_object(const suscan_object_t *);
void suscan_device_spec_destroy(suscan_device_spec_t *);
void suscan_device_spec_reset(suscan_device_spec_t *);
suscan_device_properties_t *suscan_device_spec_properties(const suscan_device_spec_t *);
const char *suscan_device_spec_analyzer(const suscan_device_spec_t *);
const char *suscan_device_spec_source(const suscan_device_spec_t *);
const char *suscan_device_spec_get(const suscan_device_spec_t *, const char *);
char *suscan_device_spec_to_uri(const suscan_device_spec_t *);
uint64_t suscan_device_spec_uuid(const suscan_device_spec_t *);
SUBOOL suscan_device_spec_set_analyzer(suscan_device_spec_t *, const char *);
SUBOOL suscan_device_spec_set_source(suscan_device_spec_t *, const char *);
SUBOOL suscan_device_spec_set(suscan_device_spec_t *, const char *, const char *);
SUBOOL suscan_device_spec_set_traits(suscan_device_spec_t *, const strmap_t *);
SUBOOL suscan_device_spec_set_params(suscan_device_spec_t *, const strmap_t *);
void suscan_device_spec_update_uuid(suscan_device_spec_t *);
#ifdef __cplusplus
}
#endif
#endif
