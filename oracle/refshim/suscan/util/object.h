/* refshim: <suscan/util/object.h> (libsuscan's XML object tree, absent): the calls of include/Suscan/Object.h */
#ifndef REFSHIM_SUSCAN_OBJECT_H
#define REFSHIM_SUSCAN_OBJECT_H
#include <sigutils/types.h>
#ifdef __cplusplus
extern "C" {
#endif
enum suscan_object_type { SUSCAN_OBJECT_TYPE_OBJECT, SUSCAN_OBJECT_TYPE_SET, SUSCAN_OBJECT_TYPE_FIELD };
typedef struct suscan_object suscan_object_t;
suscan_object_t *suscan_object_new(enum suscan_object_type type);
suscan_object_t *suscan_object_copy(const suscan_object_t *);
void  suscan_object_destroy(suscan_object_t *);
suscan_object_t *suscan_object_from_xml(const char *url, const void *data, size_t size);
SUBOOL suscan_object_to_xml(const suscan_object_t *, void **data, size_t *size);
const char *suscan_object_get_class(const suscan_object_t *);
SUBOOL suscan_object_set_class(suscan_object_t *, const char *);
enum suscan_object_type suscan_object_get_type(const suscan_object_t *);
suscan_object_t *suscan_object_get_field(const suscan_object_t *, const char *);
SUBOOL suscan_object_set_field(suscan_object_t *, const char *, suscan_object_t *);
unsigned int suscan_object_field_count(const suscan_object_t *);
suscan_object_t *suscan_object_get_field_by_index(const suscan_object_t *, unsigned int);
int      suscan_object_get_field_int(const suscan_object_t *, const char *, int dfl);
SUBOOL   suscan_object_get_field_bool(const suscan_object_t *, const char *, SUBOOL dfl);
unsigned int suscan_object_get_field_uint(const suscan_object_t *, const char *, unsigned int dfl);
SUFLOAT  suscan_object_get_field_float(const suscan_object_t *, const char *, SUFLOAT dfl);
const char *suscan_object_get_field_value(const suscan_object_t *, const char *);
SUBOOL suscan_object_set_field_int(suscan_object_t *, const char *, int);
SUBOOL suscan_object_set_field_uint(suscan_object_t *, const char *, unsigned int);
SUBOOL suscan_object_set_field_bool(suscan_object_t *, const char *, SUBOOL);
SUBOOL suscan_object_set_field_float(suscan_object_t *, const char *, SUFLOAT);
SUBOOL suscan_object_set_field_value(suscan_object_t *, const char *, const char *);
SUBOOL suscan_object_set_value(suscan_object_t *, const char *);
const char *suscan_object_get_name(const suscan_object_t *);
const char *suscan_object_get_value(const suscan_object_t *);
unsigned int suscan_object_set_get_count(const suscan_object_t *);
suscan_object_t *suscan_object_set_get(const suscan_object_t *, unsigned int);
SUBOOL suscan_object_set_put(suscan_object_t *, unsigned int, suscan_object_t *);
SUBOOL suscan_object_set_delete(suscan_object_t *, unsigned int);
SUBOOL suscan_object_set_append(suscan_object_t *, suscan_object_t *);
void   suscan_object_set_clear(suscan_object_t *);
void   suscan_object_clear_fields(suscan_object_t *);
#ifdef __cplusplus
}
#endif
#endif
