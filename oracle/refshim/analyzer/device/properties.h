/* refshim: <analyzer/device/properties.h> (absent; control plane) */
#ifndef REFSHIM_DEVICE_PROPERTIES_H
#define REFSHIM_DEVICE_PROPERTIES_H
#include <analyzer/device/spec.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct suscan_device_gain_desc { char *name; SUFLOAT min, max, step, def; } suscan_device_gain_desc_t;
struct suscan_device_properties {
  char *label; const char *analyzer; const char *source; uint64_t uuid;
  SUFREQ freq_min, freq_max; unsigned channels;
  char **antenna_list; unsigned antenna_count;
  double *samp_rate_list; unsigned samp_rate_count;
};
suscan_device_properties_t *suscan_device_properties_dup(const suscan_device_properties_t *);
void suscan_device_properties_destroy(suscan_device_properties_t *);
SUBOOL suscan_device_properties_match(const suscan_device_properties_t *, const suscan_device_spec_t *);
suscan_device_spec_t *suscan_device_properties_make_spec(const suscan_device_properties_t *);
uint64_t suscan_device_properties_uuid(const suscan_device_properties_t *);
const char *suscan_device_properties_get(const suscan_device_properties_t *, const char *);
char *suscan_device_properties_uri(const suscan_device_properties_t *);
int suscan_device_properties_get_all_gains(const suscan_device_properties_t *, suscan_device_gain_desc_t ***);
#ifdef __cplusplus
}
#endif
#endif
