/* refshim: <analyzer/device/facade.h> (absent; control plane) */
#ifndef REFSHIM_DEVICE_FACADE_H
#define REFSHIM_DEVICE_FACADE_H
#include <analyzer/device/properties.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct suscan_device_facade suscan_device_facade_t;
suscan_device_facade_t *suscan_device_facade_instance(void);
int suscan_device_facade_get_all_devices(suscan_device_facade_t *, suscan_device_properties_t ***);
suscan_device_properties_t *suscan_device_facade_get_device_by_uuid(suscan_device_facade_t *, uint64_t);
void suscan_device_facade_discover_all(suscan_device_facade_t *);
SUBOOL suscan_device_facade_start_discovery(suscan_device_facade_t *, const char *);
SUBOOL suscan_device_facade_stop_discovery(suscan_device_facade_t *, const char *);
char *suscan_device_facade_wait_for_devices(suscan_device_facade_t *, unsigned int);
#ifdef __cplusplus
}
#endif
#endif
