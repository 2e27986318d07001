/* refshim: <analyzer/source.h>: the source config API served by the product is in suscan_amd.h; the control-plane
 * remainder (XML, device specs, metadata guessing, live sources) is declared here only */
#ifndef REFSHIM_ANALYZER_SOURCE_H
#define REFSHIM_ANALYZER_SOURCE_H
#include <sigutils/types.h>
#include <suscan_amd.h>
#include <suscan/util/object.h>
#include <analyzer/device/spec.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct suscan_source { suscan_source_config_t *config; } suscan_source_t;   /* Suscan/Source.cpp:652 reads ->config */
struct suscan_source_metadata { uint32_t guessed; unsigned int sample_rate; SUFREQ frequency; struct timeval start_time; enum suscan_source_format format; };
#define SUSCAN_SOURCE_CONFIG_GUESS_FREQ       1
#define SUSCAN_SOURCE_CONFIG_GUESS_SAMP_RATE  2
#define SUSCAN_SOURCE_CONFIG_GUESS_START_TIME 4
#define SUSCAN_SOURCE_CONFIG_GUESS_FORMAT     8
suscan_source_config_t *suscan_source_config_from_object(const suscan_object_t *);
suscan_object_t *suscan_source_config_to_object(const suscan_source_config_t *);
SUBOOL suscan_source_config_guess_metadata(const suscan_source_config_t *, struct suscan_source_metadata *);
SUBOOL suscan_source_config_set_device_spec(suscan_source_config_t *, const suscan_device_spec_t *);
suscan_device_spec_t *suscan_source_config_get_device_spec(const suscan_source_config_t *);
SUBOOL suscan_source_config_register(suscan_source_config_t *);
SUBOOL suscan_source_config_walk(SUBOOL (*func)(suscan_source_config_t *, void *), void *);
suscan_source_t *suscan_source_new(suscan_source_config_t *);
void suscan_source_destroy(suscan_source_t *);
#ifdef __cplusplus
}
#endif
#endif
