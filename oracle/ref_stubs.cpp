/*
 * oracle/ref_stubs.cpp -- TEST INFRASTRUCTURE ONLY.  Link-time stand-ins for the libsuscan CONTROL PLANE the reference's
 * wrappers mention next to the hot path (XML object tree, device discovery, config database, orbits, SuWidgets): out
 * of scope (SURVEY.md section 8 / DESIGN.md section 0), never reached by the path, so each one aborts loudly if called.
 * Everything the path does call -- suscan_analyzer_*, suscan_mq_*, suscan_config_*, suscan_source_config_*,
 * suscan_source_info_* -- resolves to libsigdigger_amd.so; oracle/Makefile.ref links with --no-undefined.
 */
#include <cstdio>
#include <cstdlib>
#include <sigutils/types.h>
#include <suscan_amd.h>
#include <analyzer/source.h>
#include <analyzer/device/facade.h>
#include <suscan/util/object.h>
#include <suscan/util/confdb.h>
#include <sgdp4/sgdp4.h>
#include <SuWidgetsHelpers.h>
#include <Suscan/Device.h>

#define OUT_OF_SCOPE(name) do { std::fprintf(stderr, "refstub: %s is libsuscan control plane, not served\n", name); std::abort(); } while (0)
#define STUB(ret, name, args) ret name args { OUT_OF_SCOPE(#name); }

extern "C" {
STUB(suscan_object_t *, suscan_object_new, (enum suscan_object_type))
STUB(suscan_object_t *, suscan_object_copy, (const suscan_object_t *))
void suscan_object_destroy(suscan_object_t *) {}
STUB(suscan_object_t *, suscan_object_from_xml, (const char *, const void *, size_t))
STUB(SUBOOL, suscan_object_to_xml, (const suscan_object_t *, void **, size_t *))
STUB(const char *, suscan_object_get_class, (const suscan_object_t *))
STUB(SUBOOL, suscan_object_set_class, (suscan_object_t *, const char *))
STUB(enum suscan_object_type, suscan_object_get_type, (const suscan_object_t *))
STUB(suscan_object_t *, suscan_object_get_field, (const suscan_object_t *, const char *))
STUB(SUBOOL, suscan_object_set_field, (suscan_object_t *, const char *, suscan_object_t *))
STUB(unsigned int, suscan_object_field_count, (const suscan_object_t *))
STUB(suscan_object_t *, suscan_object_get_field_by_index, (const suscan_object_t *, unsigned int))
STUB(int, suscan_object_get_field_int, (const suscan_object_t *, const char *, int))
STUB(SUBOOL, suscan_object_get_field_bool, (const suscan_object_t *, const char *, SUBOOL))
STUB(unsigned int, suscan_object_get_field_uint, (const suscan_object_t *, const char *, unsigned int))
STUB(SUFLOAT, suscan_object_get_field_float, (const suscan_object_t *, const char *, SUFLOAT))
STUB(const char *, suscan_object_get_field_value, (const suscan_object_t *, const char *))
STUB(SUBOOL, suscan_object_set_field_int, (suscan_object_t *, const char *, int))
STUB(SUBOOL, suscan_object_set_field_uint, (suscan_object_t *, const char *, unsigned int))
STUB(SUBOOL, suscan_object_set_field_bool, (suscan_object_t *, const char *, SUBOOL))
STUB(SUBOOL, suscan_object_set_field_float, (suscan_object_t *, const char *, SUFLOAT))
STUB(SUBOOL, suscan_object_set_field_value, (suscan_object_t *, const char *, const char *))
STUB(SUBOOL, suscan_object_set_value, (suscan_object_t *, const char *))
STUB(const char *, suscan_object_get_name, (const suscan_object_t *))
STUB(const char *, suscan_object_get_value, (const suscan_object_t *))
STUB(unsigned int, suscan_object_set_get_count, (const suscan_object_t *))
STUB(suscan_object_t *, suscan_object_set_get, (const suscan_object_t *, unsigned int))
STUB(SUBOOL, suscan_object_set_put, (suscan_object_t *, unsigned int, suscan_object_t *))
STUB(SUBOOL, suscan_object_set_delete, (suscan_object_t *, unsigned int))
STUB(SUBOOL, suscan_object_set_append, (suscan_object_t *, suscan_object_t *))
STUB(void, suscan_object_set_clear, (suscan_object_t *))
STUB(void, suscan_object_clear_fields, (suscan_object_t *))

STUB(suscan_config_context_t *, suscan_config_context_lookup, (const char *))
STUB(suscan_config_context_t *, suscan_config_context_assert, (const char *))
STUB(const suscan_object_t *, suscan_config_context_get_list, (const suscan_config_context_t *))
STUB(SUBOOL, suscan_config_context_put, (suscan_config_context_t *, suscan_object_t *))
STUB(SUBOOL, suscan_config_context_remove, (suscan_config_context_t *, suscan_object_t *))
STUB(void, suscan_config_context_flush, (suscan_config_context_t *))
STUB(void, suscan_config_context_set_save, (suscan_config_context_t *, SUBOOL))
STUB(SUBOOL, suscan_confdb_use, (const char *))
STUB(SUBOOL, suscan_confdb_save_all, (void))
STUB(const char *, suscan_confdb_get_local_tle_path, (void))

STUB(suscan_source_config_t *, suscan_source_config_from_object, (const suscan_object_t *))
STUB(suscan_object_t *, suscan_source_config_to_object, (const suscan_source_config_t *))
STUB(SUBOOL, suscan_source_config_guess_metadata, (const suscan_source_config_t *, struct suscan_source_metadata *))
STUB(SUBOOL, suscan_source_config_set_device_spec, (suscan_source_config_t *, const suscan_device_spec_t *))
STUB(suscan_device_spec_t *, suscan_source_config_get_device_spec, (const suscan_source_config_t *))
STUB(suscan_source_t *, suscan_source_new, (suscan_source_config_t *))
STUB(void, suscan_source_destroy, (suscan_source_t *))

/* a default-constructed Suscan::DeviceSpec lives inside every Source::Config (include/Suscan/Source.h:46) */
suscan_device_spec_t *suscan_device_spec_new(void) { return static_cast<suscan_device_spec_t *>(std::calloc(1, 8)); }
void suscan_device_spec_destroy(suscan_device_spec_t *s) { std::free(s); }
STUB(suscan_device_spec_t *, suscan_device_spec_copy, (const suscan_device_spec_t *))
STUB(suscan_device_spec_t *, suscan_device_spec_from_uri, (const char *))
STUB(suscan_device_spec_t *, suscan_device_spec_from_object, (const suscan_object_t *))
STUB(void, suscan_device_spec_reset, (suscan_device_spec_t *))
STUB(suscan_device_properties_t *, suscan_device_spec_properties, (const suscan_device_spec_t *))
STUB(const char *, suscan_device_spec_analyzer, (const suscan_device_spec_t *))
STUB(const char *, suscan_device_spec_source, (const suscan_device_spec_t *))
STUB(const char *, suscan_device_spec_get, (const suscan_device_spec_t *, const char *))
STUB(char *, suscan_device_spec_to_uri, (const suscan_device_spec_t *))
STUB(uint64_t, suscan_device_spec_uuid, (const suscan_device_spec_t *))
STUB(SUBOOL, suscan_device_spec_set_analyzer, (suscan_device_spec_t *, const char *))
STUB(SUBOOL, suscan_device_spec_set_source, (suscan_device_spec_t *, const char *))
STUB(SUBOOL, suscan_device_spec_set, (suscan_device_spec_t *, const char *, const char *))
STUB(SUBOOL, suscan_device_spec_set_traits, (suscan_device_spec_t *, const strmap_t *))
STUB(SUBOOL, suscan_device_spec_set_params, (suscan_device_spec_t *, const strmap_t *))
STUB(void, suscan_device_spec_update_uuid, (suscan_device_spec_t *))
STUB(suscan_device_properties_t *, suscan_device_properties_dup, (const suscan_device_properties_t *))
void suscan_device_properties_destroy(suscan_device_properties_t *) {}
STUB(SUBOOL, suscan_device_properties_match, (const suscan_device_properties_t *, const suscan_device_spec_t *))
STUB(suscan_device_spec_t *, suscan_device_properties_make_spec, (const suscan_device_properties_t *))
STUB(uint64_t, suscan_device_properties_uuid, (const suscan_device_properties_t *))
STUB(const char *, suscan_device_properties_get, (const suscan_device_properties_t *, const char *))
STUB(char *, suscan_device_properties_uri, (const suscan_device_properties_t *))
STUB(int, suscan_device_properties_get_all_gains, (const suscan_device_properties_t *, suscan_device_gain_desc_t ***))
STUB(suscan_device_facade_t *, suscan_device_facade_instance, (void))
STUB(int, suscan_device_facade_get_all_devices, (suscan_device_facade_t *, suscan_device_properties_t ***))
STUB(suscan_device_properties_t *, suscan_device_facade_get_device_by_uuid, (suscan_device_facade_t *, uint64_t))
STUB(void, suscan_device_facade_discover_all, (suscan_device_facade_t *))
STUB(SUBOOL, suscan_device_facade_start_discovery, (suscan_device_facade_t *, const char *))
STUB(SUBOOL, suscan_device_facade_stop_discovery, (suscan_device_facade_t *, const char *))
STUB(char *, suscan_device_facade_wait_for_devices, (suscan_device_facade_t *, unsigned int))

STUB(strmap_t *, strmap_new, (void))
STUB(SUBOOL, strmap_set, (strmap_t *, const char *, const char *))
void strmap_destroy(strmap_t *) {}

void orbit_finalize(orbit_t *o) { if (o) { std::free(o->name); o->name = nullptr; } }
STUB(int, orbit_init_from_data, (orbit_t *, const char *, size_t))
STUB(int, orbit_init_from_file, (orbit_t *, const char *))
}

QString SuWidgetsHelpers::formatQuantity(qreal value, int, QString const &units, bool) { return QString::number(value) + " " + units; }
QString SuWidgetsHelpers::formatQuantity(qreal value, QString const &units) { return QString::number(value) + " " + units; }

/* Suscan/Device.cpp (device discovery wrappers) is control plane and is not compiled; the members Suscan/Source.cpp
 * references: an empty DeviceSpec lives in every Source::Config (include/Suscan/Source.h:46) */
namespace Suscan {
DeviceProperties::DeviceProperties() {}
DeviceProperties::~DeviceProperties() {}
DeviceSpec::DeviceSpec() {}
DeviceSpec::~DeviceSpec() {}
DeviceSpec::DeviceSpec(DeviceSpec &&) { OUT_OF_SCOPE("DeviceSpec(DeviceSpec &&)"); }
DeviceSpec &DeviceSpec::operator=(DeviceSpec &&) { return *this; }
DeviceSpec DeviceSpec::wrap(suscan_device_spec_t *) { OUT_OF_SCOPE("DeviceSpec::wrap"); }
const DeviceProperties *DeviceSpec::properties() const { OUT_OF_SCOPE("DeviceSpec::properties"); }
std::list<std::string> DeviceProperties::gains() const { OUT_OF_SCOPE("DeviceProperties::gains"); }
}
