// tuning.cpp -- sdk::tuning(): the one place the SUAMD_* tuning environment is read (tuning.hpp has the fields and what they do)
#include "tuning.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <strings.h>

namespace sdk {
namespace {

const TuningField kFields[] = {
#define SUAMD_TUNING_ROW(f, env, def, lo, hi, doc) {#f, env, def, lo, hi, doc, &Tuning::f},
  SUAMD_TUNING_FIELDS(SUAMD_TUNING_ROW)
#undef SUAMD_TUNING_ROW
};
constexpr unsigned kCount = sizeof kFields / sizeof kFields[0];

Tuning g_tuning;
std::once_flag g_once;

// the few variables that rounds 1-5 gave words instead of numbers
bool word_value(const TuningField &f, const char *e, long long *v)
{
  if (!std::strcmp(f.name, "st_kernel")) {
    if (!std::strcmp(e, "wg")) { *v = 2; return true; }
    if (!std::strcmp(e, "wave")) { *v = 1; return true; }
    if (!std::strcmp(e, "pair")) { *v = 0; return true; }
  } else if (!std::strcmp(f.name, "psd_large")) {
    if (!std::strcmp(e, "passes")) { *v = 0; return true; }
    if (!std::strcmp(e, "twotrip")) { *v = 1; return true; }
  } else if (!std::strcmp(f.name, "analyzer_stage_priority")) {
    if (!strcasecmp(e, "off")) { *v = -2; return true; }
  } else if (!std::strcmp(f.name, "analyzer_debug") || !std::strcmp(f.name, "analyzer_trace") || !std::strcmp(f.name, "analyzer_poison_rows")) {
    *v = 1; return true;                                       // (set at all = on, as before)
  }
  return false;
}

void from_env(Tuning &t)
{
  t = Tuning();
  for (const TuningField &f : kFields) {
    const char *e = std::getenv(f.env);
    if (!e || !*e) continue;
    long long v = 0;
    if (!word_value(f, e, &v)) {
      char *end = nullptr;
      v = std::strtoll(e, &end, 10);
      if (end == e) continue;                                  // not a number: the default stays
    }
    if (v >= f.lo && v <= f.hi) t.*(f.member) = v;
  }
}

}  // namespace

Tuning &tuning()
{
  std::call_once(g_once, [] { from_env(g_tuning); });
  return g_tuning;
}

const TuningField *tuning_fields(unsigned *count) { if (count) *count = kCount; return kFields; }

bool tuning_set(const char *name, long long value)
{
  if (!name) return false;
  Tuning &t = tuning();
  for (const TuningField &f : kFields)
    if (!std::strcmp(f.name, name) || !std::strcmp(f.env, name)) {
      if (value < f.lo || value > f.hi) return false;
      t.*(f.member) = value;
      return true;
    }
  return false;
}

bool tuning_get(const char *name, long long *value)
{
  if (!name || !value) return false;
  Tuning &t = tuning();
  for (const TuningField &f : kFields)
    if (!std::strcmp(f.name, name) || !std::strcmp(f.env, name)) { *value = t.*(f.member); return true; }
  return false;
}

void tuning_reset() { from_env(tuning()); }

}  // namespace sdk
