// analyzer_config.cpp -- the passive objects of the suscan ABI (include/suscan_amd.h): the message queue, source
// configurations, inspector configuration descriptors / instances (the key vocabulary of
// Default/GenericInspector/InspectorCtl/*.cpp) and source-info records.  Host code only; the worker is analyzer.cpp.
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "analyzer_internal.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// message queue
struct MQImpl {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::pair<uint32_t, void *>> q;
};
MQImpl *impl(struct suscan_mq *mq) { return static_cast<MQImpl *>(mq->impl); }

// ------------------------------------------------------------------------------------------
// inspector config descriptors: key vocabulary of Default/GenericInspector/InspectorCtl/*.cpp
struct FieldDef { const char *name; suscan_field_type type; double def; };
const FieldDef kPskFields[] = {
  {"agc.enabled", SUSCAN_FIELD_TYPE_BOOLEAN, 1}, {"agc.gain", SUSCAN_FIELD_TYPE_FLOAT, 1},
  {"afc.costas-order", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"afc.bits-per-symbol", SUSCAN_FIELD_TYPE_INTEGER, 1},
  {"afc.offset", SUSCAN_FIELD_TYPE_FLOAT, 0}, {"afc.loop-bw", SUSCAN_FIELD_TYPE_FLOAT, 100},
  {"mf.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"mf.roll-off", SUSCAN_FIELD_TYPE_FLOAT, .35},
  {"clock.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"clock.baud", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"clock.gain", SUSCAN_FIELD_TYPE_FLOAT, .2}, {"clock.phase", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"clock.running", SUSCAN_FIELD_TYPE_BOOLEAN, 1},
  {"equalizer.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"equalizer.rate", SUSCAN_FIELD_TYPE_FLOAT, 1e-3},
  {"equalizer.locked", SUSCAN_FIELD_TYPE_BOOLEAN, 0},
};
const FieldDef kFskFields[] = {
  {"agc.enabled", SUSCAN_FIELD_TYPE_BOOLEAN, 1}, {"agc.gain", SUSCAN_FIELD_TYPE_FLOAT, 1},
  {"mf.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"mf.roll-off", SUSCAN_FIELD_TYPE_FLOAT, .35},
  {"clock.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"clock.baud", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"clock.gain", SUSCAN_FIELD_TYPE_FLOAT, .2}, {"clock.phase", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"clock.running", SUSCAN_FIELD_TYPE_BOOLEAN, 1},
  {"fsk.bits-per-symbol", SUSCAN_FIELD_TYPE_INTEGER, 1}, {"fsk.phase", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"fsk.quad-demod", SUSCAN_FIELD_TYPE_BOOLEAN, 1},
};

// "ask": amplitude keying, optionally carrier-locked by a PLL (InspectorCtl/AskControl.cpp:53-76)
const FieldDef kAskFields[] = {
  {"agc.enabled", SUSCAN_FIELD_TYPE_BOOLEAN, 1}, {"agc.gain", SUSCAN_FIELD_TYPE_FLOAT, 1},
  {"mf.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"mf.roll-off", SUSCAN_FIELD_TYPE_FLOAT, .35},
  {"clock.type", SUSCAN_FIELD_TYPE_INTEGER, 0}, {"clock.baud", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"clock.gain", SUSCAN_FIELD_TYPE_FLOAT, .2}, {"clock.phase", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"clock.running", SUSCAN_FIELD_TYPE_BOOLEAN, 1},
  {"ask.bits-per-symbol", SUSCAN_FIELD_TYPE_INTEGER, 1}, {"ask.use-pll", SUSCAN_FIELD_TYPE_BOOLEAN, 0},
  {"ask.loop-bw", SUSCAN_FIELD_TYPE_FLOAT, 100}, {"ask.offset", SUSCAN_FIELD_TYPE_FLOAT, 0},
  {"ask.channel", SUSCAN_FIELD_TYPE_INTEGER, 0},
};

struct DescHolder {
  suscan_config_desc_t desc{};
  std::vector<suscan_field> fields;
  std::vector<suscan_field *> ptrs;
  std::vector<double> defaults;
  DescHolder(const char *name, const FieldDef *defs, size_t n)
  {
    fields.resize(n); ptrs.resize(n); defaults.resize(n);
    for (size_t i = 0; i < n; ++i) {
      fields[i].type = defs[i].type; fields[i].optional = SU_TRUE;
      fields[i].name = const_cast<char *>(defs[i].name); fields[i].desc = const_cast<char *>("");
      ptrs[i] = &fields[i]; defaults[i] = defs[i].def;
    }
    desc.global_name = const_cast<char *>(name);
    desc.field_list = n ? ptrs.data() : nullptr;
    desc.field_count = (unsigned)n;
  }
};
DescHolder &psk_desc() { static DescHolder d("psk", kPskFields, sizeof kPskFields / sizeof kPskFields[0]); return d; }
DescHolder &fsk_desc() { static DescHolder d("fsk", kFskFields, sizeof kFskFields / sizeof kFskFields[0]); return d; }
DescHolder &ask_desc() { static DescHolder d("ask", kAskFields, sizeof kAskFields / sizeof kAskFields[0]); return d; }
DescHolder &raw_desc() { static DescHolder d("raw", nullptr, 0); return d; }
// "power": mean channel power over windows of power.integrate-samples (Default/RMSInspector/RMSInspector.cpp:415,438)
const FieldDef kPowerFields[] = { {"power.integrate-samples", SUSCAN_FIELD_TYPE_INTEGER, 1000} };
DescHolder &power_desc() { static DescHolder d("power", kPowerFields, 1); return d; }
// "audio": Default/Audio/AudioProcessor.cpp:251-270 (demodulator 1 AM, 2 FM, 3 USB, 4 LSB, 5 RAW)
const FieldDef kAudioFields[] = {
  {"audio.volume", SUSCAN_FIELD_TYPE_FLOAT, 1}, {"audio.cutoff", SUSCAN_FIELD_TYPE_FLOAT, 15000},
  {"audio.sample-rate", SUSCAN_FIELD_TYPE_INTEGER, 44100}, {"audio.demodulator", SUSCAN_FIELD_TYPE_INTEGER, 2},
  {"audio.squelch", SUSCAN_FIELD_TYPE_BOOLEAN, 0}, {"audio.squelch-level", SUSCAN_FIELD_TYPE_FLOAT, 0.5},
  {"agc.enabled", SUSCAN_FIELD_TYPE_BOOLEAN, 0}, {"agc.ts", SUSCAN_FIELD_TYPE_FLOAT, 0.2},
};
DescHolder &audio_desc() { static DescHolder d("audio", kAudioFields, sizeof kAudioFields / sizeof kAudioFields[0]); return d; }
DescHolder *holder_for(const char *cls)
{
  if (!cls) return nullptr;
  if (!std::strcmp(cls, "audio")) return &audio_desc();
  if (!std::strcmp(cls, "psk")) return &psk_desc();
  if (!std::strcmp(cls, "fsk")) return &fsk_desc();
  if (!std::strcmp(cls, "ask")) return &ask_desc();
  if (!std::strcmp(cls, "raw")) return &raw_desc();
  if (!std::strcmp(cls, "power")) return &power_desc();
  return nullptr;
}


}  // namespace

namespace suan {

double cfg_get(const suscan_config_t *cfg, const char *name, double dflt)
{
  if (!cfg) return dflt;
  struct suscan_field_value *v = suscan_config_get_value(cfg, name);
  if (!v) return dflt;
  switch (v->field->type) {
    case SUSCAN_FIELD_TYPE_INTEGER: return (double)v->as_int;
    case SUSCAN_FIELD_TYPE_BOOLEAN: return v->as_bool ? 1.0 : 0.0;
    case SUSCAN_FIELD_TYPE_FLOAT:   return (double)v->as_float;
    default: return dflt;
  }
}

}  // namespace suan

using suan::dupstr;

// ==========================================================================================
extern "C" {

SUBOOL suscan_mq_init(struct suscan_mq *mq)
{
  if (!mq) return SU_FALSE;
  mq->impl = new (std::nothrow) MQImpl;
  return mq->impl ? SU_TRUE : SU_FALSE;
}

void suscan_mq_finalize(struct suscan_mq *mq)
{
  if (!mq || !mq->impl) return;
  MQImpl *q = impl(mq);
  for (auto &e : q->q) suscan_analyzer_dispose_message(e.first, e.second);
  delete q;
  mq->impl = nullptr;
}

void *suscan_mq_read(struct suscan_mq *mq, uint32_t *type)
{
  MQImpl *q = impl(mq);
  std::unique_lock<std::mutex> lk(q->m);
  q->cv.wait(lk, [&] { return !q->q.empty(); });
  auto e = q->q.front();
  q->q.pop_front();
  if (type) *type = e.first;
  return e.second;
}

SUBOOL suscan_mq_poll(struct suscan_mq *mq, uint32_t *type, void **msg)
{
  MQImpl *q = impl(mq);
  std::lock_guard<std::mutex> lk(q->m);
  if (q->q.empty()) return SU_FALSE;
  auto e = q->q.front();
  q->q.pop_front();
  if (type) *type = e.first;
  if (msg) *msg = e.second;
  return SU_TRUE;
}

SUBOOL suscan_mq_write(struct suscan_mq *mq, uint32_t type, void *msg)
{
  MQImpl *q = impl(mq);
  {
    std::lock_guard<std::mutex> lk(q->m);
    q->q.emplace_back(type, msg);
  }
  q->cv.notify_one();
  return SU_TRUE;
}

// ---- source config ----
suscan_source_config_t *suscan_source_config_new(const char *type, enum suscan_source_format fmt)
{
  auto *c = new (std::nothrow) suscan_source_config;
  if (!c) return nullptr;
  c->type = type ? type : "file";
  c->format = fmt;
  return c;
}
void suscan_source_config_destroy(suscan_source_config_t *c) { delete c; }
void suscan_source_config_set_samp_rate(suscan_source_config_t *c, unsigned int r) { if (c) c->samp_rate = r; }
void suscan_source_config_set_freq(suscan_source_config_t *c, SUFREQ f) { if (c) c->freq = f; }
SUBOOL suscan_source_config_set_path(suscan_source_config_t *c, const char *p)
{
  if (!c || !p) return SU_FALSE;
  c->path = p;
  return SU_TRUE;
}
void suscan_source_config_set_loop(suscan_source_config_t *c, SUBOOL l) { if (c) c->loop = l != 0; }
suscan_source_config_t *suscan_source_config_clone(const suscan_source_config_t *c) { return c ? new (std::nothrow) suscan_source_config(*c) : nullptr; }
const char *suscan_source_config_get_label(const suscan_source_config_t *c) { return c ? c->label.c_str() : nullptr; }
SUBOOL suscan_source_config_set_label(suscan_source_config_t *c, const char *l) { if (!c || !l) return SU_FALSE; c->label = l; return SU_TRUE; }
const char *suscan_source_config_get_type(const suscan_source_config_t *c) { return c ? c->type.c_str() : nullptr; }
enum suscan_source_format suscan_source_config_get_format(const suscan_source_config_t *c) { return c ? c->format : SUSCAN_SOURCE_FORMAT_AUTO; }
void suscan_source_config_set_type_format(suscan_source_config_t *c, const char *t, enum suscan_source_format f)
{
  if (!c) return;
  if (t) c->type = t;
  c->format = f;
}
const char *suscan_source_config_get_path(const suscan_source_config_t *c) { return c && !c->path.empty() ? c->path.c_str() : nullptr; }
SUFREQ suscan_source_config_get_freq(const suscan_source_config_t *c) { return c ? c->freq : 0; }
SUFREQ suscan_source_config_get_lnb_freq(const suscan_source_config_t *c) { return c ? c->lnb_freq : 0; }
void suscan_source_config_set_lnb_freq(suscan_source_config_t *c, SUFREQ f) { if (c) c->lnb_freq = f; }
unsigned int suscan_source_config_get_samp_rate(const suscan_source_config_t *c) { return c ? c->samp_rate : 0; }
unsigned int suscan_source_config_get_average(const suscan_source_config_t *c) { return c ? c->average : 1; }
SUBOOL suscan_source_config_set_average(suscan_source_config_t *c, unsigned int a) { if (!c || a < 1) return SU_FALSE; c->average = a; return SU_TRUE; }
SUFLOAT suscan_source_config_get_bandwidth(const suscan_source_config_t *c) { return c ? c->bandwidth : 0; }
void suscan_source_config_set_bandwidth(suscan_source_config_t *c, SUFLOAT b) { if (c) c->bandwidth = b; }
SUFLOAT suscan_source_config_get_ppm(const suscan_source_config_t *c) { return c ? c->ppm : 0; }
void suscan_source_config_set_ppm(suscan_source_config_t *c, SUFLOAT p) { if (c) c->ppm = p; }
SUBOOL suscan_source_config_get_loop(const suscan_source_config_t *c) { return c && c->loop ? SU_TRUE : SU_FALSE; }
SUBOOL suscan_source_config_get_dc_remove(const suscan_source_config_t *c) { return c && c->dc_remove ? SU_TRUE : SU_FALSE; }
void suscan_source_config_set_dc_remove(suscan_source_config_t *c, SUBOOL v) { if (c) c->dc_remove = v != 0; }
SUBOOL suscan_source_config_get_iq_balance(const suscan_source_config_t *c) { return c && c->iq_balance ? SU_TRUE : SU_FALSE; }
void suscan_source_config_set_iq_balance(suscan_source_config_t *c, SUBOOL v) { if (c) c->iq_balance = v != 0; }
void suscan_source_config_get_start_time(const suscan_source_config_t *c, struct timeval *tv) { if (c && tv) *tv = c->start_time; }
void suscan_source_config_set_start_time(suscan_source_config_t *c, struct timeval tv) { if (c) c->start_time = tv; }
SUBOOL suscan_source_config_file_is_valid(const suscan_source_config_t *c)
{
  if (!c || c->type != "file" || c->path.empty()) return SU_FALSE;
  FILE *fp = std::fopen(c->path.c_str(), "rb");
  if (!fp) return SU_FALSE;
  std::fclose(fp);
  return SU_TRUE;
}
SUBOOL suscan_source_config_get_end_time(const suscan_source_config_t *c, struct timeval *tv)
{
  // raw payload length / bytes per sample / rate after the start time (a WAV / SigMF container's header is a few
  // dozen bytes: below the microsecond at any rate this path is used at)
  if (!c || !tv || c->type != "file" || c->path.empty() || c->samp_rate == 0) return SU_FALSE;
  FILE *fp = std::fopen(c->path.c_str(), "rb");
  if (!fp) return SU_FALSE;
  std::fseek(fp, 0, SEEK_END);
  const long bytes = std::ftell(fp);
  std::fclose(fp);
  if (bytes < 0) return SU_FALSE;
  unsigned bps = 8;
  switch (c->format) {
    case SUSCAN_SOURCE_FORMAT_RAW_UNSIGNED8: case SUSCAN_SOURCE_FORMAT_RAW_SIGNED8: bps = 2; break;
    case SUSCAN_SOURCE_FORMAT_RAW_SIGNED16: case SUSCAN_SOURCE_FORMAT_WAV: bps = 4; break;
    default: break;
  }
  const double t = (double)c->start_time.tv_sec + 1e-6 * (double)c->start_time.tv_usec + (double)(bytes / bps) / (double)c->samp_rate;
  tv->tv_sec = (time_t)t;
  tv->tv_usec = (suseconds_t)((t - std::floor(t)) * 1e6);
  return SU_TRUE;
}
SUBOOL suscan_source_config_is_real_time(const suscan_source_config_t *) { return SU_FALSE; }
SUBOOL suscan_source_config_is_seekable(const suscan_source_config_t *c) { return c && c->type == "file" ? SU_TRUE : SU_FALSE; }
SUBOOL suscan_source_config_get_freq_limits(const suscan_source_config_t *c, SUFREQ *mn, SUFREQ *mx)
{
  if (!c || !mn || !mx) return SU_FALSE;
  *mn = -3e11; *mx = 3e11;                                   // what suscan_analyzer_new reports in the source info
  return SU_TRUE;
}
const char *suscan_source_config_get_antenna(const suscan_source_config_t *c) { return c && c->has_antenna ? c->antenna.c_str() : nullptr; }
SUBOOL suscan_source_config_set_antenna(suscan_source_config_t *c, const char *a) { if (!c || !a) return SU_FALSE; c->antenna = a; c->has_antenna = true; return SU_TRUE; }
SUFLOAT suscan_source_config_get_gain(const suscan_source_config_t *c, const char *n)
{
  if (!c || !n) return 0;
  auto it = c->gains.find(n);
  return it == c->gains.end() ? 0 : it->second;
}
SUBOOL suscan_source_config_set_gain(suscan_source_config_t *c, const char *n, SUFLOAT v) { if (!c || !n) return SU_FALSE; c->gains[n] = v; return SU_TRUE; }
const char *suscan_source_config_get_param(const suscan_source_config_t *c, const char *k)
{
  if (!c || !k) return nullptr;
  auto it = c->params.find(k);
  return it == c->params.end() ? nullptr : it->second.c_str();
}
void suscan_source_config_clear_params(suscan_source_config_t *c) { if (c) c->params.clear(); }
SUBOOL suscan_source_config_walk_params(const suscan_source_config_t *c,
                                        SUBOOL (*func)(const suscan_source_config_t *, const char *, const char *, void *), void *priv)
{
  if (!c || !func) return SU_FALSE;
  for (const auto &kv : c->params) if (!func(c, kv.first.c_str(), kv.second.c_str(), priv)) return SU_FALSE;
  return SU_TRUE;
}
SUBOOL suscan_source_config_set_param(suscan_source_config_t *c, const char *k, const char *v)
{
  if (!c || !k || !v) return SU_FALSE;
  c->params[k] = v;
  return SU_TRUE;
}

// ---- config ----
const suscan_config_desc_t *suscan_inspector_config_desc(const char *cls)
{
  DescHolder *h = holder_for(cls);
  return h ? &h->desc : nullptr;
}

suscan_config_t *suscan_config_new(const suscan_config_desc_t *desc)
{
  if (!desc) return nullptr;
  auto *c = static_cast<suscan_config_t *>(std::calloc(1, sizeof(suscan_config_t)));
  c->desc = desc;
  c->values = static_cast<suscan_field_value **>(std::calloc(desc->field_count ? desc->field_count : 1, sizeof(void *)));
  DescHolder *h = holder_for(desc->global_name);
  for (unsigned i = 0; i < desc->field_count; ++i) {
    auto *v = static_cast<suscan_field_value *>(std::calloc(1, sizeof(suscan_field_value)));
    v->field = desc->field_list[i];
    const double d = h ? h->defaults[i] : 0;
    switch (v->field->type) {
      case SUSCAN_FIELD_TYPE_INTEGER: v->as_int = (uint64_t)d; break;
      case SUSCAN_FIELD_TYPE_BOOLEAN: v->as_bool = d != 0; break;
      case SUSCAN_FIELD_TYPE_FLOAT:   v->as_float = (SUFLOAT)d; break;
      default: break;
    }
    c->values[i] = v;
  }
  return c;
}

suscan_config_t *suscan_config_dup(const suscan_config_t *cfg)
{
  if (!cfg) return nullptr;
  suscan_config_t *c = suscan_config_new(cfg->desc);
  for (unsigned i = 0; c && i < cfg->desc->field_count; ++i) {
    c->values[i]->set = cfg->values[i]->set;
    c->values[i]->as_int = cfg->values[i]->as_int;
  }
  return c;
}

void suscan_config_destroy(suscan_config_t *cfg)
{
  if (!cfg) return;
  for (unsigned i = 0; i < cfg->desc->field_count; ++i) {
    if (cfg->values[i]) { std::free(cfg->values[i]->as_string); std::free(cfg->values[i]); }
  }
  std::free(cfg->values);
  std::free(cfg);
}

struct suscan_field_value *suscan_config_get_value(const suscan_config_t *cfg, const char *name)
{
  if (!cfg || !name) return nullptr;
  for (unsigned i = 0; i < cfg->desc->field_count; ++i)
    if (!std::strcmp(cfg->desc->field_list[i]->name, name)) return cfg->values[i];
  return nullptr;
}

SUBOOL suscan_config_set_integer(suscan_config_t *cfg, const char *name, uint64_t v)
{
  auto *f = suscan_config_get_value(cfg, name);
  if (!f || f->field->type != SUSCAN_FIELD_TYPE_INTEGER) return SU_FALSE;
  f->as_int = v; f->set = SU_TRUE;
  return SU_TRUE;
}
SUBOOL suscan_config_set_float(suscan_config_t *cfg, const char *name, SUFLOAT v)
{
  auto *f = suscan_config_get_value(cfg, name);
  if (!f || f->field->type != SUSCAN_FIELD_TYPE_FLOAT) return SU_FALSE;
  f->as_int = 0; f->as_float = v; f->set = SU_TRUE;
  return SU_TRUE;
}
SUBOOL suscan_config_set_bool(suscan_config_t *cfg, const char *name, SUBOOL v)
{
  auto *f = suscan_config_get_value(cfg, name);
  if (!f || f->field->type != SUSCAN_FIELD_TYPE_BOOLEAN) return SU_FALSE;
  f->as_int = 0; f->as_bool = v ? SU_TRUE : SU_FALSE; f->set = SU_TRUE;
  return SU_TRUE;
}
SUBOOL suscan_config_set_string(suscan_config_t *cfg, const char *name, const char *v)
{
  auto *f = suscan_config_get_value(cfg, name);
  if (!f || !v || (f->field->type != SUSCAN_FIELD_TYPE_STRING && f->field->type != SUSCAN_FIELD_TYPE_FILE)) return SU_FALSE;
  char *dup = strdup(v);
  if (!dup) return SU_FALSE;
  std::free(f->as_string); f->as_string = dup; f->set = SU_TRUE;
  return SU_TRUE;
}
SUBOOL suscan_config_desc_has_prefix(const suscan_config_desc_t *desc, const char *prefix)
{
  if (!desc || !prefix) return SU_FALSE;
  const size_t n = std::strlen(prefix);
  for (unsigned i = 0; i < desc->field_count; ++i)
    if (!std::strncmp(desc->field_list[i]->name, prefix, n)) return SU_TRUE;
  return SU_FALSE;
}

// ---- source info (include/Suscan/Analyzer.h:50-105: init / deep copy / finalize) ----
void suscan_source_info_init(struct suscan_source_info *info)
{
  if (info) std::memset(info, 0, sizeof *info);
}
void suscan_source_info_finalize(struct suscan_source_info *info)
{
  if (!info) return;
  std::free(info->antenna);
  for (unsigned i = 0; i < info->gain_count; ++i) {
    if (info->gain_list[i]) std::free(info->gain_list[i]->name);
    std::free(info->gain_list[i]);
  }
  std::free(info->gain_list);
  for (unsigned i = 0; i < info->antenna_count; ++i) std::free(info->antenna_list[i]);
  std::free(info->antenna_list);
  std::memset(info, 0, sizeof *info);
}
SUBOOL suscan_source_info_init_copy(struct suscan_source_info *dst, const struct suscan_source_info *src)
{
  if (!dst || !src) return SU_FALSE;
  *dst = *src;
  dst->antenna = src->antenna ? strdup(src->antenna) : nullptr;
  dst->gain_list = nullptr; dst->antenna_list = nullptr;
  if (src->gain_count) {
    dst->gain_list = static_cast<suscan_source_gain_info **>(std::calloc(src->gain_count, sizeof(void *)));
    for (unsigned i = 0; i < src->gain_count; ++i) {
      dst->gain_list[i] = static_cast<suscan_source_gain_info *>(std::malloc(sizeof(suscan_source_gain_info)));
      *dst->gain_list[i] = *src->gain_list[i];
      dst->gain_list[i]->name = src->gain_list[i]->name ? strdup(src->gain_list[i]->name) : nullptr;
    }
  }
  if (src->antenna_count) {
    dst->antenna_list = static_cast<char **>(std::calloc(src->antenna_count, sizeof(char *)));
    for (unsigned i = 0; i < src->antenna_count; ++i) dst->antenna_list[i] = strdup(src->antenna_list[i]);
  }
  return SU_TRUE;
}

}  // extern "C"
