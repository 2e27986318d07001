// analyzer_internal.hpp -- what analyzer.cpp (the worker, the shards, the inspector chains) and analyzer_config.cpp (message
// queue, source / inspector configuration objects of the suscan ABI) share.  Not installed.
#pragma once
#include <sys/time.h>
#include <cstring>
#include <map>
#include <string>

#include "../../include/suscan_amd.h"

// ==========================================================================================
struct suscan_source_config {
  std::string type;
  enum suscan_source_format format;
  unsigned samp_rate = 1000000;
  double freq = 0;
  std::string path;
  bool loop = false;
  std::map<std::string, std::string> params;
  // recorded and read back (Suscan/Source.cpp getters); a file / generator source does not act on them
  std::string label = "Unlabeled source", antenna;
  bool has_antenna = false, dc_remove = false, iq_balance = false;
  double lnb_freq = 0;
  unsigned average = 1;
  float bandwidth = 0, ppm = 0;
  struct timeval start_time{0, 0};
  std::map<std::string, float> gains;
};

namespace suan {
// a field of an inspector configuration as a number (integer / boolean / float fields; `dflt` when absent)
double cfg_get(const suscan_config_t *cfg, const char *name, double dflt);
inline char *dupstr(const char *s) { return s ? strdup(s) : nullptr; }
}  // namespace suan
