/* examples/analyzer_live_bench.c -- throughput of the live path from plain C: what sigdigger_amd/livebench.py measures
 * through ctypes, with a C consumer thread instead of a Python one, so that the figure describes the LIBRARY (the sharded
 * analyzer behind suscan_analyzer_*: SUAMD_DEVICES=0,1,...) and not the interpreter that reads its queue.  N heterogeneous
 * PSK inspectors (own carrier / bandwidth / baud / Costas order / loop bandwidth: Default/GenericInspector/InspectorCtl/ sources
 * vocabulary), a looping capture in the page cache, unthrottled; one message per inspector and block is read, counted and
 * disposed (Suscan/Analyzer.cpp:63-103).
 *
 *   gcc -O2 -std=c99 examples/analyzer_live_bench.c -Iinclude -Lsigdigger_amd -lsigdigger_amd \
 *       -Wl,-rpath,$PWD/sigdigger_amd -lm -o analyzer_live_bench
 *   SUAMD_DEVICES=0,1,2,3,4,5,6,7 SUAMD_ANALYZER_BCAST=rccl ./analyzer_live_bench capture.raw 512 60 [class]
 * prints one JSON line.  class: psk (default) or raw -- channel samples without a demodulator chain behind them, i.e. next to no
 * GPU work per message: what the source thread, the queue and this consumer sustain by themselves. */
#define _POSIX_C_SOURCE 200809L
#include <suscan_amd.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
  struct suscan_mq mq;
  struct suscan_analyzer_params params = suscan_analyzer_params_INITIALIZER;
  suscan_source_config_t *config;
  suscan_analyzer_t *an;
  const unsigned fs = 50000000u, nfft = 8192u;
  const unsigned long block = 1ul << 21;
  unsigned n_insp, nblocks, k, configured = 0, psd = 0;
  unsigned long long symbols = 0, samples_msgs = 0;
  double t0 = 0, spacing, dt = 0, worker = 0;
  int running = 1, measuring = 0, done = 0;

  const char *cls = argc > 4 ? argv[4] : "psk";
  const int psk = strcmp(cls, "psk") == 0;

  if (argc < 4) { fprintf(stderr, "usage: %s capture.raw inspectors blocks [psk|raw]\n", argv[0]); return 2; }
  n_insp = (unsigned)atoi(argv[2]);
  nblocks = (unsigned)atoi(argv[3]);
  if (!suscan_mq_init(&mq)) return 1;
  config = suscan_source_config_new("file", SUSCAN_SOURCE_FORMAT_RAW_FLOAT32);
  suscan_source_config_set_samp_rate(config, fs);
  if (!suscan_source_config_set_path(config, argv[1])) return 1;
  suscan_source_config_set_loop(config, SU_TRUE);
  params.detector_params.window_size = nfft;
  params.psd_update_int = (SUFLOAT)((double)block / (double)fs);
  an = suscan_analyzer_new(&params, config, &mq);
  suscan_source_config_destroy(config);
  if (!an) { fprintf(stderr, "suscan_analyzer_new failed\n"); return 1; }
  suscan_analyzer_set_throttle_async(an, 0, 0);
  spacing = 0.9 * (double)fs / (double)n_insp;
  if (spacing > 300e3) spacing = 300e3;
  for (k = 0; k < n_insp; ++k) {
    struct sigutils_channel ch = sigutils_channel_INITIALIZER;
    const double fc = ((double)k - 0.5 * (double)n_insp + 0.5) * spacing;
    const double bw = (100e3 + 10e3 * (double)(k % 7)) * spacing / 300e3;
    ch.fc = fc; ch.f_lo = fc - bw / 2; ch.f_hi = fc + bw / 2; ch.bw = (SUFLOAT)bw; ch.ft = 100e6;
    if (!suscan_analyzer_open_ex_async(an, cls, &ch, SU_TRUE, -1, 1000 + k)) return 1;
  }
  while (running) {
    uint32_t type = 0;
    void *msg = suscan_analyzer_read(an, &type);
    switch (type) {
      case SUSCAN_WORKER_MSG_TYPE_HALT:
        running = 0;
        break;
      case SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR: {
        struct suscan_analyzer_inspector_msg *m = msg;
        if (m->kind == SUSCAN_ANALYZER_INSPECTOR_MSGKIND_OPEN) {
          const unsigned i = m->req_id - 1000;
          suscan_config_t *c2;
          if (!psk) { ++configured; break; }
          c2 = suscan_config_dup(m->config);
          suscan_config_set_integer(c2, "afc.costas-order", 1 + i % 3);
          suscan_config_set_float(c2, "afc.loop-bw", (SUFLOAT)(50.0 + 5.0 * (double)(i % 11)));
          suscan_config_set_integer(c2, "clock.type", 1);
          suscan_config_set_float(c2, "clock.baud", (SUFLOAT)((20e3 + 1e3 * (double)(i % 13)) * spacing / 300e3));
          suscan_analyzer_set_inspector_config_async(an, m->handle, c2, 2000 + i);
          suscan_config_destroy(c2);
        } else if (m->kind == SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_CONFIG) ++configured;
        break;
      }
      case SUSCAN_ANALYZER_MESSAGE_TYPE_PSD:
        if (configured == n_insp && !measuring) { measuring = 1; t0 = now_s(); psd = 0; symbols = 0; samples_msgs = 0; }
        ++psd;
        if (measuring && !done && psd == nblocks) {
          dt = now_s() - t0;
          worker = (double)suscan_analyzer_get_measured_samp_rate(an);
          done = 1;
          suscan_analyzer_req_halt(an);
        }
        break;
      case SUSCAN_ANALYZER_MESSAGE_TYPE_SAMPLES:
        if (measuring && !done) { symbols += ((struct suscan_analyzer_sample_batch_msg *)msg)->sample_count; ++samples_msgs; }
        break;
      case SUSCAN_ANALYZER_MESSAGE_TYPE_EOS:
        running = 0;
        break;
      default:
        break;
    }
    if (type != SUSCAN_WORKER_MSG_TYPE_HALT) suscan_analyzer_dispose_message(type, msg);
  }
  suscan_analyzer_destroy(an);
  suscan_mq_finalize(&mq);
  if (!done) { printf("{\"error\": \"halted before the measurement finished (%u of %u inspectors configured)\"}\n", configured, n_insp); return 1; }
  printf("{\"consumer\": \"C (examples/analyzer_live_bench.c)\", \"class\": \"%s\", \"inspectors\": %u, \"blocks\": %u, \"block_samples\": %lu, "
         "\"value_MSps\": %.3f, \"ms_per_block\": %.4f, \"worker_MSps\": %.3f, \"symbols_Msps\": %.3f, \"sample_messages_per_s\": %.0f}\n",
         cls, n_insp, nblocks, block, (double)nblocks * (double)block / dt / 1e6, dt / nblocks * 1e3, worker / 1e6,
         (double)symbols / dt / 1e6, (double)samples_msgs / dt);
  return 0;
}
