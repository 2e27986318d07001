/* examples/analyzer_pump.c -- the live path from plain C: what Suscan::Analyzer's constructor and
 * AsyncThread::run (Suscan/Analyzer.cpp:601-623, 63-103) do with libsuscan, done with libsigdigger_amd:
 * create the analyzer on a file source, pump messages until EOS / HALT, dispose every message once.
 * For every PSD message it applies what PSDMessage's constructor applies (fftshift + dB,
 * Suscan/Messages/PSDMessage.cpp:29-38) and prints the strongest bin.
 *
 *   gcc -std=c99 examples/analyzer_pump.c -Iinclude -Lsigdigger_amd -lsigdigger_amd \
 *       -Wl,-rpath,$PWD/sigdigger_amd -lm -o analyzer_pump
 *   ./analyzer_pump capture.raw <samp_rate> <fft_size>
 */
#include <suscan_amd.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
  struct suscan_mq mq;
  struct suscan_analyzer_params params = suscan_analyzer_params_INITIALIZER;
  suscan_source_config_t *config;
  suscan_analyzer_t *analyzer;
  unsigned frames = 0;
  int running = 1;

  if (argc < 4) { fprintf(stderr, "usage: %s capture.raw samp_rate fft_size\n", argv[0]); return 2; }
  if (!suscan_mq_init(&mq)) return 1;
  config = suscan_source_config_new("file", SUSCAN_SOURCE_FORMAT_AUTO);
  suscan_source_config_set_samp_rate(config, (unsigned)atoi(argv[2]));
  suscan_source_config_set_freq(config, 100e6);
  if (!suscan_source_config_set_path(config, argv[1])) return 1;
  params.detector_params.window_size = (unsigned)atoi(argv[3]);
  params.psd_update_int = (SUFLOAT)(16.0 * atoi(argv[3]) / atoi(argv[2]));      /* 16 frames per PSD message */
  analyzer = suscan_analyzer_new(&params, config, &mq);
  suscan_source_config_destroy(config);
  if (!analyzer) { fprintf(stderr, "suscan_analyzer_new failed\n"); return 1; }

  while (running) {
    uint32_t type = 0;
    void *msg = suscan_analyzer_read(analyzer, &type);           /* blocks, like AsyncThread::run */
    switch (type) {
      case SUSCAN_WORKER_MSG_TYPE_HALT:
        running = 0;
        break;
      case SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INIT: {
        struct suscan_analyzer_status_msg *st = msg;
        if (st->code != SUSCAN_ANALYZER_INIT_SUCCESS) fprintf(stderr, "source: %s\n", st->err_msg ? st->err_msg : "failed");
        break;
      }
      case SUSCAN_ANALYZER_MESSAGE_TYPE_PSD: {
        struct suscan_analyzer_psd_msg *p = msg;
        SUSCOUNT i, n = p->psd_size, best = 0;
        for (i = 0; i < n / 2; ++i) {                            /* PSDMessage::PSDMessage, in place */
          SUFLOAT a = p->psd_data[i], b = p->psd_data[i + n / 2];
          p->psd_data[i] = 10.f * log10f(b + 1e-8f);
          p->psd_data[i + n / 2] = 10.f * log10f(a + 1e-8f);
        }
        for (i = 1; i < n; ++i) if (p->psd_data[i] > p->psd_data[best]) best = i;
        printf("psd %u: peak %.2f dB at %+.1f Hz\n", frames++, p->psd_data[best],
               ((double)best - (double)n / 2) * p->samp_rate / (double)n);
        break;
      }
      case SUSCAN_ANALYZER_MESSAGE_TYPE_EOS:
        printf("end of stream after %u spectra\n", frames);
        break;
      default:
        break;
    }
    if (type != SUSCAN_WORKER_MSG_TYPE_HALT) suscan_analyzer_dispose_message(type, msg);
  }
  suscan_analyzer_destroy(analyzer);
  suscan_mq_finalize(&mq);
  return 0;
}
