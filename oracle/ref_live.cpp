/*
 * oracle/ref_live.cpp -- TEST INFRASTRUCTURE ONLY (oracle/Makefile.ref -> oracle/_ref/ref_live).
 *
 * The reference's own live-path classes -- Suscan::Analyzer (Suscan/Analyzer.cpp), Suscan::Source::Config
 * (Suscan/Source.cpp), Suscan::AnalyzerParams, Suscan::AnalyzerRequestTracker and the Suscan::*Message wrappers,
 * compiled UNCHANGED from /root/reference and linked against libsigdigger_amd.so -- driving the GPU analyzer the way
 * SigDigger's UI does (App/Application.cpp:429, UIMediator/InspectorMediator.cpp): construct the analyzer on a file
 * source, let the request tracker open a "psk" inspector and set its id, push an inspector config, collect
 * PSDMessage / SamplesMessage objects from the Qt signals until end of stream, then halt.
 *
 * usage: ref_live <iq.f32> <samp_rate> <window_size> <chan_fc_hz> <chan_bw_hz> <out.bin> [class = psk | raw] [baud]
 * out.bin: u32 magic 'RLV1', u32 npsd, u32 psd_size, u32 nbatches, u64 nsamples, f32 fs, f32 equiv_fs, f32 bandwidth,
 *          u32 inspector_id_seen, then psd_size floats (first PSD frame as PSDMessage delivers it: shifted dB),
 *          then nsamples complex64 (all SamplesMessage payloads in order).
 * exit 0 on EOS after at least one PSD frame, 2 on read error / init failure, 3 on timeout.
 */
#include <QCoreApplication>
#include <QTimer>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <Suscan/Analyzer.h>
#include <Suscan/AnalyzerRequestTracker.h>

int main(int argc, char **argv)
{
  if (argc < 7) { std::fprintf(stderr, "usage: %s iq.f32 samp_rate window chan_fc chan_bw out.bin\n", argv[0]); return 64; }
  QCoreApplication app(argc, argv);
  const std::string path = argv[1];
  const unsigned fs = (unsigned)std::atol(argv[2]);
  const unsigned window = (unsigned)std::atol(argv[3]);
  const double chan_fc = std::atof(argv[4]), chan_bw = std::atof(argv[5]);
  const char *out_path = argv[6];
  const std::string cls = argc > 7 ? argv[7] : "psk";
  const float baud = argc > 8 ? (float)std::atof(argv[8]) : 0.f;

  Suscan::Source::Config cfg("file", SUSCAN_SOURCE_FORMAT_RAW_FLOAT32);
  cfg.setPath(path);
  cfg.setSampleRate(fs);
  cfg.setFreq(100e6);
  cfg.setLoop(false);
  if (!cfg.fileIsValid()) { std::fprintf(stderr, "ref_live: %s is not readable\n", path.c_str()); return 2; }

  Suscan::AnalyzerParams params;
  params.windowSize = window;
  params.windowFunction = Suscan::AnalyzerParams::BLACKMANN_HARRIS;
  params.mode = Suscan::AnalyzerParams::CHANNEL;
  params.psdUpdateInterval = 0.01f;

  std::vector<float> first_psd;
  std::vector<SUCOMPLEX> samples;
  unsigned npsd = 0, nbatches = 0, seen_id = 0;
  float bb_fs = 0, equiv_fs = 0, bandwidth = 0;
  int rc = 3;
  bool opened_requested = false;

  Suscan::Analyzer *an = nullptr;
  try {
    an = new Suscan::Analyzer(params, cfg);
  } catch (Suscan::Exception &e) {
    std::fprintf(stderr, "ref_live: %s\n", e.what());
    return 2;
  }
  Suscan::AnalyzerRequestTracker tracker;
  tracker.setAnalyzer(an);
  QObject::connect(an, &Suscan::Analyzer::inspector_message, &tracker, &Suscan::AnalyzerRequestTracker::onInspectorMessage);

  QObject::connect(an, &Suscan::Analyzer::source_info_message, [&](const Suscan::SourceInfoMessage &m) {
    if (opened_requested) return;
    opened_requested = true;
    if (m.info()->getSampleRate() != fs) std::fprintf(stderr, "ref_live: source info reports %lu S/s\n", (unsigned long)m.info()->getSampleRate());
    Suscan::Channel ch;
    ch.fc = chan_fc; ch.ft = 0; ch.bw = chan_bw;
    ch.fLow = -0.5 * chan_bw; ch.fHigh = 0.5 * chan_bw;
    tracker.requestOpen(cls, ch, QVariant(), false);
  });
  QObject::connect(&tracker, &Suscan::AnalyzerRequestTracker::opened, [&](Suscan::AnalyzerRequest const &req) {
    bb_fs = (float)req.basebandRate; equiv_fs = req.equivRate; bandwidth = req.bandwidth;
    if (cls != "psk") return;                                 // "raw": the channel samples as they are
    Suscan::Config c(req.config);                             // dup'd by the tracker (AnalyzerRequestTracker.cpp:139-141)
    c.set("afc.costas-order", (uint64_t)2);
    c.set("afc.bits-per-symbol", (uint64_t)2);
    c.set("clock.type", (uint64_t)1);
    c.set("clock.baud", (SUFLOAT)(baud > 0 ? baud : equiv_fs / 8.f));
    c.set("clock.running", true);
    an->setInspectorConfig(req.handle, c);
  });
  QObject::connect(&tracker, &Suscan::AnalyzerRequestTracker::error, [&](Suscan::AnalyzerRequest const &, const std::string &what) {
    std::fprintf(stderr, "ref_live: open failed: %s\n", what.c_str());
    rc = 2; app.quit();
  });
  QObject::connect(an, &Suscan::Analyzer::psd_message, [&](const Suscan::PSDMessage &m) {
    if (npsd++ == 0) first_psd.assign(m.get(), m.get() + m.size());
  });
  QObject::connect(an, &Suscan::Analyzer::samples_message, [&](const Suscan::SamplesMessage &m) {
    ++nbatches; seen_id = m.getInspectorId();
    samples.insert(samples.end(), m.getSamples(), m.getSamples() + m.getCount());
  });
  QObject::connect(an, &Suscan::Analyzer::status_message, [&](const Suscan::StatusMessage &m) {
    if (m.getCode() < 0) { std::fprintf(stderr, "ref_live: status %d: %s\n", m.getCode(), m.getMessage().toStdString().c_str()); rc = 2; app.quit(); }
  });
  QObject::connect(an, &Suscan::Analyzer::eos, [&]() { rc = npsd > 0 ? 0 : 2; app.quit(); });
  QObject::connect(an, &Suscan::Analyzer::read_error, [&]() { rc = 2; app.quit(); });
  QObject::connect(an, &Suscan::Analyzer::halted, [&]() { if (rc == 3) rc = 2; app.quit(); });
  QTimer::singleShot(120000, [&]() { std::fprintf(stderr, "ref_live: timeout\n"); app.quit(); });

  app.exec();
  delete an;                                                  // ~Analyzer: halt, join the reader thread, destroy

  FILE *fp = std::fopen(out_path, "wb");
  if (!fp) return 2;
  const uint32_t hdr[4] = { 0x31564c52u, npsd, (uint32_t)first_psd.size(), nbatches };
  const uint64_t ns = samples.size();
  const float rates[3] = { bb_fs, equiv_fs, bandwidth };
  std::fwrite(hdr, sizeof hdr, 1, fp);
  std::fwrite(&ns, sizeof ns, 1, fp);
  std::fwrite(rates, sizeof rates, 1, fp);
  std::fwrite(&seen_id, sizeof seen_id, 1, fp);
  std::fwrite(first_psd.data(), sizeof(float), first_psd.size(), fp);
  std::fwrite(static_cast<const void *>(samples.data()), sizeof(SUCOMPLEX), samples.size(), fp);
  std::fclose(fp);
  std::fprintf(stderr, "ref_live: %u PSD frames, %u batches, %llu samples, rc %d\n", npsd, nbatches, (unsigned long long)ns, rc);
  return rc;
}
