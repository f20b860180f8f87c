// integration/compress_engine.cpp -- compression with the WHOLE predictor on the device: what runner.cpp's
// RunCompression + Compress do (runner.cpp:101-119,162-212) around the chunk pipeline. Same command line for the two
// compress modes, same preprocessor (the reference's own object code, out of the engine's scope by contract), same
// container, same bits -- and no reference model object anywhere in the link (oracle/Makefile target `engine`):
//
//     cmix_engine -c [dictionary] input output        cmix_engine -n input output
//
// Per chunk: cmx_pipeline_submit = PPMd + the fxcm / paq8 text parsers on this thread, every learning stage and the final
// mixing network on the MI355X; the probabilities come back a chunk at a time and feed the arithmetic coder
// (cmx_encoder_*). tests/test_gpu_dropin.py compares its files with the reference binary's, byte for byte.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "preprocess/preprocessor.h"   // pulls in integration/predictor_engine.h through its PREDICTOR_H guard

char* dictionary_path = NULL;  // the dictionary's path, handed to the fxcm stage (cmx_pipeline_enable_fxcm)

namespace {
const size_t kChunk = 4096;
const int kMinVocabFileSize = 10000;  // runner.cpp:14

int Help() {
  printf("look-ahead compressor (cmix v21 container):\n"
         "    with dictionary:    cmix_engine -c [dictionary] [input] [output]\n"
         "    without dictionary: cmix_engine -c [input] [output]\n"
         "    no preprocessing:   cmix_engine -n [input] [output]\n"
         "decompress with the per-bit build (cmix_hybrid -d) or the reference.\n");
  return -1;
}

void CompressEngine(Predictor* P, const std::vector<uint8_t>& data, cmx_encoder_t* enc) {
  const size_t N = data.size();
  if (N == 0) return;
  const int dev = P->device();
  const size_t C = P->chunk(), T = 8 * C;
  // CMX_PIPELINE_SLOTS chunks in flight: each owns a layer-0 matrix and a p[] buffer on the device; p[] comes back and is coded as
  // soon as the chunk has left the mixing network, so memory does not grow with the input and output appears as it goes
  constexpr size_t R = CMX_PIPELINE_SLOTS;
  float* d_layer0[R]; float* d_p[R];
  for (size_t i = 0; i < R; ++i) {
    d_layer0[i] = (float*)cmx_device_alloc(dev, T * CMX_N_INPUTS * sizeof(float));
    d_p[i] = (float*)cmx_device_alloc(dev, T * sizeof(float));
    if (!d_layer0[i] || !d_p[i]) Predictor::Die();
  }
  float* const p = (float*)cmx_host_alloc(T * sizeof(float));   // page-locked
  if (!p) Predictor::Die();
  const size_t nchunks = (N + C - 1) / C;
  auto len = [&](size_t c) { return c + 1 < nchunks ? C : N - c * C; };
  auto drain = [&](size_t c) {   // chunk c: wait, copy its probabilities back, code its bytes
    if (cmx_pipeline_fetch(P->pipe(), c, p)) Predictor::Die();   // waits for this chunk only
    if (cmx_encoder_encode_bytes(enc, p, data.data() + c * C, len(c))) Predictor::Die();
  };
  for (size_t c = 0; c < nchunks; ++c) {
    if (c >= R) drain(c - R);   // frees slot c % R
    if (cmx_pipeline_submit(P->pipe(), data.data() + c * C, len(c), d_layer0[c % R], d_p[c % R])) Predictor::Die();
    fprintf(stderr, "\rprogress: %.2f%%", 100.0 * (c + 1) / nchunks);
  }
  for (size_t c = nchunks > R ? nchunks - R : 0; c < nchunks; ++c) drain(c);
  if (cmx_pipeline_sync(P->pipe())) Predictor::Die();
  for (size_t i = 0; i < R; ++i) { cmx_device_free(dev, d_layer0[i]); cmx_device_free(dev, d_p[i]); }
  cmx_host_free(p);
}
}  // namespace

int main(int argc, char* argv[]) {
  setenv("GPU_MAX_HW_QUEUES", "16", 0);   // one hardware queue per stage stream (HIP's default of 4 would serialise the stages)
  if (argc < 4 || argc > 5 || strlen(argv[1]) != 2 || argv[1][0] != '-' || (argv[1][1] != 'c' && argv[1][1] != 'n'))
    return Help();
  const bool enable_preprocess = argv[1][1] == 'c';
  std::string input_path = argv[2], output_path = argv[3];
  FILE* dictionary = NULL;
  if (argc == 5) {
    if (!enable_preprocess) return Help();
    dictionary = fopen(argv[2], "rb");
    if (!dictionary) return Help();
    dictionary_path = argv[2];
    input_path = argv[3];
    output_path = argv[4];
  }
  const std::string temp_path = output_path + ".cmix.temp";
  // ---- the vocabulary-independent stages of the engine (mixing network, paq8 and fxcm stages: ~16 GB of tables to
  // allocate and initialise) are built on a library thread WHILE this thread preprocesses the input and scans the vocabulary ----
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  const bool timing = getenv("CMIX_TIMING") != NULL;
  {
    const char* dev = getenv("CMIX_DEVICE");
    if (!getenv("CMIX_NO_PREWARM") && cmx_prewarm(dev ? atoi(dev) : 0, dictionary_path, 1)) Predictor::Die();
  }
  // ---- runner.cpp:166-186: the reference's preprocessor into a temp file ----
  FILE* data_in = fopen(input_path.c_str(), "rb");
  FILE* temp_out = data_in ? fopen(temp_path.c_str(), "wb") : NULL;
  if (!data_in || !temp_out) return Help();
  fseek(data_in, 0L, SEEK_END);
  const unsigned long long input_bytes = ftell(data_in);
  fseek(data_in, 0L, SEEK_SET);
  if (enable_preprocess) preprocessor::Encode(data_in, temp_out, false, input_bytes, temp_path, dictionary);
  else preprocessor::NoPreprocess(data_in, temp_out, input_bytes);
  fclose(data_in);
  fclose(temp_out);
  std::vector<uint8_t> data;
  {
    FILE* f = fopen(temp_path.c_str(), "rb");
    if (!f) return Help();
    fseek(f, 0L, SEEK_END);
    data.resize(ftell(f));
    fseek(f, 0L, SEEK_SET);
    if (!data.empty() && fread(data.data(), 1, data.size(), f) != data.size()) return Help();
    fclose(f);
    remove(temp_path.c_str());
  }
  // ---- runner.cpp:196-205: vocabulary, header, predictor, pretraining ----
  std::vector<bool> vocab(256, false);
  if (data.size() < (size_t)kMinVocabFileSize) std::fill(vocab.begin(), vocab.end(), true);
  else for (uint8_t b : data) vocab[b] = true;
  uint8_t v8[256], hdr[CMX_HEADER_MAX];
  for (int i = 0; i < 256; ++i) v8[i] = vocab[i];
  const size_t hn = cmx_header_write(data.size(), v8, dictionary != NULL, hdr);
  if (!hn) Predictor::Die();
  const double t_pre = since();
  Predictor p(vocab, kChunk);
  const double t_ready = since();
  if (enable_preprocess) preprocessor::Pretrain(&p, dictionary);
  p.FlushPretrain();
  const double t_trained = since();
  // ---- runner.cpp:101-119: the coding loop, a chunk at a time ----
  cmx_encoder_t* enc = cmx_encoder_create();
  CompressEngine(&p, data, enc);
  cmx_encoder_flush(enc);
  FILE* out = fopen(output_path.c_str(), "wb");
  if (!out) return Help();
  fwrite(hdr, 1, hn, out);
  fwrite(cmx_encoder_data(enc), 1, cmx_encoder_size(enc), out);
  const unsigned long long output_bytes = hn + cmx_encoder_size(enc);
  fclose(out);
  cmx_encoder_destroy(enc);
  if (timing)
    fprintf(stderr, "\ncmix_engine timing: preprocessing + vocabulary %.2f s | engine ready +%.2f s (%s) | pretraining +%.2f s | coding +%.2f s | total %.2f s\n", t_pre,
            t_ready - t_pre, getenv("CMIX_NO_PREWARM") ? "built after preprocessing" : "vocabulary-independent stages built during preprocessing", t_trained - t_ready,
            since() - t_trained, since());
  printf("\r%llu bytes -> %llu bytes (engine: all model families on the device).\n", input_bytes, output_bytes);
  return 0;
}
