// integration/compress_lookahead.cpp -- compression with look-ahead: what runner.cpp's RunCompression + Compress do
// (runner.cpp:101-119,162-212), restated around the chunk pipeline. Same command line for the two compress modes, same
// preprocessor (the reference's own object code), same container, same bits:
//
//     cmix_lookahead -c [dictionary] input output        cmix_lookahead -n input output
//
// Per chunk of CHUNK bytes: cmx_pipeline_begin (PPMd on this thread; context stage and LSTM on the MI355X),
// cmx_pipeline_hints (the LSTM's per-bit lstmpr/lstmex, which fxcm reads), then the two host model families run over
// the chunk on two threads -- they only depend on the bits and on those hints, never on the mixer -- writing their
// 2022 columns per bit, and cmx_pipeline_finish uploads the columns and runs the mixing network while the host is
// already in the next chunk. The probabilities come back once and feed the arithmetic coder (cmx_encoder_*).
// With CMX_FXCM_DEVICE=1 in the environment the fxcm family runs as a device stage instead (cmx_pipeline_enable_fxcm):
// no hints come back, only paq8 runs on a host thread and only its 1591 columns are uploaded.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "preprocess/preprocessor.h"   // pulls in integration/predictor_lookahead.h through its PREDICTOR_H guard

char* dictionary_path = NULL;  // read by fxcm (fxcmv1.cpp:48,412-428)
int lstmpr = 0, lstmex = 0;    // read by fxcm (fxcmv1.cpp:47,4740-4753)

namespace {
const size_t kChunk = 1024;
const int kMinVocabFileSize = 10000;  // runner.cpp:14

int Help() {
  printf("look-ahead compressor (cmix v21 container):\n"
         "    with dictionary:    cmix_lookahead -c [dictionary] [input] [output]\n"
         "    without dictionary: cmix_lookahead -c [input] [output]\n"
         "    no preprocessing:   cmix_lookahead -n [input] [output]\n"
         "decompress with the per-bit build (cmix_hybrid -d) or the reference.\n");
  return -1;
}

inline unsigned DiscretizeHint(float p) {  // predictor.cpp:180-182
  volatile float prod = 4094 * p;
  return (unsigned)(1 + prod);
}

void CompressLookahead(Predictor* P, const std::vector<uint8_t>& data, cmx_encoder_t* enc) {
  const size_t N = data.size();
  if (N == 0) return;
  const int dev = P->device();
  const size_t C = P->chunk(), T = 8 * C;
  // CMX_PIPELINE_SLOTS chunks in flight: each owns a layer-0 matrix and a p[] buffer on the device; p[] comes back and is coded
  // as soon as the chunk has left the mixing network, so memory does not grow with the input and output appears as it goes
  constexpr size_t R = CMX_PIPELINE_SLOTS;
  float* d_layer0[R]; float* d_p[R];
  for (size_t i = 0; i < R; ++i) {
    d_layer0[i] = (float*)cmx_device_alloc(dev, T * CMX_N_INPUTS * sizeof(float));
    d_p[i] = (float*)cmx_device_alloc(dev, T * sizeof(float));
    if (!d_layer0[i] || !d_p[i]) Predictor::Die();
  }
  float* cols = (float*)cmx_host_alloc(T * 2022 * sizeof(float));
  if (!cols) Predictor::Die();
  std::vector<float> p(T);
  std::vector<float> hint_p(T + 1);
  std::vector<int> hint_ex(T + 1);
  const size_t nchunks = (N + C - 1) / C;
  auto len = [&](size_t c) { return c + 1 < nchunks ? C : N - c * C; };
  auto drain = [&](size_t c) {   // chunk c: wait, copy its probabilities back, code its bytes
    if (cmx_pipeline_fetch(P->pipe(), c, p.data())) Predictor::Die();   // waits for this chunk only
    if (cmx_encoder_encode_bytes(enc, p.data(), data.data() + c * C, len(c))) Predictor::Die();
  };
  if (cmx_pipeline_begin(P->pipe(), data.data(), len(0), d_layer0[0])) Predictor::Die();
  for (size_t c = 0; c < nchunks; ++c) {
    const size_t n = len(c);
    const uint8_t* bytes = data.data() + c * C;
    // CMX_FXCM_DEVICE=1: fxcm runs as a device stage behind cmx_pipeline_begin (its hints never leave the device)
    if (!P->fxcm_on_device() && cmx_pipeline_hints(P->pipe(), hint_p.data(), hint_ex.data())) Predictor::Die();
    // the host model families over the chunk: Predict() then Perceive(bit) per bit, as Predictor::Predict / Perceive
    // order them (predictor.cpp:363-369,422-425,462-467); each only touches its own state and its own columns
    std::thread tp([&] {
      Model* m = P->paq8();
      for (size_t t = 0; t < 8 * n; ++t) {
        const std::valarray<float>& o = m->Predict();
        float* row = cols + t * 2022 + 431;
        for (size_t i = 0; i < o.size(); ++i) row[i] = o[i];
        m->Perceive((bytes[t >> 3] >> (7 - (t & 7))) & 1);
      }
    });
    if (!P->fxcm_on_device()) {
      Model* m = P->fxcm();
      for (size_t t = 0; t < 8 * n; ++t) {
        const std::valarray<float>& o = m->Predict();
        float* row = cols + t * 2022;
        for (size_t i = 0; i < o.size(); ++i) row[i] = o[i];
        lstmpr = (int)DiscretizeHint(hint_p[t + 1]);  // what Perceive leaves behind before fxcm's turn (:462-466)
        lstmex = hint_ex[t + 1];
        m->Perceive((bytes[t >> 3] >> (7 - (t & 7))) & 1);
      }
    }
    // fxcm is the lighter family: while paq8 is still at it, this thread runs PPMd for the next chunk and enqueues
    // its context stage and LSTM, whose hints are ready long before the next iteration asks for them
    if (c + 1 >= R) drain(c + 1 - R);   // frees slot (c + 1) % R: its layer-0 matrix and p[] buffer
    if (c + 1 < nchunks && cmx_pipeline_begin(P->pipe(), bytes + C, len(c + 1), d_layer0[(c + 1) % R])) Predictor::Die();
    tp.join();
    if (P->fxcm_on_device()) {   // only paq8's 1591 columns (434..2024) go up; rows stay 2022 floats apart
      std::vector<float> pq(8 * n * 1591);
      for (size_t t = 0; t < 8 * n; ++t) memcpy(&pq[t * 1591], cols + t * 2022 + 431, 1591 * sizeof(float));
      if (cmx_pipeline_finish_cols(P->pipe(), pq.data(), 434, 1591, d_p[c % R])) Predictor::Die();
    } else if (cmx_pipeline_finish(P->pipe(), cols, d_p[c % R])) Predictor::Die();
    fprintf(stderr, "\rprogress: %.2f%%", 100.0 * (c + 1) / nchunks);
  }
  for (size_t c = nchunks + 1 > R ? nchunks + 1 - R : 0; c < nchunks; ++c) drain(c);
  if (cmx_pipeline_sync(P->pipe())) Predictor::Die();
  for (size_t i = 0; i < R; ++i) { cmx_device_free(dev, d_layer0[i]); cmx_device_free(dev, d_p[i]); }
  cmx_host_free(cols);
}
}  // namespace

int main(int argc, char* argv[]) {
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
  Predictor p(vocab, kChunk);
  if (enable_preprocess) preprocessor::Pretrain(&p, dictionary);
  p.FlushPretrain();
  // ---- runner.cpp:101-119: the coding loop, a chunk at a time ----
  cmx_encoder_t* enc = cmx_encoder_create();
  CompressLookahead(&p, data, enc);
  cmx_encoder_flush(enc);
  FILE* out = fopen(output_path.c_str(), "wb");
  if (!out) return Help();
  fwrite(hdr, 1, hn, out);
  fwrite(cmx_encoder_data(enc), 1, cmx_encoder_size(enc), out);
  const unsigned long long output_bytes = hn + cmx_encoder_size(enc);
  fclose(out);
  cmx_encoder_destroy(enc);
  printf("\r%llu bytes -> %llu bytes (look-ahead chunk pipeline).\n", input_bytes, output_bytes);
  return 0;
}
