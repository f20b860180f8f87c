// fxcm_rec.h -- the interface between the two halves of the fxcm stage: the host text parser
// (fxcm_parser_host.cpp) emits one FxByteRec per input byte, the device numeric core (fxcm_dev.h, launched by
// fxcm_stage.hip) consumes a chunk of them. Also the static geometry of the model (reference
// src/models/fxcmv1.cpp:3313-3405: 31 hashed context maps holding 81 context slots, listed here in the order
// modelPrediction mixes them, :4601-4640) that both halves index by.
#ifndef CMX_FXCM_REC_H
#define CMX_FXCM_REC_H
#include <stdint.h>

enum { FX_NMAPS = 31, FX_NSLOTS = 81, FX_NSSCM = 7, FX_OUTPUTS = 431, FX_TX = 512, FX_NMIX1 = 10 };

// One row per context map, in mixing order. kind 0/1/2 = ContextMap (64-byte buckets of 7 slots) / ContextMap1 (32 B, 3)
// / ContextMap2 (128 B, 14; the table is twice the size argument, :1443). idx = index in the reference's cmC / cmC1 / cmC2
// arrays; size = Init's size argument in bytes; C = contexts per byte; prm = row of the c_r / c_s / c_s3 / c_s4 parameter
// vectors (:3222-3230); sta = state table (0..5 = STA1, 2, 4, 5, 6, 7); keep = E::get's keep flag; u = the st2 input is
// emitted; st2 = which st2 table (0 zeros, 1 st2_p1, 2 st2_p2).
struct FxMapDef { uint8_t kind, idx, C, prm, sta, keep, u, st2; uint32_t size; };
#define FXG (4096u * 4096u)
static const FxMapDef FX_MAPS[FX_NMAPS] = {
    {2, 0, 3, 0, 4, 0xf0, 1, 1, 8 * FXG},       {2, 1, 1, 1, 4, 0xf0, 1, 1, 16 * FXG},      {2, 2, 1, 2, 4, 0xf0, 1, 1, 8 * FXG},
    {2, 3, 1, 3, 4, 0xf0, 1, 1, 8 * FXG},       {2, 4, 2, 4, 4, 0xf0, 1, 1, 8 * FXG},       {2, 5, 6, 5, 4, 0xf0, 1, 1, 8 * FXG},
    {2, 6, 1, 6, 0, 0x00, 1, 1, FXG / 64},      {2, 7, 1, 7, 3, 0xf0, 1, 1, 2 * FXG},       {2, 8, 4, 8, 2, 0x00, 1, 1, 4 * FXG},
    {1, 0, 2, 9, 4, 0x00, 0, 0, 32 * 4096},     {1, 1, 3, 10, 5, 0x00, 1, 1, 2 * 32 * 4096}, {1, 2, 4, 11, 1, 0x00, 1, 1, 32 * 4096},
    {1, 4, 5, 12, 5, 0x00, 1, 1, 16 * 4096},    {0, 0, 7, 13, 1, 0x00, 1, 1, 16 * 4096},    {0, 1, 3, 14, 3, 0xf0, 0, 0, 64 * 2 * 4096},
    {0, 2, 2, 15, 1, 0xf0, 0, 0, 2 * 4096},     {1, 3, 2, 16, 0, 0x00, 0, 0, 128 * 4096},   {2, 9, 4, 17, 4, 0xf0, 1, 1, 8 * FXG},
    {2, 10, 6, 18, 3, 0xf0, 1, 1, 8 * FXG},     {2, 11, 5, 19, 3, 0xf0, 1, 1, 8 * FXG},     {2, 12, 2, 20, 4, 0xf0, 1, 1, 8 * FXG},
    {2, 13, 2, 21, 4, 0xf0, 1, 1, 16 * FXG},    {0, 3, 2, 22, 1, 0x00, 1, 2, 32 * 4096},    {2, 14, 1, 23, 4, 0xf0, 1, 1, 2 * FXG},
    {2, 15, 1, 24, 0, 0x00, 0, 0, 8 * 64 * 4096}, {0, 4, 1, 25, 0, 0xf0, 1, 1, 512 * 4096}, {0, 5, 1, 26, 0, 0xf0, 1, 1, 512 * 4096},
    {2, 16, 1, 17, 4, 0xf0, 1, 1, FXG / 2},     {2, 17, 2, 17, 4, 0xf0, 1, 1, 2 * FXG},     {1, 6, 1, 5, 4, 0x00, 0, 1, 16 * 4096},
    {1, 7, 4, 12, 1, 0x00, 1, 1, 16 * 4096}};
#undef FXG

// What the parser hands over for one byte (all values as they stand when byte_update returns, i.e. what the eight
// per-bit passes of modelPrediction for this byte position read).
struct FxByteRec {
  uint32_t cx[FX_NSLOTS];   // hashed contexts as ContextMap::set stores them (:1057-1065), slot = first slot of the map (mixing order) + call index
  uint32_t skip[3];         // bit s set: slot s was skipped (sets(), :1066-1070)
  uint32_t rcm_cx;          // RunContextMap::set context (:4327)
  uint32_t sscm[FX_NSSCM];  // SmallStationaryContextMap contexts (:4577-4583), unmasked
  uint32_t mh[4];           // MatchModel2's hashes: t[9], t[7], t[5], last word of the sentence list (:3608-3640)
  uint32_t s2, s3, s3R, s2R;                         // stream2b, stream3b, stream3bR, stream2bR
  uint32_t AH1, AH2, x5;                             // APM contexts (:4592-4593)
  uint32_t deccode;                                  // selector of mixer 8 (decoded dictionary word / 2-bit stream)
  uint8_t pc1;                                       // the parser's c1 ("&!" arrives as a space)
  uint8_t BrFc, FcIdx, words, numbers, isPar;        // BrFcIdx, FcIdx, words, numbers, isParagraph
  uint8_t pad[2];
};

#ifdef __cplusplus
extern "C" {
#endif
// host parser (fxcm_parser_host.cpp). dictionary_path may be NULL (cmix -n / -c without a dictionary).
struct FxParser;
struct FxParser* fxp_create(const char* dictionary_path);
void fxp_destroy(struct FxParser* p);
// consume n bytes, write n records; returns 0, or -1 if a map received a number of contexts other than its C
int fxp_run(struct FxParser* p, const uint8_t* bytes, int n, struct FxByteRec* out);
// test hook: the position in the block (several thresholds of the model depend on it, up to 463 139 793)
void fxp_set_blpos(struct FxParser* p, int blpos);
#ifdef __cplusplus
}
#endif
#endif
