// fxcm_parser_host.cpp -- HOST stage of the fxcm model (cmix v21 vendors it as src/models/fxcmv1.cpp): the text
// parser. Everything fxcm's modelPrediction does at a byte boundary (fxcmv1.cpp:3802-4600) is a pure function of the
// byte stream (and of cmix's WRT dictionary): quantised 2/3/4-bit byte streams, words / stems / word classes
// (Word + EnglishStemmer, :2302-3216), sentence / paragraph / stream word lists (WordsContext :2157-2274), brackets,
// quotes, first characters (BracketContext :1932-1998), wiki tables and columns (ColumnContext :2000-2155), numbers,
// indirect histories. It ends in 81 ContextMap::set()/sets() calls, 7 SmallStationaryContextMap::set(), one
// RunContextMap::set() and a handful of registers the per-bit mixer selectors read. This file runs that part on a
// host thread -- like the PPMd stage, it is branchy pointer/byte work that runs ahead of the device -- and emits one
// FxByteRec per input byte; the learned tables (context maps, state maps, mixers, APMs, match models) live on the
// device and are driven by cmx_fxcm_chunk_kernel (fxcm_dev.h / fxcm_stage.hip).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fxcm_rec.h"
#include "cmx_fxcm_tables.h"
#include "cmx_fxcm_stem_tables.h"

namespace {
#define FX_MAX_WORD 64
struct P8Word { uint8_t Letters[FX_MAX_WORD]; uint8_t Start, End; };   // Word's letters (fxcmv1.cpp:2302-2314)
struct FxWord { P8Word w; uint32_t Hash, Type, Suffix, Preffix; };
#define COUNT(a) ((int)(sizeof(a) / sizeof((a)[0])))
static void p8w_add(P8Word* w, int c) {   // the stemmer only appends literal lower-case letters
  if (w->End < FX_MAX_WORD - 1) { w->End += (w->Letters[w->End] > 0); w->Letters[w->End] = (uint8_t)c; }
}
static uint8_t p8w_at(const P8Word* w, int i) { return (w->End - w->Start >= (uint8_t)i) ? w->Letters[w->Start + (uint8_t)i] : 0; }
static uint8_t p8w_back(const P8Word* w, int i) { return (w->End - w->Start >= (uint8_t)i) ? w->Letters[w->End - (uint8_t)i] : 0; }
static uint32_t p8w_len(const P8Word* w) { return w->Letters[w->Start] != 0 ? (uint32_t)(w->End - w->Start + 1) : 0; }
static int p8w_eq(const P8Word* w, const char* s) {
  const size_t len = strlen(s);
  return (size_t)(w->End - w->Start + (w->Letters[w->Start] != 0)) == len && memcmp(&w->Letters[w->Start], s, len) == 0;
}
static int p8w_ends(const P8Word* w, const char* s) {
  const size_t len = strlen(s);
  return p8w_len(w) > len && memcmp(&w->Letters[w->End - len + 1], s, len) == 0;
}
static int p8w_starts(const P8Word* w, const char* s) {
  const size_t len = strlen(s);
  return p8w_len(w) > len && memcmp(&w->Letters[w->Start], s, len) == 0;
}
static int p8w_change_suffix(P8Word* w, const char* old_suffix, const char* new_suffix) {
  const size_t len = strlen(old_suffix);
  if (p8w_len(w) > len && memcmp(&w->Letters[w->End - len + 1], old_suffix, len) == 0) {
    const size_t n = strlen(new_suffix);
    if (n > 0) {
      const int lim = (FX_MAX_WORD - 1 < w->End + (int)n ? FX_MAX_WORD - 1 : w->End + (int)n) - w->End;
      memcpy(&w->Letters[w->End - (int)len + 1], new_suffix, (size_t)lim);
      const int e = w->End - (int)len + (int)n;
      w->End = (uint8_t)(FX_MAX_WORD - 1 < e ? FX_MAX_WORD - 1 : e);
    } else w->End -= (uint8_t)len;
    return 1;
  }
  return 0;
}
static int p8w_matches_any(const P8Word* w, const char* const* a, int count) {
  const size_t len = p8w_len(w);
  int i = 0;
  for (; i < count && (len != strlen(a[i]) || memcmp(&w->Letters[w->Start], a[i], len) != 0); i++) {}
  return i < count;
}
// ---- Word + EnglishStemmer (fxcmv1.cpp:2302-3216) ----
/* EngWordTypeFlags / ...Negation / ...Suffix :2370-2413 (bit positions are part of the hashed state) */
enum { FXT_Verb = 1, FXT_Noun = 2, FXT_Adjective = 4, FXT_Plural = 8, FXT_PastTense = (1 << 5) | 1, FXT_PresentParticiple = (1 << 4) | 1,
       FXT_AdjectiveSuperlative = (1 << 5) | 4, FXT_AdjectiveWithout = (1 << 6) | 4, FXT_AdjectiveFull = (1 << 7) | 4, FXT_AdverbOfManner = 1 << 8,
       FXT_Suffix = 1 << 9, FXT_Prefix = 1 << 10, FXT_Male = 1 << 11, FXT_Female = 1 << 13, FXT_Article = 1 << 14, FXT_Conjunction = 1 << 15,
       FXT_Adposition = 1 << 16, FXT_Number = 1 << 17, FXT_ConjunctiveAdverb = 1 << 19 };
enum { FXP_Negation = 1, FXP_PrefixIrr = 2 | 1, FXP_PrefixOver = 4, FXP_PrefixUnder = 8, FXP_PrefixUnn = 16 | 1, FXP_PrefixNon = 32 | 1,
       FXP_PrefixAnti = 64 | 1, FXP_PrefixDis = 128 | 1 };
enum { FXS_SuffixIVE = 1 << 8 };
static const char FXW_Vowels[] = {'a', 'e', 'i', 'o', 'u', 'y'}, FXW_Doubles[] = {'b', 'd', 'f', 'g', 'm', 'n', 'p', 'r', 't'},
                  FXW_LiEndings[] = {'c', 'd', 'e', 'g', 'h', 'k', 'm', 'n', 'r', 't'}, FXW_NonShortConsonants[] = {'w', 'x', 'Y'},
                  FXW_nAllowed[] = {'a', 'e', 'i', 'o'}, FXW_Allowed1[] = {'b', 'c', 'd', 'f', 'g', 'k', 'p', 't', 'y', 'z'},
                  FXW_Allowed2[] = {'a', 'i', 'o', 'u'}, FXW_Allowed[] = {'a', 'd', 'e', 'i', 'l', 'r', 'u'};   /* :2417-2424, :2915-2940 */

static int in_set(int c, const char* a, int n) { int i = 0; for (; i < n && (char)c != a[i]; i++) {} return i < n; }
static int suffix_in_rn(const P8Word* x, uint32_t rn, const char* suffix) { return x->Start != x->End && rn <= p8w_len(x) - (uint32_t)strlen(suffix); }
static void fxw_add(FxWord* W, int c) {  /* Word::operator+= :2315-2320: signed char, bytes >= 0x80 are dropped */
  P8Word* x = &W->w;
  if ((signed char)c > 0 && x->End < FX_MAX_WORD - 1) { x->End += (x->Letters[x->End] > 0); x->Letters[x->End] = (uint8_t)c; }
}
#define w (&W->w)

static int fx_is_vowel(int c) { return in_set(c, FXW_Vowels, COUNT(FXW_Vowels)); }
#define V(c) fx_is_vowel(c)
#define CONS(c) (!fx_is_vowel(c))
#define B(i) p8w_back(w, i)
#define F(i) p8w_at(w, i)
static uint32_t fxs_region(FxWord* W, uint32_t from) {  /* GetRegion :2689-2700: ends at Length(), not Start + Length() */
  int has_vowel = 0;
  for (int i = w->Start + (int)from; i <= w->End; i++) {
    if (V(w->Letters[i])) { has_vowel = 1; continue; }
    else if (has_vowel) return (uint32_t)(i - w->Start + 1);
  }
  return p8w_len(w);
}
static uint32_t fxs_region1(FxWord* W) {
  for (int i = 0; i < COUNT(FXW_ExceptionsRegion1); i++)
    if (p8w_starts(w, FXW_ExceptionsRegion1[i])) return (uint32_t)strlen(FXW_ExceptionsRegion1[i]);
  return fxs_region(W, 0);
}
static int fxs_short_syllable(FxWord* W) {
  if (w->End == w->Start) return 0;
  if (w->End == w->Start + 1) return V(B(1)) && CONS(B(0));
  return CONS(B(2)) && V(B(1)) && CONS(B(0)) && !in_set(B(0), FXW_NonShortConsonants, COUNT(FXW_NonShortConsonants));
}
static int fxs_short_word(FxWord* W) { return fxs_short_syllable(W) && fxs_region1(W) == p8w_len(w); }
static int fxs_has_vowels(FxWord* W) { for (int i = w->Start; i <= w->End; i++) if (V(w->Letters[i])) return 1; return 0; }
static void fxs_hash(FxWord* W) {  /* Hash :2683-2687: one 32-bit hash of the stem */
  W->Hash = 0xb0a710ad;
  for (int i = w->Start; i <= w->End; i++) W->Hash = W->Hash * 263 * 32 + w->Letters[i];
}
static int fxs_prefixes(FxWord* W) {  /* ProcessPrefixes :2752-2771: the prefix kind goes to its own flag word */
  int skip, kind;
  if (p8w_starts(w, "irr") && p8w_len(w) > 5 && (F(3) == 'a' || F(3) == 'e')) { skip = 2; kind = FXP_PrefixIrr; }
  else if (p8w_starts(w, "over") && p8w_len(w) > 5) { skip = 4; kind = FXP_PrefixOver; }
  else if (p8w_starts(w, "under") && p8w_len(w) > 6) { skip = 5; kind = FXP_PrefixUnder; }
  else if (p8w_starts(w, "unn") && p8w_len(w) > 5) { skip = 2; kind = FXP_PrefixUnn; }
  else if (p8w_starts(w, "non") && p8w_len(w) > (uint32_t)(5 + (F(3) == '-'))) { skip = 2 + (F(3) == '-'); kind = FXP_PrefixNon; }
  else if (p8w_starts(w, "anti") && p8w_len(w) > 6 && F(4) == '-') { skip = 5; kind = FXP_PrefixAnti; }
  else if (p8w_starts(w, "dis") && p8w_len(w) > 5 && F(3) == '-') { skip = 3; kind = FXP_PrefixDis; }
  else return 0;
  w->Start += (uint8_t)skip; W->Type |= FXT_Prefix; W->Preffix |= (uint32_t)kind;
  return 1;
}
static int fxs_superlatives(FxWord* W) {  
  if (p8w_ends(w, "est") && p8w_len(w) > 4) {
    const uint8_t keep = w->End;
    w->End -= 3;
    W->Type |= FXT_AdjectiveSuperlative;
#define UNDO() do { w->End = keep; W->Type &= ~(uint64_t)FXT_AdjectiveSuperlative; } while (0)
    if (B(0) == B(1) && B(0) != 'r' && !(p8w_len(w) >= 4 && memcmp("sugg", &w->Letters[w->End - 3], 4) == 0)) {
      w->End -= (((B(0) != 'f' && B(0) != 'l' && B(0) != 's') || (p8w_len(w) > 4 && B(1) == 'l' && (B(2) == 'u' || B(3) == 'u' || B(3) == 'v'))) &&
                 (!(p8w_len(w) == 3 && B(1) == 'd' && B(2) == 'o')));
      if (p8w_len(w) == 2 && (F(0) != 'i' || F(1) != 'n')) UNDO();
    } else {
      switch (B(0)) {
        case 'd': case 'k': case 'm': case 'y': break;
        case 'g':
          if (!(p8w_len(w) > 3 && (B(1) == 'n' || B(1) == 'r') && memcmp("cong", &w->Letters[w->End - 3], 4) != 0)) UNDO();
          else w->End += (B(2) == 'a');
          break;
        case 'i': w->Letters[w->End] = 'y'; break;
        case 'l':
          if (w->End == w->Start + 1 || memcmp("mo", &w->Letters[w->End - 2], 2) == 0) UNDO();
          else w->End += CONS(B(1));
          break;
        case 'n': if (p8w_len(w) < 3 || CONS(B(1)) || CONS(B(2))) UNDO(); break;
        case 'r':
          if (p8w_len(w) > 3 && V(B(1)) && V(B(2))) w->End += (B(2) == 'u') && (B(1) == 'a' || B(1) == 'i');
          else UNDO();
          break;
        case 's': w->End++; break;
        case 'w': if (!(p8w_len(w) > 2 && V(B(1)))) UNDO(); break;
        case 'h': if (!(p8w_len(w) > 2 && CONS(B(1)))) UNDO(); break;
        default: w->End += 3; W->Type &= ~(uint64_t)FXT_AdjectiveSuperlative;
      }
    }
#undef UNDO
  }
  return (W->Type & FXT_AdjectiveSuperlative) > 0;
}
static int fxs_step0(FxWord* W) {
  for (int i = 0; i < COUNT(FXW_SuffixesStep0); i++)
    if (p8w_ends(w, FXW_SuffixesStep0[i])) { w->End -= (uint8_t)strlen(FXW_SuffixesStep0[i]); W->Type |= FXT_Plural; return 1; }
  return 0;
}
static int fxs_step1a(FxWord* W) {
  if (p8w_ends(w, "sses")) { w->End -= 2; W->Type |= FXT_Plural; return 1; }
  if (p8w_ends(w, "ied") || p8w_ends(w, "ies")) {
    W->Type |= (B(0) == 'd') ? FXT_PastTense : FXT_Plural;
    w->End -= 1 + (p8w_len(w) > 4);
    return 1;
  }
  if (p8w_ends(w, "us") || p8w_ends(w, "ss")) return 0;
  if (B(0) == 's' && p8w_len(w) > 2)
    for (int i = w->Start; i <= w->End - 2; i++)
      if (V(w->Letters[i])) { w->End--; W->Type |= FXT_Plural; return 1; }
  if (p8w_ends(w, "n't") && p8w_len(w) > 4) {
    switch (B(3)) {
      case 'a': if (B(4) == 'c') w->End -= 2; else p8w_change_suffix(w, "n't", "ll"); break;
      case 'i': p8w_change_suffix(w, "in't", "m"); break;
      case 'o': if (B(4) == 'w') p8w_change_suffix(w, "on't", "ill"); else w->End -= 3; break;
      default: w->End -= 3;
    }
    W->Type |= FXT_Prefix; W->Preffix |= FXP_Negation;  /* a suffix filed as a prefix */
    return 1;
  }
  if (p8w_ends(w, "hood") && p8w_len(w) > 7) { w->End -= 4; return 1; }
  return 0;
}
static int fxs_step1b(FxWord* W, uint32_t R1) {
  for (int i = 0; i < COUNT(FXW_SuffixesStep1b); i++) {
    if (!p8w_ends(w, FXW_SuffixesStep1b[i])) continue;
    if (i < 2) {
      if (suffix_in_rn(w, R1, FXW_SuffixesStep1b[i])) w->End -= (uint8_t)(1 + i * 2);
    } else {
      const uint8_t j = w->End;
      w->End -= (uint8_t)strlen(FXW_SuffixesStep1b[i]);
      if (!fxs_has_vowels(W)) { w->End = j; return 0; }
      if (p8w_ends(w, "at") || p8w_ends(w, "bl") || p8w_ends(w, "iz") || fxs_short_word(W)) p8w_add(w, 'e');
      else if (p8w_len(w) > 2) {
        if (B(0) == B(1) && in_set(B(0), FXW_Doubles, COUNT(FXW_Doubles))) w->End--;
        else if (i == 2 || i == 3) {
          switch (B(0)) {
            case 'c': case 's': case 'v': w->End += !(p8w_ends(w, "ss") || p8w_ends(w, "ias")); break;
            case 'd': w->End += V(B(1)) && (!in_set(B(2), FXW_nAllowed, COUNT(FXW_nAllowed))); break;
            case 'k': w->End += p8w_ends(w, "uak"); break;
            case 'l': w->End += in_set(B(1), FXW_Allowed1, COUNT(FXW_Allowed1)) || (in_set(B(1), FXW_Allowed2, COUNT(FXW_Allowed2)) && CONS(B(2))); break;
          }
        } else if (i >= 4) {
          switch (B(0)) {
            case 'd': if (V(B(1)) && B(2) != 'a' && B(2) != 'e' && B(2) != 'o') p8w_add(w, 'e'); break;
            case 'g':
              if (in_set(B(1), FXW_Allowed, COUNT(FXW_Allowed)) ||
                  (B(1) == 'n' && (B(2) == 'e' || (B(2) == 'u' && B(3) != 'b' && B(3) != 'd') ||
                                   (B(2) == 'a' && (B(3) == 'r' || (B(3) == 'h' && B(4) == 'c'))) ||
                                   (p8w_ends(w, "ring") && (B(4) == 'c' || B(4) == 'f')))))
                p8w_add(w, 'e');
              break;
            case 'l':
              if (!(B(1) == 'l' || B(1) == 'r' || B(1) == 'w' || (V(B(1)) && V(B(2))))) p8w_add(w, 'e');
              if (p8w_ends(w, "uell") && p8w_len(w) > 4 && B(4) != 'q') w->End--;
              break;
            case 'r':
              if (((B(1) == 'i' && B(2) != 'a' && B(2) != 'e' && B(2) != 'o') ||
                   (B(1) == 'a' && (!(B(2) == 'e' || B(2) == 'o' || (B(2) == 'l' && B(3) == 'l')))) ||
                   (B(1) == 'o' && (!(B(2) == 'o' || (B(2) == 't' && B(3) != 's')))) || B(1) == 'c' || B(1) == 't') &&
                  (!p8w_ends(w, "str")))
                p8w_add(w, 'e');
              break;
            case 't': if (B(1) == 'o' && B(2) != 'g' && B(2) != 'l' && B(2) != 'i' && B(2) != 'o') p8w_add(w, 'e'); break;
            case 'u': if (!(p8w_len(w) > 3 && V(B(1)) && V(B(2)))) p8w_add(w, 'e'); break;
            case 'z':
              if (p8w_ends(w, "izz") && p8w_len(w) > 3 && (B(3) == 'h' || B(3) == 'u')) w->End--;
              else if (B(1) != 't' && B(1) != 'z') p8w_add(w, 'e');
              break;
            case 'k': if (p8w_ends(w, "uak")) p8w_add(w, 'e'); break;
            case 'b': case 'c': case 's': case 'v':
              if (!((B(0) == 'b' && (B(1) == 'm' || B(1) == 'r')) || p8w_ends(w, "ss") || p8w_ends(w, "ias") || p8w_eq(w, "zinc"))) p8w_add(w, 'e');
              break;
          }
        }
      }
    }
    W->Type |= FXW_TypesStep1b[i];
    return 1;
  }
  return 0;
}
static int fxs_step1c(FxWord* W) {  /* :3020-3026 (no case folding here) */
  if (p8w_len(w) > 2 && B(0) == 'y' && CONS(B(1))) { w->Letters[w->End] = 'i'; return 1; }
  return 0;
}
static int fxs_step2(FxWord* W, uint32_t R1) {
  for (int i = 0; i < COUNT(FXW_SuffixesStep2); i++)
    if (p8w_ends(w, FXW_SuffixesStep2[i][0]) && suffix_in_rn(w, R1, FXW_SuffixesStep2[i][0])) {
      p8w_change_suffix(w, FXW_SuffixesStep2[i][0], FXW_SuffixesStep2[i][1]);
      W->Type |= FXW_TypesStep2[i]; W->Suffix |= FXW_TypesStep2Suffix[i];
      return 1;
    }
  if (p8w_ends(w, "logi") && suffix_in_rn(w, R1, "ogi")) { w->End--; return 1; }
  else if (p8w_ends(w, "li")) {
    if (suffix_in_rn(w, R1, "li") && in_set(B(2), FXW_LiEndings, COUNT(FXW_LiEndings))) { w->End -= 2; W->Type |= FXT_AdverbOfManner; return 1; }
    else if (p8w_len(w) > 3) {
      switch (B(2)) {
        case 'b': w->Letters[w->End] = 'e'; W->Type |= FXT_AdverbOfManner; return 1;
        case 'i': if (p8w_len(w) > 4) { w->End -= 2; W->Type |= FXT_AdverbOfManner; return 1; } break;
        case 'l': if (p8w_len(w) > 5 && (B(3) == 'a' || B(3) == 'u')) { w->End -= 2; W->Type |= FXT_AdverbOfManner; return 1; } break;
        case 's': w->End -= 2; W->Type |= FXT_AdverbOfManner; return 1;
        case 'e': case 'g': case 'm': case 'n': case 'r': case 'w':
          if (p8w_len(w) > (uint32_t)(4 + (B(2) == 'r'))) { w->End -= 2; W->Type |= FXT_AdverbOfManner; return 1; }
      }
    }
  }
  return 0;
}
static int fxs_step3(FxWord* W, uint32_t R1, uint32_t R2) {
  int res = 0;
  for (int i = 0; i < COUNT(FXW_SuffixesStep3); i++)
    if (p8w_ends(w, FXW_SuffixesStep3[i][0]) && suffix_in_rn(w, R1, FXW_SuffixesStep3[i][0])) {
      p8w_change_suffix(w, FXW_SuffixesStep3[i][0], FXW_SuffixesStep3[i][1]);
      W->Type |= FXW_TypesStep3[i]; W->Suffix |= FXW_TypesStep3Suffix[i];
      res = 1;
      break;
    }
  if (p8w_ends(w, "ative") && suffix_in_rn(w, R2, "ative")) { w->End -= 5; W->Type |= FXT_Suffix; W->Suffix |= FXS_SuffixIVE; return 1; }
  if (p8w_len(w) > 5 && p8w_ends(w, "less")) { w->End -= 4; W->Type |= FXT_AdjectiveWithout; return 1; }
  return res;
}
static int fxs_step4(FxWord* W, uint32_t R2) {
  int res = 0;
  for (int i = 0; i < COUNT(FXW_SuffixesStep4); i++)
    if (p8w_ends(w, FXW_SuffixesStep4[i]) && suffix_in_rn(w, R2, FXW_SuffixesStep4[i])) {
      w->End -= (uint8_t)(strlen(FXW_SuffixesStep4[i]) - (i > 17));
      if (i != 10 || B(0) != 'm') { W->Type |= FXW_TypesStep4[i]; W->Suffix |= FXW_TypesStep4Suffix[i]; }
      if (i == 0 && p8w_ends(w, "nti")) { w->End--; res = 1; continue; }
      return 1;
    }
  return res;
}
static int fxs_step5(FxWord* W, uint32_t R1, uint32_t R2) {
  if (B(0) == 'e' && !p8w_eq(w, "here")) {
    if (suffix_in_rn(w, R2, "e")) w->End--;
    else if (suffix_in_rn(w, R1, "e")) { w->End--; w->End += fxs_short_syllable(W); }
    else return 0;
    return 1;
  } else if (p8w_len(w) > 1 && B(0) == 'l' && suffix_in_rn(w, R2, "l") && B(1) == 'l') { w->End--; return 1; }
  return 0;
}
static int fx_stem(FxWord* W, int blpos) {  /* Stem :3143-3206 */
  int res = 0, cnt = 0;
  while (w->Start != w->End && F(0) == '\'') { res = 1; w->Start++; cnt++; }  /* TrimStartingApostrophe :2729-2750 */
  while (w->Start != w->End && B(0) == '\'') { if (cnt == 0) break; w->End--; cnt--; }
  if (B(0) == '-') w->End--;
  if (fxs_prefixes(W)) res = 1;
  if (fxs_superlatives(W)) res = 1;
  for (int i = 0; i < COUNT(FXW_Exceptions1); i++)
    if (p8w_eq(w, FXW_Exceptions1[i][0])) {
      if (i < 11) {
        const size_t len = strlen(FXW_Exceptions1[i][1]);
        memcpy(&w->Letters[w->Start], FXW_Exceptions1[i][1], len);
        w->End = (uint8_t)(w->Start + len - 1);
      }
      fxs_hash(W);
      W->Type |= FXW_TypesExceptions1[i];
      return i < 11;
    }
  if (F(0) == 'y') w->Letters[w->Start] = 'Y';  /* MarkYsAsConsonants */
  for (int i = w->Start + 1; i <= w->End; i++)
    if (V(w->Letters[i - 1]) && w->Letters[i] == 'y') w->Letters[i] = 'Y';
  const uint32_t R1 = fxs_region1(W), R2 = fxs_region(W, R1);
  if (fxs_step0(W)) res = 1;
  if (fxs_step1a(W)) res = 1;
  for (int i = 0; i < COUNT(FXW_Exceptions2); i++)
    if (p8w_eq(w, FXW_Exceptions2[i])) { fxs_hash(W); W->Type |= FXW_TypesExceptions2[i]; return res; }
  if (fxs_step1b(W, R1)) res = 1;
  if (fxs_step1c(W)) res = 1;
  if (fxs_step2(W, R1)) res = 1;
  if (fxs_step3(W, R1, R2)) res = 1;
  if (fxs_step4(W, R2)) res = 1;
  if (fxs_step5(W, R1, R2)) res = 1;
  for (uint8_t i = w->Start; i <= w->End; i++)
    if (w->Letters[i] == 'Y') w->Letters[i] = 'y';
  if (!W->Type || W->Type == FXT_Plural) {  /* closed word classes */
    static const struct { const char* const* list; int n; uint32_t flag; } CLS[] = {
        {FXW_MaleWords, COUNT(FXW_MaleWords), FXT_Male}, {FXW_FemaleWords, COUNT(FXW_FemaleWords), FXT_Female},
        {FXW_ArticleWords, COUNT(FXW_ArticleWords), FXT_Article}, {FXW_ConjWords, COUNT(FXW_ConjWords), FXT_Conjunction},
        {FXW_ApoWords, COUNT(FXW_ApoWords), FXT_Adposition}, {FXW_ConAdVerPrepWords, COUNT(FXW_ConAdVerPrepWords), FXT_ConjunctiveAdverb},
        {FXW_VerbWords1, COUNT(FXW_VerbWords1), FXT_Verb}, {FXW_Numbers, COUNT(FXW_Numbers), FXT_Number}};
    for (int k = 0; k < 8; k++) {
      if (k == 6 && !(blpos < 451531986)) continue;  /* the auxiliary-verb list is switched off late in a 1 GB input (:3198) */
      if (p8w_matches_any(w, CLS[k].list, CLS[k].n)) { res = 1; W->Type |= CLS[k].flag; break; }
    }
  }
  fxs_hash(W);
  return res;
}
#undef w

// ---- the parser (fxcmv1.cpp:1882-2300, :3222-3279, :3680-4600) ----
enum { LF = 10, ESCAPE = 12, SPACE = 32, HTLINK = 31, HTML = 30, APOSTROPHE = 39, QUOTATION = 34, FIRSTUPPER = 64, UPPER = 7, TEXTDATA = 96,
       COLON = 'J', SEMICOLON = 'K', LESSTHAN = 'L', EQUALS = 'M', GREATERTHAN = 'N', QUESTION = 'O', SQUAREOPEN = 91, SQUARECLOSE = 93,
       CURLYOPENING = 'P', VERTICALBAR = 'Q', CURLYCLOSE = 'R', WIKIHEADER = GREATERTHAN, WIKITABLE = '-' };   /* :1852-1876, :2004-2005 (WRT-swapped alphabet) */
enum { T_Verb = 1, T_Noun = 2, T_Adjective = 4, T_Plural = 8, T_PresentParticiple = (1 << 4) | 1, T_AdverbOfManner = 1 << 8, T_Suffix = 1 << 9,
       T_Prefix = 1 << 10, T_Male = 1 << 11, T_Female = 1 << 13, T_Article = 1 << 14, T_Conjunction = 1 << 15, T_Adposition = 1 << 16,
       T_Number = 1 << 17, T_ConjunctiveAdverb = 1 << 19 };
enum { BMASK = 0xffffff, CBMASK = 0xfff, MAXLEN = 62, MINLEN_RM = 3, LEN1 = 5, LEN2 = 7, LEN3 = 9 };
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a < b ? b : a; }

/* ---- vec<T,S> :1882-1930: a stack that wraps at its capacity and never really erases ---- */
typedef struct { int cxt[512]; int size; } Vec512;
#define VPUSH(v, cap, e) do { (v)->cxt[(v)->size++] = (e); (v)->size &= (cap) - 1; } while (0)
#define VPOP(v) do { if ((v)->size > 0) { (v)->cxt[(v)->size] = 0; (v)->size--; } } while (0)
#define VRESET(v) do { (v)->cxt[0] = 0; (v)->size = 0; } while (0)
#define VTOP(v) ((v)->cxt[(v)->size - 1])

/* ---- BracketContext<T> :1932-1998 ---- */
typedef struct { uint32_t context; Vec512 active, distance; const uint16_t* element; int count, doPop, limit, bits; uint32_t cxt, dst; } Bracket;
static void br_init(Bracket* b, const uint16_t* el, int n, int pop, int bits, int limit) {
  memset(b, 0, sizeof *b);
  b->element = el; b->count = n; b->doPop = pop; b->bits = bits; b->limit = limit;
}
static void br_reset(Bracket* b) { VRESET(&b->active); VRESET(&b->distance); b->context = b->cxt = b->dst = 0; }
static void br_update(Bracket* b, int byte) {
  int pop = 0;
  if (b->active.size != 0) {
    int end = 0;
    for (int i = 0; i < b->count; i += 2) if (b->element[i] == VTOP(&b->active) && b->element[i + 1] == byte) end = 1;
    if (end || VTOP(&b->distance) >= b->limit) { VPOP(&b->active); VPOP(&b->distance); pop = b->doPop; }
    else VTOP(&b->distance)++;
  }
  if (!pop) {
    int found = 0;
    for (int i = 0; i < b->count; i += 2) if (b->element[i] == byte) { found = 1; break; }
    if (found) { VPUSH(&b->active, 512, byte); VPUSH(&b->distance, 512, 0); }
  }
  if (b->active.size != 0) {
    const uint32_t tmask = (1u << b->bits) - 1;
    b->cxt = (uint32_t)VTOP(&b->active) & tmask;
    b->dst = (uint32_t)imin(VTOP(&b->distance), (1 << b->bits) - 1) & tmask;
    b->context = (1u << b->bits) * b->cxt + b->dst;
  } else b->context = b->cxt = b->dst = 0;
}
static int br_last(const Bracket* b) { return b->active.size > 1 ? b->active.cxt[b->active.size - 2] : 0; }
static const uint16_t kBrackets[8] = {'(', ')', CURLYOPENING, CURLYCLOSE, '[', ']', LESSTHAN, GREATERTHAN};
static const uint16_t kQuotes[4] = {APOSTROPHE, APOSTROPHE, QUOTATION, QUOTATION};
static const uint16_t kFchar[20] = {FIRSTUPPER, LF, TEXTDATA, LF, COLON, LF, LESSTHAN, GREATERTHAN, EQUALS, LF, SQUAREOPEN, SQUARECLOSE, CURLYOPENING, CURLYCLOSE,
                                    '*', LF, VERTICALBAR, LF, HTLINK, LF};
static const uint16_t kHtml[2] = {'&' * 256 + 'L', '&' * 256 + 'N'};

/* ---- ColumnContext :2000-2155: the last four lines, and cell positions of wiki tables / page headers ---- */
typedef struct { uint32_t linepos; uint8_t fc; uint8_t bytes[2048]; int size; } Line;
typedef struct { uint32_t cxt[32]; int size; } CellRow;
typedef struct {
  Line col[4]; CellRow cell[4];
  int rows, cellCount, cells, abovecellpos, abovecellpos1, NL, isTemp, limit;
  uint8_t nlChar;
} Columns;
static int col_len(const Columns* c, int i, int l) { return imin(l ? l : c->limit, c->col[(c->rows - i) & 3].size + 1); }
static int col_lastfc(const Columns* c, int i) { return c->col[(c->rows - i) & 3].fc; }
static int col_b(const Columns* c, int i, int j) {  /* colb: index -1 (first byte of a line, j = 1) lands on a padding byte of the reference's struct: 0 */
  const int k = col_len(c, 0, 0) - (1 + j);
  return col_len(c, 0, 0) < col_len(c, i, 0) && k >= 0 ? c->col[(c->rows - i) & 3].bytes[k] : 0;
}
static int cells_count(const Columns* c) { return c->cell[(c->cells - 1) & 3].size; }
static int cell_pos(const Columns* c, int id) { return (int)c->cell[(c->cells - 1) & 3].cxt[imin(cells_count(c) - 1, id)]; }
static void cell_new_row(Columns* c, int blpos) {
  c->cells = (c->cells + 1) & 3;
  CellRow* r = &c->cell[c->cells];
  r->cxt[0] = 0; r->size = 0;
  r->cxt[r->size++] = (uint32_t)blpos; r->size &= 31;
  c->cellCount = c->abovecellpos = c->abovecellpos1 = 0;
}
static void cell_step(Columns* c, int newcell) {
  if (c->abovecellpos) { c->abovecellpos++; if (c->abovecellpos > c->abovecellpos1) c->abovecellpos = c->abovecellpos1 = 0; }
  if (newcell && cells_count(c) > 0) { c->abovecellpos = cell_pos(c, c->cellCount - 1); c->abovecellpos1 = cell_pos(c, c->cellCount); }
}
static void col_update(Columns* c, int byte, uint32_t b2, int blpos, int isPre) {
  if (b2 == ((CURLYOPENING << 16) + (CURLYOPENING << 8) + VERTICALBAR)) c->nlChar = WIKITABLE;
  else if (b2 == ((VERTICALBAR << 16) + (CURLYCLOSE << 8) + CURLYCLOSE)) { c->nlChar = LF; for (int i = 0; i < 4; i++) { c->cell[i].cxt[0] = 0; c->cell[i].size = 0; } }
  if (byte != CURLYOPENING && (b2 & 0xff00) == (CURLYOPENING << 8) && (b2 & 0xff0000) != (CURLYOPENING << 16)) c->isTemp = 1;
  else if (c->isTemp && byte == CURLYCLOSE) c->isTemp = 0;
  c->NL = 0;
  Line* ln = &c->col[c->rows];
  ln->bytes[ln->size++] = (uint8_t)byte; ln->size &= 2047;
  if (byte == LF) {
    c->rows = (c->rows + 1) & 3;
    ln = &c->col[c->rows];
    ln->bytes[0] = 0; ln->size = 0;
    ln->fc = 0;
    ln->linepos = (uint32_t)(blpos - 1);
  } else if (col_len(c, 0, 0) == 2) {
    ln->fc = (uint8_t)imin(byte, TEXTDATA);
    c->NL = 1;
    if (ln->fc == GREATERTHAN && !isPre) c->nlChar = WIKIHEADER;
    if (ln->fc == SQUAREOPEN && c->nlChar == WIKIHEADER) c->nlChar = LF;
  }
  if (c->nlChar == WIKITABLE) {  /* {| |- | || |} */
    if ((b2 & 0xffff) == (WIKITABLE + VERTICALBAR * 256)) cell_new_row(c, blpos);
    int newcell = 0;
    if ((b2 & 0xffff) == (VERTICALBAR + VERTICALBAR * 256) || (b2 & 0xffff00) == ((VERTICALBAR + LF * 256) * 256)) {
      CellRow* r = &c->cell[c->cells];
      r->cxt[r->size++] = (uint32_t)blpos; r->size &= 31;
      c->cellCount++; newcell = 1;
    }
    cell_step(c, newcell);
  }
  if (c->nlChar == WIKIHEADER) {  /* the header block of a filtered wiki page: one '>' per field */
    if ((b2 & 0xffff) == (WIKIHEADER + LF * 256)) cell_new_row(c, blpos);
    else {
      int newcell = 0;
      if ((b2 & 0xff) == WIKIHEADER) { CellRow* r = &c->cell[c->cells]; r->cxt[r->size++] = (uint32_t)blpos; r->size &= 31; c->cellCount++; newcell = 1; }
      cell_step(c, newcell);
    }
  }
}

/* ---- WordsContext :2157-2274: the words of the current sentence / paragraph / stream with their surroundings ---- */
typedef struct {
  uint16_t sbytes[256]; uint32_t type[256], stem[256]; uint8_t capital[256];
  int n_sbytes, n_type, n_stem, n_capital;
  uint32_t fword, ftype; uint8_t pbyte; int wordcount, upper, ref;
} Words;
static void wc_reset(Words* w) {
  w->sbytes[0] = 0; w->n_sbytes = 0; w->type[0] = 0; w->n_type = 0; w->stem[0] = 0; w->n_stem = 0; w->capital[0] = 0; w->n_capital = 0;
  w->fword = w->ftype = 0; w->pbyte = 0; w->wordcount = w->upper = w->ref = 0;
}
static void wc_set(Words* w, int b, int a) { w->pbyte = (uint8_t)b; w->upper = a; }
static void wc_update(Words* w, uint32_t word, int b, uint32_t t, uint32_t s) {
  if (w->fword == 0) w->fword = word;
  w->sbytes[w->n_sbytes++] = (uint16_t)(w->pbyte * 256 + b); w->n_sbytes &= 255;
  w->type[w->n_type++] = t; w->n_type &= 255;
  w->stem[w->n_stem++] = s; w->n_stem &= 255;
  w->capital[w->n_capital++] = (uint8_t)w->upper; w->n_capital &= 255;
  w->pbyte = 0; w->wordcount++;
  if (w->ftype == 0 && t) w->ftype = t;
}
static void wc_remove(Words* w) {
  if (w->n_stem) {
    if (w->n_sbytes > 0) { w->sbytes[w->n_sbytes] = 0; w->n_sbytes--; }
    if (w->n_type > 0) { w->type[w->n_type] = 0; w->n_type--; }
    if (w->n_stem > 0) { w->stem[w->n_stem] = 0; w->n_stem--; }
    if (w->n_capital > 0) { w->capital[w->n_capital] = 0; w->n_capital--; }
    w->wordcount--;
  }
}
static uint32_t wc_word(const Words* w, int i) { return w->n_stem >= i ? w->stem[w->n_stem - i] : 0; }
static uint32_t wc_sbytes(const Words* w, int i) { return w->n_sbytes >= i ? w->sbytes[(w->n_sbytes - i) & 255] : 0; }
static uint32_t wc_type(const Words* w, int i) { return w->n_type >= i ? w->type[w->n_type - i] : 0; }
static uint32_t wc_capital(const Words* w, int i) { return w->n_capital >= i ? w->capital[w->n_capital - i] : 0; }
static uint32_t wc_last(const Words* w, int j, uint32_t t, int or_zero) {  /* Last / LastIf */
  if (t == 0) return wc_word(w, j);
  if (w->n_type >= j)
    for (int i = j; i < w->n_type; i++) if (wc_type(w, i) & t) return wc_word(w, i);
  return or_zero ? 0 : wc_word(w, j);
}
static void wc_remove_words(Words* w, int len, int c, int d, int left) {  /* removeWordsL / removeWordsR */
#define SIDE(i) (left ? (wc_sbytes(w, i) >> 8) : (wc_sbytes(w, i) & 0xff))
  if ((wc_sbytes(w, 1) & 0xff) == (uint32_t)d)
    for (int i = 1; i < len; i++)
      if (SIDE(i) == (uint32_t)c) {
        while (SIDE(1) != (uint32_t)c) wc_remove(w);
        wc_remove(w);
        break;
      }
#undef SIDE
}


struct FxParserState {
  int blpos; uint32_t c4;
  uint32_t t[14];
  int c1, c2, c3;
  uint8_t words, spaces, numbers;
  uint32_t word0, word00, word1, word2, word3, wshift, x4, x5, firstWord, linkword, senword;
  uint32_t number0, number1, numlen0, numlen1, mybenum;
  uint32_t FcIdx, BrFcIdx, AH1, AH2;
  int nl, nl1, col, fc;
  uint32_t t1[0x100], t2[0x10000];
  int wp[0x10000];
  uint16_t* ind3;
  uint32_t indirectBrByte, indirectByte, indirectWord0Pos, indirectWord, u8w, context1_ind3, cxtind3, lastWT;
  uint32_t o3bState, n3bState, stream3bR, stream3b, stream3bMask, stream3bMask1, stream3bRMask1, stream3bRMask2;
  uint32_t o2bState, n2bState, stream2bR, stream2b, stream2bMask, o4bState, n4bState, stream4bR, stream4b;
  uint8_t* buffer; int pos;
  uint8_t cwbuf[0x1000]; int cwpos;
  FxWord StemWords[4]; int cWord, pWord, StemIndex;
  int dcw, dcwl;
  uint32_t sVerb;
  int lastArt, isNowiki, isText, isMath, isPre, isParagraph, utf8left, deccode;
  /* WRT dictionary (:352-437): the decoded word of the last codeword and the one before the last ':' */
  char** dictW; int sizeDict, lastCW; const char *so, *colonstr;
  Bracket brcxt, qocxt, fccxt, htcxt;
  Columns colcxt;
  Words worcxt, worcxt1, worcxt2;
  /* emission: the record being filled, calls received per map, the reference's three map arrays as indices into FX_MAPS */
  FxByteRec* rec; int cn[FX_NMAPS], base[FX_NMAPS];
  int cmC[6], cmC1[8], cmC2[18];
  uint32_t* scmA; uint32_t rcmA;
};
typedef FxParserState FxModel;
// ContextMap::set (:1057-1065) / sets (:1066-1070): the stored context depends on the call index within the map
static void fx_cm_set(FxModel* m, int map, uint32_t cx) {
  const uint32_t i = (uint32_t)m->cn[map]++;
  if (i >= FX_MAPS[map].C) return;
  cx = cx * 987654323u + i;
  cx = cx << 16 | cx >> 16;
  m->rec->cx[m->base[map] + i] = cx * 123456791u + i;
}
static void fx_cm_skip(FxModel* m, int map) {
  const uint32_t i = (uint32_t)m->cn[map]++;
  if (i >= FX_MAPS[map].C) return;
  const uint32_t s = (uint32_t)m->base[map] + i;
  m->rec->cx[s] = 0;
  m->rec->skip[s >> 5] |= 1u << (s & 31);
}
static void fx_sscm_set(uint32_t* slot, uint32_t ctx) { *slot = ctx; }
static void fx_rcm_set(uint32_t* slot, uint32_t cx, int) { *slot = cx; }
static uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {  /* hash :2276-2279 (c defaults to 0xffffffff) */
  const uint32_t h = a * 110002499u + b * 30005491u + c * 50004239u;
  return h ^ h >> 9 ^ a >> 3 ^ b >> 3 ^ c >> 4;
}
static int char_swap(int c) {  /* charSwap :2281-2287: undo cmix's WRT character swap */
  if (c >= '{' && c < 127) c += 'P' - '{';
  else if (c >= 'P' && c < 'T') c -= 'P' - '{';
  else if ((c >= ':' && c <= '?') || (c >= 'J' && c <= 'O')) c ^= 0x70;
  if (c == 'X' || c == '`') c ^= 'X' ^ '`';
  return c;
}
static uint8_t kFcy[128], kFcq[128];   /* :3680-3700, filled by init_class_tables */
static void init_class_tables() {
  kFcy['"'] = 5; kFcy['\''] = 6; kFcy['('] = 1; kFcy['L'] = 4; kFcy['P'] = 2; kFcy['['] = 3;
  kFcq['*'] = 6; kFcq['@'] = 1; kFcq['J'] = 3; kFcq['L'] = 4; kFcq['M'] = 5; kFcq['P'] = 2; kFcq['Q'] = 7; kFcq['['] = 2; kFcq['`'] = 2;
}
static const uint32_t kPrimes[14] = {0, 257, 251, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193, 191};
#define BUF(i) ((int)m->buffer[((uint32_t)m->pos - (uint32_t)(i)) & BMASK])
#define BUFR(i) ((int)m->buffer[(uint32_t)(i) & BMASK])
#define BUFFER1(i) ((int)m->cwbuf[((uint32_t)m->cwpos - (uint32_t)(i)) & CBMASK])

static int get_wt(uint32_t t) {  /* getWT :3706-3722 */
  if (t & T_Verb) return 1;
  if (t & T_Noun) return 2;
  if (t & T_Adjective) return 3;
  if (t & T_Male) return 4;
  if (t & T_Female) return 5;
  if (t & T_Article) return 6;
  if (t & T_Conjunction) return 7;
  if (t & T_Adposition) return 8;
  if (t & T_ConjunctiveAdverb) return 9;
  if (t & T_AdverbOfManner) return 11;
  if (t & T_Suffix) return 12;
  if (t & T_Prefix) return 13;
  if (t & T_Plural) return 10;
  return t ? 14 : 15;
}
/* setbuf + setbufstem :3724-3772: the un-swapped text goes to a 4 KB buffer; letters build the current word, any other
 * character closes it: stem, classify, file it in the sentence / paragraph / stream word lists */
static void set_buf(FxModel* m, int ch) {
  const char c = (char)ch;
  m->cwbuf[m->cwpos & CBMASK] = (uint8_t)c;
  m->cwpos++;
  FxWord* cw = &m->StemWords[m->cWord];
  if ((c >= 'a' && c <= 'z') || (c == APOSTROPHE && m->c2 != APOSTROPHE) || (c == '-' && p8w_len(&cw->w) > 0)) { fxw_add(cw, c); return; }
  if (p8w_len(&cw->w) > 0 && c == SQUARECLOSE && m->fccxt.cxt != HTLINK && m->isParagraph) return;  /* [dog]s stays one word */
  if (p8w_len(&cw->w) == 0) return;
  fx_stem(cw, m->blpos);
  m->StemIndex = (m->StemIndex + 1) & 3;
  m->pWord = m->cWord;
  m->cWord = m->StemIndex;
  memset(&m->StemWords[m->cWord], 0, sizeof(FxWord));
  FxWord* pw = &m->StemWords[m->pWord];
  if (pw->Type & T_Verb) m->sVerb = pw->Hash;
  if (m->lastArt) pw->Type |= T_Noun;
  m->lastArt = (pw->Type == T_Article && BUFFER1(5) == SPACE && BUFFER1(4) == 't' && BUFFER1(3) == 'h' && BUFFER1(2) == 'e');
  uint32_t whash = m->isMath ? m->word0 : pw->Hash;
  m->lastWT = m->lastWT * 16 + (uint32_t)get_wt(pw->Type);
  if (pw->Type == T_Number && wc_type(&m->worcxt, 1) == T_Number) {  /* multi-word numbers become one entry */
    const uint32_t sb = wc_sbytes(&m->worcxt, 1);
    whash = whash + wc_word(&m->worcxt, 1);
    wc_remove(&m->worcxt);
    wc_set(&m->worcxt, (int)(sb >> 8), 0);
  }
  wc_update(&m->worcxt, m->word0, m->c1, pw->Type, whash);
  if ((pw->Type & (T_Conjunction + T_Article + T_Male + T_Female + T_Number + T_ConjunctiveAdverb)) == 0 && m->brcxt.cxt != LESSTHAN)
    wc_update(&m->worcxt1, m->word0, m->c1, pw->Type, whash);
  if ((pw->Type & (T_Conjunction + T_Article + T_Male + T_Female + T_Adposition + T_Number + T_AdverbOfManner + T_ConjunctiveAdverb)) == 0 &&
      m->brcxt.cxt != LESSTHAN && pw->Type)
    wc_update(&m->worcxt2, m->word0, m->c1, pw->Type, whash);
}

static const char kEmpty[1] = {0};
static int decode_codeword(int cw) {  /* decodeCodeWord :389-411: 1 to 3 codeword bytes (128..255) -> dictionary index */
  enum { d1 = 80, d2 = 32 };
#define SYM(c) ((c) >= 128 ? (c) - 128 : 0)   /* codeword2sym after dosym() :424-431 */
  int c = cw & 255, i;
  if (SYM(c) < d1) return SYM(c);
  i = d1 * (SYM(c) - d1);
  c = (cw >> 8) & 255;
  if (SYM(c) < d1) return i + SYM(c) + d1;
  i = (i - d1 * d2) * d2;
  i += d1 * (SYM(c) - d1);
  c = (cw >> 16) & 255;
  return i + SYM(c) + 80 * 49;
#undef SYM
}
static void set_buf(FxModel* m, int ch);
static void proc_word(FxModel* m) {  /* procWord :3782-3795: a finished codeword is decoded and its letters go through the word parser */
  if (m->dcwl > 0) {
    if (m->dcwl == 2) m->dcw = (m->dcw / 256) + (m->dcw & 255) * 256;
    if (m->dcwl == 3) m->dcw = ((m->dcw / 256) / 256) + (m->dcw & 0xff00) + (m->dcw & 255) * 256 * 256;
    if (m->dcwl > 3) return;
    if (m->dictW) {
      const int j = decode_codeword(m->dcw);
      if (j > 0 && j < m->sizeDict) { m->lastCW = j; m->so = m->dictW[j]; }
    }
    m->dcw = m->dcwl = 0;
    for (const char* p = m->so; *p; ++p) set_buf(m, *p);   /* an undecodable codeword replays the previous word (:3790-3794) */
  }
}
#define CM(k) fx_cm_set(m, m->cmC[k],
#define CM1(k) fx_cm_set(m, m->cmC1[k],
#define CM2(k) fx_cm_set(m, m->cmC2[k],
#define SKIP(map) fx_cm_skip(m, map)

/* the byte-boundary half of modelPrediction :3802-4600 */
static void byte_update(FxModel* m) {
  int i;
  uint32_t h = 0, j;
  uint32_t c4 = m->c4;
  m->c3 = m->c2; m->c2 = m->c1; m->c1 = (int)(c4 & 0xff);
  int c1 = m->c1, c2 = m->c2; const int c3 = m->c3;
  m->n2bState = FX_WRT_2B[c1]; m->n3bState = FX_WRT_3B[c1]; m->n4bState = FX_WRT_4B[c1];
  m->stream2b = m->stream2b * 4 + m->n2bState;
  m->stream4b = m->stream4b * 16 + m->n4bState;
  m->buffer[m->pos & BMASK] = (uint8_t)c1;
  m->pos++;
  if (c2 == GREATERTHAN && m->isText) {  /* the line after <text ...> starts a paragraph */
    m->isText = 0;
    if (c1 == APOSTROPHE || c1 == FIRSTUPPER) {
      col_update(&m->colcxt, LF, 0, m->blpos, m->isPre);
      wc_reset(&m->worcxt); wc_reset(&m->worcxt1);
      m->fc = m->isParagraph = 0; m->firstWord = 0;
      m->nl1 = m->nl; m->nl = m->pos - 2;
    }
  }
  col_update(&m->colcxt, c1, c4 & 0xffffff, m->blpos, m->isPre);
  if (c1 < 'a') br_update(&m->brcxt, c1);
  if (c1 == SPACE && c2 == LESSTHAN) br_update(&m->brcxt, GREATERTHAN);
  CM(4) (m->brcxt.context << 8) + (uint32_t)c1);
  br_update(&m->qocxt, c1);
  if (m->htcxt.cxt && c2 == 'L' && (c1 == SPACE || c1 == '!' || c1 < 128)) br_update(&m->htcxt, '&' * 256 + 'N');
  br_update(&m->htcxt, (int)(c4 & 0xffff));

  if (c1 == '$' || c1 == SQUARECLOSE || c1 == VERTICALBAR || c1 == ')' || c1 == SQUAREOPEN) {  /* these end an order-X context */
    if (c1 != c2) for (i = 13; i > 0; --i) m->t[i] = m->t[i - 1] * kPrimes[i];
    m->x4 = (m->x4 << 8) + (uint32_t)c2;
    m->stream2b = m->stream2b * 4 + m->n2bState;
    m->stream2bR = (m->stream2bR << 2) + m->n2bState;
    m->stream3bR = (m->stream3bR << 3) + m->n3bState;
  }
  m->x4 = (m->x4 << 8) + (uint32_t)c1;
  for (i = 13; i > 0; --i) m->t[i] = m->t[i - 1] * kPrimes[i] + (uint32_t)c1 + (uint32_t)i * 256;
  if (m->fc == SPACE && c1 == SPACE) { SKIP(m->cmC2[0]); SKIP(m->cmC2[0]); SKIP(m->cmC2[0]); }
  else for (i = 3; i < 6; ++i) CM2(0) m->t[i]);
  CM2(1) m->t[6]);
  CM2(2) m->t[8]);
  CM2(3) m->t[13]);

  m->words = (uint8_t)(m->words << 1); m->spaces = (uint8_t)(m->spaces << 1); m->numbers = (uint8_t)(m->numbers << 1);
  j = (uint32_t)c1;
  if ((j - 'a') <= ('z' - 'a') || (c1 > 127 && c2 != ESCAPE)) {   /* a letter (or a WRT codeword byte) */
    if (m->word0 == 0) {
      if (m->isMath && c2 == '/' && c3 == LESSTHAN) m->isMath = 0;
      int reChar = c2;
      if (c2 == FIRSTUPPER || c2 == UPPER) {
        if (c3 != APOSTROPHE) reChar = c3;
        else if (BUF(4) != APOSTROPHE) reChar = BUF(4);
        else if (BUF(5) != APOSTROPHE) reChar = BUF(5);
        else if (BUF(6) != APOSTROPHE) reChar = BUF(6);
        else reChar = c3;
      } else if (c2 == '/' && c3 == LESSTHAN) reChar = c3;
      wc_set(&m->worcxt, reChar & 255, c2 == FIRSTUPPER ? 1 : 0);
      wc_set(&m->worcxt1, reChar & 255, 0);
    }
    m->words |= 1;
    m->word0 = m->word0 * 2104 + j;
    m->word00 = m->word0;
    h = m->word0 * 271; m->u8w = 0;
    if (m->brcxt.cxt == SQUAREOPEN && m->fccxt.cxt != HTLINK && m->fc != HTML) m->linkword = m->linkword * 2104 + j;
    if (m->isParagraph && m->fccxt.cxt != HTLINK && !m->colcxt.isTemp) m->senword = m->senword * 2104 + j;
    const int word3bit = m->words & 7;
    if ((word3bit == 5 && c2 == APOSTROPHE) || (word3bit == 1 && c3 == SQUARECLOSE && c2 == APOSTROPHE) || (word3bit == 1 && (m->numbers & 4) && c2 == APOSTROPHE))
      br_update(&m->qocxt, (int)m->qocxt.cxt);
    if (c1 > 127) {   /* a codeword byte: try to decode what there is of it */
      m->dcw = (int)((uint32_t)m->dcw * 256 + (uint32_t)c1); m->dcwl++;
      if (m->blpos > 6) {
        int dcw2 = 0;
        if (m->dcwl == 2) dcw2 = (m->dcw / 256) + (m->dcw & 255) * 256;
        else if (m->dcwl == 3) dcw2 = ((m->dcw / 256) / 256) + (m->dcw & 0xff00) + (m->dcw & 255) * 256 * 256;
        const int k = m->dictW ? decode_codeword(dcw2) : 0;
        if (k > 0 && k < m->sizeDict) m->deccode = k;
      }
    } else if (m->dcw) { proc_word(m); if (m->blpos < 448131719) m->deccode = m->lastCW; }
    if (c1 == 10 || c1 == 9 || (c1 > 31 && c1 < 128)) set_buf(m, char_swap(c1));
  } else {
    if (m->word0) { proc_word(m); if (m->blpos < 448131719) m->deccode = m->lastCW; }
    else m->deccode = (int)(0x10000 + (m->stream2b & 0xffff));
    if (c1 == 10 || c1 == 9 || (c1 > 31 && c1 < 128)) set_buf(m, char_swap(c1));
    if (c1 >= '0' && c1 <= '9') {   /* numbers: (number), (number.number), (number,number) */
      m->numbers = (uint8_t)(m->numbers + 1);
      if ((m->numbers & 4) && c2 == ',') { m->number0 = m->number1; m->number1 = 0; m->numlen0 = m->numlen1; m->numlen1 = 0; }
      if (m->mybenum && m->numlen1 <= 2) { m->number0 = m->number1; m->number1 = 0; m->numlen0 = m->numlen1; m->numlen1 = 0; }
      m->number0 = m->number0 * 10 + (uint32_t)(c1 & 0x0f);
      m->numlen0 = (uint32_t)imin(19, (int)(m->numlen0 + 1)); m->mybenum = 0;
    } else {
      if (m->numlen0 || (m->numbers & 0xf) == 0) { m->number1 = m->number0; m->numlen1 = m->numlen0; m->number0 = m->numlen0 = 0; }
      if (m->numlen1 <= 2 && m->numlen1 && (m->numbers & 5) == 5 && m->numlen0 == 0 && c2 == '.') m->mybenum = 2;
      else if (m->numlen1 <= 2 && m->numlen1 && (m->numbers & 2) && m->numlen0 == 0 && c1 == '.') m->mybenum = 1;
      else if (m->mybenum == 1 && c1 != '.') m->mybenum = 0;
    }
    const int word3bit = m->words & 7;
    if ((word3bit == 4 && c1 == SPACE && c2 == APOSTROPHE) || (c1 == FIRSTUPPER && (m->numbers & 4) && c2 == APOSTROPHE) ||
        (word3bit == 4 && c1 == FIRSTUPPER && c2 == APOSTROPHE) || (word3bit == 4 && (m->numbers & 1) && c2 == APOSTROPHE))
      br_update(&m->qocxt, (int)m->qocxt.cxt);
    if (m->word00 && m->fccxt.cxt != SQUAREOPEN) m->word00 = 0;
    Words* wc = &m->worcxt;
    if (m->word0) {   /* a word just ended */
      if (m->blpos > 463139793 || (m->StemWords[m->pWord].Type & (T_ConjunctiveAdverb + T_Conjunction)) == 0) {
        m->word3 = m->word2 * 47; m->word2 = m->word1 * 53; m->word1 = m->word0 * 83;
      }
      if (wc_type(wc, 1) == T_Number) { m->stream3bR = (m->stream3bR << 7) + 1; m->stream3b = (m->stream3b << 7) + 1; }
      if (m->firstWord == 0 && m->fccxt.cxt != SQUAREOPEN) m->firstWord = m->word0;
      if (wc_type(wc, 1) & T_Conjunction) { m->stream3bR <<= 7; m->stream3b <<= 7; if (m->isParagraph) m->senword = 0; }
      if (wc_type(wc, 1) & T_Article) { m->stream3bR = (m->stream3bR << 7) + 2; m->stream3b = (m->stream3b << 7) + 2; }
      if ((wc_type(wc, 1) & T_Adposition) || (m->isParagraph && (wc_type(wc, 1) & T_PresentParticiple))) {
        m->stream2bR = (m->stream2bR << 2) + (m->stream2bR & 3);
        m->stream2b = (m->stream2b << 2) + (m->stream2b & 3);
      }
      if ((wc_type(wc, 1) & T_AdverbOfManner) && m->isParagraph) wc_remove(wc);
      if ((wc_type(wc, 1) & T_Noun) && (wc_type(wc, 2) & T_Article)) {   /* article + noun become one entry */
        m->stream3bR = (m->stream3bR << 6) + 1; m->stream3b = (m->stream3b << 6) + 1;
        const uint32_t sb = wc_sbytes(wc, 1), w = wc_word(wc, 1), t = wc_type(wc, 1), ca = wc_capital(wc, 1);
        wc_remove(wc); wc_remove(wc);
        wc_set(wc, (int)(sb >> 8), (int)ca);
        wc_update(wc, w, c1, t, w);
      }
      m->stream3bRMask2 = m->stream3bRMask1;
      m->stream3bMask1 = m->stream3bMask;
      m->stream3bMask = m->stream2bMask = m->stream3bRMask1 = 0;
    } else if (c1 == VERTICALBAR && m->colcxt.isTemp) {
      const uint32_t sb = wc_sbytes(wc, 1), w = wc_word(wc, 1), t = wc_type(wc, 1), ca = wc_capital(wc, 1);
      wc_remove(wc);
      wc_set(wc, (int)(sb >> 8), (int)ca);
      wc_update(wc, w, c1, t, w);
    }
    /* <text>, <nowiki>, <math>, <pre>, </page> boundaries, recognised through the last decoded dictionary word (:4028-4045) */
#define SO_IS(str) (strcmp(m->so, str) == 0)
    const int lt = char_swap(LESSTHAN);
    if (BUFFER1(6) == lt && BUFFER1(5) == 't' && !m->isText && c1 == SPACE && SO_IS("text")) { m->isText = 1; m->so = kEmpty; }
    if (BUFFER1(8) == lt && !m->isNowiki && SO_IS("nowiki")) m->isNowiki = 1;
    else if (BUFFER1(9) == '/' && c1 == GREATERTHAN && m->isNowiki && SO_IS("nowiki")) { m->isNowiki = m->isPre = 0; m->so = kEmpty; }
    if (m->isMath && ((c1 == SPACE && col_lastfc(&m->colcxt, 0) != COLON) || c1 == ',') && c2 == GREATERTHAN && SO_IS("math")) { m->isMath = 0; m->so = kEmpty; }
    if (m->isMath && c1 == '/' && c2 == LESSTHAN && c3 == GREATERTHAN && BUFFER1(4) == 'h') { m->isMath = 0; m->so = kEmpty; }
    if (!m->isNowiki && BUFFER1(6) == lt && BUFFER1(5) == 'm' && !m->isMath && c1 != '.' && BUFFER1(7) != '&' && BUFFER1(8) != '&' && SO_IS("math")) m->isMath = 1;
    else if (BUFFER1(6) == '/' && (c1 == GREATERTHAN || c1 == '&') && m->isMath && SO_IS("math")) { m->isMath = 0; m->so = kEmpty; }
    if (BUFFER1(5) == lt && c1 == GREATERTHAN && BUFFER1(4) == 'p' && !m->isPre && SO_IS("pre")) { m->isPre = 1; m->so = kEmpty; }
    else if (BUFFER1(5) == '/' && c1 == GREATERTHAN && BUFFER1(4) == 'p' && SO_IS("pre")) { m->isPre = 0; m->so = kEmpty; }
    if (BUFFER1(6) == '/' && c1 == GREATERTHAN && BUFFER1(5) == 'p' && SO_IS("page")) m->isPre = m->isMath = m->isNowiki = 0;
#undef SO_IS

    m->wp[m->word0 & 0xffff] = m->pos;
    m->word0 = h = 0;
    if (m->linkword && c1 == COLON) m->linkword = 0;
    if (c1 == '-' && c2 == SPACE) { wc_reset(&m->worcxt1); m->sVerb = 0; }
    if (c1 == SPACE) m->spaces++;
    else if (c1 == LF) {
      m->fc = m->isParagraph = 0; m->firstWord = 0; m->lastWT = 0;
      m->nl1 = m->nl; m->nl = m->pos - 1;
      m->stream3bR <<= 7;
      m->stream2b |= 0x3fc;
      m->words = 0xfc;
      wc_reset(&m->worcxt); wc_reset(&m->worcxt1);
      m->stream2bR <<= 2;
      m->stream4b |= 0xfff0;
      if (c2 == LF) m->isNowiki = 0;
    } else if (c1 == '.' || c1 == ')' || c1 == QUESTION) {
      m->lastWT *= 16;
      m->stream3bR <<= 7; m->stream3b <<= 7;
      m->words |= 0xfe;
      m->x5 = (m->x5 << 8) + (c4 & 0xff);
      m->stream2b |= 204;
      m->stream4b = ((m->stream4b & 0xffff0) << 8) + (m->stream4b & 0xf);
      m->stream2bR &= 0xffffffc0;
      if (c1 == '.') {
        m->wshift = 1;
        if (!(m->fccxt.cxt == SQUAREOPEN || m->fccxt.cxt == '(' || m->colcxt.nlChar == WIKITABLE || col_lastfc(&m->colcxt, 0) == '*')) wc_reset(&m->worcxt);
        m->senword = 0;
      }
      if (c1 == ')') m->senword = 0;
    } else if (c1 == ',') { m->words |= 0xfc; m->senword = 0; }
    else if (c1 == '(') m->senword = 0;
    else if (c1 == SEMICOLON) wc_reset(&m->worcxt);
    else if (c1 == COLON) {
      m->stream3b = (m->stream3b & 0xfffffff8) + 4;
      m->stream2b |= 12;
      m->x5 = (m->x5 << 8) + (c4 & 0xff);
      m->senword = 0;
    } else if (c1 == CURLYCLOSE || c1 == CURLYOPENING) {
      m->words |= 0xfc;
      m->stream3bR &= 0xffffffc0;
      m->x5 = (m->x5 << 8) + (c4 & 0xff);
      m->stream3b = (m->stream3b & 0xfffffff8) + 3;
    } else if (c1 == SQUARECLOSE) { m->stream3b = (m->stream3b & 0xfffffff8) + 3; m->linkword = 0; }
    else if (c1 == LESSTHAN || c2 == '&') m->words |= 0xfc;
    else if (c1 == '-' && col_lastfc(&m->colcxt, 0) == '*' && m->brcxt.cxt != SQUAREOPEN && m->isParagraph == 0) { m->isParagraph = 1; m->fc = FIRSTUPPER; }
    else if (c1 == EQUALS) {
      m->stream3b = (m->stream3b & 0xfffffff8) + 4;
      m->c2 = c2 = '.';
      m->words = (uint8_t)(m->words * 2);
    }
    if (c1 == '!' && c2 == '&') {   /* "&nbsp;" arrives as "&!" and counts as a space */
      m->c1 = c1 = SPACE;
      c4 = (c4 & 0xffffff00) + SPACE;
      m->stream2b = (m->stream2b & 0xfffffffc) + FX_WRT_2B[SPACE];
      m->stream3b = (m->stream3b & 0xfffffff8) + FX_WRT_3B[SPACE];
    } else if (col_lastfc(&m->colcxt, 0) == '*' && (c1 == ',' || c1 == SPACE) && c2 == SQUARECLOSE && m->isParagraph == 0) { m->isParagraph = 1; m->fc = FIRSTUPPER; }
  }

  m->x5 = (m->x5 << 8) + (c4 & 0xff);
  if (m->o2bState != m->n2bState) { m->stream2bR = (m->stream2bR << 2) + m->n2bState; m->o2bState = m->n2bState; }   /* non-repeating streams */
  m->stream2bMask = (m->stream2bMask << 2) + 3;
  if (m->o3bState != m->n3bState) {
    m->stream3bR = (m->stream3bR << 3) + m->n3bState;
    m->stream3bRMask1 = (m->stream3bRMask1 << 3) + 7;
    m->stream3bRMask2 = (m->stream3bRMask2 << 3) + 7;
    m->o3bState = m->n3bState;
  }
  m->stream3b = (m->stream3b << 3) + m->n3bState;
  m->stream3bMask = (m->stream3bMask << 3) + 7;
  m->stream3bMask1 = (m->stream3bMask1 << 3) + 7;
  const uint32_t brcontext = m->brcxt.cxt & 255;
  m->BrFcIdx = 0;
  if (m->brcxt.context) m->BrFcIdx = kFcy[brcontext & 127];
  if (m->brcxt.context == 0 && m->qocxt.context) m->BrFcIdx = kFcy[(m->qocxt.context >> 8) & 127];

  Columns* cc = &m->colcxt;
  m->col = col_len(cc, 0, 0);
  int above = m->buffer[(uint32_t)(m->nl1 + m->col) & BMASK], above1 = m->buffer[(uint32_t)(m->nl1 + m->col - 1) & BMASK];
  if (cc->nlChar == WIKIHEADER) { above = col_b(cc, 1, 0); above1 = col_b(cc, 1, 1); }
  if (cc->NL) {
    if ((int)(cc->col[cc->rows & 3].linepos + 2 - cc->col[(cc->rows - 1) & 3].linepos) < 4) {   /* two empty lines reset the nesting contexts */
      br_reset(&m->fccxt); br_reset(&m->brcxt); br_reset(&m->qocxt); br_reset(&m->htcxt);
    }
    m->fc = col_lastfc(cc, 0);
    if (m->fc == WIKIHEADER) br_reset(&m->fccxt);
    m->isParagraph = (m->fc == FIRSTUPPER);
    br_update(&m->fccxt, m->fc);
  }
  if (m->col > 2 && c1 > FIRSTUPPER && !m->isMath) {
    if (m->fccxt.cxt == VERTICALBAR && (c1 == SQUARECLOSE || c1 == CURLYCLOSE)) while (m->fccxt.cxt == VERTICALBAR) br_update(&m->fccxt, LF);
    if ((m->fccxt.cxt == COLON || m->fccxt.cxt == HTLINK) && c1 == SQUARECLOSE) while (m->fccxt.cxt == COLON || m->fccxt.cxt == HTLINK) br_update(&m->fccxt, LF);
    if (c1 < 128) br_update(&m->fccxt, c1);
  }
  if (c1 == COLON && (m->words & 2) == 2) m->colonstr = m->so;   /* the decoded dictionary word before ':' */
  if (c1 == SPACE && m->fccxt.cxt == COLON && col_lastfc(cc, 0) != COLON && cc->nlChar != WIKITABLE && strcmp(m->colonstr, "image") != 0)
    while (m->fccxt.cxt == COLON) br_update(&m->fccxt, LF);
  if (c1 == COLON && (strcmp(m->colonstr, "category") == 0 || strcmp(m->colonstr, "wikipedia") == 0)) { br_update(&m->fccxt, LF); wc_remove(&m->worcxt); }
  if (c1 == SPACE && c2 == LESSTHAN) br_update(&m->fccxt, GREATERTHAN);
  if (m->fccxt.cxt == COLON && c2 == '/' && c1 == '/') { br_update(&m->fccxt, LF); br_update(&m->fccxt, HTLINK); }
  if (col_lastfc(cc, 0) == SQUAREOPEN && c1 == SPACE && m->isParagraph == 0 && (c2 == SQUARECLOSE || c3 == SQUARECLOSE)) {
    m->fc = FIRSTUPPER; m->isParagraph = 1;
    br_reset(&m->fccxt); br_update(&m->fccxt, m->fc);
  }
  if (m->fc == SPACE && c1 != SPACE) {
    m->fc = imin(c1, TEXTDATA);
    m->isParagraph = (m->fc == FIRSTUPPER);
    br_update(&m->fccxt, m->fc);
  }
  const uint32_t fccontext = m->fccxt.cxt & 255;
  if (m->BrFcIdx == 0 && m->fccxt.context) m->BrFcIdx = kFcy[fccontext & 127];
  m->FcIdx = kFcq[fccontext & 127];
  CM(5) (m->fccxt.context & 0xff00) + (uint32_t)c1 + (m->stream2b & 12) * 256 + ((brcontext + (uint32_t)br_last(&m->brcxt)) << 24));

  if (m->fc == '*' && c1 != SPACE) m->fc = imin(c1, TEXTDATA);
  if (m->fc == '&' && c1 == LESSTHAN) m->fc = HTML;
  if (c2 == GREATERTHAN && m->fc == LESSTHAN && c1 == APOSTROPHE) m->fc = APOSTROPHE;
  if ((col_lastfc(cc, 0) == APOSTROPHE || (m->fc == APOSTROPHE && col_lastfc(cc, 0) != '*')) && c1 == SPACE && (c2 == APOSTROPHE || c3 == APOSTROPHE)) {
    m->fc = FIRSTUPPER; m->isParagraph = 1;
    br_reset(&m->fccxt); br_update(&m->fccxt, m->fc);
  }
  if (m->fc != FIRSTUPPER && (c4 & 0xffffff) == 0x4a2f2f) m->fc = HTLINK;
  wc_remove_words(&m->worcxt, 8, '(', ')', 1); wc_remove_words(&m->worcxt1, 8, '(', ')', 1);
  wc_remove_words(&m->worcxt, 8, SQUAREOPEN, VERTICALBAR, 1); wc_remove_words(&m->worcxt1, 8, SQUAREOPEN, VERTICALBAR, 1);
  wc_remove_words(&m->worcxt, 8, LESSTHAN, COLON, 1);
  if (cc->isTemp) wc_remove_words(&m->worcxt, 10, EQUALS, VERTICALBAR, 0);
  wc_remove_words(&m->worcxt, 8, LESSTHAN, GREATERTHAN, 1); wc_remove_words(&m->worcxt1, 8, LESSTHAN, GREATERTHAN, 1);

  /* indirect histories */
  m->indirectWord = (c4 >> 8) & 0xffff;
  m->t2[m->indirectWord] = (m->t2[m->indirectWord] << 8) | (uint32_t)c1;
  m->indirectWord = c4 & 0xffff;
  m->indirectWord = m->indirectWord | (m->t2[m->indirectWord] << 16);
  m->indirectByte = (c4 >> 8) & 0xff;
  m->t1[m->indirectByte] = (m->t1[m->indirectByte] << 8) | (uint32_t)c1;
  m->indirectByte = (uint32_t)c1 | (m->t1[c1] << 8);
  m->t1[brcontext] = (m->t1[brcontext] << 2) | (m->stream2b & 3);
  m->indirectBrByte = (m->stream3b & 7) | (m->t1[brcontext] << 3);
  m->indirectWord0Pos = (uint32_t)(m->pos - m->wp[m->word0 & 0xffff]);
  if (m->indirectWord0Pos > 255) m->indirectWord0Pos = 256 + ((uint32_t)c1 << 16);
  else m->indirectWord0Pos = m->indirectWord0Pos + ((uint32_t)BUF(m->indirectWord0Pos) << 8) + ((uint32_t)c1 << 16);
  m->ind3[m->context1_ind3] = (uint16_t)((m->cxtind3 * 32 + (uint32_t)c1) & (0x2000000 - 1));
  m->context1_ind3 = (m->context1_ind3 * 32 + (uint32_t)c1) & (0x2000000 - 1);
  m->cxtind3 = m->ind3[m->context1_ind3];
  if (c2 == 12) {   /* escaped UTF-8 */
    if (m->utf8left == 0) {
      if ((c1 >> 5) == 6) { m->utf8left = 1; m->u8w = m->u8w * 191 + (uint32_t)c1; }
      else if ((c1 >> 4) == 0xE) { m->utf8left = 2; m->u8w = m->u8w * 191 + (uint32_t)c1; }
      else if ((c1 >> 3) == 0x1E) { m->utf8left = 3; m->u8w = m->u8w * 191 + (uint32_t)c1; }
      else m->utf8left = 0;
    } else { m->utf8left--; if ((c1 >> 6) != 2) m->utf8left = 0; }
  }
  h = h + (uint32_t)c1;

  /* ---- the 81 context slots, in the order the maps receive them (:4323-4596) ---- */
  Words *wc = &m->worcxt, *wc1 = &m->worcxt1, *wc2 = &m->worcxt2;
  const uint32_t s2 = m->stream2b, s3 = m->stream3b, s3R = m->stream3bR, word0 = m->word0, word00 = m->word00, BrFc = m->BrFcIdx, uc1 = (uint32_t)c1;
  const uint32_t fc = (uint32_t)m->fc, col = (uint32_t)m->col;
  const int lastfc = col_lastfc(cc, 0), utf8left = m->utf8left, isPar = m->isParagraph;
  fx_rcm_set(&m->rcmA, m->word3 * 53 + uc1 + 193 * (s3 & 0x7fff), c1);
  if (m->col < 2 || m->fc == SPACE) { SKIP(m->cmC2[4]); SKIP(m->cmC2[4]); SKIP(m->cmC2[17]); }
  else {
    CM2(4) word00 + (m->number0 * 191 + m->numlen0) + m->u8w);
    if (lastfc == '&' || utf8left) SKIP(m->cmC2[4]); else CM2(4) h + m->word1);
    if (m->brcxt.cxt == LESSTHAN) SKIP(m->cmC2[17]); else CM2(17) wc_word(wc1, 1) * 53 + wc_word(wc1, 2) * 11 + h + (m->lastWT & 0xf));
  }
  if (c1 == ESCAPE || m->col < 2 || utf8left || m->fc == SPACE) SKIP(m->cmC2[5]); else CM2(5) h + m->word2 * 71);
  if (m->fc == SPACE || m->brcxt.cxt == LESSTHAN) { for (i = 0; i < 5; i++) SKIP(m->cmC2[5]); }
  else {
    CM2(5) wc_word(wc, 4) * 53 + wc_word(wc1, 1) + h + (s3 & 511));
    CM2(5) wc_last(wc, 4, wc_type(wc, 4) ^ T_Verb, 0) * 53 + m->sVerb + h + (s3R & 63));
    CM2(5) wc->fword * 53 + wc_word(wc1, 1) + h + (s3 & 63));
    CM2(5) wc_word(wc2, 1) + wc_word(wc2, 2) * 11 + word00 + uc1);
    const uint32_t lastParVerb = wc_last(wc2, 1, wc_type(wc, 1) & T_Verb, 1);
    if (lastParVerb) CM2(5) lastParVerb * 11 + word00 + uc1); else SKIP(m->cmC2[5]);
  }
  CM1(6) h + (wc_type(wc, 1) & 0x1FF) + wc_word(wc1, 1));
  CM2(6) ((s2 & 15) << 16) + (m->t[2] & 0xffff));
  if (c1 == ESCAPE || utf8left || fccontext == CURLYOPENING) CM2(7) 0); else CM2(7) m->indirectBrByte);
  CM2(8) (m->indirectBrByte & 0x7ff) * 32 + ((m->stream4b & 0xfff0) << 16) + BrFc);
  CM2(8) (s3R & 0x3fffffff) * 4 + (s2 & 3));
  CM2(8) fccontext * 4 + ((s3R & 0x3ffff) << 9) + BrFc);
  if (fccontext == HTLINK) SKIP(m->cmC2[8]); else CM2(8) (c4 & 0xffffff) + ((s2 << 18) & 0xff000000));
  CM1(0) (uint32_t)lastfc | (fccontext << 15) | ((s3 & 63) << 7) | (brcontext << 24));
  CM1(0) (uint32_t)lastfc | ((c4 & 0xffffff) << 8));
  CM1(1) (s2 & 3) + word00 * 11);
  CM1(1) c4 & 0xffff);
  CM1(1) ((fc << 11) | uc1) + ((s2 & 3) << 18));
  CM1(2) (s2 & 15) + ((s3 & 7) << 6));
  CM1(2) uc1 | ((col * (uint32_t)(c1 == SPACE)) << 8) | ((s2 & 15) << 16));
  CM1(2) isPar ? m->firstWord : (fc << 11));
  if (c1 == ESCAPE || m->fc == SPACE || utf8left) SKIP(m->cmC1[2]); else CM1(2) 91u * 83u * wc_word(wc, 1) + 89u * word0);
  if (m->fc == SPACE) SKIP(m->cmC1[4]); else CM1(4) uc1 + ((s3 & 0xe38) << 6));
  CM1(4) wc->fword * 11 + BrFc);
  CM1(4) uc1 + word0 + m->number0 * 191);
  CM1(4) ((c4 & 0xffff) << 16) | (fccontext << 8) | fc);
  CM1(4) ((s3R & 0xfff) << 8) + (s2 & 0xfc));
  if (c1 == ESCAPE) { for (i = 0; i < 6; i++) SKIP(m->cmC[0]); }
  else {
    if (isPar == 1) {
      CM(0) wc->fword * 3191 + (s2 & 3));
      CM(0) h + m->firstWord * 89);
      CM(0) word0 * 53 + uc1 + BrFc);
    } else {
      CM(0) (uint32_t)above | ((s3 & 0x3f) << 9) | ((uint32_t)col_len(cc, 0, 0) << 19) | ((s2 & 3) << 16));
      CM(0) h + m->firstWord * 89);
      CM(0) (uint32_t)above | (uc1 << 16) | ((col + m->numlen0 + BrFc) << 8) | ((uint32_t)above1 << 24));
    }
    const uint32_t cellb = (uint32_t)BUFR(cc->abovecellpos);
    if (lastfc == '*') {
      CM(0) (word0 + (fccontext << 8)) | (BrFc << 16));
      CM(0) uc1);
      CM(0) word0);
    } else {
      CM(0) FX_WRT_2B[cellb] | (fccontext << 8) | (BrFc << 16));
      CM(0) cellb | (uc1 << 8));
      CM(0) word0 + FX_WRT_2B[cellb]);
    }
  }
  CM(1) (s3 & 0x7fff) * word0 + BrFc);
  CM(1) (m->x4 & 0xff0000ff) | ((s3 & 0xe07) << 8));
  CM(1) (m->indirectBrByte & 0xffff) | ((s3 & 0x38) << 16));
  if (m->isMath) SKIP(m->cmC[0]); else CM(0) (m->indirectByte & 0xff00) + 257u * wc_word(wc, 1) * 53u + uc1);
  CM(2) (uc1 << 8) | (m->indirectByte >> 2) | (fc << 16));
  CM(2) (c4 & 0xffff) + (uint32_t)(c2 == c3 ? 1 : 0));
  CM1(3) ((s3 & m->stream3bMask) * 256) | (s2 & m->stream2bMask & 255));
  CM1(3) m->x4);
  CM2(9) 257u * m->StemWords[m->pWord].Hash + fccontext + 193u * (s3 & m->stream3bMask));
  CM2(9) fc | ((m->stream2bR & 0xfff) << 9) | (uc1 << 24));
  CM2(16) wc->fword * 83 + (s2 & 15) * 11 + brcontext);
  CM2(17) wc_last(wc, 1, T_Verb, 0) + wc_word(wc, 1) * 83 + h);
  CM2(9) (m->x4 & 0xffff00) + brcontext + (fccontext << 24));
  if (m->linkword) CM2(9) m->linkword);
  else if (m->isMath) SKIP(m->cmC2[9]);
  else if (m->senword) CM2(9) m->senword * 1471 + uc1);
  else if (m->fc == HTML || brcontext == LESSTHAN) SKIP(m->cmC2[9]);
  else CM2(9) 0);
  CM2(10) m->indirectByte);
  CM2(10) ((m->indirectByte & 0xffff00) >> 4) | (s2 & m->stream2bMask & 0xf) | ((s3 & 0xfff) << 20));
  CM2(10) (m->x4 >> 16) | ((s2 & 255) << 24));
  if (c1 > 127) CM2(10) ((((s2 & 12) * 256) + uc1) << 11) | ((m->indirectWord & 0xffffff) >> 16));
  else CM2(10) (uc1 << 11) | (BrFc << 8) | ((m->indirectWord & 0xffffff) >> 16));
  if (m->isMath) SKIP(m->cmC2[10]); else CM2(10) (fccontext * 4 + BrFc) | ((c4 & 0xffff) << 9) | ((s2 & 0xff) << 24));
  CM2(10) (m->indirectWord >> 16) | ((s2 & 0x3c) << 25) | ((s3 & 0x1ff) << 16));
  CM2(11) (uint32_t)m->words + ((uint32_t)m->spaces << 8) + ((s2 & 15) << 16) + (((s3R >> 3) & 511) << 21) + ((uint32_t)isPar << 30));
  CM2(11) uc1 + ((s3 << 5) & 0x1fffff00));
  CM2(11) m->stream2bR * 16 + BrFc);
  CM2(11) ((m->indirectByte & 0xffff) >> 8) + ((64 * m->stream2bR) & 0x3ffff00) + (brcontext << 25));
  if (fccontext == FIRSTUPPER && brcontext == SQUAREOPEN) SKIP(m->cmC2[11]); else CM2(11) m->indirectWord0Pos | ((m->indirectByte & 0xff00) << 16));
  CM2(12) (m->x4 & 0x80f00000) + ((m->x4 & 0x0000f0ff) << 12));
  if (isPar == 1) {
    if (c1 == ESCAPE || fccontext == HTLINK || fccontext == CURLYOPENING || m->isMath || m->isPre) SKIP(m->cmC2[12]);
    else CM2(12) h + wc_word(wc, 1) * 53u * 79u + wc_word(wc, 3) * 53u * 47u * 71u);
  } else {
    if (fccontext == HTLINK || brcontext == LESSTHAN || m->htcxt.cxt) SKIP(m->cmC2[12]);
    else if (m->col == 31) CM2(12) c4 << 16);
    else CM2(12) (uint32_t)above | ((c4 & 0xffff) << 16) | ((uint32_t)above1 << 8));
  }
  const int esc_word = (wc_sbytes(wc, 0) >> 8) == '\\';
  if (c1 == ESCAPE || utf8left || fccontext == CURLYOPENING || fccontext == HTLINK || m->fc == HTML || m->htcxt.cxt || m->fc == SPACE || m->isPre || c1 == '&' ||
      brcontext == LESSTHAN || m->isMath || m->col < 2 || esc_word) { SKIP(m->cmC2[13]); SKIP(m->cmC2[13]); }
  else {
    CM2(13) wc_word(wc, 1) * 83u * 1471u - word0 * 53 + wc_word(wc, 2));
    CM2(13) h + wc_word(wc, 2) * 53u * 79u + wc_word(wc, 3) * 53u * 47u * 71u);
  }
  CM(3) ((s3R & 7) << 10) + (s2 & 3) + fc * 4 + (BrFc << 24));
  CM(3) (m->linkword ? m->linkword : word0) * 3301 + m->number0 * 3191);
  if (c1 == ESCAPE || utf8left || fccontext == CURLYOPENING || fccontext == HTLINK || m->fc == SPACE || m->fc == HTML || brcontext == LESSTHAN || m->col < 2 ||
      m->isMath || esc_word) SKIP(m->cmC2[14]);
  else CM2(14) BrFc + wc_word(wc, 2) * (s3R & m->stream3bRMask2) + (wc_type(wc, 1) & 0x1ff));
  if (c1 == ESCAPE || utf8left || m->fc == SPACE) { for (i = 0; i < 4; i++) SKIP(m->cmC1[7]); }
  else {
    CM1(7) wc_word(wc1, 1) + word00);
    CM1(7) wc_word(wc, 2) + word0 * 191 + (s3R & 63));
    CM1(7) word0 * 191 + (s3R & 63));
    CM1(7) (m->indirectWord0Pos & 0xffff) * 191 + word0 + (s3R & 63));
  }
  fx_sscm_set(&m->scmA[0], uc1);
  fx_sscm_set(&m->scmA[1], (uint32_t)(c2 * isPar));
  fx_sscm_set(&m->scmA[2], (m->indirectWord & 0xffffff) >> 16);
  fx_sscm_set(&m->scmA[3], s3 & 0x1ff);
  fx_sscm_set(&m->scmA[4], s2 & 0xff);
  fx_sscm_set(&m->scmA[5], brcontext);
  fx_sscm_set(&m->scmA[6], (uint32_t)isPar + 2 * (s3R & 0x3f));
  if (m->wshift || c1 == LF) {
    m->word3 = m->word3 * 47; m->word2 = m->word2 * 53; m->word1 = m->word1 * 83;
    m->wshift = 0;
    if (c1 == LF) m->sVerb = 0;
  }
  CM2(15) (BrFc * 256) + fc + ((s3R & 0xFFF) << 16));
  m->AH1 = hash3(m->x5 & 255, (m->x5 >> 8) & 255, (m->x5 >> 16) & 0x80ff);
  m->AH2 = hash3(19, m->x5 & 0x80ffff, 0xffffffffu);
}

// dosym + loaddict + wfgets :352-381, :413-433: one word per line, index = line number
static void load_dictionary(FxModel* m, const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return;
  m->dictW = (char**)calloc(44516, sizeof(char*));
  char* line = (char*)malloc(8192 * 8);
  int n = 0, c;
  for (;;) {
    int i = 0;
    while (i < 8192 * 8 - 1 && (c = getc(f)) != EOF) { line[i++] = (char)c; if (c == '\n') { line[i - 1] = 0; break; } }
    line[i] = 0;
    if (i == 0 || n >= 44516) break;
    m->dictW[n] = (char*)calloc((size_t)i + 1, 1);
    memcpy(m->dictW[n], line, (size_t)i);
    n++;
  }
  free(line);
  fclose(f);
  m->sizeDict = n;
}
}  // namespace

struct FxParser { FxParserState s; };

extern "C" FxParser* fxp_create(const char* dictionary_path) {
  init_class_tables();
  FxParser* p = (FxParser*)calloc(1, sizeof(FxParser));
  FxModel* m = &p->s;
  m->AH2 = 0x765BA55C;                          // :3262
  m->n3bState = m->n2bState = 0xffffffff;
  m->buffer = (uint8_t*)calloc(BMASK + 1, 1);
  m->ind3 = (uint16_t*)calloc(0x2000000, 2);
  br_init(&m->brcxt, kBrackets, 8, 0, 8, 256);
  br_init(&m->qocxt, kQuotes, 4, 1, 8, 256);
  br_init(&m->fccxt, kFchar, 20, 0, 8, 256);
  br_init(&m->htcxt, kHtml, 2, 0, 16, 0xfff);
  m->colcxt.nlChar = LF; m->colcxt.limit = 31;
  m->cWord = 0; m->pWord = 3;
  m->so = m->colonstr = kEmpty;
  if (dictionary_path && dictionary_path[0]) load_dictionary(m, dictionary_path);
  int b = 0;
  for (int k = 0; k < FX_NMAPS; k++) {
    m->base[k] = b; b += FX_MAPS[k].C;
    const FxMapDef& d = FX_MAPS[k];
    (d.kind == 0 ? m->cmC : d.kind == 1 ? m->cmC1 : m->cmC2)[d.idx] = k;
  }
  return p;
}
extern "C" void fxp_destroy(FxParser* p) {
  if (!p) return;
  FxModel* m = &p->s;
  if (m->dictW) { for (int i = 0; i < m->sizeDict; i++) free(m->dictW[i]); free(m->dictW); }
  free(m->buffer); free(m->ind3); free(p);
}
extern "C" void fxp_set_blpos(FxParser* p, int blpos) { p->s.blpos = blpos; }
extern "C" int fxp_run(FxParser* p, const uint8_t* bytes, int n, FxByteRec* out) {
  FxModel* m = &p->s;
  int bad = 0;
  for (int k = 0; k < n; k++) {
    FxByteRec* r = &out[k];
    memset(r, 0, sizeof *r);
    m->rec = r; m->scmA = r->sscm;
    memset(m->cn, 0, sizeof m->cn);
    m->c4 = (m->c4 << 8) + bytes[k];            // update1 :4760-4764
    ++m->blpos;
    byte_update(m);
    for (int i = 0; i < FX_NMAPS; i++) bad |= (m->cn[i] != FX_MAPS[i].C);
    r->rcm_cx = m->rcmA;
    r->mh[0] = m->t[LEN3]; r->mh[1] = m->t[LEN2]; r->mh[2] = m->t[LEN1]; r->mh[3] = wc_word(&m->worcxt, 1);
    r->s2 = m->stream2b; r->s3 = m->stream3b; r->s3R = m->stream3bR; r->s2R = m->stream2bR;
    r->AH1 = m->AH1; r->AH2 = m->AH2; r->x5 = m->x5;
    r->deccode = (uint32_t)m->deccode;
    r->pc1 = (uint8_t)m->c1;
    r->BrFc = (uint8_t)m->BrFcIdx; r->FcIdx = (uint8_t)m->FcIdx; r->words = m->words; r->numbers = m->numbers; r->isPar = (uint8_t)m->isParagraph;
  }
  return bad ? -1 : 0;
}
