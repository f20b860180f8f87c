/* oracle/fxcm_stem.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of fxcm's Word and EnglishStemmer (reference src/models/fxcmv1.cpp:2302-3216), a derivative of the
 * paq8 stemmer restated in oracle/paq8_stem.c (whose Word helpers it shares): word-class, suffix and prefix flags live
 * in three separate words; one 32-bit stem hash; more prefixes ("anti-", "dis-"); apostrophes trimmed from both ends;
 * regions end at Length(); no case folding (the model feeds lower-case letters); closed word classes (articles,
 * conjunctions, adpositions, auxiliary verbs, numbers) recognised after stemming. Word / suffix lists and the
 * per-suffix flag words are dumped from the live reference build (oracle/fxcm_stem_tables.h). Pinned against the
 * reference's own class in tests/test_oracle_fxcmcore.py. */
#include <ctype.h>
#include <stdint.h>
#include <string.h>

#include "fxcm_stem.h"
#include "fxcm_stem_tables.h"
#define COUNT(a) ((int)(sizeof(a) / sizeof((a)[0])))

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
void fxw_add(FxWord* W, int c) {  /* Word::operator+= :2315-2320: signed char, bytes >= 0x80 are dropped */
  P8Word* x = &W->w;
  if ((signed char)c > 0 && x->End < P8_MAX_WORD - 1) { x->End += (x->Letters[x->End] > 0); x->Letters[x->End] = (uint8_t)c; }
}
#define w (&W->w)

int fx_is_vowel(int c) { return in_set(c, FXW_Vowels, COUNT(FXW_Vowels)); }
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
int fx_stem(FxWord* W, int blpos) {  /* Stem :3143-3206 */
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

/* test entry: stem one word given as a C string */
int orc_fx_stem_word(const char* s, int blpos, uint8_t* letters64, int* start_end, uint32_t* hash_type_suffix_prefix) {
  FxWord W;
  memset(&W, 0, sizeof W);
  for (const char* p = s; *p; ++p) fxw_add(&W, *p);
  const int r = fx_stem(&W, blpos);
  memcpy(letters64, W.w.Letters, 64);
  start_end[0] = W.w.Start; start_end[1] = W.w.End;
  hash_type_suffix_prefix[0] = W.Hash; hash_type_suffix_prefix[1] = W.Type; hash_type_suffix_prefix[2] = W.Suffix; hash_type_suffix_prefix[3] = W.Preffix;
  return r;
}
