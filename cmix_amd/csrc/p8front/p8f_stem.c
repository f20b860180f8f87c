/* p8front/p8f_stem.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's Word (reference src/models/paq8.cpp:1545-1622) and EnglishStemmer (:1764-2431), a Porter2
 * derivative with prefix / superlative handling and word-class flags, used by wordModel and TextModel. The suffix and
 * exception lists are data extracted from the reference (p8f_stem_tables.h, scripts/gen_paq8_stem_tables.py);
 * the control flow below follows the reference step by step because the stem, the flags and the hashes of every word
 * have to come out identical. Pinned in tests/test_p8front_twins.py (THIS file's objects against the reference's own class on a 6 000-word vocabulary
 * and its inflected forms; tests/test_oracle_paq8core.py pins the oracle's twin, oracle/paq8_stem.c, the same way). */
#include <ctype.h>
#include <stdint.h>
#include <string.h>

#include "p8f_stem.h"

/* English::Flags :1670-1699 (bit positions are part of the hashed state) */
enum {
  EN_Verb = 1 << 0, EN_Noun = 1 << 1, EN_Adjective = 1 << 2, EN_Plural = 1 << 3, EN_Male = 1 << 4, EN_Female = 1 << 5, EN_Negation = 1 << 6,
  EN_PastTense = (1 << 7) | EN_Verb, EN_PresentParticiple = (1 << 8) | EN_Verb, EN_AdjectiveSuperlative = (1 << 9) | EN_Adjective,
  EN_AdjectiveWithout = (1 << 10) | EN_Adjective, EN_AdjectiveFull = (1 << 11) | EN_Adjective, EN_AdverbOfManner = 1 << 12,
  EN_SuffixNESS = 1 << 13, EN_SuffixITY = (1 << 14) | EN_Noun, EN_SuffixCapable = 1 << 15, EN_SuffixNCE = 1 << 16, EN_SuffixNT = 1 << 17,
  EN_SuffixION = 1 << 18, EN_SuffixAL = (1 << 19) | EN_Adjective, EN_SuffixIC = (1 << 20) | EN_Adjective, EN_SuffixIVE = 1 << 21,
  EN_SuffixOUS = (1 << 22) | EN_Adjective, EN_PrefixOver = 1 << 23, EN_PrefixUnder = 1 << 24
};
enum { FR_Verb = 1, FR_Noun = 2, FR_Adjective = 4, FR_Plural = 8 };
enum { DE_Verb = 1, DE_Noun = 2, DE_Adjective = 4, DE_Plural = 8, DE_Female = 16 };
#include "p8f_stem_tables.h"
#define COUNT(a) ((int)(sizeof(a) / sizeof((a)[0])))

uint64_t p8f_hash2(uint64_t a, uint64_t b);
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c);

/* ---- Word ---- */
void p8w_init(P8Word* w) { memset(w, 0, sizeof *w); }
void p8w_add(P8Word* w, int c) {
  if (w->End < P8_MAX_WORD - 1) {
    w->End += (w->Letters[w->End] > 0);
    w->Letters[w->End] = (uint8_t)tolower((char)c);
  }
}
uint8_t p8w_at(const P8Word* w, int i) { return (w->End - w->Start >= (uint8_t)i) ? w->Letters[w->Start + (uint8_t)i] : 0; }
uint8_t p8w_back(const P8Word* w, int i) { return (w->End - w->Start >= (uint8_t)i) ? w->Letters[w->End - (uint8_t)i] : 0; }
uint32_t p8w_len(const P8Word* w) { return w->Letters[w->Start] != 0 ? (uint32_t)(w->End - w->Start + 1) : 0; }
void p8w_hashes(P8Word* w) {
  w->Hash[0] = 0xc01dflu; w->Hash[1] = ~w->Hash[0];
  for (int i = w->Start; i <= w->End; i++) {
    const uint8_t l = w->Letters[i];
    w->Hash[0] ^= p8f_hash3(w->Hash[0], l, (uint64_t)i);
    w->Hash[1] ^= p8f_hash2(w->Hash[1], ((l & 0x80) == 0) ? l & 0x5F : ((l & 0xC0) == 0x80) ? l & 0x3F : ((l & 0xE0) == 0xC0) ? l & 0x1F
                                                      : ((l & 0xF0) == 0xE0) ? l & 0xF : l & 0x7);
  }
  w->Hash[2] = (~w->Hash[0]) ^ w->Hash[1];
  w->Hash[3] = (~w->Hash[1]) ^ w->Hash[0];
}
int p8w_eq(const P8Word* w, const char* s) {
  const size_t len = strlen(s);
  return (size_t)(w->End - w->Start + (w->Letters[w->Start] != 0)) == len && memcmp(&w->Letters[w->Start], s, len) == 0;
}
int p8w_ends(const P8Word* w, const char* s) {
  const size_t len = strlen(s);
  return p8w_len(w) > len && memcmp(&w->Letters[w->End - len + 1], s, len) == 0;
}
int p8w_starts(const P8Word* w, const char* s) {
  const size_t len = strlen(s);
  return p8w_len(w) > len && memcmp(&w->Letters[w->Start], s, len) == 0;
}
int p8w_change_suffix(P8Word* w, const char* old_suffix, const char* new_suffix) {
  const size_t len = strlen(old_suffix);
  if (p8w_len(w) > len && memcmp(&w->Letters[w->End - len + 1], old_suffix, len) == 0) {
    const size_t n = strlen(new_suffix);
    if (n > 0) {
      const int lim = (P8_MAX_WORD - 1 < w->End + (int)n ? P8_MAX_WORD - 1 : w->End + (int)n) - w->End;
      memcpy(&w->Letters[w->End - (int)len + 1], new_suffix, (size_t)lim);
      const int e = w->End - (int)len + (int)n;
      w->End = (uint8_t)(P8_MAX_WORD - 1 < e ? P8_MAX_WORD - 1 : e);
    } else w->End -= (uint8_t)len;
    return 1;
  }
  return 0;
}
int p8w_matches_any(const P8Word* w, const char* const* a, int count) {
  const size_t len = p8w_len(w);
  int i = 0;
  for (; i < count && (len != strlen(a[i]) || memcmp(&w->Letters[w->Start], a[i], len) != 0); i++) {}
  return i < count;
}
static int in_set(int c, const char* a, int n) { int i = 0; for (; i < n && (char)c != a[i]; i++) {} return i < n; }

/* ---- Stemmer base :1729-1762 ---- */
static uint32_t region(const P8Word* w, uint32_t from, int (*is_vowel)(int)) {
  int has_vowel = 0;
  for (int i = w->Start + (int)from; i <= w->End; i++) {
    if (is_vowel(w->Letters[i])) { has_vowel = 1; continue; }
    else if (has_vowel) return (uint32_t)(i - w->Start + 1);
  }
  return w->Start + p8w_len(w);
}
static int suffix_in_rn(const P8Word* w, uint32_t rn, const char* suffix) {
  return w->Start != w->End && rn <= p8w_len(w) - (uint32_t)strlen(suffix);
}

/* ---- EnglishStemmer ---- */
int p8_en_is_vowel(int c) { return in_set(c, EN_Vowels, COUNT(EN_Vowels)); }
#define V(c) p8_en_is_vowel(c)
#define CONS(c) (!p8_en_is_vowel(c))
#define B(i) p8w_back(w, i)
#define F(i) p8w_at(w, i)
static uint32_t en_region1(const P8Word* w) {
  for (int i = 0; i < COUNT(EN_ExceptionsRegion1); i++)
    if (p8w_starts(w, EN_ExceptionsRegion1[i])) return (uint32_t)strlen(EN_ExceptionsRegion1[i]);
  return region(w, 0, p8_en_is_vowel);
}
static int en_short_syllable(const P8Word* w) {
  if (w->End == w->Start) return 0;
  if (w->End == w->Start + 1) return V(B(1)) && CONS(B(0));
  return CONS(B(2)) && V(B(1)) && CONS(B(0)) && !in_set(B(0), EN_NonShortConsonants, COUNT(EN_NonShortConsonants));
}
static int en_short_word(const P8Word* w) { return en_short_syllable(w) && en_region1(w) == p8w_len(w); }
static int en_has_vowels(const P8Word* w) { for (int i = w->Start; i <= w->End; i++) if (V(w->Letters[i])) return 1; return 0; }
static void en_hash(P8Word* w) {  /* EnglishStemmer::Hash :2350-2362 */
  w->Hash[2] = w->Hash[3] = 0xb0a710ad;
  for (int i = w->Start; i <= w->End; i++) {
    const uint8_t l = w->Letters[i];
    w->Hash[2] = w->Hash[2] * 263 * 32 + l;
    if (V(l)) w->Hash[3] = w->Hash[3] * 997 * 8 + (uint64_t)(int64_t)(l / 4 - 22);
    else if (l >= 'b' && l <= 'z') w->Hash[3] = w->Hash[3] * 271 * 32 + (uint64_t)(l - 97);
    else w->Hash[3] = w->Hash[3] * 11 * 32 + l;
  }
}
static int en_prefixes(P8Word* w) {  /* ProcessPrefixes :1934-1949 */
  if (p8w_starts(w, "irr") && p8w_len(w) > 5 && (F(3) == 'a' || F(3) == 'e')) { w->Start += 2; w->Type |= EN_Negation; }
  else if (p8w_starts(w, "over") && p8w_len(w) > 5) { w->Start += 4; w->Type |= EN_PrefixOver; }
  else if (p8w_starts(w, "under") && p8w_len(w) > 6) { w->Start += 5; w->Type |= EN_PrefixUnder; }
  else if (p8w_starts(w, "unn") && p8w_len(w) > 5) { w->Start += 2; w->Type |= EN_Negation; }
  else if (p8w_starts(w, "non") && p8w_len(w) > (uint32_t)(5 + (F(3) == '-'))) { w->Start += 2 + (F(3) == '-'); w->Type |= EN_Negation; }
  else return 0;
  return 1;
}
static int en_superlatives(P8Word* w) {  /* ProcessSuperlatives :1950-2043 */
  if (p8w_ends(w, "est") && p8w_len(w) > 4) {
    const uint8_t keep = w->End;
    w->End -= 3;
    w->Type |= EN_AdjectiveSuperlative;
#define UNDO() do { w->End = keep; w->Type &= ~(uint64_t)EN_AdjectiveSuperlative; } while (0)
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
        default: w->End += 3; w->Type &= ~(uint64_t)EN_AdjectiveSuperlative;
      }
    }
#undef UNDO
  }
  return (w->Type & EN_AdjectiveSuperlative) > 0;
}
static int en_step0(P8Word* w) {
  for (int i = 0; i < COUNT(EN_SuffixesStep0); i++)
    if (p8w_ends(w, EN_SuffixesStep0[i])) { w->End -= (uint8_t)strlen(EN_SuffixesStep0[i]); w->Type |= EN_Plural; return 1; }
  return 0;
}
static int en_step1a(P8Word* w) {  /* :2054-2103 */
  if (p8w_ends(w, "sses")) { w->End -= 2; w->Type |= EN_Plural; return 1; }
  if (p8w_ends(w, "ied") || p8w_ends(w, "ies")) {
    w->Type |= (B(0) == 'd') ? EN_PastTense : EN_Plural;
    w->End -= 1 + (p8w_len(w) > 4);
    return 1;
  }
  if (p8w_ends(w, "us") || p8w_ends(w, "ss")) return 0;
  if (B(0) == 's' && p8w_len(w) > 2)
    for (int i = w->Start; i <= w->End - 2; i++)
      if (V(w->Letters[i])) { w->End--; w->Type |= EN_Plural; return 1; }
  if (p8w_ends(w, "n't") && p8w_len(w) > 4) {
    switch (B(3)) {
      case 'a': if (B(4) == 'c') w->End -= 2; else p8w_change_suffix(w, "n't", "ll"); break;
      case 'i': p8w_change_suffix(w, "in't", "m"); break;
      case 'o': if (B(4) == 'w') p8w_change_suffix(w, "on't", "ill"); else w->End -= 3; break;
      default: w->End -= 3;
    }
    w->Type |= EN_Negation;
    return 1;
  }
  if (p8w_ends(w, "hood") && p8w_len(w) > 7) { w->End -= 4; return 1; }
  return 0;
}
static int en_step1b(P8Word* w, uint32_t R1) {  /* :2104-2222 */
  for (int i = 0; i < COUNT(EN_SuffixesStep1b); i++) {
    if (!p8w_ends(w, EN_SuffixesStep1b[i])) continue;
    if (i < 2) {
      if (suffix_in_rn(w, R1, EN_SuffixesStep1b[i])) w->End -= (uint8_t)(1 + i * 2);
    } else {
      const uint8_t j = w->End;
      w->End -= (uint8_t)strlen(EN_SuffixesStep1b[i]);
      if (!en_has_vowels(w)) { w->End = j; return 0; }
      if (p8w_ends(w, "at") || p8w_ends(w, "bl") || p8w_ends(w, "iz") || en_short_word(w)) p8w_add(w, 'e');
      else if (p8w_len(w) > 2) {
        if (B(0) == B(1) && in_set(B(0), EN_Doubles, COUNT(EN_Doubles))) w->End--;
        else if (i == 2 || i == 3) {
          switch (B(0)) {
            case 'c': case 's': case 'v': w->End += !(p8w_ends(w, "ss") || p8w_ends(w, "ias")); break;
            case 'd': w->End += V(B(1)) && (!in_set(B(2), EN_nAllowed, COUNT(EN_nAllowed))); break;
            case 'k': w->End += p8w_ends(w, "uak"); break;
            case 'l': w->End += in_set(B(1), EN_Allowed1, COUNT(EN_Allowed1)) || (in_set(B(1), EN_Allowed2, COUNT(EN_Allowed2)) && CONS(B(2))); break;
          }
        } else if (i >= 4) {
          switch (B(0)) {
            case 'd': if (V(B(1)) && B(2) != 'a' && B(2) != 'e' && B(2) != 'o') p8w_add(w, 'e'); break;
            case 'g':
              if (in_set(B(1), EN_Allowed, COUNT(EN_Allowed)) ||
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
    w->Type |= EN_TypesStep1b[i];
    return 1;
  }
  return 0;
}
static int en_step1c(P8Word* w) {
  if (p8w_len(w) > 2 && tolower(B(0)) == 'y' && CONS(B(1))) { w->Letters[w->End] = 'i'; return 1; }
  return 0;
}
static int en_step2(P8Word* w, uint32_t R1) {  /* :2231-2287 */
  for (int i = 0; i < COUNT(EN_SuffixesStep2); i++)
    if (p8w_ends(w, EN_SuffixesStep2[i][0]) && suffix_in_rn(w, R1, EN_SuffixesStep2[i][0])) {
      p8w_change_suffix(w, EN_SuffixesStep2[i][0], EN_SuffixesStep2[i][1]);
      w->Type |= EN_TypesStep2[i];
      return 1;
    }
  if (p8w_ends(w, "logi") && suffix_in_rn(w, R1, "ogi")) { w->End--; return 1; }
  else if (p8w_ends(w, "li")) {
    if (suffix_in_rn(w, R1, "li") && in_set(B(2), EN_LiEndings, COUNT(EN_LiEndings))) { w->End -= 2; w->Type |= EN_AdverbOfManner; return 1; }
    else if (p8w_len(w) > 3) {
      switch (B(2)) {
        case 'b': w->Letters[w->End] = 'e'; w->Type |= EN_AdverbOfManner; return 1;
        case 'i': if (p8w_len(w) > 4) { w->End -= 2; w->Type |= EN_AdverbOfManner; return 1; } break;
        case 'l': if (p8w_len(w) > 5 && (B(3) == 'a' || B(3) == 'u')) { w->End -= 2; w->Type |= EN_AdverbOfManner; return 1; } break;
        case 's': w->End -= 2; w->Type |= EN_AdverbOfManner; return 1;
        case 'e': case 'g': case 'm': case 'n': case 'r': case 'w':
          if (p8w_len(w) > (uint32_t)(4 + (B(2) == 'r'))) { w->End -= 2; w->Type |= EN_AdverbOfManner; return 1; }
      }
    }
  }
  return 0;
}
static int en_step3(P8Word* w, uint32_t R1, uint32_t R2) {
  int res = 0;
  for (int i = 0; i < COUNT(EN_SuffixesStep3); i++)
    if (p8w_ends(w, EN_SuffixesStep3[i][0]) && suffix_in_rn(w, R1, EN_SuffixesStep3[i][0])) {
      p8w_change_suffix(w, EN_SuffixesStep3[i][0], EN_SuffixesStep3[i][1]);
      w->Type |= EN_TypesStep3[i];
      res = 1;
      break;
    }
  if (p8w_ends(w, "ative") && suffix_in_rn(w, R2, "ative")) { w->End -= 5; w->Type |= EN_SuffixIVE; return 1; }
  if (p8w_len(w) > 5 && p8w_ends(w, "less")) { w->End -= 4; w->Type |= EN_AdjectiveWithout; return 1; }
  return res;
}
static int en_step4(P8Word* w, uint32_t R2) {
  int res = 0;
  for (int i = 0; i < COUNT(EN_SuffixesStep4); i++)
    if (p8w_ends(w, EN_SuffixesStep4[i]) && suffix_in_rn(w, R2, EN_SuffixesStep4[i])) {
      w->End -= (uint8_t)(strlen(EN_SuffixesStep4[i]) - (i > 17));
      if (i != 10 || B(0) != 'm') w->Type |= EN_TypesStep4[i];
      if (i == 0 && p8w_ends(w, "nti")) { w->End--; res = 1; continue; }
      return 1;
    }
  return res;
}
static int en_step5(P8Word* w, uint32_t R1, uint32_t R2) {
  if (B(0) == 'e' && !p8w_eq(w, "here")) {
    if (suffix_in_rn(w, R2, "e")) w->End--;
    else if (suffix_in_rn(w, R1, "e")) { w->End--; w->End += en_short_syllable(w); }
    else return 0;
    return 1;
  } else if (p8w_len(w) > 1 && B(0) == 'l' && suffix_in_rn(w, R2, "l") && B(1) == 'l') { w->End--; return 1; }
  return 0;
}
int p8_en_stem(P8Word* w) {  /* Stem :2363-2429 */
  if (p8w_len(w) < 2) { en_hash(w); return 0; }
  int res = (w->Start != w->End && F(0) == '\'');  /* TrimStartingApostrophe */
  w->Start += (uint8_t)res;
  res |= en_prefixes(w);
  res |= en_superlatives(w);
  for (int i = 0; i < COUNT(EN_Exceptions1); i++)
    if (p8w_eq(w, EN_Exceptions1[i][0])) {
      if (i < 11) {
        const size_t len = strlen(EN_Exceptions1[i][1]);
        memcpy(&w->Letters[w->Start], EN_Exceptions1[i][1], len);
        w->End = (uint8_t)(w->Start + (uint8_t)(len - 1));
      }
      en_hash(w);
      w->Type |= EN_TypesExceptions1[i];
      w->Language = LANG_English;
      return i < 11;
    }
  if (F(0) == 'y') w->Letters[w->Start] = 'Y';  /* MarkYsAsConsonants */
  for (int i = w->Start + 1; i <= w->End; i++)
    if (V(w->Letters[i - 1]) && w->Letters[i] == 'y') w->Letters[i] = 'Y';
  const uint32_t R1 = en_region1(w), R2 = region(w, R1, p8_en_is_vowel);
  res |= en_step0(w);
  res |= en_step1a(w);
  for (int i = 0; i < COUNT(EN_Exceptions2); i++)
    if (p8w_eq(w, EN_Exceptions2[i])) {
      en_hash(w);
      w->Type |= EN_TypesExceptions2[i];
      w->Language = LANG_English;
      return res;
    }
  res |= en_step1b(w, R1);
  res |= en_step1c(w);
  res |= en_step2(w, R1);
  res |= en_step3(w, R1, R2);
  res |= en_step4(w, R2);
  res |= en_step5(w, R1, R2);
  for (uint8_t i = w->Start; i <= w->End; i++)
    if (w->Letters[i] == 'Y') w->Letters[i] = 'y';
  if (!w->Type || w->Type == EN_Plural) {
    if (p8w_matches_any(w, EN_MaleWords, COUNT(EN_MaleWords))) { res = 1; w->Type |= EN_Male; }
    else if (p8w_matches_any(w, EN_FemaleWords, COUNT(EN_FemaleWords))) { res = 1; w->Type |= EN_Female; }
  }
  if (!res) res = p8w_matches_any(w, EN_CommonWords, COUNT(EN_CommonWords));
  en_hash(w);
  if (res) w->Language = LANG_English;
  return res;
}

/* ---- FrenchStemmer :2433-2822 (a Snowball-French derivative working on Latin-1 letters) ---- */
#undef V
#undef CONS
int p8_fr_is_vowel(int c) { return in_set(c, FR_Vowels, COUNT(FR_Vowels)); }
#define V(c) p8_fr_is_vowel(c)
#define CONS(c) (!p8_fr_is_vowel(c))
#define SLEN(s) ((uint8_t)strlen(s))
#define ENDS_IN(s, rn) (p8w_ends(w, s) && suffix_in_rn(w, rn, s))
static void fr_utf8(P8Word* w) {  /* ConvertUTF8 :2487-2497: 0xC3 xx pairs folded to one Latin-1 letter, in place */
  for (int i = w->Start; i < w->End; i++) {
    const uint8_t n = w->Letters[i + 1], c = (uint8_t)(n + (n < 0xA0 ? 0x60 : 0x40));
    if (w->Letters[i] == 0xC3 && (V(c) || (n & 0xDF) == 0x87)) {
      w->Letters[i] = c;
      if (i + 1 < w->End) memmove(&w->Letters[i + 1], &w->Letters[i + 2], (size_t)(w->End - i - 1));
      w->End--;
    }
  }
}
static void fr_mark(P8Word* w) {  /* MarkVowelsAsConsonants :2498-2513 */
  uint8_t* L = w->Letters;
  for (int i = w->Start; i <= w->End; i++) {
    if (L[i] == 'i' || L[i] == 'u') {
      if (i > w->Start && i < w->End && (V(L[i - 1]) || (L[i - 1] == 'q' && L[i] == 'u')) && V(L[i + 1])) L[i] = (uint8_t)toupper(L[i]);
    } else if (L[i] == 'y') {
      if ((i > w->Start && V(L[i - 1])) || (i < w->End && V(L[i + 1]))) L[i] = 'Y';
    }
  }
}
static uint32_t fr_rv(const P8Word* w) {  /* GetRV :2514-2526 */
  const uint32_t len = p8w_len(w);
  if (len >= 3 && ((V(w->Letters[w->Start]) && V(w->Letters[w->Start + 1])) || p8w_starts(w, "par") || p8w_starts(w, "col") || p8w_starts(w, "tap")))
    return (uint32_t)w->Start + 3;
  for (int i = w->Start + 1; i <= w->End; i++)
    if (V(w->Letters[i])) return (uint32_t)i + 1;
  return w->Start + len;
}
static void fr_ic(P8Word* w, uint32_t R2) {  /* "ic": dropped inside R2, respelt "iqU" outside */
  if (suffix_in_rn(w, R2, "ic")) w->End -= 2;
  else p8w_change_suffix(w, "c", "qU");
}
static int fr_step1(P8Word* w, uint32_t RV, uint32_t R1, uint32_t R2, int* force2a) {  /* :2527-2664; the suffix list is scanned in groups */
  const char* const* S = FR_SuffixesStep1;
  int i = 0;
  for (; i < 11; i++)
    if (ENDS_IN(S[i], R2)) { w->End -= SLEN(S[i]); if (i == 3) w->Type |= FR_Adjective; return 1; }
  for (; i < 17; i++)
    if (ENDS_IN(S[i], R2)) { w->End -= SLEN(S[i]); if (p8w_ends(w, "ic")) p8w_change_suffix(w, "c", "qU"); return 1; }
  for (; i < 25; i++)
    if (ENDS_IN(S[i], R2)) {
      w->End -= (uint8_t)(SLEN(S[i]) - 1 - (i < 19) * 2);
      if (i > 22) { w->End += 2; w->Letters[w->End] = 't'; }
      return 1;
    }
  for (; i < 27; i++)
    if (ENDS_IN(S[i], R1) && CONS(p8w_back(w, SLEN(S[i])))) { w->End -= SLEN(S[i]); return 1; }
  for (; i < 29; i++)
    if (ENDS_IN(S[i], RV)) {
      w->End -= SLEN(S[i]);
      if (ENDS_IN("iv", R2)) {
        w->End -= 2;
        if (ENDS_IN("at", R2)) w->End -= 2;
      } else if (p8w_ends(w, "eus")) {
        if (suffix_in_rn(w, R2, "eus")) w->End -= 3;
        else if (suffix_in_rn(w, R1, "eus")) w->Letters[w->End] = 'x';
      } else if (ENDS_IN("abl", R2) || ENDS_IN("iqU", R2)) w->End -= 3;
      else if (ENDS_IN("i\xE8r", RV) || ENDS_IN("I\xE8r", RV)) { w->End -= 2; w->Letters[w->End] = 'i'; }
      return 1;
    }
  for (; i < 31; i++)
    if (ENDS_IN(S[i], R2)) {
      w->End -= SLEN(S[i]);
      if (p8w_ends(w, "abil")) {
        if (suffix_in_rn(w, R2, "abil")) w->End -= 4;
        else { w->End--; w->Letters[w->End] = 'l'; }
      } else if (p8w_ends(w, "ic")) fr_ic(w, R2);
      else if (ENDS_IN("iv", R2)) w->End -= 2;
      return 1;
    }
  for (; i < 35; i++)
    if (ENDS_IN(S[i], R2)) {
      w->End -= SLEN(S[i]);
      if (ENDS_IN("at", R2)) {
        w->End -= 2;
        if (p8w_ends(w, "ic")) fr_ic(w, R2);
      }
      return 1;
    }
  for (; i < 37; i++)
    if (p8w_ends(w, S[i])) {
      if (suffix_in_rn(w, R2, S[i])) { w->End -= SLEN(S[i]); return 1; }
      if (suffix_in_rn(w, R1, S[i])) { p8w_change_suffix(w, S[i], "eux"); return 1; }
    }
  for (; i < COUNT(FR_SuffixesStep1); i++)
    if (ENDS_IN(S[i], RV + 1) && V(p8w_back(w, SLEN(S[i])))) { w->End -= SLEN(S[i]); *force2a = 1; return 1; }
  if (p8w_ends(w, "eaux") || p8w_eq(w, "eaux")) { w->End--; w->Type |= FR_Plural; return 1; }
  if (ENDS_IN("aux", R1)) { w->End--; w->Letters[w->End] = 'l'; w->Type |= FR_Plural; return 1; }
  if (ENDS_IN("amment", RV)) { p8w_change_suffix(w, "amment", "ant"); *force2a = 1; return 1; }
  if (ENDS_IN("emment", RV)) { p8w_change_suffix(w, "emment", "ent"); *force2a = 1; return 1; }
  return 0;
}
static int fr_step2a(P8Word* w, uint32_t RV) {
  for (int i = 0; i < COUNT(FR_SuffixesStep2a); i++) {
    const char* s = FR_SuffixesStep2a[i];
    if (ENDS_IN(s, RV + 1) && CONS(p8w_back(w, SLEN(s)))) { w->End -= SLEN(s); if (i == 31) w->Type |= FR_Verb; return 1; }
  }
  return 0;
}
static int fr_step2b(P8Word* w, uint32_t RV, uint32_t R2) {
  for (int i = 0; i < COUNT(FR_SuffixesStep2b); i++) {
    const char* s = FR_SuffixesStep2b[i];
    if (!ENDS_IN(s, RV)) continue;
    if (s[0] == 'a' || s[0] == '\xE2') {
      w->End -= SLEN(s);
      if (ENDS_IN("e", RV)) w->End--;
      return 1;
    }
    if (i != 14 || suffix_in_rn(w, R2, s)) { w->End -= SLEN(s); return 1; }
  }
  return 0;
}
static int fr_step4(P8Word* w, uint32_t RV, uint32_t R2) {
  int res = 0;
  if (p8w_len(w) >= 2 && w->Letters[w->End] == 's' && !in_set(p8w_back(w, 1), FR_SetStep4, COUNT(FR_SetStep4))) { w->End--; res = 1; }
  for (int i = 0; i < COUNT(FR_SuffixesStep4); i++) {
    const char* s = FR_SuffixesStep4[i];
    if (!ENDS_IN(s, RV)) continue;
    if (i == 2) {          /* ion: only after s / t, inside R2 */
      const uint8_t prec = p8w_back(w, 3);
      if (suffix_in_rn(w, R2, s) && suffix_in_rn(w, RV + 1, s) && (prec == 's' || prec == 't')) { w->End -= 3; return 1; }
    } else if (i == 5) { w->End--; return 1; }
    else if (i == 6) { if (p8w_ends(w, "gu\xEB")) { w->End--; return 1; } }
    else { p8w_change_suffix(w, s, "i"); return 1; }
  }
  return res;
}
static void fr_hash(P8Word* w) {  /* :2766-2778; the seed is ~0xeff1cace as a 32-bit value */
  w->Hash[2] = w->Hash[3] = 0x100e3531u;
  for (int i = w->Start; i <= w->End; i++) {
    const uint8_t l = w->Letters[i];
    w->Hash[2] = w->Hash[2] * 251 * 32 + l;
    if (V(l)) w->Hash[3] = w->Hash[3] * 997 * 16 + l;
    else if (l >= 'b' && l <= 'z') w->Hash[3] = w->Hash[3] * 271 * 32 + (uint64_t)(l - 97);
    else w->Hash[3] = w->Hash[3] * 11 * 32 + l;
  }
}
int p8_fr_stem(P8Word* w) {  /* Stem :2779-2821 */
  fr_utf8(w);
  if (p8w_len(w) < 2) { fr_hash(w); return 0; }
  for (int i = 0; i < COUNT(FR_Exceptions); i++)
    if (p8w_eq(w, FR_Exceptions[i][0])) {
      const size_t len = strlen(FR_Exceptions[i][1]);
      memcpy(&w->Letters[w->Start], FR_Exceptions[i][1], len);
      w->End = (uint8_t)(w->Start + (uint8_t)(len - 1));
      fr_hash(w);
      w->Type |= FR_TypesExceptions[i];
      w->Language = LANG_French;
      return 1;
    }
  fr_mark(w);
  const uint32_t RV = fr_rv(w), R1 = region(w, 0, p8_fr_is_vowel), R2 = region(w, R1, p8_fr_is_vowel);
  int next = 0, res = fr_step1(w, RV, R1, R2, &next);
  next |= !res;
  if (next) {
    next = !fr_step2a(w, RV);
    res |= !next;
    if (next) res |= fr_step2b(w, RV, R2);
  }
  if (res) {  /* Step3 */
    uint8_t* last = &w->Letters[w->End];
    if (*last == 'Y') *last = 'i';
    else if (*last == 0xE7) *last = 'c';
  } else res |= fr_step4(w, RV, R2);
  { int s5 = 0;  /* Step5: undouble */
    for (int i = 0; i < COUNT(FR_SuffixesStep5) && !s5; i++) if (p8w_ends(w, FR_SuffixesStep5[i])) { w->End--; s5 = 1; }
    res |= s5; }
  for (int i = w->End; i >= w->Start; i--)  /* Step6: unaccent the last vowel when consonants follow it */
    if (V(w->Letters[i])) {
      if (i < w->End && (w->Letters[i] & 0xFE) == 0xE8) { w->Letters[i] = 'e'; res |= 1; }
      break;
    }
  for (int i = w->Start; i <= w->End; i++) w->Letters[i] = (uint8_t)tolower(w->Letters[i]);
  if (!res) res = p8w_matches_any(w, FR_CommonWords, COUNT(FR_CommonWords));
  fr_hash(w);
  if (res) w->Language = LANG_French;
  return res;
}

/* ---- GermanStemmer :2831-3004 ---- */
#undef V
#undef CONS
int p8_de_is_vowel(int c) { return in_set(c, DE_Vowels, COUNT(DE_Vowels)); }
#define V(c) p8_de_is_vowel(c)
static int de_valid_ending(int c, int include_r) { return in_set(c, DE_Endings, COUNT(DE_Endings)) || (include_r && (char)c == 'r'); }
static void de_hash(P8Word* w) {  /* :2958-2970; the seed is ~0xbea7ab1e as a 32-bit value */
  w->Hash[2] = w->Hash[3] = 0x415854e1u;
  for (int i = w->Start; i <= w->End; i++) {
    const uint8_t l = w->Letters[i];
    w->Hash[2] = w->Hash[2] * 263 * 32 + l;
    if (V(l)) w->Hash[3] = w->Hash[3] * 997 * 16 + l;
    else if (l >= 'b' && l <= 'z') w->Hash[3] = w->Hash[3] * 251 * 32 + (uint64_t)(l - 97);
    else w->Hash[3] = w->Hash[3] * 11 * 32 + l;
  }
}
static int de_step1(P8Word* w, uint32_t R1) {
  for (int i = 0; i < COUNT(DE_SuffixesStep1); i++)
    if (ENDS_IN(DE_SuffixesStep1[i], R1)) {
      w->End -= SLEN(DE_SuffixesStep1[i]);
      if (i >= 3) w->End -= (uint8_t)p8w_ends(w, "niss");
      return 1;
    }
  if (ENDS_IN("s", R1) && de_valid_ending(p8w_back(w, 1), 1)) { w->End--; return 1; }
  return 0;
}
static int de_step2(P8Word* w, uint32_t R1) {
  for (int i = 0; i < COUNT(DE_SuffixesStep2); i++)
    if (ENDS_IN(DE_SuffixesStep2[i], R1)) { w->End -= SLEN(DE_SuffixesStep2[i]); return 1; }
  if (ENDS_IN("st", R1) && p8w_len(w) > 5 && de_valid_ending(p8w_back(w, 2), 0)) { w->End -= 2; return 1; }
  return 0;
}
static int de_step3(P8Word* w, uint32_t R1, uint32_t R2) {  /* :2912-2953 */
  const char* const* S = DE_SuffixesStep3;
  int i = 0;
  for (; i < 2; i++)
    if (ENDS_IN(S[i], R2)) {
      w->End -= SLEN(S[i]);
      if (p8w_ends(w, "ig") && p8w_back(w, 2) != 'e' && suffix_in_rn(w, R2, "ig")) w->End -= 2;
      if (i) w->Type |= DE_Noun;
      return 1;
    }
  for (; i < 5; i++)
    if (ENDS_IN(S[i], R2) && p8w_back(w, SLEN(S[i])) != 'e') { w->End -= SLEN(S[i]); if (i > 2) w->Type |= DE_Adjective; return 1; }
  for (; i < COUNT(DE_SuffixesStep3); i++)
    if (ENDS_IN(S[i], R2)) {
      w->End -= SLEN(S[i]);
      if ((p8w_ends(w, "er") || p8w_ends(w, "en")) && suffix_in_rn(w, R1, "e?")) w->End -= 2;
      if (i > 5) w->Type |= DE_Noun | DE_Female;
      return 1;
    }
  if (ENDS_IN("keit", R2)) {
    w->End -= 4;
    if (ENDS_IN("lich", R2)) w->End -= 4;
    else if (ENDS_IN("ig", R2)) w->End -= 2;
    w->Type |= DE_Noun | DE_Female;
    return 1;
  }
  return 0;
}
/* The reference closes the gap left by a folded UTF-8 pair with memcpy() on OVERLAPPING ranges (:2852, source one byte
 * above the destination) -- undefined behaviour whose outcome depends on how the compiler expands the call. The
 * reference as built here (g++ 11 -O3, oracle/Makefile) expands it inline: head block, then tail block, then the
 * 8-byte-aligned middle, loads and stores interleaved -- so letters after an umlaut come out doubled / dropped when 5
 * or more follow it. This reproduces that expansion (Letters is 8-byte aligned in Word, as in the reference's
 * calloc'ed caches); pinned by the stemmer test against the reference build. */
static void de_close_gap(uint8_t* L, int d, int n) {
  uint8_t t[8];
  const int s = d + 1;
#define BLOCK(off, len) do { memcpy(t, L + s + (off), (len)); memcpy(L + d + (off), t, (len)); } while (0)
  if (n >= 8) {
    BLOCK(0, 8);
    BLOCK(n - 8, 8);
    const int k = 8 - (d & 7);
    for (int j = 0; j < (n - k) >> 3; j++) BLOCK(k + 8 * j, 8);
  } else if (n & 4) { BLOCK(0, 4); BLOCK(n - 4, 4); }
  else if (n) { L[d] = L[s]; if (n & 2) BLOCK(n - 2, 2); }
#undef BLOCK
}
int p8_de_stem(P8Word* w) {  /* Stem :2971-3002 */
  for (int i = w->Start; i < w->End; i++) {  /* ConvertUTF8 :2846-2856 */
    const uint8_t n = w->Letters[i + 1], c = (uint8_t)(n + (n < 0x9F ? 0x60 : 0x40));
    if (w->Letters[i] == 0xC3 && (V(c) || c == 0xDF)) {
      w->Letters[i] = c;
      if (i + 1 < w->End) de_close_gap(w->Letters, i + 1, w->End - i - 1);
      w->End--;
    }
  }
  if (p8w_len(w) < 2) { de_hash(w); return 0; }
  for (int i = w->Start; i <= w->End; i++)  /* ReplaceSharpS */
    if (w->Letters[i] == 0xDF) {
      w->Letters[i] = 's';
      if (i + 1 < P8_MAX_WORD) {
        memmove(&w->Letters[i + 2], &w->Letters[i + 1], (size_t)(P8_MAX_WORD - i - 2));
        w->Letters[i + 1] = 's';
        w->End += (w->End < P8_MAX_WORD - 1);
      }
    }
  for (int i = w->Start + 1; i < w->End; i++) {  /* MarkVowelsAsConsonants */
    const uint8_t c = w->Letters[i];
    if ((c == 'u' || c == 'y') && V(w->Letters[i - 1]) && V(w->Letters[i + 1])) w->Letters[i] = (uint8_t)toupper(c);
  }
  uint32_t R1 = region(w, 0, p8_de_is_vowel);
  const uint32_t R2 = region(w, R1, p8_de_is_vowel);
  if ((int)R1 > 3) R1 = 3;  /* min(3, R1) on ints */
  int res = de_step1(w, R1);
  res |= de_step2(w, R1);
  res |= de_step3(w, R1, R2);
  for (int i = w->Start; i <= w->End; i++) {
    uint8_t* l = &w->Letters[i];
    if (*l == 0xE4) *l = 'a';
    else if (*l == 0xF6 || *l == 0xFC) *l -= 0x87;
    else *l = (uint8_t)tolower(*l);
  }
  if (!res) res = p8w_matches_any(w, DE_CommonWords, COUNT(DE_CommonWords));
  de_hash(w);
  if (res) w->Language = LANG_German;
  return res;
}

/* test entry: stem one word given as a C string (letters are added the way wordModel adds them) */
int p8f_stem_word(int lang, const char* s, uint8_t* letters64, int* start_end, uint64_t* type_lang, uint64_t* hash4_after_stem,
                     uint64_t* hash4_gethashes) {
  P8Word w;
  p8w_init(&w);
  for (const char* p = s; *p; ++p) p8w_add(&w, *p);
  const int r = lang == LANG_French ? p8_fr_stem(&w) : lang == LANG_German ? p8_de_stem(&w) : p8_en_stem(&w);
  memcpy(letters64, w.Letters, 64);
  start_end[0] = w.Start; start_end[1] = w.End;
  type_lang[0] = w.Type; type_lang[1] = w.Language;
  memcpy(hash4_after_stem, w.Hash, 32);
  p8w_hashes(&w);
  memcpy(hash4_gethashes, w.Hash, 32);
  return r;
}
int p8f_en_stem_word(const char* s, uint8_t* letters64, int* start_end, uint64_t* type_lang, uint64_t* hash4_after_stem,
                        uint64_t* hash4_gethashes) {
  return p8f_stem_word(LANG_English, s, letters64, start_end, type_lang, hash4_after_stem, hash4_gethashes);
}
