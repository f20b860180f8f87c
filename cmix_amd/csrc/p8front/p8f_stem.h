/* p8front/p8f_stem.h -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns). Word (reference src/models/paq8.cpp:1545-1622) and the stemmers'
 * interface shared by p8f_stem.c and the models that use them. */
#ifndef CMX_P8F_STEM_H
#define CMX_P8F_STEM_H
#include <stdint.h>
#include <string.h>

#define P8_MAX_WORD 64
typedef struct {
  uint8_t Letters[P8_MAX_WORD];
  uint8_t Start, End;
  uint64_t Hash[4], Type, Language;
} P8Word;

enum { LANG_Unknown, LANG_English, LANG_French, LANG_German, LANG_Count };
enum { LANG_Verb = 1 << 0, LANG_Noun = 1 << 1 };

void p8w_init(P8Word* w);
void p8w_add(P8Word* w, int c);                 /* operator+= */
uint8_t p8w_at(const P8Word* w, int i);         /* operator[] : i-th letter from the start */
uint8_t p8w_back(const P8Word* w, int i);       /* operator() : i-th letter from the end */
uint32_t p8w_len(const P8Word* w);
void p8w_hashes(P8Word* w);                     /* GetHashes */
int p8w_eq(const P8Word* w, const char* s);
int p8w_ends(const P8Word* w, const char* s);
int p8w_starts(const P8Word* w, const char* s);
int p8w_change_suffix(P8Word* w, const char* old_suffix, const char* new_suffix);
int p8w_matches_any(const P8Word* w, const char* const* a, int count);
int p8_en_stem(P8Word* w);                      /* EnglishStemmer::Stem */
int p8_en_is_vowel(int c);
int p8_fr_stem(P8Word* w);                      /* FrenchStemmer::Stem */
int p8_fr_is_vowel(int c);
int p8_de_stem(P8Word* w);                      /* GermanStemmer::Stem */
int p8_de_is_vowel(int c);
#endif
