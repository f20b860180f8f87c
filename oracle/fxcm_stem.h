/* oracle/fxcm_stem.h -- TEST INFRASTRUCTURE ONLY. fxcm's Word (reference src/models/fxcmv1.cpp:2302-2367): the letters
 * (shared helpers: oracle/paq8_stem.h) plus one stem hash and three flag words. */
#ifndef ORACLE_FXCM_STEM_H
#define ORACLE_FXCM_STEM_H
#include "paq8_stem.h"
typedef struct { P8Word w; uint32_t Hash, Type, Suffix, Preffix; } FxWord;
void fxw_add(FxWord* W, int c);          /* Word::operator+= */
int fx_stem(FxWord* W, int blpos);       /* EnglishStemmer::Stem; blpos = position in the block (one word list depends on it) */
int fx_is_vowel(int c);
#endif
