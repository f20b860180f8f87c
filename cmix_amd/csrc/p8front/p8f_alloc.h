/* p8f_alloc.h -- force-included into the front end's C files: every calloc / malloc of a front end is tracked, so
 * p8f_front_free() releases a stream's models in one sweep (the models are plain C structs with no destructors). */
#ifndef CMX_P8F_ALLOC_H
#define CMX_P8F_ALLOC_H
#include <stdlib.h>
void* p8f_tracked_calloc(size_t n, size_t size);
#define calloc(n, s) p8f_tracked_calloc((n), (s))
#define malloc(s) p8f_tracked_calloc(1, (s))

/* Character classes as the reference sees them: it never calls setlocale(), so its <ctype.h> is the "C" locale's. A host program
 * that has (Python does at start-up) must not change what a word is: ASCII-only classes, whatever LC_CTYPE says. Arguments may be
 * negative (a `char` >= 0x80), as in the reference; such values belong to no class and map to themselves. */
#include <ctype.h>
#undef isalpha
#undef ispunct
#undef isspace
#undef isdigit
#undef isupper
#undef islower
#undef isalnum
#undef tolower
#undef toupper
static inline int p8f_isupper(int c) { return c >= 'A' && c <= 'Z'; }
static inline int p8f_islower(int c) { return c >= 'a' && c <= 'z'; }
static inline int p8f_isalpha(int c) { return p8f_isupper(c) || p8f_islower(c); }
static inline int p8f_isdigit(int c) { return c >= '0' && c <= '9'; }
static inline int p8f_isalnum(int c) { return p8f_isalpha(c) || p8f_isdigit(c); }
static inline int p8f_isspace(int c) { return c == ' ' || (c >= 9 && c <= 13); }
static inline int p8f_ispunct(int c) { return c > 32 && c < 127 && !p8f_isalnum(c); }
static inline int p8f_tolower(int c) { return p8f_isupper(c) ? c + 32 : c; }
static inline int p8f_toupper(int c) { return p8f_islower(c) ? c - 32 : c; }
#define isalpha(c) p8f_isalpha(c)
#define ispunct(c) p8f_ispunct(c)
#define isspace(c) p8f_isspace(c)
#define isdigit(c) p8f_isdigit(c)
#define isupper(c) p8f_isupper(c)
#define islower(c) p8f_islower(c)
#define isalnum(c) p8f_isalnum(c)
#define tolower(c) p8f_tolower(c)
#define toupper(c) p8f_toupper(c)
#endif
