/* orc_alloc.h -- TEST INFRASTRUCTURE ONLY (force-included into every oracle/*.c by the Makefile).
 * The oracle's models are plain C structs built from dozens of calloc()s with no destructor (they mirror reference objects that
 * live as long as the process). A test that builds several of them would keep every table until pytest exits, so allocations are
 * tracked: between orc_scope_begin() and orc_scope_pause() / _end() every block the oracle allocates is tagged, and orc_scope_free(tag)
 * / _end() frees the tagged blocks that are still live. Outside a scope nothing changes (blocks are freed only by free()). */
#ifndef ORC_ALLOC_H
#define ORC_ALLOC_H
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
void* orc_t_calloc(size_t n, size_t s);
void* orc_t_malloc(size_t s);
void* orc_t_realloc(void* p, size_t s);
void orc_t_free(void* p);
uint32_t orc_scope_begin(void);      /* start tagging; returns the tag */
void orc_scope_pause(void);          /* stop tagging (blocks keep their tag) */
void orc_scope_free(uint32_t tag);   /* free the live blocks that carry the tag */
void orc_scope_end(void);            /* pause + free the current tag */
size_t orc_live_bytes(void);
#ifndef ORC_ALLOC_IMPL
#define calloc(n, s) orc_t_calloc((n), (s))
#define malloc(s) orc_t_malloc((s))
#define realloc(p, s) orc_t_realloc((p), (s))
#define free(p) orc_t_free((p))
#endif
#endif
