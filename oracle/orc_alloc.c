/* orc_alloc.c -- TEST INFRASTRUCTURE ONLY: the oracle's tracked allocator (see orc_alloc.h). */
#define ORC_ALLOC_IMPL
#include "orc_alloc.h"
#include <stdint.h>
/* (the Makefile force-includes orc_alloc.h before this file's first line: here the names mean libc's) */
#undef calloc
#undef malloc
#undef realloc
#undef free

typedef struct Hdr {
  struct Hdr *prev, *next;
  size_t size;
  uint32_t scope, magic;
} Hdr; /* 32 bytes: the block behind it keeps malloc()'s 16-byte alignment */

#define ORC_MAGIC 0x0AC1EB10u
static Hdr g_head = {&g_head, &g_head, 0, 0, 0};
static uint32_t g_scope = 0, g_seq = 0;
static size_t g_live = 0;
static volatile int g_lock = 0;
static void lock(void) { while (__atomic_exchange_n(&g_lock, 1, __ATOMIC_ACQUIRE)) {} }
static void unlock(void) { __atomic_store_n(&g_lock, 0, __ATOMIC_RELEASE); }

static void* link_block(Hdr* h, size_t size) {
  if (!h) return NULL;
  h->size = size;
  h->magic = ORC_MAGIC;
  lock();
  h->scope = g_scope;
  h->next = g_head.next; h->prev = &g_head;
  g_head.next->prev = h; g_head.next = h;
  g_live += size;
  unlock();
  return (void*)(h + 1);
}
static void unlink_block(Hdr* h) {
  lock();
  h->prev->next = h->next; h->next->prev = h->prev;
  g_live -= h->size;
  unlock();
  h->magic = 0;
}
void* orc_t_calloc(size_t n, size_t s) {
  if (s && n > (SIZE_MAX - sizeof(Hdr)) / s) return NULL;
  return link_block((Hdr*)calloc(1, n * s + sizeof(Hdr)), n * s);   /* one calloc: large tables stay lazily zeroed pages */
}
void* orc_t_malloc(size_t s) { return link_block((Hdr*)malloc(s + sizeof(Hdr)), s); }
void orc_t_free(void* p) {
  if (!p) return;
  Hdr* h = (Hdr*)p - 1;
  if (h->magic != ORC_MAGIC) abort();   /* a pointer this allocator never returned (or freed twice): a bug in the oracle */
  unlink_block(h);
  free(h);
}
void* orc_t_realloc(void* p, size_t s) {
  if (!p) return orc_t_malloc(s);
  Hdr* h = (Hdr*)p - 1;
  if (h->magic != ORC_MAGIC) abort();
  const uint32_t scope = h->scope;
  unlink_block(h);
  Hdr* n = (Hdr*)realloc(h, s + sizeof(Hdr));
  if (!n) { link_block(h, h->size); h->scope = scope; return NULL; }
  void* r = link_block(n, s);
  n->scope = scope;
  return r;
}
uint32_t orc_scope_begin(void) {
  lock();
  if (!++g_seq) ++g_seq;
  g_scope = g_seq;
  const uint32_t tag = g_scope;
  unlock();
  return tag;
}
void orc_scope_pause(void) { lock(); g_scope = 0; unlock(); }
void orc_scope_free(uint32_t tag) {
  if (!tag) return;
  lock();
  if (g_scope == tag) g_scope = 0;
  Hdr* h = g_head.next;
  while (h != &g_head) {
    Hdr* nx = h->next;
    if (h->scope == tag) {
      h->prev->next = h->next; h->next->prev = h->prev;
      g_live -= h->size;
      h->magic = 0;
      free(h);
    }
    h = nx;
  }
  unlock();
}
void orc_scope_end(void) { orc_scope_free(g_scope); }
size_t orc_live_bytes(void) { return g_live; }
