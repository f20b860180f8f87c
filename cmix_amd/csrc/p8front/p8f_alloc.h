/* p8f_alloc.h -- force-included into the front end's C files: every calloc / malloc of a front end is tracked, so
 * p8f_front_free() releases a stream's models in one sweep (the models are plain C structs with no destructors). */
#ifndef CMX_P8F_ALLOC_H
#define CMX_P8F_ALLOC_H
#include <stdlib.h>
void* p8f_tracked_calloc(size_t n, size_t size);
#define calloc(n, s) p8f_tracked_calloc((n), (s))
#define malloc(s) p8f_tracked_calloc(1, (s))
#endif
