// tests/cpu_emul/th_rt_emul.h -- TEST-ONLY stand-in for tidehunter_amd/csrc/th_rt.h: the same runtime layer over malloc / memcpy
// and the fiber emulation of hip_emul.h.  Pre-included by tests/cpu_emul/Makefile (-include); it defines TH_RT_H, so the product
// header's HIP half is skipped.  Never part of libtidehunter_amd.so.
#ifndef TH_RT_H
#define TH_RT_H
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "hip_emul.h"
typedef int th_stream_t;
#define TH_CHECK(x) (x)
static inline int th_rt_malloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : -1; }
static inline void th_rt_forget_error(void) { }
static inline int th_rt_free(void *p) { free(p); return 0; }
static inline int th_rt_h2d(void *d, const void *h, size_t n, th_stream_t) { memcpy(d, h, n); return 0; }
static inline int th_rt_d2h(void *h, const void *d, size_t n, th_stream_t) { memcpy(h, d, n); return 0; }
static inline int th_rt_memset(void *d, int v, size_t n, th_stream_t) { memset(d, v, n); return 0; }
static inline int th_rt_sync(th_stream_t) { return 0; }
static inline int th_rt_host_alloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : -1; }
static inline int th_rt_host_free(void *p) { free(p); return 0; }
typedef int th_event_t;
static inline int th_rt_event_create(th_event_t *e) { *e = 0; return 0; }
static inline int th_rt_event_record(th_event_t, th_stream_t) { return 0; }
static inline int th_rt_event_wait(th_event_t) { return 0; }
static inline int th_rt_event_destroy(th_event_t) { return 0; }
static inline int th_rt_sync_event(th_stream_t, th_event_t) { return 0; }
static inline void th_rt_prefer_blocking() {}
#define TH_LAUNCH(kernel, grid, block, stream, ...) hipemul::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })
#define TH_LANE() ((int)(threadIdx.x & 63))
static inline int th_rt_device_count(int *n) { *n = 1; return 0; }
static inline int th_rt_set_device(int) { return 0; }
static inline int th_rt_stream_create(th_stream_t *s) { *s = 0; return 0; }
static inline int th_rt_stream_destroy(th_stream_t) { return 0; }
static inline int th_rt_stream_wait_event(th_stream_t, th_event_t) { return 0; }
static inline int th_rt_mem_info(size_t *fr, size_t *tot) { *fr = *tot = (size_t)1 << 40; return 0; }
struct th_timer_t { int x; };
static inline void th_rt_timer_begin(th_timer_t *, th_stream_t) {}
static inline void th_rt_timer_end(th_timer_t *, th_stream_t) {}
static inline float th_rt_timer_ms(th_timer_t *) { return 0.f; }
static inline void th_rt_timer_destroy(th_timer_t *) {}
#define TH_ARENA_MIN_BYTES ((size_t)0)
#define TH_PERSISTENT_WAVES(dflt) ((dflt) > 4096 ? 3 : 2)
#endif
