// th_rt.h -- the thin runtime layer the pipeline talks to: HIP runtime calls, kernels launched on a HIP stream.
//
// (The test-only CPU build of the kernel sources, tests/cpu_emul/, pre-includes its own version of this layer --
// tests/cpu_emul/th_rt_emul.h defines TH_RT_H -- so that this product header carries no test code.)
#ifndef TH_RT_H
#define TH_RT_H
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
typedef hipStream_t th_stream_t;
#define TH_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[tidehunter_amd] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)
static inline int th_rt_malloc(void **p, size_t n) { TH_CHECK(hipMalloc(p, n ? n : 1)); return 0; }
static inline void th_rt_forget_error(void) { (void)hipGetLastError(); }      // after a failure that is handled (an allocation that was only tried): the next launch check must not see it
static inline int th_rt_free(void *p) { if (p) TH_CHECK(hipFree(p)); return 0; }
static inline int th_rt_h2d(void *d, const void *h, size_t n, th_stream_t s) { if (n) TH_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); return 0; }
static inline int th_rt_d2h(void *h, const void *d, size_t n, th_stream_t s) { if (n) TH_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); return 0; }
static inline int th_rt_memset(void *d, int v, size_t n, th_stream_t s) { if (n) TH_CHECK(hipMemsetAsync(d, v, n, s)); return 0; }
static inline int th_rt_sync(th_stream_t s) { TH_CHECK(hipStreamSynchronize(s)); return 0; }
static inline int th_rt_host_alloc(void **p, size_t n) { TH_CHECK(hipHostMalloc(p, n ? n : 1, hipHostMallocDefault)); return 0; }
static inline int th_rt_host_free(void *p) { if (p) TH_CHECK(hipHostFree(p)); return 0; }
typedef hipEvent_t th_event_t;
// Every event the host waits on is a BLOCKING one: hipEventSynchronize then sleeps on the completion signal's interrupt instead of
// spinning on it.  hipStreamSynchronize spins (with the default device schedule) for as long as the kernels run -- one core per
// pipeline thread, 16 of the 30 CPU-seconds round 3 spent per million reads -- so the pipeline never calls it on its hot path: it
// records such an event behind the work and waits for that (th_rt_sync_event).  TH_SPIN_SYNC=1 restores the spinning waits (A/B runs).
static inline bool th_rt_spin() { static const bool on = getenv("TH_SPIN_SYNC") && atoi(getenv("TH_SPIN_SYNC")); return on; }
static inline int th_rt_event_create(th_event_t *e) { TH_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming | (th_rt_spin() ? 0 : hipEventBlockingSync))); return 0; }
static inline int th_rt_event_record(th_event_t e, th_stream_t s) { TH_CHECK(hipEventRecord(e, s)); return 0; }
static inline int th_rt_event_wait(th_event_t e) { TH_CHECK(hipEventSynchronize(e)); return 0; }
static inline int th_rt_event_destroy(th_event_t e) { TH_CHECK(hipEventDestroy(e)); return 0; }
// drains stream s without spinning: a blocking event behind everything enqueued so far
static inline int th_rt_sync_event(th_stream_t s, th_event_t e) { if (th_rt_spin()) { TH_CHECK(hipStreamSynchronize(s)); return 0; } TH_CHECK(hipEventRecord(e, s)); TH_CHECK(hipEventSynchronize(e)); return 0; }
// TH_BLOCKING_DEVICE=1 (opt-in; the TideHunter-amd executable sets it for its own process): asks the runtime for blocking waits everywhere else too
// (hipStreamSynchronize in slow paths, hipFree, ...).  It is a per-device, PROCESS-WIDE scheduling flag -- it would also change how an embedding application
// (torch, another library) waits -- so the library never sets it by itself; the hot path does not need it: its three waits per batch
// are the blocking events above.  Only honoured before the device's context exists.
static inline void th_rt_prefer_blocking() { const char *e = getenv("TH_BLOCKING_DEVICE"); if (e && atoi(e) && !th_rt_spin()) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); } }
// a launch the runtime rejects (resource limits, wrong device, ...) must not pass silently: the enclosing function returns
// th_launch_failed(...) (an int, defined by the pipeline before its first launch)
#define TH_LAUNCH(kernel, grid, block, stream, ...) do { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__); \
        const hipError_t le_ = hipGetLastError(); if (le_ != hipSuccess) return th_launch_failed(#kernel, hipGetErrorString(le_)); } while (0)
#define TH_LANE() ((int)(threadIdx.x & 63))
static inline int th_rt_device_count(int *n) { return hipGetDeviceCount(n) == hipSuccess ? 0 : -1; }
static inline int th_rt_set_device(int d) { return hipSetDevice(d) == hipSuccess ? 0 : -1; }
static inline int th_rt_stream_create(th_stream_t *s) { TH_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); return 0; }
static inline int th_rt_stream_destroy(th_stream_t s) { TH_CHECK(hipStreamDestroy(s)); return 0; }
static inline int th_rt_stream_wait_event(th_stream_t s, th_event_t e) { TH_CHECK(hipStreamWaitEvent(s, e, 0)); return 0; }
static inline int th_rt_mem_info(size_t *fr, size_t *tot) { return hipMemGetInfo(fr, tot) == hipSuccess ? 0 : -1; }
// device time of a stretch of one stream: a pair of timing events recorded around it
struct th_timer_t { hipEvent_t a, b; };
static inline void th_rt_timer_begin(th_timer_t *t, th_stream_t s) { hipEventCreate(&t->a); hipEventCreate(&t->b); hipEventRecord(t->a, s); }
static inline void th_rt_timer_end(th_timer_t *t, th_stream_t s) { hipEventRecord(t->b, s); }
static inline float th_rt_timer_ms(th_timer_t *t) { float ms = 0; hipEventSynchronize(t->b); hipEventElapsedTime(&ms, t->a, t->b); return ms; }
static inline void th_rt_timer_destroy(th_timer_t *t) { hipEventDestroy(t->a); hipEventDestroy(t->b); }
#define TH_ARENA_MIN_BYTES ((size_t)96 << 20)      // batches that need less than this allocate buffer by buffer
#define TH_PERSISTENT_WAVES(dflt) (dflt)          // size of a persistent grid (the test build shrinks them)
#endif

// Host arrays that device copies read from or write to live in pinned memory: a copy between HBM and PAGEABLE host memory is staged
// by the runtime and moved by a blit kernel, which needs a compute unit -- and next to another stream's persistent grid (the two
// concurrent halves of a batch, the file driver's two workers) waits for one, up to a whole kernel (seen in the kernel trace: copies
// averaging 19 ms).  Pinned buffers go through the DMA engines.  std::vector with this allocator keeps the code unchanged; growth
// re-pins, so the vectors are members that keep their capacity from batch to batch.
#include <vector>
#include <new>
template <class T> struct th_pinned_alloc {
    typedef T value_type;
    th_pinned_alloc() {}
    template <class U> th_pinned_alloc(const th_pinned_alloc<U> &) {}
    static bool pin() { static const bool on = !(getenv("TH_NOPIN") && atoi(getenv("TH_NOPIN"))); return on; }     // TH_NOPIN=1: plain malloc (A/B measurements)
    T *allocate(size_t n) { void *p = nullptr; if (!pin()) { p = malloc(n * sizeof(T) + 1); if (!p) throw std::bad_alloc(); return (T *)p; }
                            if (th_rt_host_alloc(&p, n * sizeof(T)) != 0 || !p) throw std::bad_alloc(); return (T *)p; }
    void deallocate(T *p, size_t) { if (!pin()) free(p); else th_rt_host_free(p); }
    template <class U> bool operator==(const th_pinned_alloc<U> &) const { return true; }
    template <class U> bool operator!=(const th_pinned_alloc<U> &) const { return false; }
};
template <class T> using th_pvec = std::vector<T, th_pinned_alloc<T>>;
