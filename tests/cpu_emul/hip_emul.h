// tests/cpu_emul/hip_emul.h -- TEST-ONLY lock-step emulation of the HIP execution model on the CPU.
//
// The product (tidehunter_amd/libtidehunter_amd.so) is HIP-only and never includes this file.  The
// test build (tests/cpu_emul/Makefile; it pre-includes th_hw_emul.h / th_rt_emul.h in place of the product's th_hw.h / th_rt.h) compiles the SAME kernel sources with g++ so that
// `pytest -m "not gpu"` and the sanitizers can exercise kernel logic where no GPU exists.
//
// Model: one OS thread; every HIP thread of a block is a ucontext fiber; a wave is 64 consecutive
// threadIdx.x.  Cross-lane operations (__shfl*, __ballot, ...) and __syncthreads() are rendezvous
// points: a fiber deposits its value and yields until its whole wave (or block) has arrived.
// Requirement on kernels: every lane of a wave reaches every cross-lane operation (wave-uniform
// control flow around them) and no lane returns before the last such operation -- which is also what
// makes them well defined on the hardware.
#pragma once
#include <ucontext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>
#include <algorithm>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3_e { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };

namespace hipemul {
constexpr int WAVE = 64;
struct WaveSync { uint64_t slot[2][WAVE]; long arrive[2]; };
struct Fiber { ucontext_t ctx; char *stack; bool done; long n_wave_ops; long n_block_ops; };
struct State {
    ucontext_t sched;
    std::vector<Fiber> fib;
    std::vector<WaveSync> wave;
    long block_arrive[2];
    int cur, nthreads;
    std::function<void()> body;
};
inline State &S() { static State s; return s; }
extern "C" inline void fiber_entry() { State &s = S(); s.body(); s.fib[s.cur].done = true; swapcontext(&s.fib[s.cur].ctx, &s.sched); }
}  // namespace hipemul

static uint3_e threadIdx, blockIdx;
static dim3 blockDim, gridDim;

namespace hipemul {
inline void yield() { State &s = S(); int me = s.cur; swapcontext(&s.fib[me].ctx, &s.sched); }
inline int wave_size_of(int tid) { State &s = S(); int w0 = tid / WAVE * WAVE; return std::min(WAVE, s.nthreads - w0); }
// deposit v, wait for the wave, return pointer to the wave's slots for this op
inline const uint64_t *wave_exchange(uint64_t v)
{
    State &s = S(); int tid = s.cur, w = tid / WAVE, lane = tid % WAVE;
    Fiber &f = s.fib[tid]; int b = (int)(f.n_wave_ops & 1);
    WaveSync &ws = s.wave[w];
    ws.slot[b][lane] = v; ++ws.arrive[b];
    long need = (long)wave_size_of(tid) * (f.n_wave_ops / 2 + 1);
    while (ws.arrive[b] < need) yield();
    ++f.n_wave_ops;
    return ws.slot[b];
}
inline void block_barrier()
{
    State &s = S(); Fiber &f = s.fib[s.cur]; int b = (int)(f.n_block_ops & 1);
    ++s.block_arrive[b];
    long need = (long)s.nthreads * (f.n_block_ops / 2 + 1);
    while (s.block_arrive[b] < need) yield();
    ++f.n_block_ops;
}
// the emulation keeps one global scheduler state (and `__shared__` is a plain static): launches from several host threads
// (two device workers of the file driver) are serialised here
inline std::mutex &launch_mutex() { static std::mutex m; return m; }
template <class F> void launch(dim3 grid, dim3 block, F body)
{
    std::lock_guard<std::mutex> lock(launch_mutex());
    State &s = S();
    int nt = (int)(block.x * block.y * block.z);
    const size_t STK = 256 * 1024;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.nthreads = nt; s.fib.assign(nt, Fiber()); s.wave.assign((nt + WAVE - 1) / WAVE, WaveSync());
        for (auto &w : s.wave) memset(&w, 0, sizeof(w));
        s.block_arrive[0] = s.block_arrive[1] = 0;
        s.body = body;
        gridDim = grid; blockDim = block; blockIdx = { bx, by, bz };
        for (int t = 0; t < nt; ++t) {
            Fiber &f = s.fib[t];
            f.stack = (char *)malloc(STK); f.done = false; f.n_wave_ops = f.n_block_ops = 0;
            getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STK; f.ctx.uc_link = &s.sched;
            makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        }
        int alive = nt; long spins = 0;
        while (alive > 0) {
            for (int t = 0; t < nt; ++t) {
                if (s.fib[t].done) continue;
                s.cur = t;
                threadIdx = { (unsigned)(t % block.x), (unsigned)(t / block.x % block.y), (unsigned)(t / (block.x * block.y)) };
                swapcontext(&s.sched, &s.fib[t].ctx);
                if (s.fib[t].done) --alive;
            }
            if (++spins > 2000000000L) { fprintf(stderr, "[hip_emul] deadlock?\n"); abort(); }
        }
        for (int t = 0; t < nt; ++t) free(s.fib[t].stack);
    }
}
}  // namespace hipemul

// every read of threadIdx after a yield must see this fiber's value: the scheduler restores it on resume.
static inline void __syncthreads() { hipemul::block_barrier(); }
static inline int __lane_id_e() { return hipemul::S().cur % hipemul::WAVE; }

template <class T> static inline T __shfl(T v, int src, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shfl");
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    const uint64_t *sl = hipemul::wave_exchange(u);
    int lane = __lane_id_e(), base = lane / width * width;
    int s = base + ((src % width + width) % width);
    int ws = hipemul::wave_size_of(hipemul::S().cur);
    if (s >= ws) s = lane;
    T r; memcpy(&r, &sl[s], sizeof(T)); return r;
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    const uint64_t *sl = hipemul::wave_exchange(u);
    int lane = __lane_id_e(), base = lane / width * width, s = lane - (int)d;
    if (s < base) s = lane;
    T r; memcpy(&r, &sl[s], sizeof(T)); return r;
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    const uint64_t *sl = hipemul::wave_exchange(u);
    int lane = __lane_id_e(), base = lane / width * width, s = lane + (int)d;
    if (s >= base + width || s >= hipemul::wave_size_of(hipemul::S().cur)) s = lane;
    T r; memcpy(&r, &sl[s], sizeof(T)); return r;
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64)
{
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    const uint64_t *sl = hipemul::wave_exchange(u);
    int lane = __lane_id_e(), s = lane ^ m;
    if (s / width != lane / width || s >= hipemul::wave_size_of(hipemul::S().cur)) s = lane;
    T r; memcpy(&r, &sl[s], sizeof(T)); return r;
}
static inline unsigned long long __ballot(int p)
{
    const uint64_t *sl = hipemul::wave_exchange(p ? 1 : 0);
    unsigned long long m = 0; int ws = hipemul::wave_size_of(hipemul::S().cur);
    for (int i = 0; i < ws; ++i) if (sl[i]) m |= 1ull << i;
    return m;
}
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { int ws = hipemul::wave_size_of(hipemul::S().cur); unsigned long long full = ws == 64 ? ~0ull : ((1ull << ws) - 1); return __ballot(p) == full; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }

template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
using std::max;
using std::min;
