// th_dev.h -- device-side common definitions and wave-level helpers (gfx950: 64-lane waves).
#pragma once
#include "th_rt.h"

#define TH_WAVE 64
#define TH_NEG (-0x20000000)

// Parameters the kernels read (a device-friendly projection of mini_tandem_para, src/tidehunter.h:47-61)
struct ThDevPara {
    int k;                       // k-mer length (-k)
    uint32_t min_p, max_p;       // period limits, cast as at src/tandem_hit.c:235
    double one_minus_div;        // 1 - max_div (src/partition.c:202,252: the one double product on the path)
    int min_copy;
    int match, mismatch, go1, ge1, go2, ge2;   // POA scoring (src/abpoa_cons.c:12-28)
};

// one accepted chain (chain_t, src/tandem_chain.h:30-34) as the extract kernel emits it
struct ThChainRec { int read, cell_off, len, est_start, est_period, first_end, last_start, pad; };

// one consensus job = one run of >= min_copy+1 boundaries between -1 separators (src/gen_cons.c:192-205)
struct ThTask {
    int read;        // read index in the batch
    int pos_off;     // first boundary of the run in the compact par_pos array
    int pos_n;       // boundaries in the run (units = pos_n - 1)
    int unit_off;    // first POA unit in the unit table
    int n_units;     // units that pass the filter at src/abpoa_cons.c:43 (n_seqs)
    int sum_len, max_len;
    int cons_off;    // consensus / coverage output offset
    int cons_cap;
    int aln_off;     // first entry of this task in the per-unit iden_n array
    int pad0, pad1;
};
struct ThUnit { int start, len; };   // start = index of the unit's first base inside the read

// Point where data written by some lanes of a wave (typically lane 0 after a sequential section) must be
// visible to the other lanes.  On the hardware the wave runs in lock-step, so program order plus a
// workgroup-scope fence is enough; the CPU emulation (tests only) runs lanes as independent fibers and
// needs a real rendezvous here.
#ifdef TH_CPU_EMUL
#define TH_WAVE_FENCE() do { (void)hipemul::wave_exchange(0); } while (0)
#else
#define TH_WAVE_FENCE() __threadfence_block()
#endif

// ---- wave helpers (every lane of the wave must call these) ----
__device__ __forceinline__ int th_wave_incl_max(int v)
{
    const int lane = TH_LANE();
#pragma unroll
    for (int d = 1; d < TH_WAVE; d <<= 1) { int o = __shfl_up(v, d); if (lane >= d) v = max(v, o); }
    return v;
}
__device__ __forceinline__ int th_wave_excl_max(int v, int seed)
{
    int inc = th_wave_incl_max(v);
    int e = __shfl_up(inc, 1);
    return TH_LANE() == 0 ? seed : max(e, seed);
}
__device__ __forceinline__ int th_wave_incl_sum(int v)
{
    const int lane = TH_LANE();
#pragma unroll
    for (int d = 1; d < TH_WAVE; d <<= 1) { int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}
__device__ __forceinline__ int th_wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int th_wave_min(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int th_wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ int th_ctz64(unsigned long long m) { return m ? __ffsll((long long)m) - 1 : 64; }
__device__ __forceinline__ int th_msb64(unsigned long long m) { return 63 - __clzll((long long)m); }   // m != 0
