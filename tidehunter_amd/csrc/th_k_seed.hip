// th_k_seed.hip -- stage 0 (encoding) and stage 1 (k-mer self-hit collection) for gfx950.
//
// Replaces get_bseq (src/seq.c:77-86), direct_hash (src/tandem_hit.c:37-56),
// radix_sort_hash (src/ksort.h:101-151) and collect_hash_hit (src/tandem_hit.c:171-225).
//
// One workgroup per read.  The (key<<32|pos) words of the read are bitonic-sorted in LDS (reads with
// up to TH_SORT_LDS k-mers; 128 KiB of the CU's 160 KiB) or in an HBM scratch segment (longer reads).
// Each sorted word then walks back inside its equal-key group to the nearest earlier occurrence
// at distance >= min_p.  Because -w 1 emits at most one word per position, "sort hits by
// (end, period)" (src/tandem_hit.c:223) is an order-preserving compaction over positions, done with
// a workgroup prefix sum instead of a second sort.
#include "th_dev.h"

#define TH_SORT_LDS 16384
#define TH_SEED_BLOCK 256

__device__ __forceinline__ uint8_t th_code(uint8_t c)
{
    // src/seq.c:15-32: A/a 0, C/c 1, G/g 2, T/t 3, bytes 0..3 themselves, '-' 5, all else 4
    uint8_t l = c | 0x20;
    uint8_t r = 4;
    if (c <= 3) r = c;
    if (l == 'a') r = 0;
    if (l == 'c') r = 1;
    if (l == 'g') r = 2;
    if (l == 't') r = 3;
    if (c == '-') r = 5;
    return r;
}

__global__ void th_k_encode(const uint8_t *__restrict__ seq, uint8_t *__restrict__ bseq, long long total)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // 16 bases per thread per trip: one dwordx4 load and store per lane (coalesced 1 KiB per wave)
    long long nvec = total >> 4;
    const uint4 *s4 = (const uint4 *)seq; uint4 *d4 = (uint4 *)bseq;
    for (long long v = i; v < nvec; v += stride) {
        uint4 x = s4[v], y;
        uint32_t in[4] = { x.x, x.y, x.z, x.w }, out[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) o |= (uint32_t)th_code((uint8_t)(in[w] >> (8 * b))) << (8 * b);
            out[w] = o;
        }
        y.x = out[0]; y.y = out[1]; y.z = out[2]; y.w = out[3];
        d4[v] = y;
    }
    for (long long t = (nvec << 4) + i; t < total; t += stride) bseq[t] = th_code(seq[t]);
}

// hit_end/hit_per: per-read segments at roff[r]; tmp: per-position scratch (period or 0)
__global__ void __launch_bounds__(TH_SEED_BLOCK)
th_k_hits(const uint8_t *__restrict__ bseq, const uint32_t *__restrict__ roff, ThDevPara P,
          int *__restrict__ hit_end, int *__restrict__ hit_per, int *__restrict__ hit_n,
          uint32_t *__restrict__ tmp, unsigned long long *__restrict__ gsort, const long long *__restrict__ gsort_off)
{
    __shared__ unsigned long long lds[TH_SORT_LDS];
    __shared__ int s_cnt[TH_SEED_BLOCK + 1];
    const int tid = threadIdx.x, r = blockIdx.x;
    const uint32_t off = roff[r];
    const int L = (int)(roff[r + 1] - off), k = P.k;
    const uint8_t *b = bseq + off;
    const int ne = L - k + 1;                     // k-mer slots; slot e ends at position e+k-1
    if (ne < 1) { if (tid == 0) hit_n[r] = 0; return; }
    int N = 1; while (N < ne) N <<= 1;
    unsigned long long *buf = N <= TH_SORT_LDS ? lds : gsort + gsort_off[r];

    // 1. words: key<<32 | pos of the k-mer's LAST base; any code >= 4 in the window kills the word
    for (int e = tid; e < N; e += TH_SEED_BLOCK) {
        unsigned long long v = ~0ull;
        if (e < ne) {
            uint32_t key = 0; bool ok = true;
            for (int x = 0; x < k; ++x) { uint8_t c = b[e + x]; ok = ok && c < 4; key = key << 2 | (c & 3); }
            if (ok) v = (unsigned long long)key << 32 | (uint32_t)(e + k - 1);
        }
        buf[e] = v;
    }
    for (int p = tid; p < L; p += TH_SEED_BLOCK) tmp[off + p] = 0;
    __syncthreads();

    // 2. bitonic sort, ascending on the full 64-bit word (key major, pos minor); sentinels sink to the end
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (N >> 1); t += TH_SEED_BLOCK) {
                int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                bool asc = (lo & size) == 0;
                unsigned long long a = buf[lo], c = buf[hi];
                if ((a > c) == asc) { buf[lo] = c; buf[hi] = a; }
            }
            __syncthreads();
        }

    // 3. nearest earlier occurrence at distance >= min_p (src/tandem_hit.c:192-196)
    for (int x = tid; x < ne; x += TH_SEED_BLOCK) {
        unsigned long long v = buf[x];
        if (v == ~0ull) continue;
        uint32_t key = (uint32_t)(v >> 32), pos = (uint32_t)v, per = 0;
        for (int y = x - 1; y >= 0; --y) {
            unsigned long long w = buf[y];
            if ((uint32_t)(w >> 32) != key) break;
            uint32_t p = pos - (uint32_t)w;
            if (p >= P.min_p) { per = p; break; }
        }
        if (per > P.max_p) per = 0;
        tmp[off + pos] = per;
    }
    __syncthreads();

    // 4. order-preserving compaction over positions == ascending (end, period)
    const int chunk = (L + TH_SEED_BLOCK - 1) / TH_SEED_BLOCK;
    const int p0 = min(L, tid * chunk), p1 = min(L, p0 + chunk);
    int c = 0;
    for (int p = p0; p < p1; ++p) c += tmp[off + p] != 0;
    s_cnt[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < TH_SEED_BLOCK; ++t) { int x = s_cnt[t]; s_cnt[t] = acc; acc += x; }
        s_cnt[TH_SEED_BLOCK] = acc;
        hit_n[r] = acc;
    }
    __syncthreads();
    int o = s_cnt[tid];
    for (int p = p0; p < p1; ++p) {
        uint32_t per = tmp[off + p];
        if (per) { hit_end[off + o] = p; hit_per[off + o] = (int)per; ++o; }
    }
}
