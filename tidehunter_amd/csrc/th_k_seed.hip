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
#define TH_SEED_BLOCK 512

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

// total: a multiple of 16 (the upload pads the batch with 'N').  Besides the one-byte codes every later stage reads, the bases leave here
// 2-BIT PACKED (pk2: base a in bits 2 (a & 15) .. of word a >> 4) with a mask of the positions that are not A/C/G/T (nmask: bit a & 31 of
// word a >> 5): the k-mer words of th_k_hits are cut out of these streams, four dword loads per k-mer slot instead of k byte loads.
__global__ void th_k_encode(const uint8_t *__restrict__ seq, uint8_t *__restrict__ bseq, long long total, uint32_t *__restrict__ pk2, uint16_t *__restrict__ nmask)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // 16 bases per thread per trip: one dwordx4 load and store per lane (coalesced 1 KiB per wave)
    long long nvec = total >> 4;
    const uint4 *s4 = (const uint4 *)seq; uint4 *d4 = (uint4 *)bseq;
    for (long long v = i; v < nvec; v += stride) {
        uint4 x = s4[v], y;
        uint32_t in[4] = { x.x, x.y, x.z, x.w }, out[4], pk = 0, nm = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t c = th_code((uint8_t)(in[w] >> (8 * b)));
                o |= c << (8 * b); pk |= (c & 3u) << (2 * (4 * w + b)); nm |= (c >= 4u ? 1u : 0u) << (4 * w + b);
            }
            out[w] = o;
        }
        y.x = out[0]; y.y = out[1]; y.z = out[2]; y.w = out[3];
        d4[v] = y;
        pk2[v] = pk; nmask[v] = (uint16_t)nm;
    }
    if (i == 0) { pk2[nvec] = 0; pk2[nvec + 1] = 0; nmask[nvec] = 0xffff; nmask[nvec + 1] = 0xffff; nmask[nvec + 2] = 0xffff; nmask[nvec + 3] = 0xffff; }      // (the words a window may reach beyond the last base)
    for (long long t = (nvec << 4) + i; t < total; t += stride) bseq[t] = th_code(seq[t]);
}

// hit_end/hit_per: per-read segments at roff[r]; tmp: per-position scratch (period or 0)
// W = unsigned long long: words key<<32 | pos (any k <= 16, any read length; LDS up to 16384 k-mers, else HBM scratch).
// W = uint32_t:           words key<<PB | pos when every word of the batch fits 32 bits and every read fits the LDS buffer:
//                         PB 14 / CAP 16384 (k <= 9, reads < 16384 bases: 64 KiB, two workgroups per CU) or
//                         PB 15 / CAP 32768 (k <= 8, reads < 32768 bases: 128 KiB); half the LDS traffic of the 64-bit sort.
__device__ __forceinline__ bool th_env_bitonic(const ThDevPara &P) { return P.sort_bitonic != 0; }     // TH_SORT_BITONIC=1: the bitonic sort everywhere (A/B runs, the tests' cross-check)
template <typename W, int PB, int CAP>
__global__ void __launch_bounds__(TH_SEED_BLOCK)
th_k_hits(const uint8_t *__restrict__ bseq, const uint32_t *__restrict__ roff, ThDevPara P,
          int *__restrict__ hit_end, int *__restrict__ hit_per, int *__restrict__ hit_n,
          uint32_t *__restrict__ tmp, W *__restrict__ gsort, const long long *__restrict__ gsort_off,
          int *__restrict__ err, int *__restrict__ n_seq_fallback, const uint32_t *__restrict__ pk2, const uint32_t *__restrict__ nmask)
{
    __shared__ W lds[CAP];
    const W NONE_W = (W)~(W)0;
    __shared__ int s_cnt[TH_SEED_BLOCK + 1];
    __shared__ uint32_t s_hist[sizeof(W) == 4 ? TH_SEED_BLOCK / TH_WAVE : 1][TH_RDX_BINS + 1];     // the radix sort's digit counters (32-bit words only)
    __shared__ uint32_t s_wkey[256], s_wpos[256]; // -w > 1: the minimizer window (key, pos)
    __shared__ int s_runs[32];                    // -w > 1 with -H: lengths of the last k runs
    const int tid = threadIdx.x, r = blockIdx.x;
    const uint32_t off = roff[r];
    const int L = (int)(roff[r + 1] - off), k = P.k;
    const uint8_t *b = bseq + off;
    const int ne = L - k + 1;                     // k-mer slots; slot e ends at position e+k-1
    if (ne < 1 || L - P.w <= 0) { if (tid == 0) hit_n[r] = 0; return; }
    int N = 1; while (N < ne) N <<= 1;
    W *buf = N <= CAP ? lds : gsort + gsort_off[r];

    if (P.w > 1) {
        // 1m. minimizer words (src/tandem_hit.c:97-157).  The reference's selection is a sequential state machine (one step per
        // base: a w-slot ring of (key, position), a tracked minimum, emissions gated by the length l of the current A/C/G/T run).
        // Without -H it is replayed in parallel from three per-position facts: the tracked minimum after step t is always the
        // smallest key of the last w slots, newest slot on ties (the `<=` of the replacement and the `>=` of the rescan); a word is
        // written only when the minimum is replaced (l >= w+k), leaves the ring (l >= w+k-1; then also the ties of its successor),
        // when the first window of a run completes (l == w+k-1: the ties of the minimum), and once at the end of the read.  Every
        // thread evaluates those rules for its steps and marks the slots whose words the reference writes.  The sequential replay
        // stays below: it serves -H, any read on which a slot would be marked twice (the reference would then write a word twice;
        // not seen on any test input), and TH_MINIMIZER_SEQ=1 (P.w < 0 on entry), the tests' cross-check of the parallel rules.
        for (int e = tid; e < N; e += TH_SEED_BLOCK) buf[e] = NONE_W;
        const bool seq_only = P.hpc || P.w_seq;
        __shared__ int s_n, s_dup;
        if (tid == 0) { s_n = 0; s_dup = seq_only ? 1 : 0; }
        __syncthreads();
        if (!seq_only) {
            const int w = P.w, wk = w + k;
            const uint32_t NONE = 0xffffffffu;
            uint32_t *KEY = (uint32_t *)hit_end + off; int *LEN = hit_per + off; uint32_t *MARK = tmp + off;   // scratch until the sort: all three are rewritten below
            for (int t = tid; t < L; t += TH_SEED_BLOCK) {
                int l = 0; while (l < wk && t - l >= 0 && b[t - l] < 4) ++l;        // run length ending at t, capped at w + k
                uint32_t key = NONE;
                if (l >= k) { key = 0; for (int x = k - 1; x >= 0; --x) key = key << 2 | b[t - x]; }
                KEY[t] = key; LEN[t] = l; MARK[t] = 0;
            }
            __syncthreads();
            for (int t = tid; t < L; t += TH_SEED_BLOCK) {
                // minimum of the ring before this step (slots t-w .. t-1) and after it (t-w+1 .. t), newest slot on ties
                uint32_t pk = NONE, ck = NONE; int ps = -1, cs = -1;
                for (int x = max(0, t - w); x <= t; ++x) {
                    const uint32_t kx = KEY[x];
                    if (x < t && kx <= pk) { pk = kx; ps = x; }
                    if (x > t - w && kx <= ck) { ck = kx; cs = x; }
                }
                const int l = LEN[t]; const uint32_t kt = KEY[t];
                if (t >= 1) {
                    if (l == wk - 1 && pk != NONE)                                  // the run's first full window: ties of the minimum (:126-131)
                        for (int x = max(0, t - w + 1); x < t; ++x) if (KEY[x] == pk && x != ps) atomicAdd(&MARK[x], 1u);
                    if (kt <= pk) { if (l >= wk && pk != NONE) atomicAdd(&MARK[ps], 1u); }          // replaced (:133-135)
                    else if (ps == t - w) {                                                          // left the ring (:136-148)
                        if (l >= wk - 1 && pk != NONE) atomicAdd(&MARK[ps], 1u);
                        if (l >= wk - 1 && ck != NONE) for (int x = t - w + 1; x <= t; ++x) if (KEY[x] == ck && x != cs) atomicAdd(&MARK[x], 1u);
                    }
                }
                if (t == L - 1 && ck != NONE) atomicAdd(&MARK[cs], 1u);            // :151
            }
            __syncthreads();
            for (int t = tid; t < L; t += TH_SEED_BLOCK) {
                const uint32_t m = TH_LOAD_L2(&MARK[t]);
                if (m) { const int n = atomicAdd(&s_n, 1); if (n < N) buf[n] = (W)((W)KEY[t] << PB | (uint32_t)t); if (m > 1) s_dup = 1; }
            }
            __syncthreads();
            if (s_dup) { for (int e = tid; e < N; e += TH_SEED_BLOCK) buf[e] = NONE_W; if (tid == 0) atomicAdd(n_seq_fallback, 1); }
            __syncthreads();
        }
        if (tid == 0 && s_dup) {
            const int w = P.w;
            const uint32_t mask = k >= 16 ? 0xffffffffu : (1u << (2 * k)) - 1u, NONE = 0xffffffffu;
            uint32_t key = 0, sel_key = NONE, sel_pos = NONE;
            int r_head = 0, r_cnt = 0, span = 0, count = 0, at = 0, sel_at = 0, n = 0;
            auto put = [&](uint32_t kk, uint32_t pp) { if (n < N) buf[n] = (W)((W)kk << PB | pp); ++n; };
            auto sweep = [&](int lo, int hi) {
                for (int j = lo; j < hi; ++j) { const uint32_t sk = s_wkey[j], sp = s_wpos[j]; if (sk == sel_key && sp != sel_pos) put(sk, sp); }
            };
            for (int i = 0; i < L; ++i) {
                const int c = b[i];
                uint32_t ck = NONE, cp = NONE;
                if (c < 4) {
                    if (P.hpc) {
                        int run = 1;
                        while (i + run < L && b[i + run] == c) ++run;
                        i += run - 1;
                        s_runs[(r_head + r_cnt++) & 31] = run; span += run;
                        if (r_cnt > k) { span -= s_runs[r_head]; r_head = (r_head + 1) & 31; --r_cnt; }
                    } else span = count + 1 < k ? count + 1 : k;
                    key = (key << 2 | (uint32_t)c) & mask;
                    ++count;
                    if (count >= k && span < 256) { ck = key; cp = (uint32_t)i; }
                } else { count = 0; r_head = r_cnt = 0; span = 0; key = 0; }
                s_wkey[at] = ck; s_wpos[at] = cp;
                if (count == w + k - 1 && sel_key != NONE) { sweep(at + 1, w); sweep(0, at); }
                if (ck <= sel_key) {
                    if (count >= w + k && sel_key != NONE) put(sel_key, sel_pos);
                    sel_key = ck; sel_pos = cp; sel_at = at;
                } else if (at == sel_at) {
                    const bool open = count >= w + k - 1;
                    if (open && sel_key != NONE) put(sel_key, sel_pos);
                    sel_key = NONE;
                    for (int j = at + 1; j < w; ++j) { const uint32_t sk = s_wkey[j]; if (sel_key >= sk) { sel_key = sk; sel_pos = s_wpos[j]; sel_at = j; } }
                    for (int j = 0; j <= at; ++j) { const uint32_t sk = s_wkey[j]; if (sel_key >= sk) { sel_key = sk; sel_pos = s_wpos[j]; sel_at = j; } }
                    if (open && sel_key != NONE) { sweep(at + 1, w); sweep(0, at + 1); }
                }
                if (++at == w) at = 0;
            }
            if (sel_key != NONE) put(sel_key, sel_pos);
            if (n > N) atomicOr(err, 8);
        }
    } else
    // 1. words: key<<32 | pos of the k-mer's LAST base; any code >= 4 in the window kills the word
    for (int e = tid; e < N; e += TH_SEED_BLOCK) {
        W v = NONE_W;
        if (e < ne && !P.hpc && pk2) {
            // the k-mer of slot e cut out of the 2-bit stream (first base in the LOWEST bits: the reference shifts the first base to the top,
            // src/tandem_hit.c:44 -- the self-hits depend on which words are EQUAL, not on their order, so any one-to-one key serves), and
            // its k mask bits: any base that is not A/C/G/T kills the word (:41-43)
            const unsigned long long a = (unsigned long long)off + (unsigned)e;
            const uint32_t wi = (uint32_t)(a >> 4), sh = 2u * (uint32_t)(a & 15), mi = (uint32_t)(a >> 5), ms = (uint32_t)(a & 31);
            const unsigned long long win = ((unsigned long long)pk2[wi + 1] << 32 | pk2[wi]) >> sh;
            const unsigned long long mwin = ((unsigned long long)nmask[mi + 1] << 32 | nmask[mi]) >> ms;
            const uint32_t key = (uint32_t)win & (k >= 16 ? 0xffffffffu : (1u << (2 * k)) - 1u);
            if (((uint32_t)mwin & ((1u << k) - 1u)) == 0) v = (W)((W)key << PB | (uint32_t)(e + k - 1));
        } else if (e < ne && !P.hpc) {
            uint32_t key = 0; bool ok = true;
            for (int x = 0; x < k; ++x) { uint8_t c = b[e + x]; ok = ok && c < 4; key = key << 2 | (c & 3); }
            if (ok) v = (W)((W)key << PB | (uint32_t)(e + k - 1));
        } else if (e < ne) {
            // -H (src/tandem_hit.c:46): a homopolymer run counts as one symbol and the word sits on the run's last
            // base; it exists if the k most recent runs are all A/C/G/T (any other code resets the count)
            const int p = e + k - 1;
            const uint8_t c0 = b[p];
            bool ok = c0 < 4 && (p == L - 1 || b[p + 1] != c0);
            uint32_t key = 0; int x = p;
            for (int rn = 0; rn < k && ok; ++rn) {
                if (x < 0) { ok = false; break; }
                const uint8_t cc = b[x];
                if (cc >= 4) { ok = false; break; }
                key |= (uint32_t)cc << (2 * rn);
                while (x >= 0 && b[x] == cc) --x;
            }
            if (ok) v = (W)((W)key << PB | (uint32_t)p);
        }
        buf[e] = v;
    }
    for (int p = tid; p < L; p += TH_SEED_BLOCK) tmp[off + p] = 0;
    __syncthreads();

    // 2. sort, ascending on the full word (key major, pos minor); sentinels sink to the end.  The 32-bit words of step 1 were written in
    // position order, so a STABLE sort by the key field is the full sort: two radix passes (k = 8) instead of the bitonic network's 33
    // passes over the buffer (round 4; 34 -> ms per 100 000 headline reads in profiles/r04_*).  The minimizer words (-w > 1) arrive in no
    // particular order, 64-bit words and reads beyond 32 words per lane keep the bitonic sort.
    if (sizeof(W) == 4 && P.w == 1 && N <= TH_RDX_STEPS * TH_SEED_BLOCK && N >= TH_SEED_BLOCK && buf == (W *)lds && !th_env_bitonic(P))
        th_radix_sort_field<TH_SEED_BLOCK>((uint32_t *)lds, N, PB, 2 * k, 0xffffffffu, s_hist, s_cnt, tid);
    else
        th_bitonic_sort<W, false, TH_SEED_BLOCK>(buf, N, tid);

    // 3. nearest earlier occurrence at distance >= min_p (src/tandem_hit.c:192-196)
    for (int x = tid; x < ne; x += TH_SEED_BLOCK) {
        const W v = buf[x];
        if (v == NONE_W) continue;
        const W PM = ((W)1 << PB) - 1;
        uint32_t key = (uint32_t)(v >> PB), pos = (uint32_t)(v & PM), per = 0;
        // one word per position is what makes step 4 a compaction; the minimizer replay never repeats a word in
        // practice (tests/test_oracle_golden.py sweeps for it), but fail loudly rather than drop a hit if it does
        if (x > 0 && buf[x - 1] == v) atomicOr(err, 8);
        for (int y = x - 1; y >= 0; --y) {
            const W w = buf[y];
            if ((uint32_t)(w >> PB) != key) break;
            uint32_t p = pos - (uint32_t)(w & PM);
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
    // exclusive prefix over the threads' counts: DPP scan inside each wave, then the (few) wave totals
    const int inc = th_wave_incl_sum(c);
    if ((tid & 63) == 63) s_cnt[tid >> 6] = inc;
    __syncthreads();
    int o = inc - c, tot = 0;
    for (int wv = 0; wv < TH_SEED_BLOCK / TH_WAVE; ++wv) { const int x = s_cnt[wv]; if (wv < (tid >> 6)) o += x; tot += x; }
    if (tid == 0) hit_n[r] = tot;
    for (int p = p0; p < p1; ++p) {
        uint32_t per = tmp[off + p];
        if (per) { hit_end[off + o] = p; hit_per[off + o] = (int)per; ++o; }
    }
}
