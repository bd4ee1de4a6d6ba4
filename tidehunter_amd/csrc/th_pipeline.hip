// th_pipeline.hip -- host-side orchestration of the device path for one batch of reads.
//
// This is what stands behind the fan-out of mini_tandem() (src/main.c:402-425): instead of
// n_thread pthreads each calling tidehunter_core() per read, the whole batch goes through
//   encode -> self-hits -> chain DP -> chain extraction -> partition -> [host: split runs]
//   -> POA consensus -> unit/consensus alignments -> small result records back to the host.
// All kernels of one batch run on one HIP stream; the host only synchronises where a later launch
// needs a count produced by an earlier one.
#include <vector>
#include <string>
#include <algorithm>
#include <chrono>
#include "th_internal.h"

static double th_wall_s() { using namespace std::chrono; return duration<double>(steady_clock::now().time_since_epoch()).count(); }

// one message per thread: th_run_file_multi drives one handle per device from its own thread
static thread_local std::string g_th_err;
extern "C" const char *th_gpu_error(void) { return g_th_err.c_str(); }
static int th_fail(const std::string &m) { g_th_err = m; fprintf(stderr, "[tidehunter_amd] %s\n", m.c_str()); return -1; }
static int th_launch_failed(const char *kernel, const char *why) { return th_fail(std::string("kernel launch failed: ") + kernel + ": " + why); }

#include "th_dev.h"
#include "th_k_seed.hip"
#include "th_k_chain.hip"
#include "th_k_partition.hip"
#include "th_k_poa.hip"
#include "th_k_align.hip"
#include "th_k_adapter.hip"

#define TH_STAGE_BYTES (16u << 20)
#define TH_POA_SCRATCH_BYTES (96LL << 30)       // upper limit of the POA scratch of one handle (HBM: 288 GB; two handles may share a device)
// seconds this thread spent in device malloc / free and the bytes it asked for (TH_TRACE prints them per run)
static thread_local double g_alloc_s = 0, g_free_s = 0; static thread_local long long g_alloc_bytes = 0; static thread_local int g_alloc_n = 0;
// One device allocation per handle, made when the handle first learns its batch size (th_gpu_reserve / the first upload), from
// which the buffers below are carved.  hipMalloc while ANOTHER stream of the process is running kernels waits for a kernel
// boundary each time (measured with two workers on one device: 28 calls took 5.9 s next to persistent kernels of 0.1-0.6 s each,
// 0.009 s on an idle device), so a worker must not allocate piecemeal while its neighbour computes.  A buffer that outgrows the
// arena falls back to its own hipMalloc.
struct Arena { char *base = nullptr; size_t cap = 0, top = 0; };
static thread_local Arena *g_arena = nullptr;
// a pinned vector is grown with a quarter of slack: re-pinning (hipHostFree + hipHostMalloc) costs milliseconds and must not recur as
// batches of slightly different sizes follow each other
template <class V> static inline void th_room(V &v, size_t n) { if (v.capacity() < n) v.reserve(n + n / 4 + 64); }

template <class T> struct DBuf {
    T *p = nullptr; size_t cap = 0; bool in_arena = false;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        double t0 = th_wall_s();
        if (p && !in_arena) th_rt_free(p);          // (an arena region that is outgrown is simply abandoned)
        g_free_s += th_wall_s() - t0;
        p = nullptr; cap = 0; in_arena = false;
        size_t want = n + n / 8 + 64;
        const size_t bytes = (want * sizeof(T) + 255) & ~(size_t)255;
        if (g_arena && g_arena->base && g_arena->top + bytes <= g_arena->cap) {
            p = (T *)(g_arena->base + g_arena->top); g_arena->top += bytes; in_arena = true; cap = want; return 0;
        }
        t0 = th_wall_s();
        if (th_rt_malloc((void **)&p, want * sizeof(T))) return -1;
        g_alloc_s += th_wall_s() - t0; g_alloc_bytes += (long long)(want * sizeof(T)); ++g_alloc_n;
        cap = want; return 0;
    }
    void release() { if (p && !in_arena) th_rt_free(p); p = nullptr; cap = 0; in_arena = false; }
};

#ifndef TH_CPU_EMUL
struct Ev { hipEvent_t a, b; };
#endif

struct th_gpu {
    th_gpu_para_t para; ThDevPara dp; int device; th_stream_t stream;
    int n = 0; long long T = 0; int max_len = 0;
    uint8_t *h_stage[2] = { nullptr, nullptr }; th_event_t stage_ev[2]; bool stage_ok = false;   // pinned staging for the upload (2 x TH_STAGE_BYTES)
    th_pvec<uint32_t> h_roff; th_pvec<long long> h_gsort_off; long long gsort_total = 0;
    DBuf<uint8_t> d_seq, d_bseq, d_tracked, d_cons;
    DBuf<uint32_t> d_roff, d_tmp;
    DBuf<uint16_t> d_xorder; DBuf<int> d_xstart, d_xend, d_xovf;
    DBuf<uint8_t> d_cons_c; DBuf<int> d_cov_c, d_cc_len, d_cc_off;   // compacted results   // candidates in extraction order (th_k_chain_sort)
    DBuf<int> d_hit_end, d_hit_per, d_hit_n, d_score, d_from, d_cells, d_cnt, d_read_ch_base, d_read_ch_n;
    DBuf<unsigned long long> d_gsort, d_eps; DBuf<long long> d_gsort_off;
    DBuf<ThChainRec> d_rec;
    DBuf<int> d_par, d_par_off, d_par_n, d_par_dst, d_parc, d_bnd, d_scratch, d_cov, d_cons_len, d_status, d_iden, d_ext;
    DBuf<ThTask> d_tasks, d_sub; DBuf<ThUnit> d_units; DBuf<ThAlnItem> d_items; DBuf<int> d_scl, d_sst;   // d_sub/d_scl/d_sst: task subsets of POA relaunches
    int n_rec = 0, reserve_n = 0; Arena arena;
    bool ad_on = false; int ad_min_cons = 0; ThAdPara h_ad; DBuf<ThAdPara> d_adp; DBuf<int> d_ad; th_pvec<int> h_ad_out; bool ad_valid = false;
    th_pvec<ThChainRec> h_rec; th_pvec<int> h_par_n, h_par_dst, h_parc, h_read_ch_base, h_read_ch_n;      // pinned: see th_rt.h
    th_pvec<ThTask> h_tasks, h_sub; th_pvec<ThUnit> h_units; th_pvec<ThAlnItem> h_items;
    th_pvec<int> h_cons_len, h_status, h_iden, h_ext, h_cov, h_cl, h_st, h_cc_off, h_cnt; th_pvec<uint8_t> h_cons;
    std::vector<th_job_t> jobs;
    double ms_kernels = 0, ms_dominant = 0, ms_k[8] = { 0 };
    int aln_wdiv = 10, aln_wadd = 24;     // first certified band of the unit alignments: half-width cons_len / wdiv + wadd
    int aln_xpct = 36;                    // flank extensions: band half-width requested from the anti-diagonal engine, percent of min(consensus, flank) (+ 8)
    int aln_wtol = 16, aln_mode = 1;      // anti-diagonal engine: accepted shortfall of its band against that half-width; 0 = strip engine only, 2 = verify against the full matrix
    int waves_part = 8192, waves_poa = 6144, waves_aln = 8192, waves_xw = 2048;   // persistent grids: 256 CUs x 4 SIMDs x resident waves
};

static int env_int(const char *name, int dflt) { const char *s = getenv(name); return s && *s ? atoi(s) : dflt; }

extern "C" th_gpu_t *th_gpu_create(const th_gpu_para_t *p, int device)
{
    th_gpu *g = new th_gpu();
    g->para = *p;
    g->dp.k = p->k; g->dp.w = p->w; g->dp.w_seq = env_int("TH_MINIMIZER_SEQ", 0) ? 1 : 0; g->dp.hpc = p->hpc; g->dp.min_p = p->min_p; g->dp.max_p = p->max_p; g->dp.one_minus_div = p->one_minus_div; g->dp.min_copy = p->min_copy;
    g->dp.match = p->match; g->dp.mismatch = p->mismatch; g->dp.go1 = p->go1; g->dp.ge1 = p->ge1; g->dp.go2 = p->go2; g->dp.ge2 = p->ge2;
    g->device = device;
    // adapter search tables (th_k_adapter.hip): per consensus letter the query positions it matches, forward and reversed query
    if (p->ad_seq[0] && p->ad_seq[1] && p->ad_seq[2] && p->ad_seq[3]) {
        g->ad_on = true; memset(&g->h_ad, 0, sizeof(g->h_ad));
        auto eq = [](char a, char b) { if (a == b) return true; if ((a ^ b) != 0x20) return false; const char x = a | 0x20; return x == 'a' || x == 'c' || x == 'g' || x == 't' || x == 'n'; };
        for (int sr = 0; sr < 4; ++sr) {
            const int m = p->ad_len[sr];
            if (m > 64 * TH_AD_MAXB) { g->ad_on = false; break; }
            g->h_ad.len[sr] = m; g->h_ad.k[sr] = p->ad_k[sr]; g->h_ad.nb[sr] = (m + 63) / 64;
            for (int c = 0; c < 5; ++c)
                for (int i = 0; i < m; ++i) {
                    if (eq(p->ad_seq[sr][i], "ACGTN"[c])) g->h_ad.peq[sr][0][c][i / 64] |= 1ull << (i & 63);
                    if (eq(p->ad_seq[sr][m - 1 - i], "ACGTN"[c])) g->h_ad.peq[sr][1][c][i / 64] |= 1ull << (i & 63);
                }
        }
        g->ad_min_cons = p->ad_len[0] + p->ad_len[3];
    }
    g->para.ad_seq[0] = g->para.ad_seq[1] = g->para.ad_seq[2] = g->para.ad_seq[3] = nullptr;     // (borrowed strings: not kept)
#ifdef TH_CPU_EMUL
    g->stream = 0;
    g->waves_part = 2; g->waves_poa = 2; g->waves_aln = 3; g->waves_xw = 2;
#else
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { th_fail("no HIP device: the TideHunter HIP path needs an AMD GPU (no CPU fallback)"); delete g; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { th_fail("hipSetDevice failed"); delete g; return nullptr; }
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { th_fail("hipStreamCreate failed"); delete g; return nullptr; }
#endif
    g->waves_part = env_int("TH_WAVES_PART", g->waves_part);
    g->waves_poa = env_int("TH_WAVES_POA", g->waves_poa);
    g->waves_aln = env_int("TH_WAVES_ALN", g->waves_aln);
    g->waves_xw = env_int("TH_WAVES_XW", g->waves_xw);
    g->aln_wdiv = std::max(1, env_int("TH_ALN_WDIV", g->aln_wdiv)); g->aln_wadd = std::max(0, env_int("TH_ALN_WADD", g->aln_wadd));
    g->aln_wtol = std::max(0, env_int("TH_ALN_WTOL", g->aln_wtol)); g->aln_mode = env_int("TH_ALN_MODE", g->aln_mode); g->aln_xpct = std::max(0, env_int("TH_ALN_XPCT", g->aln_xpct));
    return g;
}

extern "C" void th_gpu_destroy(th_gpu_t *g)
{
    if (!g) return;
#ifndef TH_CPU_EMUL
    (void)hipSetDevice(g->device);
#endif
    g_arena = nullptr;
    g->d_seq.release(); g->d_bseq.release(); g->d_tracked.release(); g->d_sub.release(); g->d_scl.release(); g->d_sst.release(); g->d_cons_c.release(); g->d_cov_c.release(); g->d_cc_len.release(); g->d_cc_off.release(); g->d_eps.release(); g->d_xovf.release(); g->d_xorder.release(); g->d_xstart.release(); g->d_xend.release(); g->d_cons.release(); g->d_roff.release(); g->d_tmp.release();
    g->d_hit_end.release(); g->d_hit_per.release(); g->d_hit_n.release(); g->d_score.release(); g->d_from.release(); g->d_cells.release();
    g->d_cnt.release(); g->d_read_ch_base.release(); g->d_read_ch_n.release(); g->d_gsort.release(); g->d_gsort_off.release(); g->d_rec.release();
    g->d_par.release(); g->d_par_off.release(); g->d_par_n.release(); g->d_par_dst.release(); g->d_parc.release(); g->d_bnd.release(); g->d_scratch.release();
    g->d_cov.release(); g->d_cons_len.release(); g->d_status.release(); g->d_iden.release(); g->d_ext.release();
    g->d_tasks.release(); g->d_units.release(); g->d_items.release(); g->d_adp.release(); g->d_ad.release();
    if (g->stage_ok) for (int b = 0; b < 2; ++b) { th_rt_host_free(g->h_stage[b]); th_rt_event_destroy(g->stage_ev[b]); }
    if (g->arena.base) th_rt_free(g->arena.base);
#ifndef TH_CPU_EMUL
    hipStreamDestroy(g->stream);
#endif
    delete g;
}

#define TRY(x) do { if ((x) != 0) return th_fail(std::string("device runtime call failed: ") + #x); } while (0)

// A handle belongs to one device, but the calling thread's current HIP device may have changed since th_gpu_create
// (torch.cuda.set_device, a second handle for another GPU, a call from another thread): every entry point re-selects it,
// otherwise allocations and launches would land on the wrong GPU.
static int th_select_device(const th_gpu *g)
{
#ifndef TH_CPU_EMUL
    if (hipSetDevice(g->device) != hipSuccess) return th_fail("hipSetDevice(" + std::to_string(g->device) + ") failed");
#endif
    g_arena = const_cast<Arena *>(&g->arena);
    return 0;
}

// The handle's arena, sized for batches of up to n reads / T bases: ~48 bytes per base of per-position arrays, the POA scratch of
// the full persistent grid at ~100 ints per base of the average read (3.7 MB per wave at the headline shape; long-period batches
// need more and take their scratch from hipMalloc), and room for the per-job arrays.
static int th_arena_create(th_gpu *g, int n, long long T)
{
    if (g->arena.base || n <= 0 || T <= 0) return 0;
    const double avg = (double)T / n;
    const size_t poa = (size_t)std::min<double>((double)TH_POA_SCRATCH_BYTES / 2, 100.0 * avg * 4.0 * std::min(n, g->waves_poa) * 1.3);
    size_t want = (size_t)(64.0 * (double)T) + (size_t)n * 1024 + poa + (256u << 20);      // (buffers carry 12.5 % slack; consensus jobs, items, par_pos: ~10 bytes per base)
#ifndef TH_CPU_EMUL
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && want > fr / 2) want = fr / 2;     // never more than half of what is free
#endif
#ifndef TH_CPU_EMUL
    if (want < (96u << 20)) return 0;                     // tiny batches (tests): plain allocations
#endif
    const double t0 = th_wall_s();
    void *p = nullptr;
    if (th_rt_malloc(&p, want)) return 0;                 // no arena: every buffer allocates for itself
    g_alloc_s += th_wall_s() - t0; g_alloc_bytes += (long long)want; ++g_alloc_n;
    g->arena.base = (char *)p; g->arena.cap = want; g->arena.top = 0;
    return 0;
}

// Sizes every per-read / per-base buffer for batches of up to n reads and T bases, so that a run of growing batches (the file
// driver starts with small ones) does not free and re-allocate gigabytes at every step.  Buffers whose size depends on the
// stage results (chain records, consensus jobs, POA scratch) still grow on demand.
extern "C" int th_gpu_reserve(th_gpu_t *g, int n, long long T)
{
    TRY(th_select_device(g));
    if (n <= 0 || T <= 0) return 0;
    if (T >= (1LL << 31)) T = (1LL << 31) - 64;
    TRY(th_arena_create(g, n, T));
    const size_t padT = (size_t)((T + 15) & ~15LL);
    TRY(g->d_seq.ensure(padT)); TRY(g->d_bseq.ensure(padT)); TRY(g->d_roff.ensure((size_t)n + 1)); TRY(g->d_gsort_off.ensure((size_t)n));
    TRY(g->d_tmp.ensure(padT)); TRY(g->d_hit_end.ensure(padT)); TRY(g->d_hit_per.ensure(padT)); TRY(g->d_hit_n.ensure(n));
    TRY(g->d_score.ensure(padT)); TRY(g->d_from.ensure(padT)); TRY(g->d_cells.ensure(padT)); TRY(g->d_tracked.ensure(padT));
    TRY(g->d_xorder.ensure(padT)); TRY(g->d_xstart.ensure(padT)); TRY(g->d_xend.ensure(padT)); TRY(g->d_eps.ensure(padT));
    TRY(g->d_read_ch_base.ensure(n)); TRY(g->d_read_ch_n.ensure(n)); TRY(g->d_rec.ensure((size_t)n * 8 + 4096));
    g->reserve_n = n;
    return 0;
}

extern "C" int th_gpu_upload(th_gpu_t *g, const char *const *seqs, const int *lens, int n)
{
    TRY(th_select_device(g));
    g->n = n; th_room(g->h_roff, (size_t)n + 1); g->h_roff.assign((size_t)n + 1, 0);
    long long T = 0; int mx = 0;
    for (int i = 0; i < n; ++i) { g->h_roff[i] = (uint32_t)T; T += lens[i]; mx = std::max(mx, lens[i]); }
    if (T >= (1LL << 31)) return th_fail("batch holds 2^31 bases or more; split it");
    if (mx >= (1 << 28)) return th_fail("a read of 2^28 bases or more is not supported (32-bit period arithmetic in th_k_chain_dp)");
    g->h_roff[n] = (uint32_t)T; g->T = T; g->max_len = mx;
    size_t padT = (size_t)((T + 15) & ~15LL);
    // reads whose sort does not fit the LDS buffers get a segment of HBM scratch
    th_room(g->h_gsort_off, (size_t)n); g->h_gsort_off.assign((size_t)n, -1); long long gs = 0;
    for (int i = 0; i < n; ++i) {
        int ne = lens[i] - g->dp.k + 1; if (ne < 1) continue;
        long long N = 1; while (N < ne) N <<= 1;
        if (N > TH_XSORT_LDS) { g->h_gsort_off[i] = gs; gs += N; }
    }
    g->gsort_total = gs;
    TRY(th_arena_create(g, n, T));                        // (no-op once the handle has one)
    TRY(g->d_seq.ensure(padT)); TRY(g->d_bseq.ensure(padT)); TRY(g->d_roff.ensure((size_t)n + 1));
    TRY(g->d_gsort_off.ensure((size_t)n)); TRY(g->d_gsort.ensure((size_t)gs));
    // The reads go to HBM through two small pinned staging buffers: while one is on its way (async H2D) the host concatenates
    // the next reads into the other.  (One pinned buffer for the whole batch meant pinning ~1 GB before the first byte moved:
    // 0.13-0.19 s per handle, and the copy could not overlap the concatenation.)
    if (!g->stage_ok) {
        for (int b = 0; b < 2; ++b) { TRY(th_rt_host_alloc((void **)&g->h_stage[b], TH_STAGE_BYTES)); TRY(th_rt_event_create(&g->stage_ev[b])); }
        g->stage_ok = true;
    }
    {
        size_t done = 0, fill = 0; int cur = 0, used[2] = { 0, 0 };
        auto flush = [&]() -> int {
            if (!fill) return 0;
            TRY(th_rt_h2d(g->d_seq.p + done, g->h_stage[cur], fill, g->stream));
            TRY(th_rt_event_record(g->stage_ev[cur], g->stream));
            used[cur] = 1; done += fill; fill = 0; cur ^= 1;
            if (used[cur]) TRY(th_rt_event_wait(g->stage_ev[cur]));      // the buffer we are about to refill must have left
            return 0;
        };
        for (int i = 0; i <= n; ++i) {
            const char *src = i < n ? seqs[i] : nullptr; size_t left = i < n ? (size_t)lens[i] : padT - (size_t)T;   // the tail is padded with 'N'
            while (left) {
                const size_t c = std::min(left, (size_t)TH_STAGE_BYTES - fill);
                if (src) { memcpy(g->h_stage[cur] + fill, src, c); src += c; } else memset(g->h_stage[cur] + fill, 'N', c);
                fill += c; left -= c;
                if (fill == TH_STAGE_BYTES) TRY(flush());
            }
        }
        TRY(flush());
    }
    TRY(th_rt_h2d(g->d_roff.p, g->h_roff.data(), ((size_t)n + 1) * 4, g->stream));
    TRY(th_rt_h2d(g->d_gsort_off.p, g->h_gsort_off.data(), (size_t)n * 8, g->stream));
    TRY(th_rt_sync(g->stream));
    return 0;
}

// ---- host: split each chain's par_pos at -1 separators into consensus jobs (src/gen_cons.c:192-205) ----
static void build_jobs(th_gpu *g)
{
    g->h_tasks.clear(); g->h_units.clear(); g->h_items.clear(); g->jobs.clear();
    th_room(g->h_tasks, g->h_par_n.size() + 16); th_room(g->h_units, g->h_parc.size() + 16);     // at most one task per chain, one unit per boundary
    const int min_copy = g->dp.min_copy;
    long long cons_top = 0; int aln_top = 0;
    // chains grouped by read, in emission order: records of read r are rec[base .. base+cnt)
    for (int r = 0; r < g->n; ++r) {
        const int L = (int)(g->h_roff[r + 1] - g->h_roff[r]);
        for (int c = g->h_read_ch_base[r]; c < g->h_read_ch_base[r] + g->h_read_ch_n[r]; ++c) {
            const int par_n = g->h_par_n[c]; const int *pp = g->h_parc.data() + g->h_par_dst[c];
            if (par_n < min_copy + 1) continue;                        // src/tidehunter.c:42
            int i = 0;
            while (i < par_n - min_copy) {
                if (pp[i] < 0) { ++i; continue; }
                int j; for (j = i + 1; j < par_n; ++j) if (pp[j] < 0) break;
                if (j - i > min_copy) {
                    ThTask t; memset(&t, 0, sizeof(t));
                    t.read = r; t.pos_off = g->h_par_dst[c] + i; t.pos_n = j - i; t.unit_off = (int)g->h_units.size();
                    for (int u = i; u < j - 1; ++u) {                  // src/abpoa_cons.c:41-48
                        const int start = pp[u], end = pp[u + 1];
                        if (start < 0 || end < 0 || start >= L - 1 || end + 1 > L) continue;
                        ThUnit x = { start + 1, end - start }; g->h_units.push_back(x);
                        ++t.n_units; t.sum_len += x.len; t.max_len = std::max(t.max_len, x.len);
                    }
                    t.cons_cap = t.n_units <= 2 ? (t.n_units ? g->h_units[t.unit_off].len : 0) : std::min(t.sum_len, 3 * t.max_len + 64);
                    t.cons_off = (int)cons_top; cons_top += (t.cons_cap + 15) & ~15;
                    t.aln_off = aln_top; aln_top += t.pos_n - 1;
                    g->h_tasks.push_back(t);
                }
                i = j + 1;
            }
        }
    }
}

// Scratch of one POA wave for task t, in ints.  Mirrors the carve in th_k_poa: 39 node / row arrays of `cap`, the overflow pool,
// the alignment-op arrays, then the cell region.  Cell region: every band cell has a 2-byte direction word; rows that store full
// cells add 16 bytes per cell -- a few per cent of the rows when the band fits the LDS ring, EVERY row when it does not (the 32-bit
// variant on long units: measured band ~ 2w + 1 + len / 30 at 15 % error).  Graph rows grow by ~9 % of the unit length per unit
// (measured: 1.27 x at 4 units of 5000, ~1.5 x at 9 units of 1000, ~2.6 x at 20 units of 200).  cell_factor is the attempt's
// safety factor (1.5 at first, doubled for the jobs that come back with "scratch too small").
static long long poa_need_ints(const ThTask &t, double cell_factor, bool noband = false, bool wide = false)
{
    if (t.n_units <= 2) return 0;
    const long long cap = t.sum_len + 2, ocap = 2LL * (t.sum_len + t.n_units + 8), opn = cap + t.max_len + 2;
    const long long graph = 39 * cap + 2 * ocap + 5 * opn + 512;
    if (noband) return (graph + (long long)(cap * ((long long)t.max_len + 4) * (cell_factor > 1.5 ? 1.25 : 1.0) * 4.5) + 63) & ~63LL;   // full rows, all stored in full
    const long long w = TH_POA_WB + t.max_len / 100;
    const long long span = 2 * w + 1 + 48 + t.max_len / 30;
    const long long rows = std::min<long long>(cap, (long long)(t.max_len * (1.0 + 0.09 * (t.n_units - 1))) + 64);
    const bool all_full = wide && span + 16 >= TH_POA_RW;
    return (graph + (long long)(rows * span * cell_factor * (all_full ? 4.5 : 1.5)) + 63) & ~63LL;
}

// per-kernel device time: HIP events recorded on the pipeline's own stream around each launch
enum { TK_SEED = 0, TK_CHAIN_DP, TK_CHAIN_X, TK_PART, TK_POA, TK_ALIGN, TK_N };
#ifndef TH_CPU_EMUL
struct KTimer {
    std::vector<Ev> ev[TK_N]; th_stream_t s;
    void begin(int k) { Ev e; hipEventCreate(&e.a); hipEventCreate(&e.b); hipEventRecord(e.a, s); ev[k].push_back(e); }
    void end(int k) { hipEventRecord(ev[k].back().b, s); }
    void collect(double *ms) {
        for (int k = 0; k < TK_N; ++k) {
            ms[k] = 0;
            for (Ev &e : ev[k]) { float t = 0; hipEventSynchronize(e.b); hipEventElapsedTime(&t, e.a, e.b); ms[k] += t; hipEventDestroy(e.a); hipEventDestroy(e.b); }
            ev[k].clear();
        }
    }
    // a run that returns early (arena overflow, a failed launch, ...) never reaches collect(): release its events here
    ~KTimer() { for (int k = 0; k < TK_N; ++k) for (Ev &e : ev[k]) { hipEventDestroy(e.a); hipEventDestroy(e.b); } }
};
#define KT_BEGIN(k) kt.begin(k)
#define KT_END(k) kt.end(k)
#else
#define KT_BEGIN(k) ((void)0)
#define KT_END(k) ((void)0)
#endif

extern "C" int th_gpu_run(th_gpu_t *g)
{
    const int n = g->n; const long long T = g->T; const size_t padT = (size_t)((T + 15) & ~15LL);
    const double t_run0 = th_wall_s(); (void)t_run0;
    TRY(th_select_device(g));
    g->jobs.clear(); g->ms_kernels = g->ms_dominant = 0; g->n_rec = 0; g->ad_valid = false;
    g->h_tasks.clear(); g->h_parc.clear();
    if (n == 0) return 0;
    for (int k = 0; k < TK_N; ++k) g->ms_k[k] = 0;
#ifndef TH_CPU_EMUL
    KTimer kt; kt.s = g->stream;
#endif
    int rec_cap = std::max(n, g->reserve_n) * 8 + 4096;
    if (getenv("TH_REC_CAP")) rec_cap = std::max(1, atoi(getenv("TH_REC_CAP")));      // test knob: forces the arena-growth path
    TRY(g->d_tmp.ensure(padT)); TRY(g->d_hit_end.ensure(padT)); TRY(g->d_hit_per.ensure(padT)); TRY(g->d_hit_n.ensure(n));
    TRY(g->d_score.ensure(padT)); TRY(g->d_from.ensure(padT)); TRY(g->d_cells.ensure(padT)); TRY(g->d_tracked.ensure(padT));
    TRY(g->d_xorder.ensure(padT)); TRY(g->d_xstart.ensure(padT)); TRY(g->d_xend.ensure(padT));
    TRY(g->d_xovf.ensure((size_t)std::min(n, g->waves_xw) * 7 * (TH_TOPN - TH_TOPL)));
    TRY(g->d_cnt.ensure(16)); TRY(g->d_read_ch_base.ensure(n)); TRY(g->d_read_ch_n.ensure(n)); TRY(g->d_rec.ensure(rec_cap));
    TRY(th_rt_memset(g->d_cnt.p, 0, 16 * sizeof(int), g->stream));

    const int enc_blocks = (int)std::min<long long>(2048, (T / 16 + 255) / 256 + 1);
    KT_BEGIN(TK_SEED);
    TH_LAUNCH(th_k_encode, enc_blocks, 256, g->stream, (const uint8_t *)g->d_seq.p, g->d_bseq.p, (long long)T);
#define HITS_LAUNCH(W, PB, CAP, GS) TH_LAUNCH((th_k_hits<W, PB, CAP>), n, TH_SEED_BLOCK, g->stream, (const uint8_t *)g->d_bseq.p, (const uint32_t *)g->d_roff.p, g->dp, \
                  g->d_hit_end.p, g->d_hit_per.p, g->d_hit_n.p, g->d_tmp.p, GS, (const long long *)g->d_gsort_off.p, g->d_cnt.p + 1, g->d_cnt.p + 8)
    // 32-bit sort words when key << PB | pos stays below the all-ones sentinel and every read sorts in LDS
    if (g->dp.k <= 9 && g->max_len < 16384) HITS_LAUNCH(uint32_t, 14, 16384, (uint32_t *)nullptr);
    else if (g->dp.k <= 8 && g->max_len < 32768) HITS_LAUNCH(uint32_t, 15, 32768, (uint32_t *)nullptr);
    else HITS_LAUNCH(unsigned long long, 32, TH_SORT_LDS, g->d_gsort.p);
    KT_END(TK_SEED); KT_BEGIN(TK_CHAIN_DP);
    if (g->max_len < 65536 && !getenv("TH_CHAIN_UNPACKED")) {      // (the environment knob exists for the CPU tests)
        TRY(g->d_eps.ensure(padT));
#define CHAIN_LAUNCH(RG) TH_LAUNCH((th_k_chain_dp<true, RG>), (n + 3) / 4, TH_CHAIN_BLOCK, g->stream, (const uint32_t *)g->d_roff.p, n, g->dp, \
                  (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, g->d_score.p, g->d_from.p, g->d_eps.p)
        const int ring = env_int("TH_CHAIN_RING", 512);
        if (ring >= 1024) CHAIN_LAUNCH(1024); else if (ring >= 768) CHAIN_LAUNCH(768); else if (ring >= 512) CHAIN_LAUNCH(512); else CHAIN_LAUNCH(0);
#undef CHAIN_LAUNCH
    } else
        TH_LAUNCH((th_k_chain_dp<false, 0>), (n + 3) / 4, TH_CHAIN_BLOCK, g->stream, (const uint32_t *)g->d_roff.p, n, g->dp,
                  (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, g->d_score.p, g->d_from.p, (unsigned long long *)nullptr);
    KT_END(TK_CHAIN_DP); KT_BEGIN(TK_CHAIN_X);
    TH_LAUNCH(th_k_chain_sort, n, TH_XS_BLOCK, g->stream, (const uint32_t *)g->d_roff.p, g->dp,
              (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, (const int *)g->d_score.p,
              g->d_xorder.p, g->d_xstart.p, g->d_xend.p);
    KT_END(TK_CHAIN_X);
    g->h_cnt.assign(16, 0); int *cnt = g->h_cnt.data();      // (pinned) 0 chain records, 1 error flags, 2-7 work-queue heads / arena tops, 8 reads whose minimizer selection fell back to the sequential replay
    // chain extraction fills an arena of chain records with one atomic per read; a batch with more chains than the arena
    // holds (8 per read on average) reruns the extraction with a larger one -- the kernels only read stage 1-2 results
    for (int attempt = 0;; ++attempt) {
        KT_BEGIN(TK_CHAIN_X);
        TH_LAUNCH(th_k_chain_extract, n, TH_XBLOCK, g->stream, (const uint32_t *)g->d_roff.p, g->dp,
                  (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p,
                  (const int *)g->d_score.p, (const int *)g->d_from.p, g->d_tracked.p, g->d_cells.p,
                  g->d_gsort.p, (const long long *)g->d_gsort_off.p, g->d_rec.p, rec_cap, g->d_cnt.p,
                  g->d_read_ch_base.p, g->d_read_ch_n.p, g->d_cnt.p + 1);
        TH_LAUNCH(th_k_chain_extract_wave, std::min(n, g->waves_xw), TH_XW_BLOCK, g->stream, (const uint32_t *)g->d_roff.p, n, g->dp,
                  (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p,
                  (const int *)g->d_score.p, (const int *)g->d_from.p, g->d_cells.p,
                  (const uint16_t *)g->d_xorder.p, (const int *)g->d_xstart.p, (const int *)g->d_xend.p, g->d_xovf.p, g->d_rec.p, rec_cap, g->d_cnt.p,
                  g->d_read_ch_base.p, g->d_read_ch_n.p, g->d_cnt.p + 1, g->d_cnt.p + 6);
        KT_END(TK_CHAIN_X);
        TRY(th_rt_d2h(cnt, g->d_cnt.p, 16 * sizeof(int), g->stream));
        TRY(th_rt_sync(g->stream));
        if (!(cnt[1] & 1)) break;
        if (attempt == 4 || (long long)rec_cap * 4 > (1LL << 30)) return th_fail("chain record arena overflow");
        rec_cap = std::max(rec_cap * 4, cnt[0] + 4096);
        TRY(g->d_rec.ensure(rec_cap));
        const int keep = cnt[1] & ~1;
        TRY(th_rt_memset(g->d_cnt.p, 0, sizeof(int), g->stream));
        TRY(th_rt_memset(g->d_cnt.p + 6, 0, sizeof(int), g->stream));      // read queue of the extraction waves
        TRY(th_rt_h2d(g->d_cnt.p + 1, &keep, sizeof(int), g->stream));
    }
    if (cnt[1] & 8) return th_fail("minimizer selection repeated a word or overflowed its buffer");
    if (cnt[8] && g->dp.w > 1 && !g->dp.hpc && !g->dp.w_seq) {
        if (getenv("TH_MINIMIZER_STRICT")) return th_fail("the parallel minimizer rules marked a slot twice on " + std::to_string(cnt[8]) + " read(s) (sequential replay used)");
        if (getenv("TH_TRACE")) fprintf(stderr, "[th] minimizer selection: %d read(s) took the sequential replay\n", cnt[8]);
    }
    if (cnt[1]) return th_fail("stage 1-2 kernels reported error flags " + std::to_string(cnt[1]));
    const int n_rec = cnt[0]; g->n_rec = n_rec;
    th_room(g->h_read_ch_base, (size_t)n); th_room(g->h_read_ch_n, (size_t)n);
    g->h_read_ch_base.resize(n); g->h_read_ch_n.resize(n);
    TRY(th_rt_d2h(g->h_read_ch_base.data(), g->d_read_ch_base.p, (size_t)n * 4, g->stream));
    TRY(th_rt_d2h(g->h_read_ch_n.data(), g->d_read_ch_n.p, (size_t)n * 4, g->stream));
    th_room(g->h_rec, (size_t)n_rec); th_room(g->h_par_n, (size_t)n_rec); th_room(g->h_par_dst, (size_t)n_rec);
    g->h_rec.resize(n_rec); g->h_par_n.assign(n_rec, 0); g->h_par_dst.assign(n_rec, 0);
    if (n_rec) {
        // ---- partition ----
        const long long bnd_part = (((long long)g->max_len + 128) * 6 + 3) & ~3LL;   // 16-byte aligned per-wave segments
        const int wp = std::min(n_rec, g->waves_part);
        TRY(g->d_par_off.ensure(n_rec)); TRY(g->d_par_n.ensure(n_rec)); TRY(g->d_par_dst.ensure(n_rec));
        TRY(g->d_bnd.ensure((size_t)(bnd_part * wp)));
        TRY(th_rt_d2h(g->h_rec.data(), g->d_rec.p, (size_t)n_rec * sizeof(ThChainRec), g->stream));
        TRY(th_rt_sync(g->stream));
        size_t par_base = 0;                                     // the regions the kernel will carve: 2 * cells + 16 per chain
        for (int c = 0; c < n_rec; ++c) par_base += 2 * (size_t)g->h_rec[c].len + 16;
        // a chain's par_pos region holds cap_mul * (2 * cells + 16) entries and the arena is sized to exactly their sum, so the
        // arena cannot overflow -- a single chain that needs more (the reference allows up to the read length,
        // src/partition.c:173) makes the partition rerun with larger regions
        for (int cap_mul = getenv("TH_PAR_MUL") ? -1 : 1;; cap_mul = cap_mul < 0 ? 1 : cap_mul * 4) {
            const int mul = cap_mul < 0 ? 0 : cap_mul;     // (TH_PAR_MUL: test knob, a first pass with empty regions)
            const size_t par_cap = (size_t)std::max(mul, 1) * par_base + 64;
            if (par_cap >= (1ull << 31)) return th_fail("par_pos arena would exceed 2^31 entries; use a smaller batch");
            TRY(g->d_par.ensure(par_cap));
            KT_BEGIN(TK_PART);
            TH_LAUNCH(th_k_partition, wp, TH_PART_BLOCK, g->stream, (const ThChainRec *)g->d_rec.p, n_rec, (const uint32_t *)g->d_roff.p,
                      (const uint8_t *)g->d_bseq.p, (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_cells.p, g->dp,
                      g->d_par.p, (int)par_cap, g->d_cnt.p + 5, g->d_par_off.p, g->d_par_n.p, g->d_bnd.p, bnd_part, g->d_cnt.p + 1, mul, g->d_cnt.p + 7);
            KT_END(TK_PART);
            TRY(th_rt_d2h(cnt, g->d_cnt.p, 16 * sizeof(int), g->stream));
            TRY(th_rt_sync(g->stream));
            if (!cnt[1]) break;
            if (cap_mul >= 64) return th_fail("partition scratch overflow (par_pos region or boundary buffer)");
            TRY(th_rt_memset(g->d_cnt.p + 1, 0, sizeof(int), g->stream));
            TRY(th_rt_memset(g->d_cnt.p + 5, 0, sizeof(int), g->stream));
            TRY(th_rt_memset(g->d_cnt.p + 7, 0, sizeof(int), g->stream));      // chain queue of the partition waves
        }
        TRY(th_rt_d2h(g->h_par_n.data(), g->d_par_n.p, (size_t)n_rec * 4, g->stream));
        TRY(th_rt_sync(g->stream));
        long long tot = 0;
        for (int c = 0; c < n_rec; ++c) { g->h_par_dst[c] = (int)tot; tot += g->h_par_n[c]; }
        th_room(g->h_parc, (size_t)tot); g->h_parc.resize((size_t)tot);
        TRY(g->d_parc.ensure((size_t)tot + 1));
        TRY(th_rt_h2d(g->d_par_dst.p, g->h_par_dst.data(), (size_t)n_rec * 4, g->stream));
        TH_LAUNCH(th_k_gather_par, n_rec, 64, g->stream, n_rec, (const int *)g->d_par.p, (const int *)g->d_par_off.p, (const int *)g->d_par_n.p,
                  (const int *)g->d_par_dst.p, g->d_parc.p);
        TRY(th_rt_d2h(g->h_parc.data(), g->d_parc.p, (size_t)tot * 4, g->stream));
        TRY(th_rt_sync(g->stream));
    } else TRY(th_rt_sync(g->stream));

    if (getenv("TH_TRACE")) fprintf(stderr, "[th] %.3f s: stages 1-3 done, %d chains (device malloc %.3f s for %.2f GB in %d calls, free %.3f s so far)\n", th_wall_s() - t_run0, n_rec, g_alloc_s, g_alloc_bytes / 1e9, g_alloc_n, g_free_s);
    build_jobs(g);
    const int nt = (int)g->h_tasks.size();
    g->jobs.resize(nt);
    for (int t = 0; t < nt; ++t) {
        const ThTask &k = g->h_tasks[t]; th_job_t &j = g->jobs[t];
        j.read = k.read; j.pos_off = k.pos_off; j.pos_n = k.pos_n; j.n_seqs = k.n_units; j.cons_off = k.cons_off; j.cons_len = 0;
        j.aln_off = k.aln_off; j.max_q_l = j.max_t_l = j.max_q_r = j.max_t_r = -1; j.status = 0;
    }
    if (nt && !g->para.only_unit) {
        for (int t = 0; t < nt; ++t) if (g->h_tasks[t].n_units <= 1) return th_fail("Not enough sequences to perform msa.");  // src/abpoa_cons.c:61
        const long long cons_total = g->h_tasks.back().cons_off + ((g->h_tasks.back().cons_cap + 15) & ~15) + 16;
        const int aln_total = g->h_tasks.back().aln_off + g->h_tasks.back().pos_n - 1;
        TRY(g->d_tasks.ensure(nt)); TRY(g->d_units.ensure(g->h_units.size() + 1)); TRY(g->d_cons.ensure((size_t)cons_total)); TRY(g->d_cov.ensure((size_t)cons_total));
        TRY(g->d_cons_len.ensure(nt)); TRY(g->d_status.ensure(nt)); TRY(g->d_iden.ensure((size_t)aln_total + 1)); TRY(g->d_ext.ensure((size_t)nt * 4));
        TRY(th_rt_h2d(g->d_tasks.p, g->h_tasks.data(), (size_t)nt * sizeof(ThTask), g->stream));
        TRY(th_rt_h2d(g->d_units.p, g->h_units.data(), g->h_units.size() * sizeof(ThUnit), g->stream));
        // ---- POA: persistent waves with private scratch; tasks that outgrow it are retried with more ----
        th_room(g->h_cons_len, (size_t)nt); th_room(g->h_status, (size_t)nt);
        g->h_cons_len.assign(nt, 0); g->h_status.assign(nt, 0);
        std::vector<int> todo(nt); for (int t = 0; t < nt; ++t) todo[t] = t;
        // Two kernel variants: the common one keeps recent rows as int16 in LDS (packed rows, 5 waves per SIMD) and gives a job
        // back with status 7 when an alignment's scores may not fit int16 (long units, unusual scoring); those jobs, and jobs
        // whose units exceed its query staging buffer, run on the 32-bit-ring variant.
        std::vector<int> todo_n, todo_w;
        for (int t : todo) (g->h_tasks[t].max_len <= TH_POA_QCAP ? todo_n : todo_w).push_back(t);
        double factor = 1.5; bool patched = false;   // patched: some results live in subset arrays
        const int poa_test_fail = std::max(0, env_int("TH_POA_TEST_FAIL", 0));   // test knob, see th_k_poa
        bool noband = false;                          // set for the rerun of jobs whose adaptive band failed
        auto run_group = [&](const std::vector<int> &grp, bool wide, std::vector<int> &again, std::vector<int> &to_wide) -> int {
            if (grp.empty()) return 0;
            long long need = 1024; int max_unit = 0;
            for (int t : grp) { need = std::max(need, poa_need_ints(g->h_tasks[t], factor, noband, wide)); max_unit = std::max(max_unit, g->h_tasks[t].max_len); }
            const int poa_flags = (noband ? 1 : 0) | poa_test_fail << 8;
            const int wv = (int)std::min<long long>(std::min<long long>((long long)grp.size(), g->waves_poa), std::max<long long>(1, (TH_POA_SCRATCH_BYTES) / (need * 4)));
            // a handle reserved for larger batches takes the scratch of the full persistent grid at once (with slack for longer units)
            const long long wv_alloc = (g->reserve_n > g->n && !noband) ? std::min<long long>(std::min<long long>(g->reserve_n, g->waves_poa), std::max<long long>(1, (TH_POA_SCRATCH_BYTES) / (need * 4))) : wv;
            TRY(g->d_scratch.ensure((size_t)(need * std::max<long long>(wv, wv_alloc))));
            const ThTask *dt = g->d_tasks.p; int *dcl = g->d_cons_len.p, *dst = g->d_status.p;
            DBuf<ThTask> &d_sub = g->d_sub; DBuf<int> &d_scl = g->d_scl, &d_sst = g->d_sst;
            const bool whole = (int)grp.size() == nt;            // every task in its original order: the main arrays serve
            if (!whole) {
                patched = true;
                th_pvec<ThTask> &sub = g->h_sub; sub.clear(); th_room(sub, grp.size());
                for (int t : grp) sub.push_back(g->h_tasks[t]);
                TRY(d_sub.ensure(sub.size())); TRY(d_scl.ensure(sub.size())); TRY(d_sst.ensure(sub.size()));
                TRY(th_rt_h2d(d_sub.p, sub.data(), sub.size() * sizeof(ThTask), g->stream));
                TRY(th_rt_sync(g->stream));
                dt = d_sub.p; dcl = d_scl.p; dst = d_sst.p;
            }
            TRY(th_rt_memset(g->d_cnt.p + 2, 0, 2 * sizeof(int), g->stream));      // job queue head
            KT_BEGIN(TK_POA);
#define POA_LAUNCH(QC, RT) TH_LAUNCH((th_k_poa<QC, RT>), wv, TH_POA_BLOCK, g->stream, dt, (int)grp.size(), (const ThUnit *)g->d_units.p, (const uint32_t *)g->d_roff.p, \
                          (const uint8_t *)g->d_bseq.p, g->dp, g->d_scratch.p, need, g->d_cons.p, g->d_cov.p, dcl, dst, g->d_cnt.p + 2, poa_flags)
            if (!wide) POA_LAUNCH(TH_POA_QCAP, short);
            else if (max_unit <= TH_POA_QCAP_WIDE) POA_LAUNCH(TH_POA_QCAP_WIDE, int);
            else POA_LAUNCH(0, int);
            KT_END(TK_POA);
            if (getenv("TH_TRACE")) { th_rt_sync(g->stream); fprintf(stderr, "[th] %.3f s: poa done (%d jobs, %d waves, %s ring, %lld MB scratch)\n", th_wall_s() - t_run0, (int)grp.size(), wv, wide ? "32-bit" : "int16", (long long)(need * wv * 4 >> 20)); }
            th_pvec<int> &cl = g->h_cl, &st = g->h_st; th_room(cl, grp.size()); th_room(st, grp.size()); cl.assign(grp.size(), 0); st.assign(grp.size(), 0);
            TRY(th_rt_d2h(cl.data(), dcl, grp.size() * 4, g->stream));
            TRY(th_rt_d2h(st.data(), dst, grp.size() * 4, g->stream));
            TRY(th_rt_sync(g->stream));
            for (size_t x = 0; x < grp.size(); ++x) {
                const int t = grp[x];
                g->h_cons_len[t] = cl[x]; g->h_status[t] = st[x];
                if (st[x] == 1 || st[x] == 2) again.push_back(t);
                else if (st[x] == 7 && !wide) to_wide.push_back(t);
                // 3 / 4 (the adaptive band came out empty or lost the global path) are rerun without a band below; 5 / 6 (in-degree
                // above 255, consensus longer than its slot) stay failed.  None of them may fail the batch: abPOA has no such
                // failure mode, and one pathological read must not stop a multi-million-read run.
            }
            return 0;
        };
        for (int attempt = 0; attempt < 8 && (!todo_n.empty() || !todo_w.empty()); ++attempt, factor *= 2) {
            std::vector<int> again_n, again_w, to_wide;
            TRY(run_group(todo_n, false, again_n, to_wide));
            todo_w.insert(todo_w.end(), to_wide.begin(), to_wide.end());     // same scratch factor: they have not run yet
            std::sort(todo_w.begin(), todo_w.end());
            TRY(run_group(todo_w, true, again_w, to_wide));
            todo_n.swap(again_n); todo_w.swap(again_w);
        }
        todo.clear(); todo.insert(todo.end(), todo_n.begin(), todo_n.end()); todo.insert(todo.end(), todo_w.begin(), todo_w.end());
        if (todo.empty()) {
            // jobs whose adaptive band failed: once more with full rows on the 32-bit variant (oracle/tho_poa.c does the same)
            std::vector<int> nb;
            for (int t = 0; t < nt; ++t) if (g->h_status[t] == 3 || g->h_status[t] == 4) nb.push_back(t);
            noband = true; factor = 1.0;
            for (int attempt = 0; attempt < 3 && !nb.empty(); ++attempt, factor *= 2) {
                std::vector<int> again, none;
                TRY(run_group(nb, true, again, none));
                nb.swap(again);
            }
            noband = false;
            todo = nb;
        }
        int n_failed = 0;
        for (int t = 0; t < nt; ++t) if (g->h_status[t] != 0 && g->h_status[t] != 1 && g->h_status[t] != 2) { g->h_cons_len[t] = 0; ++n_failed; patched = true; }
        if (n_failed) fprintf(stderr, "[tidehunter_amd] warning: %d consensus job(s) could not be aligned (POA status 3-6) and are reported without a consensus\n", n_failed);
        if (patched) {   // some results live in subset arrays: the alignment kernel reads the main one
            TRY(th_rt_h2d(g->d_cons_len.p, g->h_cons_len.data(), (size_t)nt * 4, g->stream));
            TRY(th_rt_sync(g->stream));
        }
        if (!todo.empty()) return th_fail("POA scratch still too small after 8 attempts");
#if defined(TH_POA_PROF) && !defined(TH_CPU_EMUL)
        { unsigned long long pr[20]; hipMemcpyFromSymbol(pr, HIP_SYMBOL(th_poa_prof), sizeof(pr));
          fprintf(stderr, "[poa_prof] rows %llu dp %llu traceback %llu merge %llu (cycles summed over waves)\n", pr[0], pr[1], pr[2], pr[3]);
          fprintf(stderr, "[poa_prof]   traceback: %llu tiles loaded, %llu run trips emitting %llu steps, %llu single steps\n", pr[4], pr[5], pr[6], pr[7]);
          const char *cat[6] = { "common packed", "other packed", "several predecessors", "one far predecessor", "stores full cells", "other" };
          for (int c = 0; c < 6; ++c) fprintf(stderr, "[poa_prof]   %-22s rows %12llu cycles %16llu (%.0f per row)\n", cat[c], pr[8 + c], pr[14 + c], pr[8 + c] ? (double)pr[14 + c] / pr[8 + c] : 0.0);
          unsigned long long z[20] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(th_poa_prof), z, sizeof(z)); }
#endif
        // ---- unit-vs-consensus alignments and flank extensions ----
        // Items whose two sequences fit the small staging area of the anti-diagonal engine (4 KB: 5 waves per SIMD) go first; tasks
        // whose unit + consensus fit 7 KB with at most two registers per lane (periods up to ~2500) form a second list for the same
        // register budget with a larger area; the rest a third one for the variant with a 24 KB area and up to 6 registers per lane
        // (two waves per SIMD) -- before this split the 2000-base units of config 5 ran on that variant although they need two.
        int max_cl = 0;
        const int ALN_SMALL = 4096, ALN_MID = 7168, ALN_BIG = 24576;
        std::vector<ThAlnItem> small_items, mid_items, big_items;
        for (int t = 0; t < nt; ++t) {
            max_cl = std::max(max_cl, g->h_cons_len[t]);
            const long long need2 = (long long)g->h_tasks[t].max_len + g->h_cons_len[t] + 512 * 2 + 16;
            const bool two_regs = 2 * (g->h_cons_len[t] / g->aln_wdiv + g->aln_wadd - g->aln_wtol) + 2 + 64 <= 512;     // th_ad_pick would settle for NP <= 2
            std::vector<ThAlnItem> &dst = need2 <= ALN_SMALL ? small_items : (need2 <= ALN_MID && two_regs) ? mid_items : big_items;
            for (int u = 0; u < g->h_tasks[t].pos_n - 1; ++u) { ThAlnItem it = { t, u }; dst.push_back(it); }
            ThAlnItem l = { t, -1 }, r = { t, -2 }; small_items.push_back(l); small_items.push_back(r);
        }
        th_room(g->h_items, small_items.size() + mid_items.size() + big_items.size());
        g->h_items.insert(g->h_items.end(), small_items.begin(), small_items.end());
        const int ns = (int)g->h_items.size();
        g->h_items.insert(g->h_items.end(), mid_items.begin(), mid_items.end());
        const int nm = (int)g->h_items.size();
        g->h_items.insert(g->h_items.end(), big_items.begin(), big_items.end());
        const int ni = (int)g->h_items.size();
        const long long bnd_aln = (std::max<long long>(((long long)max_cl + 128) * 4, ((long long)g->max_len + 128) * 2) + 3) & ~3LL;
        const int wa = std::min(ni, g->waves_aln);
        TRY(g->d_items.ensure(ni)); TRY(g->d_bnd.ensure((size_t)(bnd_aln * wa)));
        TRY(th_rt_h2d(g->d_items.p, g->h_items.data(), (size_t)ni * sizeof(ThAlnItem), g->stream));
        TRY(th_rt_memset(g->d_cnt.p + 7, 0, sizeof(int), g->stream));      // queue heads of the second and third list (d_cnt[4], the first one's, was zeroed at the start of the run)
        TRY(th_rt_memset(g->d_cnt.p + 9, 0, sizeof(int), g->stream));
        KT_BEGIN(TK_ALIGN);
        {
#define TH_ALN_ARGS(first, n, q) (const ThAlnItem *)g->d_items.p + (first), (n), (const ThTask *)g->d_tasks.p, (const int *)g->d_parc.p, \
                  (const uint32_t *)g->d_roff.p, (const uint8_t *)g->d_bseq.p, (const uint8_t *)g->d_cons.p, (const int *)g->d_cons_len.p, \
                  g->d_bnd.p, bnd_aln, g->d_iden.p, g->d_ext.p, g->d_cnt.p + 1, g->d_cnt.p + (q), g->aln_wdiv, g->aln_wadd, g->aln_wtol, g->aln_mode, g->aln_xpct
            if (ns) TH_LAUNCH((th_k_align<ALN_SMALL, 2>), std::min(ns, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(0, ns, 4));
            if (nm > ns) TH_LAUNCH((th_k_align<ALN_MID, 2>), std::min(nm - ns, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(ns, nm - ns, 9));
            if (ni > nm) TH_LAUNCH((th_k_align<ALN_BIG, 6>), std::min(ni - nm, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(nm, ni - nm, 7));
#undef TH_ALN_ARGS
        }
        KT_END(TK_ALIGN);
        if (getenv("TH_TRACE")) { th_rt_sync(g->stream); fprintf(stderr, "[th] %.3f s: align done (%d items)\n", th_wall_s() - t_run0, ni); }
        // ---- adapter searches on the doubled consensus (src/gen_cons.c:226-291), four per job, one per lane ----
        if (g->ad_on) {
            TRY(g->d_adp.ensure(1)); TRY(g->d_ad.ensure((size_t)nt * 12));
            TRY(th_rt_h2d(g->d_adp.p, &g->h_ad, sizeof(ThAdPara), g->stream));
            KT_BEGIN(TK_ALIGN);
            TH_LAUNCH(th_k_adapter, (nt * 4 + 63) / 64, 64, g->stream, (const ThTask *)g->d_tasks.p, nt, (const uint8_t *)g->d_cons.p, (const int *)g->d_cons_len.p,
                      (const ThAdPara *)g->d_adp.p, g->ad_min_cons, g->d_ad.p);
            KT_END(TK_ALIGN);
            th_room(g->h_ad_out, (size_t)nt * 12); g->h_ad_out.resize((size_t)nt * 12);
            TRY(th_rt_d2h(g->h_ad_out.data(), g->d_ad.p, (size_t)nt * 48, g->stream));
            g->ad_valid = true;
        }
        // results: consensus strings compacted to their lengths (the slots are sized for the worst case), coverage only on request
        th_pvec<int> &cc_off = g->h_cc_off; th_room(cc_off, (size_t)nt); cc_off.assign(nt, 0); long long cc_total = 0;
        for (int t = 0; t < nt; ++t) { cc_off[t] = (int)cc_total; cc_total += (g->h_cons_len[t] + 3) & ~3; }
        TRY(g->d_cc_len.ensure(nt)); TRY(g->d_cc_off.ensure(nt)); TRY(g->d_cons_c.ensure((size_t)cc_total + 4));
        if (g->para.want_cov) TRY(g->d_cov_c.ensure((size_t)cc_total + 4));
        TRY(th_rt_h2d(g->d_cc_len.p, g->h_cons_len.data(), (size_t)nt * 4, g->stream));
        TRY(th_rt_h2d(g->d_cc_off.p, cc_off.data(), (size_t)nt * 4, g->stream));
        TH_LAUNCH(th_k_gather_cons, nt, 128, g->stream, nt, (const ThTask *)g->d_tasks.p, (const int *)g->d_cc_len.p, (const int *)g->d_cc_off.p,
                  (const uint8_t *)g->d_cons.p, (const int *)g->d_cov.p, g->d_cons_c.p, g->para.want_cov ? g->d_cov_c.p : (int *)nullptr);
        th_room(g->h_iden, (size_t)aln_total); th_room(g->h_ext, (size_t)nt * 4); th_room(g->h_cons, (size_t)cc_total + 4);
        g->h_iden.resize(aln_total); g->h_ext.resize((size_t)nt * 4); g->h_cons.resize((size_t)cc_total + 4);
        TRY(th_rt_d2h(g->h_iden.data(), g->d_iden.p, (size_t)aln_total * 4, g->stream));
        TRY(th_rt_d2h(g->h_ext.data(), g->d_ext.p, (size_t)nt * 16, g->stream));
        TRY(th_rt_d2h(g->h_cons.data(), g->d_cons_c.p, (size_t)cc_total, g->stream));
        if (g->para.want_cov) { th_room(g->h_cov, (size_t)cc_total + 4); g->h_cov.resize((size_t)cc_total + 4); TRY(th_rt_d2h(g->h_cov.data(), g->d_cov_c.p, (size_t)cc_total * 4, g->stream)); }
        for (int t = 0; t < nt; ++t) g->jobs[t].cons_off = cc_off[t];
        TRY(th_rt_d2h(cnt, g->d_cnt.p, 16 * sizeof(int), g->stream));
        TRY(th_rt_sync(g->stream));
        if (cnt[1] & 64) return th_fail("TH_ALN_MODE=2: a banded alignment differs from the full matrix");
        if (cnt[1]) return th_fail("alignment boundary buffer overflow");
        for (int t = 0; t < nt; ++t) {
            th_job_t &j = g->jobs[t];
            j.cons_len = g->h_cons_len[t]; j.status = g->h_status[t];
            j.max_q_l = g->h_ext[4 * t]; j.max_t_l = g->h_ext[4 * t + 1]; j.max_q_r = g->h_ext[4 * t + 2]; j.max_t_r = g->h_ext[4 * t + 3];
        }
    }
    if (getenv("TH_TRACE")) fprintf(stderr, "[th] %.3f s: run done (this thread so far: device malloc %.3f s for %.2f GB in %d calls, free %.3f s)\n", th_wall_s() - t_run0, g_alloc_s, g_alloc_bytes / 1e9, g_alloc_n, g_free_s);
#ifndef TH_CPU_EMUL
    kt.collect(g->ms_k);
#endif
    g->ms_kernels = 0; g->ms_dominant = 0;
    for (int k = 0; k < TK_N; ++k) { g->ms_kernels += g->ms_k[k]; g->ms_dominant = std::max(g->ms_dominant, g->ms_k[k]); }
    return 0;
}

extern "C" int th_gpu_get(th_gpu_t *g, th_gpu_out_t *out)
{
    out->n_jobs = (int)g->jobs.size(); out->jobs = g->jobs.data(); out->par = g->h_parc.data();
    out->cons = g->h_cons.data(); out->cov = g->para.want_cov ? g->h_cov.data() : nullptr; out->iden = g->h_iden.data();
    out->adapt = g->ad_valid ? g->h_ad_out.data() : nullptr;
    out->ms_kernels = g->ms_kernels; out->ms_dominant = g->ms_dominant;
    for (int k = 0; k < TK_N; ++k) out->ms_kernel[k] = g->ms_k[k];
    return 0;
}

extern "C" long long th_gpu_debug(th_gpu_t *g, int what, void *dst, long long cap)
{
    const void *src = nullptr; long long bytes = 0;
    std::vector<int> tmp;
    if (th_select_device(g)) return -1;
    switch (what) {
    case 0: src = g->d_bseq.p; bytes = g->T; break;
    case 1: src = g->d_hit_n.p; bytes = 4LL * g->n; break;
    case 2: src = g->d_hit_end.p; bytes = 4LL * g->T; break;
    case 3: src = g->d_hit_per.p; bytes = 4LL * g->T; break;
    case 4: src = g->d_score.p; bytes = 4LL * g->T; break;
    case 5: src = g->d_from.p; bytes = 4LL * g->T; break;
    case 6: {   // chain records regrouped by read (emission order inside a read is already ascending end)
        for (int r = 0; r < g->n; ++r)
            for (int c = g->h_read_ch_base[r]; c < g->h_read_ch_base[r] + g->h_read_ch_n[r]; ++c) {
                const ThChainRec &x = g->h_rec[c];
                int v[8] = { x.read, x.cell_off, x.len, x.est_start, x.est_period, g->h_par_dst[c], g->h_par_n[c], 0 };
                tmp.insert(tmp.end(), v, v + 8);
            }
        bytes = 4LL * (long long)tmp.size();
        if (bytes > cap) return -1;
        memcpy(dst, tmp.data(), (size_t)bytes);
        return bytes;
    }
    case 7: src = g->d_cells.p; bytes = 4LL * g->T; break;
    case 8: if (4LL * (g->n + 1) > cap) return -1; memcpy(dst, g->h_roff.data(), 4 * ((size_t)g->n + 1)); return 4LL * (g->n + 1);
    case 9: bytes = 4LL * (long long)g->h_parc.size(); if (bytes > cap) return -1; memcpy(dst, g->h_parc.data(), (size_t)bytes); return bytes;
    default: return -1;
    }
    if (bytes > cap) return -1;
    if (th_rt_d2h(dst, src, (size_t)bytes, g->stream)) return -1;
    if (th_rt_sync(g->stream)) return -1;
    return bytes;
}
