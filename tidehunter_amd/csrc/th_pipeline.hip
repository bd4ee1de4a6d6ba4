// th_pipeline.hip -- host-side orchestration of the device path for one batch of reads.
//
// This is what stands behind the fan-out of mini_tandem() (src/main.c:402-425): instead of
// n_thread pthreads each calling tidehunter_core() per read, the whole batch goes through
//   encode -> self-hits -> chain DP -> chain extraction -> partition -> job tables (device)
//   -> POA consensus -> item lists (device) -> unit/consensus alignments -> small result records back to the host.
// All kernels of one batch run on one HIP stream.  The host never builds a table in the middle of the path: it reads the
// 256-byte counter block three times per batch (after the job build, after the POA, at the end) to size the next launches
// and to learn whether a slow path (an arena that was too small, a POA job that wants more scratch) has to run.
#include <vector>
#include <string>
#include <algorithm>
#include <chrono>
#include <mutex>
#include "th_internal.h"

static double th_wall_s() { using namespace std::chrono; return duration<double>(steady_clock::now().time_since_epoch()).count(); }

// one message per thread: th_run_file_multi drives one handle per device from its own threads
static thread_local std::string g_th_err;
extern "C" const char *th_gpu_error(void) { return g_th_err.c_str(); }
static int th_fail(const std::string &m) { g_th_err = m; fprintf(stderr, "[tidehunter_amd] %s\n", m.c_str()); return -1; }
static int th_launch_failed(const char *kernel, const char *why) { return th_fail(std::string("kernel launch failed: ") + kernel + ": " + why); }

#include "th_dev.h"
#include "th_k_seed.hip"
#include "th_k_chain.hip"
#include "th_k_partition.hip"
#include "th_k_poa.hip"
#include "th_k_jobs.hip"
#include "th_k_align.hip"
#include "th_k_adapter.hip"

#define TH_STAGE_BYTES (16u << 20)
#define TH_POA_SCRATCH_BYTES (96LL << 30)       // upper limit of the POA scratch of one handle (HBM: 288 GB)
// seconds this thread spent in device malloc / free and the bytes it asked for (TH_TRACE prints them per run)
static thread_local double g_alloc_s = 0, g_free_s = 0; static thread_local long long g_alloc_bytes = 0; static thread_local int g_alloc_n = 0;
// One device allocation per handle, made when the handle first learns its batch size (th_gpu_reserve / the first upload), from
// which the buffers below are carved.  hipMalloc while ANOTHER stream of the process is running kernels waits for a kernel
// boundary each time (measured with two workers on one device: 28 calls took 5.9 s next to persistent kernels of 0.1-0.6 s each,
// 0.009 s on an idle device), so nothing allocates piecemeal while kernels run.  A buffer that outgrows the arena falls back to
// its own hipMalloc.  The uploading thread and the running thread of a handle may both carve: the top is guarded.
struct Arena { char *base = nullptr; size_t cap = 0, top = 0; std::mutex mu; };
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
        if (g_arena && g_arena->base) {
            std::lock_guard<std::mutex> lk(g_arena->mu);
            if (g_arena->top + bytes <= g_arena->cap) { p = (T *)(g_arena->base + g_arena->top); g_arena->top += bytes; in_arena = true; cap = want; return 0; }
        }
        t0 = th_wall_s();
        if (th_rt_malloc((void **)&p, want * sizeof(T))) return -1;
        g_alloc_s += th_wall_s() - t0; g_alloc_bytes += (long long)(want * sizeof(T)); ++g_alloc_n;
        cap = want; return 0;
    }
    void release() { if (p && !in_arena) th_rt_free(p); p = nullptr; cap = 0; in_arena = false; }
};

// one uploaded batch: the reads concatenated in HBM and their offsets
struct ThIn {
    int n = 0; long long T = 0; int max_len = 0; long long gsort_total = 0;
    DBuf<uint8_t> d_seq; DBuf<uint32_t> d_roff; DBuf<long long> d_gsort_off;
    th_pvec<uint32_t> h_roff; th_pvec<long long> h_gsort_off;
    th_event_t ev; bool ev_ok = false;          // recorded on the copy stream behind the set's last H2D
};
// the small result records of one run, on the host (pinned: the D2H copies go through the DMA engines)
struct ThOut {
    th_pvec<th_job_t> jobs; th_pvec<int> parc, iden, ext, cov, ad, single; th_pvec<uint8_t> cons;
    int n_jobs = 0; bool ad_valid = false, single_valid = false;
    double ms_kernels = 0, ms_dominant = 0, ms_k[8] = { 0 };
};

struct th_gpu {
    th_gpu_para_t para; ThDevPara dp; int device; th_stream_t stream, cstream;
    th_event_t sync_ev; bool sync_ev_ok = false;      // the blocking event every host wait of this pipeline goes through (th_rt.h)
    ThIn in[2]; ThOut out[2]; int last_in = 0;
    uint8_t *h_stage[2] = { nullptr, nullptr }; th_event_t stage_ev[2]; bool stage_ok = false, stage_used[2] = { false, false }; int stage_cur = 0;   // pinned staging for the upload (2 x TH_STAGE_BYTES)
    DBuf<uint8_t> d_bseq, d_tracked, d_cons, d_cons_c;
    DBuf<uint32_t> d_tmp, d_pk2, d_nmask;         // d_pk2 / d_nmask: the batch's bases 2-bit packed and the mask of the positions that are not A/C/G/T (th_k_encode)
    DBuf<uint16_t> d_xorder; DBuf<int> d_xstart, d_xend, d_xovf;
    DBuf<int> d_cov_c, d_cc_off;                           // compacted results
    DBuf<int> d_hit_end, d_hit_per, d_hit_n, d_score, d_from, d_cells, d_read_ch_base, d_read_ch_n;
    DBuf<unsigned long long> d_gsort, d_eps; DBuf<ThClsEnt> d_cls;      // d_cls: the chain DP's period-class lists, TH_CLS_DUP 12-byte entries per base slot
    DBuf<ThCnt> d_cnt; th_pvec<ThCnt> h_cnt;
    DBuf<ThChainRec> d_rec;
    DBuf<int> d_par, d_par_off, d_par_n, d_par_dst, d_parc, d_bnd, d_scratch, d_cov, d_cons_len, d_status, d_iden, d_ext, d_slb;
    DBuf<int> d_rows, d_list_n, d_list_w, d_list_l, d_sublist;      // per-read / per-job counters of the table builds, POA task lists
    DBuf<ThTask> d_tasks; DBuf<ThUnit> d_units; DBuf<ThAlnItem> d_items, d_defer; DBuf<int> d_defer_s1; DBuf<th_job_t> d_jobs;
    int reserve_n = 0; Arena arena;
    int task_cap = 0, unit_cap = 0, parc_cap = 0;          // table capacities (grown when a batch overflows them)
    bool ad_on = false; int ad_min_cons = 0; ThAdPara h_ad; DBuf<ThAdPara> d_adp; DBuf<int> d_ad; bool adp_up = false;
    bool sc_on = false; ThAdRead *h_adr = nullptr; DBuf<ThAdRead> d_adr; DBuf<int> d_single; bool adr_up = false;     // -s: whole-read adapter searches
    th_pvec<ThTask> h_tasks; th_pvec<int> h_cons_len, h_status, h_sublist;     // slow paths only
    int n_sync = 0;
    // The POA's first attempt at a job sizes its scratch from an estimate of the graph's rows and band cells (th_poa_need_ints) times 1.5; a job that outgrows it is given back
    // and rerun with twice the factor.  When MOST jobs of a class (0 wide, 1 narrow, 2 long) needed reruns in a batch -- 100 kb reads of 50 copies: the graph of 50 units has
    // 5 x the rows of a unit and bands at the ring's limit, 1999 of 2011 jobs succeeded at the third attempt only, 2.4 of the step's 7.5 s were the two failed ones -- the next
    // batch of this handle starts that class at the factor the majority needed (same results: the factor sizes arrays, nothing else).  TH_POA_ADAPT=0: always 1.5.
    int poa_fexp0[3] = { 0, 0, 0 }, poa_adapt = 1;
    long long stat[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };      // cumulative, th_gpu_stats(): 0 jobs left without a consensus, 1 POA jobs given back for a rerun, 2 reruns of stages 2-3 / the job tables (arena or table too small), 3 batches, 4 consensus jobs
    int aln_wdiv = 10, aln_wadd = 24;     // first certified band of the unit alignments: half-width cons_len / wdiv + wadd
    int aln_xpct = 36;                    // flank extensions: band half-width requested from the anti-diagonal engine, percent of min(consensus, flank) (+ 8)
    int aln_probe_small = 1;              // long units whose sequences fit the two-register variant's staging area are probed and finished there (th_aln_class)
    int aln_lpt = -1;                     // first alignment launch walks its list longest flanks first: -1 when it has few items per wave (default), TH_ALN_LPT=0 never, 1 always
    int aln_slb = 1;                      // the POA leaves a score bound per unit alignment (th_poa_unit_bounds) and th_k_align starts long units from it (TH_ALN_SLB=0: off)
    int aln_probe_auto = 1;               // ... unless the batch BEFORE was mostly such units (config 4: there the dedicated launches of round 3 are faster, config 5 gains 6 %)
    int aln_wtol = 16, aln_mode = 1;      // anti-diagonal engine: accepted shortfall of its band against that half-width; 0 = strip engine only, 2 = verify against the full matrix
    // TH_REPLAY=1 -- MEASUREMENT ONLY (tools/host_ceiling.sh; the product never sets it): a pipeline's first batch of a size runs for real and its result records are kept; every
    // later batch of the same size and base count gets them copied back INSTEAD of kernels.  With an input whose batches all hold the same reads (N copies of one set, batch = set) the
    // output is still the right bytes, and reader -> upload -> record arithmetic + formatting -> writer run at whatever rate the HOST sustains: the eight-GPU host budget measured on one box
    int replay = 0; bool replay_ok = false; int replay_n = 0; long long replay_T = 0; ThOut replay_out;
    int waves_part = 8192, waves_poa = 6144, waves_aln = 8192, waves_xw = 2048, waves_xw_small = 3584;   // persistent grids: 256 CUs x 4 SIMDs x resident waves (extraction: 256 CUs x 8 / x 14 waves by their LDS)
};

static int env_int(const char *name, int dflt) { const char *s = getenv(name); return s && *s ? atoi(s) : dflt; }

extern "C" th_gpu_t *th_gpu_create(const th_gpu_para_t *p, int device)
{
    th_gpu *g = new th_gpu();
    g->para = *p;
    g->dp.k = p->k; g->dp.w = p->w; g->dp.w_seq = env_int("TH_MINIMIZER_SEQ", 0) ? 1 : 0; g->dp.sort_bitonic = env_int("TH_SORT_BITONIC", 0) ? 1 : 0; g->dp.part_full = env_int("TH_PART_FULL", 0) ? 1 : 0; g->dp.hpc = p->hpc; g->dp.min_p = p->min_p; g->dp.max_p = p->max_p; g->dp.one_minus_div = p->one_minus_div; g->dp.min_copy = p->min_copy;
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
        if (g->ad_on && p->single_copy) {          // -s: tables over all 256 byte values of a read
            g->sc_on = true; g->h_adr = new ThAdRead(); memset(g->h_adr, 0, sizeof(ThAdRead));
            for (int sr = 0; sr < 4; ++sr) {
                const int m = p->ad_len[sr];
                for (int c = 0; c < 256; ++c)
                    for (int i = 0; i < m; ++i) {
                        if (eq(p->ad_seq[sr][i], (char)c)) g->h_adr->peq[sr][0][c][i / 64] |= 1ull << (i & 63);
                        if (eq(p->ad_seq[sr][m - 1 - i], (char)c)) g->h_adr->peq[sr][1][c][i / 64] |= 1ull << (i & 63);
                    }
            }
        }
    }
    g->para.ad_seq[0] = g->para.ad_seq[1] = g->para.ad_seq[2] = g->para.ad_seq[3] = nullptr;     // (borrowed strings: not kept)
    int ndev = 0;
    if (th_rt_device_count(&ndev) || ndev <= 0) { th_fail("no HIP device: the TideHunter HIP path needs an AMD GPU (no CPU fallback)"); delete g; return nullptr; }
    if (th_rt_set_device(device)) { th_fail("hipSetDevice failed"); delete g; return nullptr; }
    th_rt_prefer_blocking();
    if (th_rt_event_create(&g->sync_ev)) { th_fail("hipEventCreate failed"); delete g; return nullptr; }
    g->sync_ev_ok = true;
    if (th_rt_stream_create(&g->stream) || th_rt_stream_create(&g->cstream)) { th_fail("hipStreamCreate failed"); delete g; return nullptr; }
    g->waves_part = env_int("TH_WAVES_PART", TH_PERSISTENT_WAVES(g->waves_part));
    g->poa_adapt = env_int("TH_POA_ADAPT", 1);
    g->waves_poa = env_int("TH_WAVES_POA", TH_PERSISTENT_WAVES(g->waves_poa));
    g->waves_aln = env_int("TH_WAVES_ALN", TH_PERSISTENT_WAVES(g->waves_aln));
    g->waves_xw = env_int("TH_WAVES_XW", TH_PERSISTENT_WAVES(g->waves_xw));
    g->waves_xw_small = env_int("TH_WAVES_XW_SMALL", TH_PERSISTENT_WAVES(g->waves_xw_small));
    g->aln_wdiv = std::max(1, env_int("TH_ALN_WDIV", g->aln_wdiv)); g->aln_wadd = std::max(0, env_int("TH_ALN_WADD", g->aln_wadd));
    // (TH_ALN_XPROBE: registers of the flank extensions' probe: one since round 4 -- four measured 17-20 ms slower on configs 4 and 5)
    g->aln_wtol = std::max(0, env_int("TH_ALN_WTOL", g->aln_wtol)); g->aln_mode = env_int("TH_ALN_MODE", g->aln_mode); g->aln_xpct = std::min(255, std::max(0, env_int("TH_ALN_XPCT", g->aln_xpct))) | std::min(4, std::max(1, env_int("TH_ALN_XPROBE", 1))) << 8 | (env_int("TH_ALN_PKSTRIP", 1) ? 0 : 1) << 11 | (env_int("TH_ALN_LONGFLANK", 1) ? 0 : 1) << 12;
    g->aln_slb = env_int("TH_ALN_SLB", 1);
    g->aln_lpt = env_int("TH_ALN_LPT", -1);
    g->replay = env_int("TH_REPLAY", 0);      // measurement only (th_gpu::replay)
    g->aln_probe_small = std::min(3, std::max(0, env_int("TH_ALN_PROBE_SMALL", 1))); g->aln_probe_auto = getenv("TH_ALN_PROBE_SMALL") ? 0 : 1;
    return g;
}

extern "C" void th_gpu_destroy(th_gpu_t *g)
{
    if (!g) return;
    (void)th_rt_set_device(g->device);
    (void)th_rt_sync(g->stream); (void)th_rt_sync(g->cstream);
    if (g->sync_ev_ok) th_rt_event_destroy(g->sync_ev);
    g_arena = nullptr;
    for (int s = 0; s < 2; ++s) { g->in[s].d_seq.release(); g->in[s].d_roff.release(); g->in[s].d_gsort_off.release(); if (g->in[s].ev_ok) th_rt_event_destroy(g->in[s].ev); }
    g->d_bseq.release(); g->d_tracked.release(); g->d_cons.release(); g->d_cons_c.release(); g->d_tmp.release(); g->d_pk2.release(); g->d_nmask.release(); g->d_xorder.release(); g->d_xstart.release(); g->d_xend.release(); g->d_xovf.release();
    g->d_cov_c.release(); g->d_cc_off.release(); g->d_hit_end.release(); g->d_hit_per.release(); g->d_hit_n.release(); g->d_score.release(); g->d_from.release(); g->d_cells.release();
    g->d_read_ch_base.release(); g->d_read_ch_n.release(); g->d_gsort.release(); g->d_eps.release(); g->d_cls.release(); g->d_cnt.release(); g->d_rec.release();
    g->d_par.release(); g->d_par_off.release(); g->d_par_n.release(); g->d_par_dst.release(); g->d_parc.release(); g->d_bnd.release(); g->d_scratch.release();
    g->d_cov.release(); g->d_cons_len.release(); g->d_status.release(); g->d_iden.release(); g->d_ext.release(); g->d_slb.release();
    g->d_rows.release(); g->d_list_n.release(); g->d_list_w.release(); g->d_list_l.release(); g->d_sublist.release();
    g->d_tasks.release(); g->d_units.release(); g->d_items.release(); g->d_defer.release(); g->d_defer_s1.release(); g->d_jobs.release(); g->d_adp.release(); g->d_ad.release(); g->d_adr.release(); g->d_single.release(); delete g->h_adr;
    if (g->stage_ok) for (int b = 0; b < 2; ++b) { th_rt_host_free(g->h_stage[b]); th_rt_event_destroy(g->stage_ev[b]); }
    if (g->arena.base) th_rt_free(g->arena.base);
    th_rt_stream_destroy(g->stream); th_rt_stream_destroy(g->cstream);
    delete g;
}

#define TRY(x) do { if ((x) != 0) return th_fail(std::string("device runtime call failed: ") + #x); } while (0)

// A handle belongs to one device, but the calling thread's current HIP device may have changed since th_gpu_create
// (torch.cuda.set_device, a second handle for another GPU, a call from another thread): every entry point re-selects it,
// otherwise allocations and launches would land on the wrong GPU.
static int th_select_device(const th_gpu *g)
{
    if (th_rt_set_device(g->device)) return th_fail("hipSetDevice(" + std::to_string(g->device) + ") failed");
    g_arena = const_cast<Arena *>(&g->arena);
    return 0;
}

// The handle's arena, sized for batches of up to n reads / T bases: ~106 bytes per base of per-position arrays (36 of them the chain DP's period-class lists, round 6) and both input sets,
// the POA scratch of the full persistent grid at ~100 ints per base of the average read (3.7 MB per wave at the headline shape;
// long-period batches need more and take their scratch from hipMalloc), and room for the per-job arrays.
static size_t th_arena_want(const th_gpu *g, int n, long long T)
{
    const double avg = (double)T / n;
    const size_t poa = (size_t)std::min<double>((double)TH_POA_SCRATCH_BYTES / 2, 100.0 * avg * 4.0 * std::min(n, g->waves_poa) * 1.3);
    return (size_t)(106.0 * (double)T) + (size_t)n * 1280 + poa + (256u << 20);             // (buffers carry 12.5 % slack; consensus jobs, items, par_pos: ~14 bytes per base)
}
// does a further pipeline of this batch size fit next to what the device already holds?  (a third of the free memory at most:
// buffers that outgrow the arena still need room)
extern "C" int th_gpu_room_for(th_gpu_t *g, int n, long long T)
{
    if (th_select_device(g)) return 0;
    if (n <= 0 || T <= 0) return 1;
    size_t fr = 0, tot = 0;
    if (th_rt_mem_info(&fr, &tot) != 0) return 0;
    return th_arena_want(g, n, T) <= fr / 3;
}
static int th_arena_create(th_gpu *g, int n, long long T)
{
    if (n <= 0 || T <= 0) return 0;
    {
        std::lock_guard<std::mutex> lk(g->arena.mu);
        if (g->arena.base) return 0;
    }
    size_t want = th_arena_want(g, n, T);
    size_t fr = 0, tot = 0;
    if (th_rt_mem_info(&fr, &tot) == 0 && want > fr / 2) want = fr / 2;     // never more than half of what is free
    if (want < TH_ARENA_MIN_BYTES) return 0;              // tiny batches (tests): plain allocations
    const double t0 = th_wall_s();
    void *p = nullptr;
    if (th_rt_malloc(&p, want)) return 0;                 // no arena: every buffer allocates for itself
    g_alloc_s += th_wall_s() - t0; g_alloc_bytes += (long long)want; ++g_alloc_n;
    std::lock_guard<std::mutex> lk(g->arena.mu);
    if (g->arena.base) { th_rt_free(p); return 0; }
    g->arena.base = (char *)p; g->arena.cap = want; g->arena.top = 0;
    return 0;
}

// the per-position work arrays of a run over T bases / n reads
static int th_ensure_work(th_gpu *g, int n, size_t padT)
{
    TRY(g->d_bseq.ensure(padT)); TRY(g->d_tmp.ensure(padT)); TRY(g->d_hit_end.ensure(padT)); TRY(g->d_hit_per.ensure(padT)); TRY(g->d_hit_n.ensure(n));
    TRY(g->d_score.ensure(padT)); TRY(g->d_from.ensure(padT)); TRY(g->d_cells.ensure(padT)); TRY(g->d_tracked.ensure(padT));
    TRY(g->d_xorder.ensure(padT)); TRY(g->d_xstart.ensure(padT)); TRY(g->d_xend.ensure(padT)); TRY(g->d_eps.ensure(padT));
    TRY(g->d_read_ch_base.ensure(n)); TRY(g->d_read_ch_n.ensure(n)); TRY(g->d_rows.ensure((size_t)n * TH_JOB_COLS));
    TRY(g->d_cnt.ensure(1));
    return 0;
}

// Sizes every per-read / per-base buffer for batches of up to n reads and T bases, so that a run of growing batches (the file
// driver starts with small ones) does not free and re-allocate gigabytes at every step.  Buffers whose size depends on the
// stage results (consensus slots, POA scratch) still grow on demand.
extern "C" int th_gpu_reserve(th_gpu_t *g, int n, long long T)
{
    TRY(th_select_device(g));
    if (n <= 0 || T <= 0) return 0;
    if (T >= (1LL << 31)) T = (1LL << 31) - 64;
    TRY(th_arena_create(g, n, T));
    const size_t padT = (size_t)((T + 15) & ~15LL);
    for (int s = 0; s < 2; ++s) { TRY(g->in[s].d_seq.ensure(padT)); TRY(g->in[s].d_roff.ensure((size_t)n + 1)); TRY(g->in[s].d_gsort_off.ensure((size_t)n)); }
    TRY(th_ensure_work(g, n, padT));
    TRY(g->d_rec.ensure((size_t)n * 8 + 4096));
    g->reserve_n = n;
    return 0;
}

extern "C" int th_gpu_upload(th_gpu_t *g, int slot, const char *const *seqs, const int *lens, int n, int wait)
{
    TRY(th_select_device(g));
    ThIn &in = g->in[slot & 1];
    in.n = n; th_room(in.h_roff, (size_t)n + 1); in.h_roff.assign((size_t)n + 1, 0);
    long long T = 0; int mx = 0;
    for (int i = 0; i < n; ++i) { in.h_roff[i] = (uint32_t)T; T += lens[i]; mx = std::max(mx, lens[i]); }
    if (T >= (1LL << 31)) return th_fail("batch holds 2^31 bases or more; split it");
    if (mx >= (1 << 28)) return th_fail("a read of 2^28 bases or more is not supported (32-bit period arithmetic in th_k_chain_dp)");
    in.h_roff[n] = (uint32_t)T; in.T = T; in.max_len = mx;
    size_t padT = (size_t)((T + 15) & ~15LL);
    // reads whose sort does not fit the LDS buffers get a segment of HBM scratch
    th_room(in.h_gsort_off, (size_t)n); in.h_gsort_off.assign((size_t)n, -1); long long gs = 0;
    for (int i = 0; i < n; ++i) {
        int ne = lens[i] - g->dp.k + 1; if (ne < 1) continue;
        long long N = 1; while (N < ne) N <<= 1;
        if (N > TH_XSORT_LDS) { in.h_gsort_off[i] = gs; gs += N; }
    }
    in.gsort_total = gs;
    TRY(th_arena_create(g, n, T));                        // (no-op once the handle has one)
    TRY(in.d_seq.ensure(padT)); TRY(in.d_roff.ensure((size_t)n + 1)); TRY(in.d_gsort_off.ensure((size_t)n));
    if (!in.ev_ok) { TRY(th_rt_event_create(&in.ev)); in.ev_ok = true; }
    // The reads go to HBM through two small pinned staging buffers: while one is on its way (async H2D) the host concatenates
    // the next reads into the other.  (One pinned buffer for the whole batch meant pinning ~1 GB before the first byte moved:
    // 0.13-0.19 s per handle, and the copy could not overlap the concatenation.)
    if (!g->stage_ok) {
        for (int b = 0; b < 2; ++b) { TRY(th_rt_host_alloc((void **)&g->h_stage[b], TH_STAGE_BYTES)); TRY(th_rt_event_create(&g->stage_ev[b])); }
        g->stage_ok = true;
    }
    {
        size_t done = 0, fill = 0; int cur = g->stage_cur;
        if (g->stage_used[cur]) TRY(th_rt_event_wait(g->stage_ev[cur]));         // (a copy of the previous upload may still be reading it)
        auto flush = [&]() -> int {
            if (!fill) return 0;
            TRY(th_rt_h2d(in.d_seq.p + done, g->h_stage[cur], fill, g->cstream));
            TRY(th_rt_event_record(g->stage_ev[cur], g->cstream));
            g->stage_used[cur] = true; done += fill; fill = 0; cur ^= 1;
            if (g->stage_used[cur]) TRY(th_rt_event_wait(g->stage_ev[cur]));      // the buffer we are about to refill must have left
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
        g->stage_cur = cur;
    }
    TRY(th_rt_h2d(in.d_roff.p, in.h_roff.data(), ((size_t)n + 1) * 4, g->cstream));
    TRY(th_rt_h2d(in.d_gsort_off.p, in.h_gsort_off.data(), (size_t)n * 8, g->cstream));
    TRY(th_rt_event_record(in.ev, g->cstream));
    if (wait) TRY(th_rt_event_wait(in.ev));
    return 0;
}

// per-kernel device time: timing events recorded on the pipeline's own stream around each launch
enum { TK_SEED = 0, TK_CHAIN_DP, TK_CHAIN_X, TK_PART, TK_POA, TK_ALIGN, TK_N };
struct KTimer {
    std::vector<th_timer_t> ev[TK_N]; th_stream_t s;
    void begin(int k) { th_timer_t e; th_rt_timer_begin(&e, s); ev[k].push_back(e); }
    void end(int k) { th_rt_timer_end(&ev[k].back(), s); }
    void collect(double *ms) {
        for (int k = 0; k < TK_N; ++k) {
            ms[k] = 0;
            for (th_timer_t &e : ev[k]) { ms[k] += th_rt_timer_ms(&e); th_rt_timer_destroy(&e); }
            ev[k].clear();
        }
    }
    // a run that returns early (arena overflow, a failed launch, ...) never reaches collect(): release its events here
    ~KTimer() { for (int k = 0; k < TK_N; ++k) for (th_timer_t &e : ev[k]) th_rt_timer_destroy(&e); }
};
#define KT_BEGIN(k) kt.begin(k)
#define KT_END(k) kt.end(k)

// the counter block comes back (pinned, 256 bytes) and the stream is drained: the pipeline's only kind of host synchronisation
static int th_fetch_cnt(th_gpu *g, ThCnt **out)
{
    g->h_cnt.resize(1);
    TRY(th_rt_d2h(g->h_cnt.data(), g->d_cnt.p, sizeof(ThCnt), g->stream));
    TRY(th_rt_sync_event(g->stream, g->sync_ev));
    ++g->n_sync;
    *out = g->h_cnt.data();
    return 0;
}
extern "C" int th_gpu_syncs(th_gpu_t *g) { return g->n_sync; }
extern "C" int th_gpu_stats(th_gpu_t *g, long long *out, int n) { for (int i = 0; i < n && i < 8; ++i) out[i] = g->stat[i]; return 5; }

// one launch of the POA kernel over a task list (both variants write consensus / length / status at the task's own index)
static int th_poa_launch(th_gpu *g, KTimer &kt, const ThIn &in, const int *list, int n_list, int cls /* 1 int16, 2 relative int16 (long units), 0 32-bit */, int max_unit, long long need, int wv, int *queue, int poa_flags)
{
    if (n_list <= 0) return 0;
    KT_BEGIN(TK_POA);
    // (the default scoring -- src/abpoa_cons.c:12-28 -- has its own instantiation of the two int16 variants with the six values compiled in; TH_POA_DEFSC=0: the generic one)
    const ThDevPara &dp = g->dp;
    const bool defsc = dp.match == 2 && dp.mismatch == 4 && dp.go1 == 4 && dp.ge1 == 2 && dp.go2 == 24 && dp.ge2 == 1 && env_int("TH_POA_DEFSC", 1) != 0;
#define POA_LAUNCH(NS, QC, RT, ...) TH_LAUNCH((NS::th_k_poa<QC, RT, ##__VA_ARGS__>), wv, TH_POA_BLOCK, g->stream, (const ThTask *)g->d_tasks.p, list, n_list, (const ThUnit *)g->d_units.p, (const uint32_t *)in.d_roff.p, \
                          (const uint8_t *)g->d_bseq.p, g->dp, g->d_scratch.p, need, g->d_cons.p, g->d_cov.p, g->d_cons_len.p, g->d_status.p, queue, poa_flags, g->aln_slb ? g->d_slb.p : (int *)nullptr)
    if (cls == 1) { if (defsc) POA_LAUNCH(thpoa_n, TH_POA_QCAP, short, true); else POA_LAUNCH(thpoa_n, TH_POA_QCAP, short); }
    else if (cls == 2) { if (defsc) POA_LAUNCH(thpoa_l, TH_POA_QCAP_LONG, short, true); else POA_LAUNCH(thpoa_l, TH_POA_QCAP_LONG, short); }
    else if (max_unit <= TH_POA_QCAP_WIDE) POA_LAUNCH(thpoa_n, TH_POA_QCAP_WIDE, int);
    else POA_LAUNCH(thpoa_n, 0, int);
#undef POA_LAUNCH
    KT_END(TK_POA);
    return 0;
}
static int th_poa_waves(const th_gpu *g, long long n_list, long long need) { return (int)std::min<long long>(std::min<long long>(n_list, g->waves_poa), std::max<long long>(1, (TH_POA_SCRATCH_BYTES) / (std::max<long long>(need, 1) * 4))); }

// ---- slow path: jobs the POA gave back.  Status 1 / 2: its scratch estimate was too small (retried with twice the cell factor);
// 7: the int16 variant's scores may overflow (rerun on the 32-bit variant); 3 / 4: the adaptive band came out empty or lost the
// global path (once more with full rows on the 32-bit variant, as oracle/tho_poa.c does).  5 / 6 (in-degree above 255, consensus
// longer than its slot) stay failed.  None of them may fail the batch: abPOA has no such failure mode, and one pathological read
// must not stop a multi-million-read run -- a job that cannot get its scratch ends up without a consensus and a warning.
static int th_poa_retries(th_gpu *g, KTimer &kt, const ThIn &in, int nt, int poa_test_fail)
{
    th_room(g->h_tasks, (size_t)nt); g->h_tasks.resize(nt); th_room(g->h_status, (size_t)nt); g->h_status.resize(nt);
    TRY(th_rt_d2h(g->h_tasks.data(), g->d_tasks.p, (size_t)nt * sizeof(ThTask), g->stream));
    TRY(th_rt_d2h(g->h_status.data(), g->d_status.p, (size_t)nt * 4, g->stream));
    TRY(th_rt_sync_event(g->stream, g->sync_ev));
    if (getenv("TH_TRACE")) {
        int c[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int t = 0; t < nt; ++t) ++c[g->h_status[t] & 7];
        fprintf(stderr, "[th] poa gave back: %d graph / scratch too small (1), %d band cells overflow (2), %d empty band (3), %d band lost the path (4), %d int16 range (7); failed for good: %d + %d\n", c[1], c[2], c[3], c[4], c[7], c[5], c[6]);
    }
    auto run_list = [&](const std::vector<int> &grp, int cls, bool noband, double factor, bool *ran) -> int {
        const bool wide = cls == 0;
        *ran = false;
        if (grp.empty()) return 0;
        long long need = 1024; int max_unit = 0;
        for (int t : grp) { const ThTask &k = g->h_tasks[t]; need = std::max(need, th_poa_need_ints(k.n_units, k.sum_len, k.max_len, factor, noband, wide)); max_unit = std::max(max_unit, k.max_len); }
        if (need * 4 > TH_POA_SCRATCH_BYTES) return 0;                       // even one wave cannot have it: the jobs keep their status
        const int wv = th_poa_waves(g, (long long)grp.size(), need);
        if (g->d_scratch.ensure((size_t)(need * wv))) { g_th_err.clear(); th_rt_forget_error(); return 0; }      // (no memory for it: likewise)
        th_room(g->h_sublist, grp.size()); g->h_sublist.assign(grp.begin(), grp.end());
        TRY(g->d_sublist.ensure(grp.size()));
        TRY(th_rt_h2d(g->d_sublist.p, g->h_sublist.data(), grp.size() * 4, g->stream));
        TH_LAUNCH(th_k_cnt_set, 1, 64, g->stream, &g->d_cnt.p->q_poa_w, 0, (int *)nullptr, 0, (int *)nullptr, 0, (int *)nullptr, 0);
        TRY(th_poa_launch(g, kt, in, g->d_sublist.p, (int)grp.size(), cls, max_unit, need, wv, &g->d_cnt.p->q_poa_w, (noband ? 1 : 0) | th_poa_fexp(factor) << 1 | poa_test_fail << 8));
        TRY(th_rt_d2h(g->h_status.data(), g->d_status.p, (size_t)nt * 4, g->stream));
        TRY(th_rt_sync_event(g->stream, g->sync_ev));
        *ran = true;
        return 0;
    };
    double factor = 1.5;
    int n_cls[3] = { 0, 0, 0 }, n_again[3][8] = { { 0 } };          // jobs per class as the batch started; jobs of the class rerun at attempt a
    for (int t = 0; t < nt; ++t) ++n_cls[std::min(std::max(g->h_tasks[t].narrow, 0), 2)];
    const double f0[3] = { (double)(1 << g->poa_fexp0[0]), (double)(1 << g->poa_fexp0[1]), (double)(1 << g->poa_fexp0[2]) };      // the factor the first attempt ran at, over 1.5
    for (int attempt = 0; attempt < 8; ++attempt, factor *= 2) {
        std::vector<int> narrow, wide, lng;
        for (int t = 0; t < nt; ++t) {
            const int st = g->h_status[t];
            if (st == 7) wide.push_back(t);                                             // (has not run on this variant yet: same factor)
            else if ((st == 1 || st == 2) && attempt > 0) (g->h_tasks[t].narrow == 1 ? narrow : g->h_tasks[t].narrow == 2 ? lng : wide).push_back(t);
        }
        if (attempt == 0 && wide.empty()) continue;
        if (narrow.empty() && wide.empty() && lng.empty()) break;
        bool ran_n = false, ran_w = false, ran_l = false;
        for (int t : wide) if (g->h_status[t] == 7) g->h_tasks[t].narrow = 0;             // from now on a 32-bit job
        if (attempt > 0) { n_again[1][attempt] = (int)narrow.size(); n_again[2][attempt] = (int)lng.size(); n_again[0][attempt] = (int)wide.size(); }
        TRY(run_list(narrow, 1, false, factor * f0[1], &ran_n));
        TRY(run_list(lng, 2, false, factor * f0[2], &ran_l));
        TRY(run_list(wide, 0, false, factor * f0[0], &ran_w));
        if (!ran_n && !ran_w && !ran_l) break;
    }
    if (g->poa_adapt && !poa_test_fail)
        for (int c = 0; c < 3; ++c) {
            int up = 0;
            for (int a = 1; a < 8; ++a) if (n_cls[c] >= 16 && 2 * n_again[c][a] > n_cls[c]) up = a;          // most of the class's jobs got there
            if (up) { g->poa_fexp0[c] = std::min(3, g->poa_fexp0[c] + up); if (getenv("TH_TRACE")) fprintf(stderr, "[th] poa class %d: the next batch starts at factor 1.5 x %d\n", c, 1 << g->poa_fexp0[c]); }
        }
    {   // jobs whose adaptive band failed: one wave at a time if that is what the scratch allows
        factor = 1.0;
        for (int attempt = 0; attempt < 3; ++attempt, factor *= 2) {
            std::vector<int> nb;
            for (int t = 0; t < nt; ++t) { const int st = g->h_status[t]; if (st == 3 || st == 4 || (attempt > 0 && (st == 1 || st == 2) && g->h_tasks[t].noband)) nb.push_back(t); }
            if (nb.empty()) break;
            for (int t : nb) g->h_tasks[t].noband = 1;                                    // (marks the jobs of the no-band rerun)
            bool ran = false;
            TRY(run_list(nb, 0, true, factor, &ran));
            if (!ran) break;
        }
    }
    int left = 0;
    for (int t = 0; t < nt; ++t) if (g->h_status[t] != 0) ++left;
    if (left) {
        // whatever is left stays without a consensus: status 5 tells the item build so (it only counts 1-4 and 7 as "retry")
        for (int t = 0; t < nt; ++t) if (g->h_status[t] != 0 && g->h_status[t] != 5 && g->h_status[t] != 6) g->h_status[t] = 5;
        TRY(th_rt_h2d(g->d_status.p, g->h_status.data(), (size_t)nt * 4, g->stream));
        TRY(th_rt_sync_event(g->stream, g->sync_ev));
    }
    return 0;
}

extern "C" int th_gpu_run(th_gpu_t *g, int in_slot, int out_slot)
{
    TRY(th_select_device(g));
    const ThIn &in = g->in[in_slot & 1]; ThOut &out = g->out[out_slot & 1];
    g->last_in = in_slot & 1;
    const int n = in.n; const long long T = in.T; const size_t padT = (size_t)((T + 15) & ~15LL);
    const double t_run0 = th_wall_s();
    const bool trace = getenv("TH_TRACE") != nullptr;
    out.n_jobs = 0; out.ad_valid = false; out.single_valid = false; out.ms_kernels = out.ms_dominant = 0; g->n_sync = 0;
    for (int k = 0; k < 8; ++k) out.ms_k[k] = 0;
    out.jobs.clear(); out.parc.clear();
    if (n == 0) return 0;
    if (g->replay && g->replay_ok && n == g->replay_n && T == g->replay_T) { out = g->replay_out; g->stat[3] += 1; g->stat[4] += out.n_jobs; return 0; }      // (measurement only: see th_gpu::replay)
    KTimer kt; kt.s = g->stream;
    int rec_cap = std::max(n, g->reserve_n) * 8 + 4096;
    if (getenv("TH_REC_CAP")) rec_cap = std::max(1, atoi(getenv("TH_REC_CAP")));      // test knob: forces the arena-growth path
    TRY(th_ensure_work(g, n, padT));
    TRY(g->d_gsort.ensure((size_t)in.gsort_total));
    TRY(g->d_xovf.ensure((size_t)std::min(n, std::max(g->waves_xw, g->waves_xw_small)) * 7 * (TH_TOPN - TH_TOPL)));
    TRY(g->d_rec.ensure(rec_cap));
    ThCnt *dc = g->d_cnt.p, *cnt = nullptr;
    if (in.ev_ok) TRY(th_rt_stream_wait_event(g->stream, in.ev));       // the batch's H2D copies (copy stream) before the first kernel

    // ---- stages 0-2: encode, self-hits, chain DP, candidate sort ----
    TH_LAUNCH(th_k_cnt_zero, 1, 64, g->stream, dc);
    const int enc_blocks = (int)std::min<long long>(2048, (T / 16 + 255) / 256 + 1);
    KT_BEGIN(TK_SEED);
    TRY(g->d_pk2.ensure(padT / 16 + 8)); TRY(g->d_nmask.ensure(padT / 32 + 8));
    TH_LAUNCH(th_k_encode, enc_blocks, 256, g->stream, (const uint8_t *)in.d_seq.p, g->d_bseq.p, (long long)padT, g->d_pk2.p, (uint16_t *)g->d_nmask.p);
#define HITS_LAUNCH(W, PB, CAP, GS) TH_LAUNCH((th_k_hits<W, PB, CAP>), n, TH_SEED_BLOCK, g->stream, (const uint8_t *)g->d_bseq.p, (const uint32_t *)in.d_roff.p, g->dp, \
                  g->d_hit_end.p, g->d_hit_per.p, g->d_hit_n.p, g->d_tmp.p, GS, (const long long *)in.d_gsort_off.p, &dc->err, &dc->min_fb, \
                  (const uint32_t *)(env_int("TH_SEED_BYTES", 0) ? nullptr : g->d_pk2.p), (const uint32_t *)g->d_nmask.p)
    // 32-bit sort words when key << PB | pos stays below the all-ones sentinel and every read sorts in LDS
    if (g->dp.k <= 9 && in.max_len < 16384) HITS_LAUNCH(uint32_t, 14, 16384, (uint32_t *)nullptr);
    else if (g->dp.k <= 8 && in.max_len < 32768) HITS_LAUNCH(uint32_t, 15, 32768, (uint32_t *)nullptr);
    else HITS_LAUNCH(unsigned long long, 32, TH_SORT_LDS, g->d_gsort.p);
#undef HITS_LAUNCH
    KT_END(TK_SEED); KT_BEGIN(TK_CHAIN_DP);
    if (in.max_len < 65536 && !getenv("TH_CHAIN_UNPACKED")) {      // (the environment knob exists for the CPU tests)
#define CHAIN_LAUNCH(RG) TH_LAUNCH((th_k_chain_dp<true, RG>), (n + 3) / 4, TH_CHAIN_BLOCK, g->stream, (const uint32_t *)in.d_roff.p, n, g->dp, \
                  (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, g->d_score.p, g->d_from.p, g->d_eps.p)
        const int ring = env_int("TH_CHAIN_RING", 512);
        // (round 6: per-read period-class lists -- th_k_chain.hip, CLS -- with the default ring; TH_CHAIN_CLS=0: every head scans the hit array, as before)
        if (ring == 512 && env_int("TH_CHAIN_CLS", 1)) {
            TRY(g->d_cls.ensure((size_t)TH_CLS_DUP * padT));
            TH_LAUNCH((th_k_chain_dp<true, 512, true>), (n + 3) / 4, TH_CHAIN_BLOCK, g->stream, (const uint32_t *)in.d_roff.p, n, g->dp,
                      (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, g->d_score.p, g->d_from.p, g->d_eps.p, g->d_cls.p);
        } else
        if (ring >= 1024) CHAIN_LAUNCH(1024); else if (ring >= 768) CHAIN_LAUNCH(768); else if (ring >= 640) CHAIN_LAUNCH(640); else if (ring >= 512) CHAIN_LAUNCH(512); else CHAIN_LAUNCH(0);
#undef CHAIN_LAUNCH
    } else
        TH_LAUNCH((th_k_chain_dp<false, 0>), (n + 3) / 4, TH_CHAIN_BLOCK, g->stream, (const uint32_t *)in.d_roff.p, n, g->dp,
                  (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, g->d_score.p, g->d_from.p, (unsigned long long *)nullptr);
    KT_END(TK_CHAIN_DP); KT_BEGIN(TK_CHAIN_X);
    TH_LAUNCH(th_k_chain_sort, n, TH_XS_BLOCK, g->stream, (const uint32_t *)in.d_roff.p, g->dp,
              (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, (const int *)g->d_score.p,
              g->d_xorder.p, g->d_xstart.p, g->d_xend.p);
    KT_END(TK_CHAIN_X);
    if (g->sc_on) {      // -s: the adapters in every whole read (independent of everything else: it only needs the uploaded bytes)
        TRY(g->d_adp.ensure(1)); TRY(g->d_adr.ensure(1)); TRY(g->d_single.ensure((size_t)n * 40));
        if (!g->adp_up) { TRY(th_rt_h2d(g->d_adp.p, &g->h_ad, sizeof(ThAdPara), g->stream)); g->adp_up = true; }
        if (!g->adr_up) { TRY(th_rt_h2d(g->d_adr.p, g->h_adr, sizeof(ThAdRead), g->stream)); g->adr_up = true; }
        KT_BEGIN(TK_ALIGN);
        TH_LAUNCH(th_k_adapter_reads, (n * 4 + 63) / 64, 64, g->stream, n, (const uint32_t *)in.d_roff.p, (const uint8_t *)in.d_seq.p, (const ThAdPara *)g->d_adp.p, (const ThAdRead *)g->d_adr.p, g->d_single.p);
        KT_END(TK_ALIGN);
        th_room(out.single, (size_t)n * 40); out.single.resize((size_t)n * 40);
        TRY(th_rt_d2h(out.single.data(), g->d_single.p, (size_t)n * 160, g->stream));
        out.single_valid = true;
    }

    // ---- stages 2-3 and the job tables, enqueued in one go; the counter block tells afterwards whether an arena was too small.
    // Chain extraction fills an arena of chain records with one atomic per read (8 records per read on average is plenty); the
    // partition carves a region of cap_mul * (2 * cells + 16) boundaries per chain from a second arena (chains own disjoint cell
    // ranges of the per-position arrays, so T + 16 per record holds every batch whose chains cover at most half of its bases;
    // a single chain that needs more than its region -- the reference allows up to the read length, src/partition.c:173 -- makes the
    // partition rerun with larger regions); the tables of the job build start at sizes that cover periods down to ~16 bases.
    if (getenv("TH_JOB_CAP")) g->task_cap = g->unit_cap = g->parc_cap = std::max(1, atoi(getenv("TH_JOB_CAP")));      // test knob: forces the table-growth path
    else { g->task_cap = std::max(g->task_cap, rec_cap); g->unit_cap = std::max<long long>(g->unit_cap, std::min<long long>(0x7fffff00LL, T / 16 + 64LL * n + 4096));
    g->parc_cap = std::max<long long>(g->parc_cap, std::min<long long>(0x7fffff00LL, T / 8 + 64LL * n + 4096)); }
    int cap_mul = getenv("TH_PAR_MUL") ? 0 : 1;                          // (TH_PAR_MUL: test knob, a first pass with regions of four entries)
    long long par_cap = std::min<long long>(0x7fffff00LL, T + 16LL * rec_cap + 64);
    const long long bnd_part = (((long long)in.max_len + 128) * 6 + 3) & ~3LL;   // 16-byte aligned per-wave segments
    bool redo_extract = true, redo_part = true;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 12) return th_fail("stage 2-3 arenas still too small after 12 attempts");
        if (redo_extract) {
            KT_BEGIN(TK_CHAIN_X);
            TH_LAUNCH(th_k_chain_extract, n, TH_XBLOCK, g->stream, (const uint32_t *)in.d_roff.p, g->dp,
                      (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p,
                      (const int *)g->d_score.p, (const int *)g->d_from.p, g->d_tracked.p, g->d_cells.p,
                      g->d_gsort.p, (const long long *)in.d_gsort_off.p, g->d_rec.p, rec_cap, &dc->rec,
                      g->d_read_ch_base.p, g->d_read_ch_n.p, &dc->err);
#define XW_LAUNCH(FMAX, FMIN, WV, Q) TH_LAUNCH((th_k_chain_extract_wave<FMAX, FMIN>), std::min(n, (WV)), TH_XW_BLOCK, g->stream, (const uint32_t *)in.d_roff.p, n, g->dp, \
                      (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_hit_n.p, \
                      (const int *)g->d_score.p, (const int *)g->d_from.p, g->d_cells.p, \
                      (const uint16_t *)g->d_xorder.p, (const int *)g->d_xstart.p, (const int *)g->d_xend.p, g->d_xovf.p, g->d_rec.p, rec_cap, &dc->rec, \
                      g->d_read_ch_base.p, g->d_read_ch_n.p, &dc->err, (Q))
            // (two instantiations, one after the other on the stream: they share the overflow segments, which are indexed by the wave)
            XW_LAUNCH(TH_XFAST_SMALL, 0, g->waves_xw_small, &dc->q_xw);
            XW_LAUNCH(TH_XFAST_MAX, TH_XFAST_SMALL, g->waves_xw, &dc->q_xw2);
#undef XW_LAUNCH
            KT_END(TK_CHAIN_X);
        }
        if (redo_part) {
            const int wp = std::min(rec_cap, g->waves_part);
            TRY(g->d_par_off.ensure(rec_cap)); TRY(g->d_par_n.ensure(rec_cap)); TRY(g->d_par_dst.ensure(rec_cap));
            TRY(g->d_bnd.ensure((size_t)(bnd_part * wp))); TRY(g->d_par.ensure((size_t)par_cap));
            KT_BEGIN(TK_PART);
            TH_LAUNCH(th_k_partition, wp, TH_PART_BLOCK, g->stream, (const ThChainRec *)g->d_rec.p, (const int *)&dc->rec, (const uint32_t *)in.d_roff.p,
                      (const uint8_t *)g->d_bseq.p, (const int *)g->d_hit_end.p, (const int *)g->d_hit_per.p, (const int *)g->d_cells.p, g->dp,
                      g->d_par.p, (int)par_cap, &dc->par_top, g->d_par_off.p, g->d_par_n.p, g->d_bnd.p, bnd_part, &dc->err, cap_mul, &dc->q_part);
            KT_END(TK_PART);
        }
        {
            TRY(g->d_tasks.ensure(g->task_cap)); TRY(g->d_jobs.ensure(g->task_cap)); TRY(g->d_list_n.ensure(g->task_cap)); TRY(g->d_list_w.ensure(g->task_cap)); TRY(g->d_list_l.ensure(g->task_cap));
            TRY(g->d_units.ensure((size_t)g->unit_cap + 1)); TRY(g->d_parc.ensure((size_t)g->parc_cap + 1));
            ThJobTables J; J.roff = in.d_roff.p; J.read_ch_base = g->d_read_ch_base.p; J.read_ch_n = g->d_read_ch_n.p;
            J.par = g->d_par.p; J.par_off = g->d_par_off.p; J.par_n = g->d_par_n.p; J.par_dst = g->d_par_dst.p; J.parc = g->d_parc.p;
            J.tasks = g->d_tasks.p; J.units = g->d_units.p; J.jobs = g->d_jobs.p; J.list_n = g->d_list_n.p; J.list_w = g->d_list_w.p; J.list_l = g->d_list_l.p; J.cls_mask = (env_int("TH_POA_LONG", 1) ? 3 : 1) | (env_int("TH_POA_FORCE_LONG", 0) ? 4 : 0); for (int c_ = 0; c_ < 3; ++c_) J.fexp0[c_] = g->poa_fexp0[c_];
            J.task_cap = g->task_cap; J.unit_cap = g->unit_cap; J.parc_cap = g->parc_cap;
            ThScanCaps caps; caps.on = 1; for (int k = 0; k < 8; ++k) caps.v[k] = 0x7fffffff;
            caps.v[0] = g->parc_cap; caps.v[1] = g->task_cap; caps.v[2] = g->unit_cap;
            KT_BEGIN(TK_PART);
            TH_LAUNCH(th_k_jobs_count, (n + 255) / 256, 256, g->stream, n, J, g->dp, g->para.only_unit, g->d_rows.p, dc);
            TH_LAUNCH((th_k_scan_rows<TH_JOB_COLS>), 1, 1024, g->stream, (const int *)nullptr, n, g->d_rows.p, &dc->par_total, caps, &dc->err);
            TH_LAUNCH(th_k_jobs_emit, (n + 255) / 256, 256, g->stream, n, J, g->dp, g->para.only_unit, (const int *)g->d_rows.p, dc);
            KT_END(TK_PART);
        }
        TRY(th_fetch_cnt(g, &cnt));                                                     // ---- host synchronisation 1 of 3 ----
        redo_extract = redo_part = false;
        if (cnt->err & TH_E_MINIMIZER) return th_fail("minimizer selection repeated a word or overflowed its buffer");
        if (cnt->err & TH_E_REC_OVERFLOW) {
            if ((long long)rec_cap * 4 > (1LL << 30)) return th_fail("chain record arena overflow");
            rec_cap = std::max(rec_cap * 4, cnt->rec + 4096); g->task_cap = std::max(g->task_cap, rec_cap);
            par_cap = std::max(par_cap, std::min<long long>(0x7fffff00LL, T + 16LL * rec_cap + 64));
            TRY(g->d_rec.ensure(rec_cap));
            redo_extract = redo_part = true;
        } else if (cnt->err & TH_E_PART_OVERFLOW) {
            // par_top is the sum of the regions the chains asked for at this cap_mul: when it exceeds the arena, the arena grows to it;
            // otherwise a region (or the boundary buffer) was too small: regions four times as large, and an arena for them
            if (cnt->par_top < 0) return th_fail("par_pos arena would exceed 2^31 entries; use a smaller batch");
            long long want = (long long)cnt->par_top + 64;
            if (want <= par_cap) {
                if (cap_mul >= 64) return th_fail("partition scratch overflow (par_pos region or boundary buffer)");
                const int old = std::max(cap_mul, 1);
                cap_mul = cap_mul < 1 ? 1 : cap_mul * 4;
                want = cap_mul == 1 ? 2LL * T + 16LL * rec_cap + 64 : (long long)cnt->par_top / old * cap_mul + 64;
            }
            if (want >= 0x7fffff00LL) return th_fail("par_pos arena would exceed 2^31 entries; use a smaller batch");
            par_cap = std::max(par_cap, want);
            redo_part = true;
        } else if (cnt->err & TH_E_JOB_CAP) {
            if (cnt->n_task > (1 << 28) || cnt->n_unit > (1 << 29)) return th_fail("job tables would exceed 2^28 jobs; use a smaller batch");
            g->task_cap = std::max(g->task_cap, cnt->n_task + cnt->n_task / 4 + 64); g->unit_cap = std::max(g->unit_cap, cnt->n_unit + cnt->n_unit / 4 + 64);
            g->parc_cap = std::max(g->parc_cap, cnt->par_total + cnt->par_total / 4 + 64);
        } else break;
        // reset what the rerun accumulates again (the error flags that are being answered, arena tops, queue heads, the build's maxima)
        TH_LAUNCH(th_k_cnt_rerun, 1, 64, g->stream, dc, redo_extract ? 1 : 0, redo_part ? 1 : 0);
        g->stat[2] += 1;
    }
    g->stat[3] += 1;
    if (cnt->min_fb && g->dp.w > 1 && !g->dp.hpc && !g->dp.w_seq) {
        if (getenv("TH_MINIMIZER_STRICT")) return th_fail("the parallel minimizer rules marked a slot twice on " + std::to_string(cnt->min_fb) + " read(s) (sequential replay used)");
        if (trace) fprintf(stderr, "[th] minimizer selection: %d read(s) took the sequential replay\n", cnt->min_fb);
    }
    if (cnt->err & TH_E_MSA_UNITS) return th_fail("Not enough sequences to perform msa.");        // src/abpoa_cons.c:61
    if (cnt->err) return th_fail("stage 1-3 kernels reported error flags " + std::to_string(cnt->err));
    const int n_rec = cnt->rec, nt = cnt->n_task, par_total = cnt->par_total, n_task_n = cnt->n_task_n, n_task_l = cnt->n_task_l, n_task_w = nt - n_task_n - n_task_l;
    const long long cons_total = (long long)cnt->cons_total + 16; const int aln_total = cnt->aln_total;
    if (trace) fprintf(stderr, "[th] %.3f s: stages 1-3 and job tables done, %d chains, %d jobs (device malloc %.3f s for %.2f GB in %d calls, free %.3f s so far)\n", th_wall_s() - t_run0, n_rec, nt, g_alloc_s, g_alloc_bytes / 1e9, g_alloc_n, g_free_s);

    th_room(out.parc, (size_t)par_total); out.parc.resize((size_t)par_total);
    th_room(out.jobs, (size_t)nt); out.jobs.resize((size_t)nt);
    TRY(th_rt_d2h(out.parc.data(), g->d_parc.p, (size_t)par_total * 4, g->stream));
    if (nt && !g->para.only_unit) {
        TRY(g->d_cons.ensure((size_t)cons_total)); TRY(g->d_cov.ensure((size_t)cons_total));
        TRY(g->d_cons_len.ensure(nt)); TRY(g->d_status.ensure(nt)); TRY(g->d_iden.ensure((size_t)aln_total + 1)); TRY(g->d_ext.ensure((size_t)nt * 4)); if (g->aln_slb) TRY(g->d_slb.ensure((size_t)aln_total + 1));
        TRY(g->d_cc_off.ensure(nt)); TRY(g->d_rows.ensure((size_t)std::max<long long>((long long)n * TH_JOB_COLS, (long long)nt * TH_ITEM_COLS)));
        TRY(g->d_items.ensure((size_t)aln_total + 2 * (size_t)nt + 1));
        if (g->aln_slb) TH_LAUNCH(th_k_fill_i32, (aln_total + 256) / 256, 256, g->stream, g->d_slb.p, aln_total + 1, TH_SLB_NONE);
        // ---- POA: persistent waves with private scratch, two kernel variants over two device-built task lists.  The common one keeps
        // recent rows as int16 in LDS (packed rows, 6 waves per SIMD); jobs whose scores may not fit int16 (long units, unusual scoring)
        // or whose units exceed its query staging buffer run on the 32-bit-ring variant.
        const int poa_test_fail = std::max(0, env_int("TH_POA_TEST_FAIL", 0));   // test knob, see th_k_poa
        const int poa_small_cap = env_int("TH_POA_TEST_SMALLCAP", 0) ? 32 : 0;  // test knob: node arrays too small on the first attempt (th_k_poa)
        const long long need_n = std::max<long long>(1024, (long long)cnt->need_n), need_w = std::max<long long>(1024, (long long)cnt->need_w), need_l = std::max<long long>(1024, (long long)cnt->need_l);
        const int wv_n = n_task_n ? th_poa_waves(g, n_task_n, need_n) : 0, wv_w = n_task_w ? th_poa_waves(g, n_task_w, need_w) : 0;
        const int wv_l = n_task_l ? std::min(th_poa_waves(g, n_task_l, need_l), g->waves_poa / 2) : 0;       // (three waves per SIMD)
        // a handle reserved for larger batches takes the scratch of the full persistent grid at once (with slack for longer units)
        const long long wv_alloc = g->reserve_n > n ? th_poa_waves(g, std::max(g->reserve_n, n_task_n), need_n) : wv_n;
        TRY(g->d_scratch.ensure((size_t)std::max<long long>(std::max<long long>(std::max<long long>(need_n * std::max<long long>(wv_n, wv_alloc), need_w * wv_w), need_l * wv_l), 1024)));
        TRY(th_poa_launch(g, kt, in, g->d_list_l.p, n_task_l, 2, cnt->max_unit, need_l, wv_l, &dc->q_poa_l, poa_small_cap | g->poa_fexp0[2] << 1 | poa_test_fail << 8));       // (the long jobs first: the short ones fill the tail)
        TRY(th_poa_launch(g, kt, in, g->d_list_n.p, n_task_n, 1, cnt->max_unit, need_n, wv_n, &dc->q_poa_n, poa_small_cap | g->poa_fexp0[1] << 1 | poa_test_fail << 8));
        TRY(th_poa_launch(g, kt, in, g->d_list_w.p, n_task_w, 0, cnt->max_unit, need_w, wv_w, &dc->q_poa_w, poa_small_cap | g->poa_fexp0[0] << 1 | poa_test_fail << 8));
        if (trace) { th_rt_sync_event(g->stream, g->sync_ev); fprintf(stderr, "[th] %.3f s: poa done (%d + %d + %d jobs, %d / %d / %d waves, %lld MB scratch)\n", th_wall_s() - t_run0, n_task_n, n_task_l, n_task_w, wv_n, wv_l, wv_w, (long long)(std::max(std::max(need_n * wv_n, need_w * wv_w), need_l * wv_l) * 4 >> 20)); }
        for (int pass = 0;; ++pass) {
            // ---- item lists of the alignment kernels and the compacted consensus offsets, per job, on the device ----
            KT_BEGIN(TK_ALIGN);
            TH_LAUNCH(th_k_items_count, (nt + 255) / 256, 256, g->stream, (const int *)&dc->n_task, (const ThTask *)g->d_tasks.p, g->d_cons_len.p, (const int *)g->d_status.p,
                      g->aln_wdiv, g->aln_wadd, g->aln_wtol | g->aln_probe_small << 16, g->d_rows.p, dc);
            { ThScanCaps nocaps; nocaps.on = 0; for (int k = 0; k < 8; ++k) nocaps.v[k] = 0;
              TH_LAUNCH((th_k_scan_rows<TH_ITEM_COLS>), 1, 1024, g->stream, (const int *)&dc->n_task, 0, g->d_rows.p, &dc->n_item_s, nocaps, &dc->err); }
            TH_LAUNCH(th_k_items_emit, (nt + 255) / 256, 256, g->stream, (const int *)&dc->n_task, (const ThTask *)g->d_tasks.p, (const int *)g->d_cons_len.p, (const int *)g->d_status.p,
                      g->aln_wdiv, g->aln_wadd, g->aln_wtol | g->aln_probe_small << 16, (const int *)g->d_rows.p, (const ThCnt *)dc, g->d_items.p, g->d_cc_off.p, g->d_jobs.p);
            KT_END(TK_ALIGN);
            TRY(th_fetch_cnt(g, &cnt));                                                 // ---- host synchronisation 2 of 3 ----
            if (pass == 0) g->stat[4] += nt;
            if (pass == 0 && cnt->n_retry) {
                g->stat[1] += cnt->n_retry;
                TRY(th_poa_retries(g, kt, in, nt, poa_test_fail));
                TH_LAUNCH(th_k_cnt_set, 1, 64, g->stream, &dc->n_retry, 0, &dc->n_failed, 0, &dc->max_cl, 0, &dc->max_need_big, 0);
                TH_LAUNCH(th_k_cnt_set, 1, 64, g->stream, &dc->n_item_long, 0, (int *)nullptr, 0, (int *)nullptr, 0, (int *)nullptr, 0);
                continue;
            }
            break;
        }
        g->stat[0] += cnt->n_failed;
        if (cnt->n_failed) fprintf(stderr, "[tidehunter_amd] warning: %d consensus job(s) could not be aligned (POA status 3-6, or no scratch for them) and are reported without a consensus\n", cnt->n_failed);
#if defined(TH_CHAIN_PROF)
        { unsigned long long cp[8]; hipMemcpyFromSymbol(cp, HIP_SYMBOL(th_chain_prof), sizeof(cp));
          fprintf(stderr, "[chain_prof] {\"heads_scanned\": %llu, \"trips\": %llu, \"pair_evaluations\": %llu, \"hits\": %llu, \"followers\": %llu, \"heads_on_a_list\": %llu, \"their_trips\": %llu, \"list_entries\": %llu, \"lists\": %llu}\n", cp[0], cp[1], cp[1] * 64ull, cp[2], cp[3], cp[4], cp[5], cp[6], cp[7]);
          unsigned long long z8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; hipMemcpyToSymbol(HIP_SYMBOL(th_chain_prof), z8, sizeof(z8)); }
#endif
#if defined(TH_POA_PROF)
        { unsigned long long pr[32]; hipMemcpyFromSymbol(pr, HIP_SYMBOL(thpoa_n::th_poa_prof), sizeof(pr));
          const char *cat[8] = { "common_packed", "packed_one_pred", "general_several_preds", "general_one_far_pred", "general_stores_full", "general_other", "packed_two_preds", "packed_three_four_preds" };
          fprintf(stderr, "[poa_prof] {\"jobs\": %llu, \"cycles\": {\"rows\": %llu, \"dp\": %llu, \"traceback\": %llu, \"merge\": %llu, \"consensus\": %llu}, "
                          "\"traceback\": {\"tiles\": %llu, \"run_trips\": %llu, \"run_steps\": %llu, \"single_steps\": %llu}, \"packed_further_trips\": %llu, \"dp_rows\": {",
                  pr[26], pr[0], pr[1], pr[2], pr[3], pr[24], pr[4], pr[5], pr[6], pr[7], pr[27]);
          for (int c = 0; c < 8; ++c) fprintf(stderr, "%s\"%s\": {\"rows\": %llu, \"cycles\": %llu}", c ? ", " : "", cat[c], pr[8 + c], pr[16 + c]);
          fprintf(stderr, ", \"packed_far\": {\"rows\": %llu, \"cycles\": %llu}}}\n", pr[28], pr[29]);
          { unsigned long long wy[8]; hipMemcpyFromSymbol(wy, HIP_SYMBOL(thpoa_n::th_poa_prof_why), sizeof(wy));
            fprintf(stderr, "[poa_prof_why] {\"rows_on_the_32_bit_path_by_reason\": {\"no_predecessor\": %llu, \"more_than_four\": %llu, \"one_64_rows_back\": %llu, \"two_outside_the_ring\": %llu, \"own_band_too_wide\": %llu, \"other\": %llu, \"job_of_a_32_bit_variant\": %llu}}\n",
                    wy[0], wy[1], wy[2], wy[3], wy[4], wy[5], wy[6]);
            unsigned long long zw[8] = { 0 }; hipMemcpyToSymbol(HIP_SYMBOL(thpoa_n::th_poa_prof_why), zw, sizeof(zw)); }
          unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(thpoa_n::th_poa_prof), z, sizeof(z));
          unsigned long long rw[8]; hipMemcpyFromSymbol(rw, HIP_SYMBOL(thpoa_n::th_poa_prof_rows), sizeof(rw));
          fprintf(stderr, "[poa_prof_rows] {\"fast_chunks\": %llu, \"general_chunks\": %llu, \"fast_rounds\": %llu, \"general_rounds\": %llu}\n", rw[0], rw[1], rw[2], rw[3]);
          fprintf(stderr, "[poa_prof_tail] {\"waves\": %d, \"waves_that_took_jobs\": %llu, \"busy_ticks_shortest\": %llu, \"busy_ticks_longest\": %llu, \"busy_ticks_mean\": %llu}\n", wv_n, rw[4], rw[5], rw[6], rw[4] ? (rw[7] << 10) / rw[4] : 0ull);
          unsigned long long z2[8] = { 0, 0, 0, 0, 0, ~0ull, 0, 0 }; hipMemcpyToSymbol(HIP_SYMBOL(thpoa_n::th_poa_prof_rows), z2, sizeof(z2));
          hipMemcpyFromSymbol(rw, HIP_SYMBOL(thpoa_l::th_poa_prof_rows), sizeof(rw));        // the long-unit variant's launches of this batch (configs 4 / 5)
          if (rw[4]) fprintf(stderr, "[poa_prof_tail_long] {\"waves_that_took_jobs\": %llu, \"busy_ticks_shortest\": %llu, \"busy_ticks_longest\": %llu, \"busy_ticks_mean\": %llu}\n", rw[4], rw[5], rw[6], (rw[7] << 10) / rw[4]);
          hipMemcpyToSymbol(HIP_SYMBOL(thpoa_l::th_poa_prof_rows), z2, sizeof(z2)); }
#endif
        // ---- unit-vs-consensus alignments and flank extensions: three item lists in one array, one kernel variant each ----
        const int ns = cnt->n_item_s, nm = ns + cnt->n_item_m, ni = nm + cnt->n_item_b, max_cl = cnt->max_cl;
        // the routing of the NEXT batch's long units (th_aln_class): probed and finished on the small variant (measured: config 5 +6 %) unless they
        // are most of the unit alignments (config 4: -5 % that way; its probes and strips keep the launches of their own)
        if (g->aln_probe_auto && aln_total > 0) g->aln_probe_small = (long long)cnt->n_item_long * 2 > (long long)aln_total ? 0 : 1;
        const long long cc_total = cnt->cc_total;
        const long long bnd_aln = (std::max<long long>(((long long)max_cl + 128) * 4, ((long long)in.max_len + 128) * 2) + 3) & ~3LL;
        const int wa = std::min(ni, g->waves_aln);
        TRY(g->d_bnd.ensure((size_t)(bnd_aln * std::max(wa, 1))));
        KT_BEGIN(TK_ALIGN);
        {
#define TH_ALN_ARGS(first, n, q) (const ThAlnItem *)g->d_items.p + (first), (n), (const ThTask *)g->d_tasks.p, (const int *)g->d_parc.p, \
                  (const uint32_t *)in.d_roff.p, (const uint8_t *)g->d_bseq.p, (const uint8_t *)g->d_cons.p, (const int *)g->d_cons_len.p, \
                  g->d_bnd.p, bnd_aln, g->d_iden.p, g->d_ext.p, &dc->err, (q), g->aln_wdiv, g->aln_wadd, g->aln_wtol
            const int big_first = nm, big_n = ni - big_first;           // (the second list is empty since round 3: th_aln_class)
            const int *no_n = nullptr, *no_s1 = nullptr; ThAlnItem *no_defer = nullptr; int *no_dn = nullptr, *no_ws = nullptr, *no_ds1 = nullptr;
            const int *slb_ = g->aln_slb ? (const int *)g->d_slb.p : (const int *)nullptr;      // (TH_ALN_SLB=0: no bounds from the POA -- round 4's probes)
            int *aln_stat = trace ? dc->aln_first : (int *)nullptr;          // (TH_TRACE: first / second bands tried and not certified)
            // TH_ALN_PROBE_SMALL=3: the first launch, too, hands what its anti-diagonal engine cannot certify (long units after their one-register probe)
            // on to the strip-engine launch at the end instead of finishing it in place
            const bool defer_small = g->aln_probe_small == 3 && g->aln_mode == 1;
            if (defer_small || (big_n && big_first < ni)) { TRY(g->d_defer.ensure((size_t)ni)); TRY(g->d_defer_s1.ensure((size_t)ni)); }
            ThAlnItem *sd = defer_small ? g->d_defer.p : no_defer; int *sdn = defer_small ? &dc->n_defer : no_dn, *sds1 = defer_small ? g->d_defer_s1.p : no_ds1;
#if defined(TH_ALN_PROF)
            auto aln_prof = [&](const char *what, int grid) {
                th_rt_sync_event(g->stream, g->sync_ev);
                unsigned long long pr[40]; hipMemcpyFromSymbol(pr, HIP_SYMBOL(th_aln_prof), sizeof(pr));
                if (pr[0]) { fprintf(stderr, "[aln_prof] {\"launch\": \"%s\", \"waves\": %d, \"waves_that_took_items\": %llu, \"items\": %llu, \"busy_ticks_shortest\": %llu, \"busy_ticks_longest\": %llu, \"busy_ticks_mean\": %llu, "
                                   "\"costliest_item_ticks\": %llu, \"costliest_item\": %llu, \"its_tlen\": %llu, \"its_cl\": %llu, \"its_bh\": %llu, \"its_path\": %llu, \"waves_by_ticks_shr_25\": [", what, grid, pr[0], pr[4], pr[1], pr[2], (pr[3] << 10) / pr[0], pr[5], pr[6],
                                   pr[7] & 0xffff, (pr[7] >> 16) & 0xffff, (pr[7] >> 32) & 0xfffffff, pr[7] >> 60);
                             for (int b = 0; b < 32; ++b) fprintf(stderr, "%s%llu", b ? ", " : "", pr[8 + b]);
                             fprintf(stderr, "]}\n"); }
                unsigned long long z[40] = { 0, ~0ull }; hipMemcpyToSymbol(HIP_SYMBOL(th_aln_prof), z, sizeof(z));
                unsigned long long cl_[12][4]; hipMemcpyFromSymbol(cl_, HIP_SYMBOL(th_aln_cls), sizeof(cl_));
                static const char *cls_name[10] = { "unit: first band, 1 register, certified", "unit: first band, 2 registers, certified", "unit: second register band", "unit: handed to the strip launch",
                                                    "unit: strip engines / full matrix", "left extension: first band certified", "left extension: more than one pass", "right extension: first band certified",
                                                    "right extension: more than one pass", "nothing to align" };
                if (pr[0]) { fprintf(stderr, "[aln_prof_classes] {\"launch\": \"%s\", \"classes\": [", what);
                             for (int c = 0; c < 10; ++c) fprintf(stderr, "%s{\"class\": \"%s\", \"items\": %llu, \"ticks\": %llu, \"cells_computed\": %llu, \"cells_certified_band\": %llu}", c ? ", " : "", cls_name[c], cl_[c][0], cl_[c][1], cl_[c][2], cl_[c][3]);
                             fprintf(stderr, "]}\n"); }
                memset(cl_, 0, sizeof(cl_)); hipMemcpyToSymbol(HIP_SYMBOL(th_aln_cls), cl_, sizeof(cl_));
            };
#define TH_ALN_PROF_AFTER(what, grid) aln_prof(what, grid)
#else
#define TH_ALN_PROF_AFTER(what, grid) ((void)0)
#endif
            // (few items per wave: longest flanks first -- th_k_align's `pass_queue`; TH_ALN_LPT=0 / 1 forces it off / on)
            int *lpt = (g->aln_lpt == 1 || (g->aln_lpt < 0 && (long long)ns < 64LL * wa)) ? dc->q_aln_pass : (int *)nullptr;
            if (ns && lpt) TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2, true>), std::min(ns, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(0, ns, &dc->q_aln_s), g->aln_mode, g->aln_xpct, no_n, sd, sdn, no_ws, no_s1, sds1, aln_stat, slb_, lpt);
            else if (ns) TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2>), std::min(ns, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(0, ns, &dc->q_aln_s), g->aln_mode, g->aln_xpct, no_n, sd, sdn, no_ws, no_s1, sds1, aln_stat, slb_);
            if (ns) TH_ALN_PROF_AFTER("first: 2 registers", std::min(ns, wa));
            if (nm > ns) TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2>), std::min(nm - ns, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(ns, nm - ns, &dc->q_aln_m), g->aln_mode, g->aln_xpct, no_n, sd, sdn, no_ws, no_s1, sds1, aln_stat, slb_);
            if (big_n && big_first < ni) {
                // the big-area variant hands the alignments its anti-diagonal engine could not certify on to a second launch of the small-area one
                // (mode 3): same results, but their strip-engine work then runs at three waves per SIMD instead of one or two
                // (16 KB of staging area when every item fits it -- units of up to ~6000 bases: eight waves per CU instead of six)
                if (cnt->max_need_big <= TH_ALN_BIG16)
                    TH_LAUNCH((th_k_align<TH_ALN_BIG16, 6>), std::min(big_n, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(big_first, big_n, &dc->q_aln_b), g->aln_mode, g->aln_xpct, no_n, g->d_defer.p, &dc->n_defer, dc->aln_wide, no_s1, g->d_defer_s1.p, (int *)nullptr, slb_);
                else
                    TH_LAUNCH((th_k_align<TH_ALN_BIG, 6>), std::min(big_n, wa), TH_ALN_BLOCK, g->stream, TH_ALN_ARGS(big_first, big_n, &dc->q_aln_b), g->aln_mode, g->aln_xpct, no_n, g->d_defer.p, &dc->n_defer, dc->aln_wide, no_s1, g->d_defer_s1.p, (int *)nullptr, slb_);
                TH_ALN_PROF_AFTER("long units: 6 registers", std::min(big_n, wa));
            }
            if (g->aln_mode == 1 && (defer_small || (big_n && big_first < ni)))
                TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2>), std::min(ni, wa), TH_ALN_BLOCK, g->stream, (const ThAlnItem *)g->d_defer.p, 0, (const ThTask *)g->d_tasks.p, (const int *)g->d_parc.p,
                          (const uint32_t *)in.d_roff.p, (const uint8_t *)g->d_bseq.p, (const uint8_t *)g->d_cons.p, (const int *)g->d_cons_len.p,
                          g->d_bnd.p, bnd_aln, g->d_iden.p, g->d_ext.p, &dc->err, &dc->q_aln_d, g->aln_wdiv, g->aln_wadd, g->aln_wtol, 3, g->aln_xpct, (const int *)&dc->n_defer, no_defer, no_dn, no_ws, (const int *)g->d_defer_s1.p, no_ds1);
            if (g->aln_mode == 1 && (defer_small || (big_n && big_first < ni))) TH_ALN_PROF_AFTER("deferred: strips", std::min(ni, wa));
#undef TH_ALN_ARGS
        }
        KT_END(TK_ALIGN);
        if (trace) { th_rt_sync_event(g->stream, g->sync_ev); fprintf(stderr, "[th] %.3f s: align done (%d items)\n", th_wall_s() - t_run0, ni); }
        // ---- adapter searches on the doubled consensus (src/gen_cons.c:226-291), four per job, one per lane ----
        if (g->ad_on) {
            TRY(g->d_adp.ensure(1)); TRY(g->d_ad.ensure((size_t)nt * 12));
            if (!g->adp_up) { TRY(th_rt_h2d(g->d_adp.p, &g->h_ad, sizeof(ThAdPara), g->stream)); g->adp_up = true; }
            KT_BEGIN(TK_ALIGN);
            TH_LAUNCH(th_k_adapter, (nt * 4 + 63) / 64, 64, g->stream, (const ThTask *)g->d_tasks.p, nt, (const uint8_t *)g->d_cons.p, (const int *)g->d_cons_len.p,
                      (const ThAdPara *)g->d_adp.p, g->ad_min_cons, g->d_ad.p);
            KT_END(TK_ALIGN);
            th_room(out.ad, (size_t)nt * 12); out.ad.resize((size_t)nt * 12);
            TRY(th_rt_d2h(out.ad.data(), g->d_ad.p, (size_t)nt * 48, g->stream));
            out.ad_valid = true;
        }
        // results: consensus strings compacted to their lengths (the slots are sized for the worst case), coverage only on request
        TRY(g->d_cons_c.ensure((size_t)cc_total + 4));
        if (g->para.want_cov) TRY(g->d_cov_c.ensure((size_t)cc_total + 4));
        TH_LAUNCH(th_k_gather_cons, nt, 128, g->stream, nt, (const ThTask *)g->d_tasks.p, (const int *)g->d_cons_len.p, (const int *)g->d_cc_off.p,
                  (const uint8_t *)g->d_cons.p, (const int *)g->d_cov.p, g->d_cons_c.p, g->para.want_cov ? g->d_cov_c.p : (int *)nullptr);
        th_room(out.iden, (size_t)aln_total); th_room(out.ext, (size_t)nt * 4); th_room(out.cons, (size_t)cc_total + 4);
        out.iden.resize(aln_total); out.ext.resize((size_t)nt * 4); out.cons.resize((size_t)cc_total + 4);
        TRY(th_rt_d2h(out.iden.data(), g->d_iden.p, (size_t)aln_total * 4, g->stream));
        TRY(th_rt_d2h(out.ext.data(), g->d_ext.p, (size_t)nt * 16, g->stream));
        TRY(th_rt_d2h(out.cons.data(), g->d_cons_c.p, (size_t)cc_total, g->stream));
        if (g->para.want_cov) { th_room(out.cov, (size_t)cc_total + 4); out.cov.resize((size_t)cc_total + 4); TRY(th_rt_d2h(out.cov.data(), g->d_cov_c.p, (size_t)cc_total * 4, g->stream)); }
    }
    TRY(th_rt_d2h(out.jobs.data(), g->d_jobs.p, (size_t)nt * sizeof(th_job_t), g->stream));
    TRY(th_fetch_cnt(g, &cnt));                                                         // ---- host synchronisation 3 of 3 ----
    if (cnt->err & TH_E_ALN_VERIFY) return th_fail("TH_ALN_MODE=2: a banded alignment differs from the full matrix");
    if (cnt->err & TH_E_ALN_BOUND) return th_fail("th_k_align: a band sized by a lower bound scored below the bound (the POA's unit bound was not one)");
    if (cnt->err) return th_fail("alignment boundary buffer overflow");
    out.n_jobs = nt;
    if (trace) fprintf(stderr, "[th] unit alignments of the small-area launch: %d first bands, %d not certified; %d second bands, %d not certified\n", cnt->aln_first[0], cnt->aln_first[1], cnt->aln_first[2], cnt->aln_first[3]);
    if (trace) fprintf(stderr, "[th] %.3f s: run done, %d host synchronisations (this thread so far: device malloc %.3f s for %.2f GB in %d calls, free %.3f s)\n", th_wall_s() - t_run0, g->n_sync, g_alloc_s, g_alloc_bytes / 1e9, g_alloc_n, g_free_s);
    kt.collect(out.ms_k);
    out.ms_kernels = 0; out.ms_dominant = 0;
    for (int k = 0; k < TK_N; ++k) { out.ms_kernels += out.ms_k[k]; out.ms_dominant = std::max(out.ms_dominant, out.ms_k[k]); }
    if (g->replay && !(g->replay_ok && n <= g->replay_n)) { g->replay_out = out; g->replay_n = n; g->replay_T = T; g->replay_ok = true; }      // (measurement only: the largest batch so far is the one replayed)
    return 0;
}

extern "C" int th_gpu_get(th_gpu_t *g, int out_slot, th_gpu_out_t *o)
{
    const ThOut &out = g->out[out_slot & 1];
    o->n_jobs = out.n_jobs; o->jobs = out.jobs.data(); o->par = out.parc.data();
    o->cons = out.cons.data(); o->cov = g->para.want_cov ? out.cov.data() : nullptr; o->iden = out.iden.data(); o->ext = out.ext.data();
    o->adapt = out.ad_valid ? out.ad.data() : nullptr;
    o->single = out.single_valid ? out.single.data() : nullptr;
    o->ms_kernels = out.ms_kernels; o->ms_dominant = out.ms_dominant;
    for (int k = 0; k < 8; ++k) o->ms_kernel[k] = out.ms_k[k];
    return 0;
}

// test hook: intermediate stages of the batch that ran last (device arrays are copied on request, nothing is kept for it)
extern "C" long long th_gpu_debug(th_gpu_t *g, int what, void *dst, long long cap)
{
    const void *src = nullptr; long long bytes = 0;
    if (th_select_device(g)) return -1;
    const ThIn &in = g->in[g->last_in];
    switch (what) {
    case 0: src = g->d_bseq.p; bytes = in.T; break;
    case 1: src = g->d_hit_n.p; bytes = 4LL * in.n; break;
    case 2: src = g->d_hit_end.p; bytes = 4LL * in.T; break;
    case 3: src = g->d_hit_per.p; bytes = 4LL * in.T; break;
    case 4: src = g->d_score.p; bytes = 4LL * in.T; break;
    case 5: src = g->d_from.p; bytes = 4LL * in.T; break;
    case 6: {   // chain records regrouped by read (emission order inside a read is already ascending end)
        ThCnt c;
        if (th_rt_d2h(&c, g->d_cnt.p, sizeof(c), g->stream) || th_rt_sync(g->stream)) return -1;
        const int n_rec = c.rec;
        std::vector<ThChainRec> rec(n_rec); std::vector<int> par_dst(n_rec), par_n(n_rec), cb(in.n), cn(in.n), tmp;
        if (th_rt_d2h(rec.data(), g->d_rec.p, (size_t)n_rec * sizeof(ThChainRec), g->stream) || th_rt_d2h(par_dst.data(), g->d_par_dst.p, (size_t)n_rec * 4, g->stream) ||
            th_rt_d2h(par_n.data(), g->d_par_n.p, (size_t)n_rec * 4, g->stream) || th_rt_d2h(cb.data(), g->d_read_ch_base.p, (size_t)in.n * 4, g->stream) ||
            th_rt_d2h(cn.data(), g->d_read_ch_n.p, (size_t)in.n * 4, g->stream) || th_rt_sync(g->stream)) return -1;
        for (int r = 0; r < in.n; ++r)
            for (int c2 = cb[r]; c2 < cb[r] + cn[r]; ++c2) {
                const ThChainRec &x = rec[c2];
                int v[8] = { x.read, x.cell_off, x.len, x.est_start, x.est_period, par_dst[c2], par_n[c2], 0 };
                tmp.insert(tmp.end(), v, v + 8);
            }
        bytes = 4LL * (long long)tmp.size();
        if (bytes > cap) return -1;
        memcpy(dst, tmp.data(), (size_t)bytes);
        return bytes;
    }
    case 7: src = g->d_cells.p; bytes = 4LL * in.T; break;
    case 8: if (4LL * (in.n + 1) > cap) return -1; memcpy(dst, in.h_roff.data(), 4 * ((size_t)in.n + 1)); return 4LL * (in.n + 1);
    case 9: {
        ThCnt c;
        if (th_rt_d2h(&c, g->d_cnt.p, sizeof(c), g->stream) || th_rt_sync(g->stream)) return -1;
        src = g->d_parc.p; bytes = 4LL * c.par_total; break;
    }
    default: return -1;
    }
    if (bytes > cap) return -1;
    if (th_rt_d2h(dst, src, (size_t)bytes, g->stream)) return -1;
    if (th_rt_sync(g->stream)) return -1;
    return bytes;
}

#include "th_debug_hooks.h"
