// th_debug_hooks.h -- test hooks: RAW pairs through the device's alignment and adapter engines (included at the end of th_pipeline.hip;
// declared in th_internal.h, not in the public header).
//
// tests/test_gpu_ref_vectors.py compares what these return with tests/golden/ref_ksw2_vectors.npz / ref_edlib_vectors.npz -- vectors
// written by the reference's OWN ksw2 / edlib code (tests/golden/make_ref_vectors.py, oracle/_ref/libthref.so) -- so that th_k_align,
// th_ksw_global_wave and th_k_adapter* are held to the reference directly, not through the oracle.  The hooks do not go through the
// pipeline: they build the tables th_k_align / th_k_adapter* read (tasks, items, boundaries) so that pair i IS job i, and launch the
// product kernels unchanged, in the variant and with the knobs the caller names.
//   src/ksw2_align.c:117-173 (ksw2_global, ksw2_global_with_cigar + ksw2_backtrack_left_end, ksw2_left_ext / ksw2_right_ext)
//   src/edlib_align.c:73-85  (edlib_align_HW)

// the full-matrix engine with the CIGAR functionals (the partition's engine), one pair per wave
__global__ void __launch_bounds__(64)
th_k_debug_global(int n, const uint8_t *__restrict__ q, const int *__restrict__ qo, const uint8_t *__restrict__ t, const int *__restrict__ to,
                  const int *__restrict__ q_left_ext, int n_ext, int *__restrict__ bnd, long long bnd_stride, int *__restrict__ iden, int *__restrict__ tback, int *__restrict__ queue, int banded)
{
    __shared__ int s_in[2][TH_WAVE][8];
    int *mybnd = bnd + (long long)blockIdx.x * bnd_stride;
    for (;;) {
        const int i = th_queue_pop(queue);
        if (i >= n) break;
        const int ql = qo[i + 1] - qo[i], tl = to[i + 1] - to[i];
        const int v0 = th_ksw_global_wave<false>(q + qo[i], ql, t + to[i], tl, 0, mybnd, nullptr);
        if (TH_LANE() == 0) iden[i] = v0;
        for (int e = 0; e < n_ext; ++e) {
            const int x = q_left_ext[(long long)i * n_ext + e];
            int tb = -1;
            const int v = x >= 0 && x <= ql ? th_ksw_global_r_wave(q + qo[i], ql, t + to[i], tl, x, mybnd, s_in, &tb, banded != 0) : -1;      // (banded: the partition's flow -- certified band, pruned wider band, full matrix)
            if (TH_LANE() == 0) tback[(long long)i * n_ext + e] = (x >= 0 && x <= ql && v == v0) ? tb : -0x7fffffff;      // (both instantiations must agree on iden_n)
        }
    }
}

// (a launch argument must be a plain pointer taken from .p beforehand, never `buf.p` itself: the test build's launch captures its arguments by value,
// and a captured DbgBuf would free the buffer a second time)
struct DbgBuf { void *p = nullptr; DbgBuf() {} DbgBuf(const DbgBuf &) = delete; ~DbgBuf() { if (p) th_rt_free(p); } int get(size_t n) { return th_rt_malloc(&p, n ? n : 1); } };
#define DBG_UP(buf, src, bytes) do { TRY((buf).get(bytes)); TRY(th_rt_h2d((buf).p, (src), (bytes), s)); } while (0)

// kind 0: ksw2_global(q, t) -> out[i] = iden_n, through th_k_align's unit path (q = unit, t = consensus)
// kind 1 / 2: ksw2_left_ext / ksw2_right_ext(q, t) -> out[2i], out[2i + 1] = (max_q, max_t), through its flank path (q = consensus, t = flank)
// cfg: [0] variant (0 <12288,2>; 1 <16384,6>; 2 <24576,6>) [1] mode (TH_ALN_MODE) [2] wdiv [3] wadd [4] wtol [5] xpct [6] defer (1: the big variants hand what
// they cannot certify to a mode-3 launch of the small one, as the pipeline does).  stats (8 ints, optional): [0] error flags [1] deferred items [2] wide first bands tried [3] failed
extern "C" int th_debug_align_pairs(int device, int kind, int n, const uint8_t *q_cat, const int *q_off, const uint8_t *t_cat, const int *t_off, const int *cfg, int *out, int *stats)
{
    if (th_rt_set_device(device)) return th_fail("hipSetDevice failed");
    if (n <= 0) return 0;
    th_stream_t s; TRY(th_rt_stream_create(&s));
    struct SG { th_stream_t s; ~SG() { th_rt_stream_destroy(s); } } sg{ s };
    // pair i is job i.  kind 0: read i = q_i with boundaries {-1, qlen - 1} (one unit: the whole read), consensus i = t_i.
    // kind 1: consensus i = q_i, read i = t_i with first boundary tlen - 1 (the flank is read[0 .. pos[0]], src/gen_cons.c:218).
    // kind 2: consensus i = q_i, read i = t_i with last boundary -1 (the flank is read[pos[last] + 1 ..], :221).
    const uint8_t *rd = kind == 0 ? q_cat : t_cat, *cs = kind == 0 ? t_cat : q_cat;
    const int *ro = kind == 0 ? q_off : t_off, *co = kind == 0 ? t_off : q_off;
    std::vector<uint32_t> roff((size_t)n + 1); std::vector<ThTask> tasks(n); std::vector<ThAlnItem> items(n); std::vector<int> parc((size_t)2 * n), clen(n);
    int max_cl = 0, max_rl = 0;
    for (int i = 0; i <= n; ++i) roff[i] = (uint32_t)ro[i];
    for (int i = 0; i < n; ++i) {
        const int rl = ro[i + 1] - ro[i], cl = co[i + 1] - co[i];
        if (rl <= 0 || cl <= 0) return th_fail("th_debug_align_pairs: empty sequence");
        max_cl = std::max(max_cl, cl); max_rl = std::max(max_rl, rl);
        ThTask T; memset(&T, 0, sizeof(T));
        T.read = i; T.pos_off = 2 * i; T.pos_n = 2; T.cons_off = co[i]; T.cons_cap = cl; T.aln_off = i; T.max_len = rl; T.n_units = 1;
        tasks[i] = T; clen[i] = cl;
        parc[2 * i] = kind == 0 ? -1 : kind == 1 ? rl - 1 : -1;
        parc[2 * i + 1] = kind == 0 ? rl - 1 : kind == 1 ? rl - 1 : -1;
        ThAlnItem it = { i, kind == 0 ? 0 : kind == 1 ? -1 : -2 }; items[i] = it;
    }
    DbgBuf d_rd, d_cs, d_roff, d_tasks, d_items, d_parc, d_clen, d_bnd, d_iden, d_ext, d_cnt, d_defer, d_ds1;
    DBG_UP(d_rd, rd, (size_t)ro[n] + 16); DBG_UP(d_cs, cs, (size_t)co[n] + 16);
    DBG_UP(d_roff, roff.data(), ((size_t)n + 1) * 4); DBG_UP(d_tasks, tasks.data(), (size_t)n * sizeof(ThTask)); DBG_UP(d_items, items.data(), (size_t)n * sizeof(ThAlnItem));
    DBG_UP(d_parc, parc.data(), (size_t)n * 8); DBG_UP(d_clen, clen.data(), (size_t)n * 4);
    const int waves = std::min(n, 2048);
    const long long bnd_aln = (std::max<long long>(((long long)max_cl + 128) * 4, ((long long)max_rl + 128) * 2) + 3) & ~3LL;
    TRY(d_bnd.get((size_t)bnd_aln * waves * 4)); TRY(d_iden.get((size_t)n * 4 + 4)); TRY(d_ext.get((size_t)n * 16)); TRY(d_cnt.get(sizeof(ThCnt)));
    TRY(d_defer.get((size_t)n * sizeof(ThAlnItem))); TRY(d_ds1.get((size_t)n * 4));
    TRY(th_rt_memset(d_iden.p, 0xff, (size_t)n * 4 + 4, s)); TRY(th_rt_memset(d_ext.p, 0xff, (size_t)n * 16, s));
    ThCnt *dc = (ThCnt *)d_cnt.p;
    TH_LAUNCH(th_k_cnt_zero, 1, 64, s, dc);
    const ThAlnItem *p_items = (const ThAlnItem *)d_items.p; const ThTask *p_tasks = (const ThTask *)d_tasks.p; const int *p_parc = (const int *)d_parc.p, *p_clen = (const int *)d_clen.p;
    const uint32_t *p_roff = (const uint32_t *)d_roff.p; const uint8_t *p_rd = (const uint8_t *)d_rd.p, *p_cs = (const uint8_t *)d_cs.p;
    int *p_bnd = (int *)d_bnd.p, *p_iden = (int *)d_iden.p, *p_ext = (int *)d_ext.p, *p_ds1 = (int *)d_ds1.p; ThAlnItem *p_defer = (ThAlnItem *)d_defer.p;
    const int variant = cfg[0], mode = cfg[1], wdiv = std::max(1, cfg[2]), wadd = cfg[3], wtol = cfg[4], xpct = cfg[5], use_defer = cfg[6];
    const int *no_n = nullptr, *no_s1 = nullptr; ThAlnItem *no_defer = nullptr; int *no_dn = nullptr, *no_ws = nullptr, *no_ds1 = nullptr;
#define DBG_ARGS(itm, cnt, q) (const ThAlnItem *)(itm), (cnt), p_tasks, p_parc, p_roff, p_rd, p_cs, p_clen, p_bnd, bnd_aln, p_iden, p_ext, &dc->err, (q), wdiv, wadd, wtol
    ThAlnItem *df = use_defer ? p_defer : no_defer; int *dfn = use_defer ? &dc->n_defer : no_dn, *dfs = use_defer ? p_ds1 : no_ds1;
    // variant 3: the LONGEST-FIRST instantiation of the two-register variant with its pass queues -- the only one that holds th_aln_long_flank (prefix probe + pruned strips);
    // stats[4] counts the flanks that took it
    if (variant == 3) TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2, true>), waves, TH_ALN_BLOCK, s, DBG_ARGS(p_items, n, &dc->q_aln_s), mode, xpct, no_n, no_defer, no_dn, no_ws, no_s1, no_ds1, dc->aln_first, (const int *)nullptr, dc->q_aln_pass);
    else if (variant == 0) TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2>), waves, TH_ALN_BLOCK, s, DBG_ARGS(p_items, n, &dc->q_aln_s), mode, xpct, no_n, no_defer, no_dn, no_ws, no_s1, no_ds1);
    else if (variant == 1) TH_LAUNCH((th_k_align<TH_ALN_BIG16, 6>), waves, TH_ALN_BLOCK, s, DBG_ARGS(p_items, n, &dc->q_aln_b), mode, xpct, no_n, df, dfn, dc->aln_wide, no_s1, dfs);
    else TH_LAUNCH((th_k_align<TH_ALN_BIG, 6>), waves, TH_ALN_BLOCK, s, DBG_ARGS(p_items, n, &dc->q_aln_b), mode, xpct, no_n, df, dfn, dc->aln_wide, no_s1, dfs);
    if (variant != 0 && use_defer && mode == 1)
        TH_LAUNCH((th_k_align<TH_ALN_SMALL, 2>), waves, TH_ALN_BLOCK, s, DBG_ARGS(p_defer, 0, &dc->q_aln_d), 3, xpct, (const int *)&dc->n_defer, no_defer, no_dn, no_ws, (const int *)p_ds1, no_ds1);
#undef DBG_ARGS
    ThCnt hc;
    TRY(th_rt_d2h(&hc, dc, sizeof(hc), s));
    if (kind == 0) TRY(th_rt_d2h(out, d_iden.p, (size_t)n * 4, s));
    else {
        std::vector<int> ext((size_t)4 * n);
        TRY(th_rt_d2h(ext.data(), d_ext.p, (size_t)n * 16, s)); TRY(th_rt_sync(s));
        for (int i = 0; i < n; ++i) { out[2 * i] = ext[4 * i + (kind == 1 ? 0 : 2)]; out[2 * i + 1] = ext[4 * i + (kind == 1 ? 1 : 3)]; }
    }
    TRY(th_rt_sync(s));
    if (stats) { stats[0] = hc.err; stats[1] = hc.n_defer; stats[2] = hc.aln_wide[0]; stats[3] = hc.aln_wide[1]; stats[4] = hc.aln_first[4]; stats[5] = stats[6] = stats[7] = 0; }
    return 0;
}

// ksw2_global_with_cigar + ksw2_backtrack_left_end on the partition's engine (th_ksw_global_wave): iden[i], and tback[i * n_ext + e] for
// q_left_ext[i * n_ext + e] (-0x7fffffff where the value is outside 0 .. qlen)
extern "C" int th_debug_global_pairs(int device, int n, const uint8_t *q_cat, const int *q_off, const uint8_t *t_cat, const int *t_off, const int *q_left_ext, int n_ext, int *iden, int *tback, int banded)
{
    if (th_rt_set_device(device)) return th_fail("hipSetDevice failed");
    if (n <= 0) return 0;
    th_stream_t s; TRY(th_rt_stream_create(&s));
    struct SG { th_stream_t s; ~SG() { th_rt_stream_destroy(s); } } sg{ s };
    int max_t = 0;
    for (int i = 0; i < n; ++i) { if (q_off[i + 1] <= q_off[i] || t_off[i + 1] <= t_off[i]) return th_fail("th_debug_global_pairs: empty sequence"); max_t = std::max(max_t, t_off[i + 1] - t_off[i]); }
    DbgBuf d_q, d_t, d_qo, d_to, d_x, d_bnd, d_iden, d_tb, d_cnt;
    DBG_UP(d_q, q_cat, (size_t)q_off[n] + 16); DBG_UP(d_t, t_cat, (size_t)t_off[n] + 16); DBG_UP(d_qo, q_off, ((size_t)n + 1) * 4); DBG_UP(d_to, t_off, ((size_t)n + 1) * 4);
    DBG_UP(d_x, q_left_ext, (size_t)n * std::max(n_ext, 1) * 4);
    const int waves = std::min(n, 2048);
    const long long stride = (((long long)max_t + 128) * 6 + 3) & ~3LL;
    TRY(d_bnd.get((size_t)stride * waves * 4)); TRY(d_iden.get((size_t)n * 4)); TRY(d_tb.get((size_t)n * std::max(n_ext, 1) * 4)); TRY(d_cnt.get(sizeof(ThCnt)));
    ThCnt *dc = (ThCnt *)d_cnt.p;
    TH_LAUNCH(th_k_cnt_zero, 1, 64, s, dc);
    const uint8_t *p_q = (const uint8_t *)d_q.p, *p_t = (const uint8_t *)d_t.p; const int *p_qo = (const int *)d_qo.p, *p_to = (const int *)d_to.p, *p_x = (const int *)d_x.p;
    int *p_bnd = (int *)d_bnd.p, *p_iden = (int *)d_iden.p, *p_tb = (int *)d_tb.p;
    TH_LAUNCH(th_k_debug_global, waves, 64, s, n, p_q, p_qo, p_t, p_to, p_x, n_ext, p_bnd, stride, p_iden, p_tb, &dc->q_part, banded);
    TRY(th_rt_d2h(iden, d_iden.p, (size_t)n * 4, s));
    if (n_ext > 0) TRY(th_rt_d2h(tback, d_tb.p, (size_t)n * n_ext * 4, s));
    TRY(th_rt_sync(s));
    return 0;
}

// the four-query equality tables of th_k_adapter / th_k_adapter_reads (src/edlib_align.c:21-27: byte equality + a/A c/C g/G t/T n/N)
static bool th_ad_eq(char a, char b) { if (a == b) return true; if ((a ^ b) != 0x20) return false; const char x = a | 0x20; return x == 'a' || x == 'c' || x == 'g' || x == 't' || x == 'n'; }
static bool th_ad_tables(const char *const seq[4], const int len[4], const int k[4], ThAdPara *ap, ThAdRead *ar)
{
    memset(ap, 0, sizeof(*ap)); if (ar) memset(ar, 0, sizeof(*ar));
    for (int sr = 0; sr < 4; ++sr) {
        const int m = len[sr];
        if (m > 64 * TH_AD_MAXB) return false;
        ap->len[sr] = m; ap->k[sr] = k[sr]; ap->nb[sr] = (m + 63) / 64;
        for (int i = 0; i < m; ++i) {
            for (int c = 0; c < 5; ++c) {
                if (th_ad_eq(seq[sr][i], "ACGTN"[c])) ap->peq[sr][0][c][i / 64] |= 1ull << (i & 63);
                if (th_ad_eq(seq[sr][m - 1 - i], "ACGTN"[c])) ap->peq[sr][1][c][i / 64] |= 1ull << (i & 63);
            }
            if (ar) for (int c = 0; c < 256; ++c) {
                if (th_ad_eq(seq[sr][i], (char)c)) ar->peq[sr][0][c][i / 64] |= 1ull << (i & 63);
                if (th_ad_eq(seq[sr][m - 1 - i], (char)c)) ar->peq[sr][1][c][i / 64] |= 1ull << (i & 63);
            }
        }
    }
    return true;
}

// edlib_align_HW(query, target, k) on the device, four queries against every target.
// doubled = 0: th_k_adapter_reads -- targets are raw bytes; out[(i * 4 + sr) * 3 ..] = (ed, start, end) of the search over the whole target
// doubled = 1: th_k_adapter -- targets are consensus CODES (0..4) and the search runs over target + target (src/gen_cons.c:228-231)
extern "C" int th_debug_adapter(int device, int doubled, const char *const queries[4], const int qlen[4], const int k[4], int n, const uint8_t *t_cat, const int *t_off, int *out)
{
    if (th_rt_set_device(device)) return th_fail("hipSetDevice failed");
    if (n <= 0) return 0;
    th_stream_t s; TRY(th_rt_stream_create(&s));
    struct SG { th_stream_t s; ~SG() { th_rt_stream_destroy(s); } } sg{ s };
    ThAdPara ap; std::vector<ThAdRead> ar(1);
    if (!th_ad_tables(queries, qlen, k, &ap, doubled ? nullptr : ar.data())) return th_fail("th_debug_adapter: a query of more than 256 symbols");
    DbgBuf d_t, d_ap, d_ar, d_roff, d_tasks, d_clen, d_out;
    DBG_UP(d_t, t_cat, (size_t)t_off[n] + 16); DBG_UP(d_ap, &ap, sizeof(ap));
    if (doubled) {
        std::vector<ThTask> tasks(n); std::vector<int> clen(n);
        for (int i = 0; i < n; ++i) { memset(&tasks[i], 0, sizeof(ThTask)); tasks[i].read = i; tasks[i].cons_off = t_off[i]; clen[i] = t_off[i + 1] - t_off[i]; }
        DBG_UP(d_tasks, tasks.data(), (size_t)n * sizeof(ThTask)); DBG_UP(d_clen, clen.data(), (size_t)n * 4);
        TRY(d_out.get((size_t)n * 48));
        const ThTask *p_tasks = (const ThTask *)d_tasks.p; const uint8_t *p_t = (const uint8_t *)d_t.p; const int *p_clen = (const int *)d_clen.p; const ThAdPara *p_ap = (const ThAdPara *)d_ap.p; int *p_out = (int *)d_out.p;
        TH_LAUNCH(th_k_adapter, (n * 4 + 63) / 64, 64, s, p_tasks, n, p_t, p_clen, p_ap, -1, p_out);
        TRY(th_rt_d2h(out, d_out.p, (size_t)n * 48, s)); TRY(th_rt_sync(s));
    } else {
        std::vector<uint32_t> roff((size_t)n + 1);
        for (int i = 0; i <= n; ++i) roff[i] = (uint32_t)t_off[i];
        DBG_UP(d_ar, ar.data(), sizeof(ThAdRead)); DBG_UP(d_roff, roff.data(), ((size_t)n + 1) * 4);
        TRY(d_out.get((size_t)n * 160));
        const uint32_t *p_roff = (const uint32_t *)d_roff.p; const uint8_t *p_t = (const uint8_t *)d_t.p; const ThAdPara *p_ap = (const ThAdPara *)d_ap.p; const ThAdRead *p_ar = (const ThAdRead *)d_ar.p; int *p_out = (int *)d_out.p;
        TH_LAUNCH(th_k_adapter_reads, (n * 4 + 63) / 64, 64, s, n, p_roff, p_t, p_ap, p_ar, p_out);
        std::vector<int> raw((size_t)n * 40);
        TRY(th_rt_d2h(raw.data(), d_out.p, (size_t)n * 160, s)); TRY(th_rt_sync(s));
        for (int i = 0; i < n; ++i) for (int sr = 0; sr < 4; ++sr) {
            const int *o = raw.data() + 40 * i + 10 * sr; int *d = out + (i * 4 + sr) * 3;
            if (o[0] >= 1) { d[0] = o[1]; d[1] = o[2]; d[2] = o[3]; } else { d[0] = -1; d[1] = -1; d[2] = -1; }
        }
    }
    return 0;
}
#undef DBG_UP
