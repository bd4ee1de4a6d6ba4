// th_k_poa_body.h -- the POA kernels, compiled once per configuration (th_k_poa.hip sets TH_POA_NS, TH_POA_RW, TH_POA_TRIPS, TH_POA_REL,
// TH_POA_OCC_NS and includes this file; no include guard on purpose).
namespace TH_POA_NS {
static_assert(TH_POA_RW + 2 <= 128 * TH_POA_TRIPS && (TH_POA_RW & 1) == 0, "a packed row covers its band from the even column at or below its start");
struct ThPoaG {
    int cap, n_node;
    // by node id
    int *nbase, *nread, *ring, *inn, *outn, *in4, *out4, *in_oh, *in_ot, *out_oh, *out_ot, *pos;
    int *ovf_val, *ovf_next, *ovf_cnt;
    int *order, *order2;
    // by row (order index)
    int *r_meta, *r_p, *r_s, *r_rem, *r_beg, *r_end, *r_mpl, *r_mpr;
    int *r_off, *r_foff;       // offset (2-byte units) of a row's direction words -- of its band start (16-bit rows) or of its even base column (8-bit rows: in-degree 1) -- / of its band start among the full cells (rows that store them)
    int *cntB, *headB, *aft;
    int *ops, *pv, *glo, *pth;
    int *upath, *ccol;         // round 5: the node of every base of every unit, unit after unit (the alignments' paths) / by node id: its column's place in the consensus, or -1
    int4 *cell; uint16_t *dcell; uint32_t cell_u2;     // cell_u2: size of the region in 2-byte units (clamped to 2^31: row offsets are ints); full cells (rows some far row will read, rows of in-degree > 4) / 16-bit direction words (all other rows)
};

__device__ __forceinline__ int th_poa_in(const ThPoaG &g, int v, int ord)
{
    if (ord < TH_POA_INL) return g.in4[v * TH_POA_INL + ord];
    int e = g.in_oh[v];
    for (int x = TH_POA_INL; x < ord; ++x) e = g.ovf_next[e];
    return g.ovf_val[e];
}
__device__ __forceinline__ int th_poa_out(const ThPoaG &g, int v, int ord)
{
    if (ord < TH_POA_INL) return g.out4[v * TH_POA_INL + ord];
    int e = g.out_oh[v];
    for (int x = TH_POA_INL; x < ord; ++x) e = g.ovf_next[e];
    return g.ovf_val[e];
}
// one lane adds edge from->to if absent; `from` and `to` are each handled by exactly one lane per path
__device__ void th_poa_add_edge(ThPoaG &g, int from, int to)
{
    const int on = g.outn[from];
    {
        int e = g.out_oh[from];
        for (int x = 0; x < on; ++x) {
            int u;
            if (x < TH_POA_INL) u = g.out4[from * TH_POA_INL + x];
            else { u = g.ovf_val[e]; e = g.ovf_next[e]; }
            if (u == to) return;
        }
    }
    if (on < TH_POA_INL) g.out4[from * TH_POA_INL + on] = to;
    else {
        const int id = atomicAdd(g.ovf_cnt, 1); g.ovf_val[id] = to; g.ovf_next[id] = -1;
        if (g.out_ot[from] < 0) g.out_oh[from] = id; else g.ovf_next[g.out_ot[from]] = id;
        g.out_ot[from] = id;
    }
    g.outn[from] = on + 1;
    const int in = g.inn[to];
    if (in < TH_POA_INL) g.in4[to * TH_POA_INL + in] = from;
    else {
        const int id = atomicAdd(g.ovf_cnt, 1); g.ovf_val[id] = from; g.ovf_next[id] = -1;
        if (g.in_ot[to] < 0) g.in_oh[to] = id; else g.ovf_next[g.in_ot[to]] = id;
        g.in_ot[to] = id;
    }
    g.inn[to] = in + 1;
}
__device__ __forceinline__ void th_poa_init_node(ThPoaG &g, int v, int b, int nread)
{
    g.nbase[v] = b; g.nread[v] = nread; g.ring[v] = v; g.inn[v] = 0; g.outn[v] = 0;
    g.in_oh[v] = g.in_ot[v] = g.out_oh[v] = g.out_ot[v] = -1;
}

// ---- first unit: chain source(0) -> bases(2..L+1) -> sink(1); order = source, bases, sink ----
__device__ void th_poa_seed(ThPoaG &g, const uint8_t *__restrict__ q, int L, int *__restrict__ upath)
{
    const int lane = TH_LANE();
    for (int j = lane; j < L; j += TH_WAVE) {
        const int v = 2 + j;
        th_poa_init_node(g, v, min((int)q[j], 4), 1);
        g.inn[v] = 1; g.in4[v * TH_POA_INL] = j == 0 ? 0 : v - 1;
        g.outn[v] = 1; g.out4[v * TH_POA_INL] = j == L - 1 ? 1 : v + 1;
        g.order[1 + j] = v; g.pos[v] = 1 + j;
        if (upath) upath[j] = v;
    }
    if (lane == 0) {
        th_poa_init_node(g, 0, 4, 0); th_poa_init_node(g, 1, 4, 0);
        g.outn[0] = 1; g.out4[0] = 2; g.inn[1] = 1; g.in4[TH_POA_INL] = L + 1;
        g.order[0] = 0; g.pos[0] = 0; g.order[L + 1] = 1; g.pos[1] = L + 1;
        *g.ovf_cnt = 0;
    }
    g.n_node = L + 2;
    TH_WAVE_FENCE();
}

__device__ __forceinline__ int *th_carve(int *&p, long long cnt) { int *r = p; p += (cnt + 3) & ~3LL; return r; }
// the arrays of one job's graph inside the wave's scratch; returns the first unused int (the cell region starts there)
__device__ __forceinline__ int *th_poa_carve(ThPoaG &g, int *my, long long cap, long long ocap, long long opn, long long upn = 0)
{
    int *p = my;
    g.nbase = th_carve(p, cap); g.nread = th_carve(p, cap); g.ring = th_carve(p, cap); g.inn = th_carve(p, cap); g.outn = th_carve(p, cap);
    g.in4 = th_carve(p, cap * TH_POA_INL); g.out4 = th_carve(p, cap * TH_POA_INL);
    g.in_oh = th_carve(p, cap); g.in_ot = th_carve(p, cap); g.out_oh = th_carve(p, cap); g.out_ot = th_carve(p, cap); g.pos = th_carve(p, cap);
    g.ovf_val = th_carve(p, ocap); g.ovf_next = th_carve(p, ocap); g.ovf_cnt = th_carve(p, 4);
    g.order = th_carve(p, cap); g.order2 = th_carve(p, cap);
    g.r_meta = th_carve(p, cap); g.r_p = th_carve(p, cap * 4); g.r_s = th_carve(p, cap * 4); g.r_rem = th_carve(p, cap);
    g.r_beg = th_carve(p, cap); g.r_end = th_carve(p, cap); g.r_mpl = th_carve(p, cap); g.r_mpr = th_carve(p, cap);
    g.r_off = th_carve(p, cap); g.r_foff = th_carve(p, cap);
    g.cntB = th_carve(p, cap); g.headB = th_carve(p, cap); g.aft = th_carve(p, cap);
    g.ops = th_carve(p, 2 * opn); g.pv = th_carve(p, opn); g.glo = th_carve(p, opn); g.pth = th_carve(p, opn + 2);
    g.upath = th_carve(p, upn); g.ccol = th_carve(p, upn ? cap : 0);      // (last: a caller that does not use them -- th_poa_rows -- carves the same offsets without)
    return p;
}

// ---- remain (longest node count to the sink) + row tables, 64 rows per trip from the end ----
// What a chunk needs from the graph -- node id, degrees, base, successor / predecessor ROW indices -- is three levels of dependent
// loads (order -> node arrays -> pos), and a successor's remain a fourth.  The phase was bound by that latency chain, not by the
// relaxation, so it is software-pipelined: each trip issues level 1 for the chunk three trips ahead, level 2 for the one two ahead,
// level 3 for the next one, and stores the PREVIOUS chunk's results; everything a trip consumes was issued a trip earlier.  The
// remain of a successor in the chunk just finished comes from that chunk's registers (a lane shuffle), not from memory.
#if defined(TH_POA_PROF)
__device__ unsigned long long th_poa_prof_rows[8];      // row-table phase: 0 fast chunks, 1 general chunks, 2 rounds of the fast loop (two per trip), 3 rounds of the general loop, 4 calls
#define TH_PROF_ROWS(k, v) do { if (TH_LANE() == 0) atomicAdd(&th_poa_prof_rows[k], (unsigned long long)(v)); } while (0)
// why a row of the DP took the one-column-per-lane path: 0 no predecessor (the source), 1 more than four, 2 one of them 64 rows or more back, 3 two or more outside the ring
// (or too wide for it), 4 its own band too wide for the ring, 5 other, 6 a job of the 32-bit variants
__device__ unsigned long long th_poa_prof_why[8];
#else
#define TH_PROF_ROWS(k, v) ((void)0)
#endif
#ifndef TH_POA_ROWS_ONE_ROUND
#define TH_POA_ROWS_ONE_ROUND 1      // one round per run scan (round 5: a trip finishes one more run whether it holds one round or two)
#endif
struct ThPoaRows2 { int u, on, in, nb, o4[TH_POA_INL], i4[TH_POA_INL]; };
struct ThPoaRows3 { int u, on, in, nb, s[TH_POA_INL], p[TH_POA_INL]; };
// (Round 5: the three stages are BRANCH-FREE.  A lane outside the chunk -- the top chunk's tail, or any lane while the pipeline drains below row 0 -- reads a row
// that exists (its index clamped) and an edge slot that does not exist reads node 0; what was read is then discarded by a select.  As predicated loads every
// `ok ? g.x[u] : 0` was a basic block of its own -- execution-mask save, branch, restore around one load: ~45 blocks per 64 rows, a quarter of the phase's
// instructions; the edge slots are read four at a time.)
__device__ __forceinline__ int th_poa_rows_l1(const ThPoaG &g, int c0, int n)
{
    return g.order[min(max(c0 + TH_LANE(), 0), n - 1)];
}
__device__ __forceinline__ void th_poa_rows_l2(const ThPoaG &g, int c0, int n, int u, ThPoaRows2 &a)
{
    const int idx = c0 + TH_LANE(); const bool ok = c0 >= 0 && idx < n;
    const int on = g.outn[u], in = g.inn[u], nb = g.nbase[u];
    const int4 o = ((const int4 *)g.out4)[u], i = ((const int4 *)g.in4)[u];          // (TH_POA_INL == 4 slots per node; slots beyond the degree hold anything)
    a.u = u; a.on = ok ? on : 0; a.in = ok ? in : 0; a.nb = ok ? nb : 0;
    a.o4[0] = o.x; a.o4[1] = o.y; a.o4[2] = o.z; a.o4[3] = o.w; a.i4[0] = i.x; a.i4[1] = i.y; a.i4[2] = i.z; a.i4[3] = i.w;
}
__device__ __forceinline__ void th_poa_rows_l3(const ThPoaG &g, int c0, int n, const ThPoaRows2 &a, ThPoaRows3 &r)
{
    r.u = a.u; r.on = a.on; r.in = a.in; r.nb = a.nb;          // (degrees are 0 in lanes outside the chunk: th_poa_rows_l2)
#pragma unroll
    for (int k = 0; k < TH_POA_INL; ++k) {
        const int ps = g.pos[k < a.on ? a.o4[k] : 0], pp = g.pos[k < a.in ? a.i4[k] : 0];
        r.s[k] = k < a.on ? ps : -1; r.p[k] = k < a.in ? pp : -1;
    }
}
__device__ __forceinline__ int th_poa_rows_body(const ThPoaG &g)
{
    const int lane = TH_LANE(), n = g.n_node;
    const int c_top = (n - 1) & ~63;
    int maxdeg = 0;
    int a1; ThPoaRows2 a2; ThPoaRows3 a3;
    // fill the pipeline (these waits are paid once per call)
    a1 = th_poa_rows_l1(g, c_top, n); th_poa_rows_l2(g, c_top, n, a1, a2); th_poa_rows_l3(g, c_top, n, a2, a3);
    a1 = th_poa_rows_l1(g, c_top - TH_WAVE, n); th_poa_rows_l2(g, c_top - TH_WAVE, n, a1, a2);
    a1 = th_poa_rows_l1(g, c_top - 2 * TH_WAVE, n);
    int prev_rem = -TH_BIG;                                   // remain of the chunk above (lane = row & 63)
    int st_idx = -1, st_rem = 0, st_meta = 0, st_p[TH_POA_INL] = { 0, 0, 0, 0 };       // results of the chunk above, stored one trip late
    for (int c0 = c_top; c0 >= 0; c0 -= TH_WAVE) {
        const int idx = c0 + lane; const bool ok = idx < n;
        const ThPoaRows3 a = a3;
        TH_WAVE_FENCE();
        th_poa_rows_l3(g, c0 - TH_WAVE, n, a2, a3);
        th_poa_rows_l2(g, c0 - 2 * TH_WAVE, n, a1, a2);
        a1 = th_poa_rows_l1(g, c0 - 3 * TH_WAVE, n);
        if (st_idx >= 0) {
            g.r_rem[st_idx] = st_rem; g.r_meta[st_idx] = st_meta;
#pragma unroll
            for (int k = 0; k < TH_POA_INL; ++k) g.r_p[st_idx * TH_POA_INL + k] = st_p[k];
        }
        TH_WAVE_FENCE();
        const int u = a.u, on = a.on, in = a.in;
        int s[TH_POA_INL], p[TH_POA_INL];
#pragma unroll
        for (int k = 0; k < TH_POA_INL; ++k) { s[k] = a.s[k]; p[k] = a.p[k]; }
        // successors outside the chunk are final: the chunk above hands its values over in registers, anything further up (rare: an
        // edge over more than 64 rows) is read from the table; in-chunk ones are relaxed in registers
        int base = (ok && u == 1) ? -1 : -TH_BIG;
        bool far = false;
#pragma unroll
        for (int k = 0; k < TH_POA_INL; ++k) {
            const int o = __shfl(prev_rem, (s[k] - c0) & 63);
            base = max(base, (unsigned)(s[k] - c0 - TH_WAVE) < (unsigned)TH_WAVE ? o + 1 : -TH_BIG);      // (selects, not branches: a divergent `if` is an execution-mask block per edge slot)
            far = far || s[k] >= c0 + 2 * TH_WAVE;
        }
        const bool big = ok && on > TH_POA_INL;
        if (__any(far || big)) {
            // (every chunk above this one is in the table by now: the last one was stored at the top of this trip)
#pragma unroll
            for (int k = 0; k < TH_POA_INL; ++k) if (s[k] >= c0 + 2 * TH_WAVE) base = max(base, g.r_rem[s[k]] + 1);
            if (big) {          // overflow successors that are already final
                int e = g.out_oh[u];
                for (int x = TH_POA_INL; x < on; ++x) { const int sx = g.pos[g.ovf_val[e]]; e = g.ovf_next[e]; if (sx >= c0 + TH_WAVE) base = max(base, g.r_rem[sx] + 1); }
            }
        }
        const bool hard = __any(big);
        // in-chunk successors are almost always 1..3 rows ahead: then a relaxation round is three lane shifts
        bool nearok = true; int has1 = 0, has2 = 0, has3 = 0;
#pragma unroll
        for (int k = 0; k < TH_POA_INL; ++k) {          // (selects, not branches; an absent slot holds -1: not in the chunk)
            const int dl = s[k] - idx; const bool inch = (unsigned)(s[k] - c0) < (unsigned)TH_WAVE;
            has1 |= (int)(inch && dl == 1); has2 |= (int)(inch && dl == 2); has3 |= (int)(inch && dl == 3);
            nearok = nearok && (!inch || (unsigned)(dl - 1) < 3u);
        }
        const bool fastrel = !hard && !__any(!nearok);
        int cur = base;
        // Round 5.  The rounds below (either kind) move a value ONE edge per round, and a chunk is mostly a chain (row i's successor is row i + 1): 64 rounds for 64
        // rows -- 860 of the phase's ~1500 vector instructions per chunk (profiles/r05_poa_phases.json: the phase was 12 % of the kernel's vector issue).
        // Along a run of rows whose next row is a successor, remain[i] >= remain[i + 1] + 1, i.e. remain + row is a SUFFIX MAXIMUM over the run: one
        // segmented scan (segment number in the high bits of the key, lanes reversed so that the prefix scan of th_hw.h serves) hands every lane the
        // best of its run, and the rounds only have to carry values across the run ends (branches of bubbles): two or three trips instead of 33.
        // Exact: the start value P obeys base <= P <= T(P) <= the fixed point of the rounds' operator T, which is monotone; the loop ends at the same
        // (unique: the graph is acyclic) fixed point it reached from `base`.
        // (the scan is repeated after every round of either loop below: a round carries values across the run ends, the scan down the whole run beneath -- one
        // more RUN final per trip, ~10 per chunk, where a round alone finishes one more ROW)
        const bool scan_ok = n < (1 << 21) - 128;
        const int seg = (int)__popcll(~__ballot(has1 != 0) >> lane) << 22;            // run ends at or above this lane (the same number all along a run, smaller above it), in the key's high bits
#define TH_POA_RUN_SCAN(cur) do { if (scan_ok) { \
            int key_ = seg | ((cur) > -(TH_BIG / 2) ? (cur) + lane + 2 : 0);    /* (0: no path yet -- a lane the rounds have not reached holds -TH_BIG plus a few rounds' drift, never a real value) */ \
            key_ = __shfl(key_, 63 - lane); key_ = th_wave_incl_max(key_); key_ = __shfl(key_, 63 - lane); \
            const int best_ = key_ & 0x3fffff; \
            if (best_) (cur) = best_ - lane - 2; } } while (0)
        TH_POA_RUN_SCAN(cur);
        if (fastrel) {
            // the usual chunk: three lane shifts, three adds, two max3 per round (an edge that does not exist adds minus "infinity";
            // values near -TH_BIG are "no path yet" and lose every maximum against a real one), two rounds per convergence test
            const int inc1 = has1 ? 1 : -TH_BIG, inc2 = has2 ? 1 : -TH_BIG, inc3 = has3 ? 1 : -TH_BIG;
            TH_PROF_ROWS(0, 1);
            for (int it = 0; it < TH_WAVE + 1; ++it) {
                TH_PROF_ROWS(2, TH_POA_ROWS_ONE_ROUND ? 1 : 2);
                int c1 = th_shl1(cur, -TH_BIG), c2 = th_shl1(c1, -TH_BIG), c3 = th_shl1(c2, -TH_BIG);
                int v2 = max(max(base, c1 + inc1), max(c2 + inc2, c3 + inc3));
                if (!TH_POA_ROWS_ONE_ROUND) {
                    c1 = th_shl1(v2, -TH_BIG); c2 = th_shl1(c1, -TH_BIG); c3 = th_shl1(c2, -TH_BIG);
                    v2 = max(max(base, c1 + inc1), max(c2 + inc2, c3 + inc3));
                }
                TH_POA_RUN_SCAN(v2);
                const bool ch = v2 != cur;
                cur = v2;
                if (!__any(ch)) break;
            }
        } else
        for (int it = 0; it < TH_WAVE + 1; ++it) {
            if (it == 0) TH_PROF_ROWS(1, 1);
            TH_PROF_ROWS(3, 1);
            int v = base;
            {
#pragma unroll
                for (int k = 0; k < TH_POA_INL; ++k) {
                    const int o = __shfl(cur, (s[k] - c0) & 63);
                    if (s[k] >= c0 && s[k] < c0 + TH_WAVE && o > -TH_BIG) v = max(v, o + 1);
                }
            }
            if (hard) {     // rare: a node with > 4 successors, some inside the chunk -> exchange through memory
                if (ok) g.r_rem[idx] = cur;
                TH_WAVE_FENCE();
                if (big) {
                    int e = g.out_oh[u];
                    for (int x = TH_POA_INL; x < on; ++x) {
                        const int sx = g.pos[g.ovf_val[e]]; e = g.ovf_next[e];
                        if (sx >= c0 && sx < c0 + TH_WAVE) { const int o = g.r_rem[sx]; if (o > -TH_BIG) v = max(v, o + 1); }
                    }
                }
                TH_WAVE_FENCE();
            }
            TH_POA_RUN_SCAN(v);
            const bool ch = ok && v != cur;
            cur = v;
            if (!__any(ch)) break;
        }
        {
            int needfull = on > TH_POA_INL;
#pragma unroll
            for (int k = 0; k < TH_POA_INL; ++k) if (s[k] >= 0 && (s[k] - idx >= TH_POA_RING || s[k] == n - 1)) needfull = 1;
            // bit 26: every predecessor (one to four) is one of the three rows before this one, and no far row reads this
            // one -- what the graph decides of the DP's bubble-row path (the band widths and cache states it also needs are known only there)
            bool ring_ok = in >= 1 && in <= TH_POA_INL && !needfull && idx > 0;
#pragma unroll
            for (int k = 0; k < TH_POA_INL; ++k) if (k < in && !(p[k] < idx && idx - p[k] < TH_POA_RING)) ring_ok = false;      // (round 5: across a 64-row boundary too -- lane (row & 63) of the DP's chunk registers keeps a row's band until row + 64 overwrites it)
            st_idx = ok ? idx : -1; st_rem = cur; st_meta = (int)ring_ok << 26 | a.nb | min(in, 255) << 8 | min(on, 255) << 16 | needfull << 24 | (int)(in == 1 && p[0] == idx - 1 && !needfull) << 25;     // bit 25: the DP's common row, as far as the graph decides it
#pragma unroll
            for (int k = 0; k < TH_POA_INL; ++k) st_p[k] = p[k];
        }
        prev_rem = cur;
        maxdeg = max(maxdeg, in);
    }
    if (st_idx >= 0) {
        g.r_rem[st_idx] = st_rem; g.r_meta[st_idx] = st_meta;
#pragma unroll
        for (int k = 0; k < TH_POA_INL; ++k) g.r_p[st_idx * TH_POA_INL + k] = st_p[k];
    }
    TH_WAVE_FENCE();
    return th_wave_max(maxdeg);
}


// The phase as a function of its own (not inlined on the device): its registers -- three chunks in flight -- are allocated apart
// from the DP row loop's, which loses its loop-invariant registers to scratch otherwise.  The graph descriptor is rebuilt from the
// wave's scratch base and the job's capacities (a by-reference descriptor would have to live in memory).
#ifdef TH_POA_ROWS_INLINE      // (measurement only: the phase inlined into the job loop)
#define TH_POA_ROWS_ATTR
#else
#define TH_POA_ROWS_ATTR TH_NOINLINE
#endif
__device__ TH_POA_ROWS_ATTR int th_poa_rows(int *my, int cap, int ocap, int opn, int n_node, int order_swapped)
{
    my = TH_UNI_PTR(int, my);
    ThPoaG g;
    th_poa_carve(g, my, TH_UNI(cap), TH_UNI(ocap), TH_UNI(opn));
    g.cap = TH_UNI(cap); g.n_node = TH_UNI(n_node);
    if (TH_UNI(order_swapped)) { int *t = g.order; g.order = g.order2; g.order2 = t; }     // th_poa_merge swaps the two order arrays after every alignment
    const int r_ = th_poa_rows_body(g);
    return r_;
}

#if defined(TH_POA_PROF)
__device__ unsigned long long th_poa_prof[32];    // 24: consensus cycles, 25: staging + seed cycles, 26: jobs; 0..3 phase cycles; 4..7 traceback events (tiles loaded, run trips, run steps, single steps); 8..15 rows per category, 16..23 their cycles
// per-wave accumulators of the row loop in LDS, one 64-bit add per row: rows in bits 40.., cycles below (per-lane register arrays of round 4 cost the loop 32 VGPRs)
__shared__ unsigned long long s_poa_prof[10];     // 8: further trips of packed rows; 9: packed rows with a far predecessor / a far reader
#define TH_PROF_ROW(cat, dt) do { if (TH_LANE() == 0) atomicAdd(&s_poa_prof[cat], (1ull << 40) + (unsigned long long)(dt)); } while (0)
#define TH_PROF_CNT(k, v) do { if (TH_LANE() == 0) atomicAdd(&th_poa_prof[k], (unsigned long long)(v)); } while (0)
// -DTH_POA_PROF=1: events only (rows per path of the row loop, further trips, traceback events); =2: s_memtime cycles per phase and per row path as well
#if TH_POA_PROF >= 2
#define TH_PROF_NOW() __builtin_amdgcn_s_memtime()
#else
#define TH_PROF_NOW() 0ull
#endif
#define TH_PROF_T0() unsigned long long pt_ = TH_PROF_NOW()
#define TH_PROF(k) do { unsigned long long n_ = TH_PROF_NOW(); if (TH_POA_PROF >= 2 && TH_LANE() == 0) atomicAdd(&th_poa_prof[k], n_ - pt_); pt_ = n_; } while (0)
#else
#define TH_PROF_CNT(k, v) ((void)0)
#define TH_PROF_T0() ((void)0)
#define TH_PROF(k) ((void)0)
#endif

// Every phase of an alignment runs inside a loop whose trip count -- ONE -- the compiler cannot see.  Measured, not designed (round 5, profiles/r05_poa_phases.json:
// the phase-attribution build with no phase repeated ran 390.4 ms against the plain product's 394.9, 195.6 against 202.2 x 10^9 vector instructions per launch):
// the opaque loops keep hipcc from moving code and live ranges across the phase boundaries of the job loop, which the row loop pays for in scalar spills.
// -DTH_POA_DUP_PHASE=k (tools/poa_phases.sh) makes phase k's count two.
#ifndef TH_POA_DUP_PHASE
#define TH_POA_DUP_PHASE (-1)
#endif
__device__ __forceinline__ int th_dup_n(int k) { int n = k == TH_POA_DUP_PHASE ? 2 : 1; TH_OPAQUE_S(n); return n; }
#define TH_DUP_N(k) th_dup_n(k)

// ---- banded DP; returns the row of the sink's best predecessor (>= 0) or a negative status ----
// The three most recent rows stay in an LDS ring (H, E1', E2' per column, int16) together with their band and
// row-maximum columns in wave-uniform registers, so a row whose predecessors are among them -- nearly all rows:
// the previous node and the members of the previous column -- issues no HBM load at all.
//
// Ring layout: a row with band [b, e] keeps column j at index j - (b & ~1) + TH_POA_PAD of each of its three arrays,
// so columns pair up (even, odd) in aligned 32-bit words.  -32768 is "minus infinity".
//
// PACKED ROWS.  A row whose only predecessor is the previous row (97 % of rows) and whose band fits the ring is
// computed two columns per lane on packed int16 pairs (v_pk_add_i16 clamp / v_pk_max_i16): 128 columns per trip,
// F1 and F2 are scanned once per lane PAIR, comparisons are sign bits of saturating differences, selections are
// v_bfi.  Such a row rewrites its whole ring slot (minus infinity outside its band), which is what lets the next
// packed row read its predecessor without bounds checks ("clean" slot).  Exactness: all real scores stay inside
// int16 (pk_ok below); an unreachable in-band cell may drift upwards from -32768 by at most M per row, which pk_ok
// keeps below every real score, and readers that work in 32 bits widen anything below `neg_thr` back to TH_NEG.
// RELATIVE SCORES (TH_POA_REL, the long-unit configuration).  The ring holds H - base, E1' - base, E2' - base for a wave-uniform `base` that
// follows the row maxima: the recurrences are translation invariant, so rows computed from ring rows need no change at all; only the
// borders convert -- full cells in HBM are absolute (+ base when written, - base when a far predecessor is read), the 32-bit path
// widens / narrows through TH_RING_LD / TH_RING_ST -- and when a row maximum leaves [-TH_POA_REL_HI, TH_POA_REL_HI] the ring is
// rebased (every few thousand rows).  Real values of the ring rows lie within ~3 x ring width + 24 of their row maximum (a cell is
// reached from the path's cell of its column by a vertical gap, from the row maximum's cell by a horizontal one), far above
// TH_POA_REL_THR; anything below that threshold is minus infinity and is stored as exactly -32768 (it cannot drift upwards).
#if TH_POA_REL
#define TH_POA_SET_RMAX(v) (L_rmax = TH_UNI(v))
#define TH_RING_LD(x) ((int)(x) < TH_POA_REL_THR ? TH_NEG : (int)(x) + base)
#define TH_RING_ST(v) ((RT)((v) - base < TH_POA_REL_THR ? -32768 : (v) - base))
#else
#define TH_POA_SET_RMAX(v) ((void)0)
#define TH_RING_LD(x) (WIDE ? (int)(x) : ((int)(x) < neg_thr ? TH_NEG : (int)(x)))
#define TH_RING_ST(v) ((RT)(WIDE ? (v) : max((v), -32768)))
#endif
// slot s of the ring may hold values in the columns of its trips 1 .. TH_TD_GET(td, s) (trip t = columns 128 t .. 128 t + 127 of the slot)
#define TH_TD_GET(td, s) (((td) >> (2 * (s))) & 3)
#define TH_TD_SET(td, s, v) ((td) = ((td) & ~(3 << (2 * (s)))) | ((v) << (2 * (s))))
#define TH_RIDX(j, beg) ((j) - ((beg) & ~1) + TH_POA_PAD)
#define TH_LC_ON (1 << 25)      // L_c (the previous row is cached, clean, in its ring slot) as the bit of the row tables' meta word it is tested against

// One packed row: lane l owns columns B0 + 128 t + 2 l and the next one; every predecessor (npk of them, ring slot q_slot*, even
// band base q_pb*) sits clean in the LDS ring.  Writes the row's ring slot and its 16-bit direction words (drow: column B0), returns
// the row maximum and its first / last column through rmax / left / right.  Forced inline: the row loop calls it twice, once
// with npk = 1 and the previous row's slot as constants (the common row: COMMON, its band never starts left of its predecessor's), once in general.
// FAR: one predecessor (edge index far_o; none if negative) is NOT in the ring -- more than three rows back, or too wide to be cached --
// and is read from its full cells in HBM (far_row: its cell of column far_pb; band far_pb..far_pe), converted to int16 with saturation
// (its values are exact 32-bit ones; minus infinity becomes -32768 like everything outside a ring row's band); full_row (or null):
// the row's own full cells {H, E1', E2', -} are written for a far reader, minus infinity widened back to TH_NEG below neg_thr.
// D8: the row's direction words are 8-bit ones (1), 16-bit ones (0), or 8-bit iff npk == 1 (2) -- the rule is "in-degree 1" everywhere
template <int QCAP, typename RT, bool COMMON, bool FAR, int D8>
__device__ __forceinline__ void th_poa_pk_row(const uint8_t *s_qs, RT (*s_ring)[3][TH_POA_RS], uint16_t *drow, const int lane, const int b, const int e, const int B0,
                                               const int myslot, const uint32_t T, const int npk,
                                               const int q_slot0, const int q_slot1, const int q_slot2, const int q_slot3, const int q_pb0, const int q_pb1, const int q_pb2, const int q_pb3,
                                               const int E1, const int E2, const uint32_t O1P, const uint32_t O2P, const uint32_t E1P, const uint32_t E2P, int &tail_dirty, int &rmax, int &left, int &right,
                                               const int far_o = -1, const int far_pb = 0, const int far_pe = -1, const int4 *far_row = nullptr, int4 *full_row = nullptr, const int neg_thr = 0, const int base = 0)
{
    // ================= packed row: lane l owns columns B0 + 128 t + 2 l and the next one =================
    uint32_t *wH = (uint32_t *)s_ring[myslot][0], *wE1 = (uint32_t *)s_ring[myslot][1], *wE2 = (uint32_t *)s_ring[myslot][2];
    int carry1 = -32768, carry2 = -32768;                   // F1 / F2 running maxima in the column-offset domain
    int i1p = 0, i2p = 0;                                   // the previous trip's F1 / F2 scans: their last lane is read only if a further trip runs (four rows in five have one trip)
    for (int t0 = 0; t0 < TH_POA_TRIPS * 128; t0 += 128) {
        const int jr = t0 + 2 * lane, j0 = B0 + jr;
        const int wi = (jr + TH_POA_PAD) >> 1;              // this lane's pair in the row being written
#if TH_POA_TRIPS == 2
        if (t0 && B0 + t0 > e) {                             // one trip was enough: clear the rest of the slot, unless it is clean already
            if (tail_dirty & (1 << myslot)) {                // (bit s of tail_dirty: slot s may hold values beyond its first 128 columns)
                if (wi < TH_POA_RS / 2) { wH[wi] = TH_PK_NEG; wE1[wi] = TH_PK_NEG; wE2[wi] = TH_PK_NEG; }
                tail_dirty &= ~(1 << myslot);
            }
            break;
        }
        if (t0) tail_dirty |= 1 << myslot;
#if defined(TH_POA_PROF)
        if (t0) TH_PROF_ROW(8, 0);
#endif
#else
        if (t0 && B0 + t0 > e) {                             // the trips so far were enough: clear the rest of the slot, unless it is clean already
            const int tdone = t0 >> 7, told = TH_TD_GET(tail_dirty, myslot);      // trips 0 .. tdone - 1 written now; trips 1 .. told held values before
            for (int tt = tdone; tt <= told; ++tt) {
                const int wc = wi + 64 * (tt - tdone);
                if (wc < TH_POA_RS / 2) { wH[wc] = TH_PK_NEG; wE1[wc] = TH_PK_NEG; wE2[wc] = TH_PK_NEG; }
            }
            TH_TD_SET(tail_dirty, myslot, tdone - 1);
            break;
        }
        if (t0 && (t0 >> 7) > TH_TD_GET(tail_dirty, myslot)) TH_TD_SET(tail_dirty, myslot, t0 >> 7);
#endif
        if (t0) { const int c1_ = TH_READLANE(i1p, 63), c2_ = TH_READLANE(i2p, 63); carry1 = t0 > 128 ? max(carry1, c1_) : c1_; carry2 = t0 > 128 ? max(carry2, c2_) : c2_; }
        const bool lo_ok = j0 >= b && j0 <= e, hi_ok = j0 + 1 <= e;
        const uint32_t IM = (lo_ok ? 0xffffu : 0u) | (hi_ok ? 0xffff0000u : 0u);
        const uint32_t NM = th_pk_neg_outside(IM);           // minus infinity in the halves outside the band: x inside | NM = th_sel(IM, x, TH_PK_NEG) as one and-or of three vector registers
        // substitution scores of the two columns from the row's table
        const uint32_t qx = s_qs[TH_LDS_SLACK_INDEX(j0, QCAP) >> 1];      // the pair's two 4-bit query codes; columns beyond e may read past the staged codes (still inside the workgroup's LDS): they are masked out below
        // T: the four biased scores of this row's node base against A, C, G, T, one per byte; code 4 (N, and the column 0 / beyond-the-
        // query filler) selects byte 0 of the other source, the bias itself.  One byte permute looks both columns up.
        const uint32_t SC = TH_PK_SUB_IMM(th_perm(32u, T, th_pk_score_sel(qx)), 32);
        // predecessor rows (all in the ring): diagonal pair (j0-1, j0) and vertical pairs (j0, j0+1) of each; anything outside
        // a row's band reads -inf (its slot is clean, indices beyond either end are sent to the padding).  The first
        // predecessor in edge order wins ties, as in the one-column-per-lane path; PMV / PE1V / PE2V hold the winners' edge
        // indices, already shifted to their place in the 16-bit direction word
        uint32_t m = 0, e1v = 0, e2v = 0, PMV = 0, PE1V = 0, PE2V = 0;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < npk) {
                const int qs = o == 0 ? q_slot0 : o == 1 ? q_slot1 : o == 2 ? q_slot2 : q_slot3, qb = o == 0 ? q_pb0 : o == 1 ? q_pb1 : o == 2 ? q_pb2 : q_pb3;
                const uint32_t *rH = (const uint32_t *)s_ring[qs][0], *rE1 = (const uint32_t *)s_ring[qs][1], *rE2 = (const uint32_t *)s_ring[qs][2];
                int pi = j0 - qb + TH_POA_PAD;
                pi = pi <= TH_POA_RS - 2 ? pi : 2;
                if (!COMMON) pi = max(pi, 2);      // (the common row starts at or right of its predecessor's even base)
                uint32_t mo, c1, c2;
                if (FAR && o == far_o) {
                    const int ja = j0 - 1, jc = j0 + 1;
                    int ha = TH_NEG, hb = TH_NEG, e1b = TH_NEG, e2b = TH_NEG, e1c = TH_NEG, e2c = TH_NEG;
                    if (ja >= far_pb && ja <= far_pe) ha = far_row[ja - far_pb].x;
                    if (j0 >= far_pb && j0 <= far_pe) { const int4 cc = far_row[j0 - far_pb]; hb = cc.x; e1b = cc.y; e2b = cc.z; }
                    if (jc >= far_pb && jc <= far_pe) { const int4 cc = far_row[jc - far_pb]; e1c = cc.y; e2c = cc.z; }
                    mo = th_pk_add(th_cvt_pk_i16(ha - base, hb - base), SC); c1 = th_cvt_pk_i16(e1b - base, e1c - base); c2 = th_cvt_pk_i16(e2b - base, e2c - base);
                } else {
                    const uint32_t Hm = rH[(pi >> 1) - 1], Hc = rH[pi >> 1];
                    c1 = rE1[pi >> 1]; c2 = rE2[pi >> 1];
                    mo = th_pk_add(th_alignbit16(Hc, Hm), SC);
                }
                if (o == 0) { m = mo; e1v = c1; e2v = c2; }
                else {
                    const uint32_t gm = th_pk_negmask_op(th_pk_sub(m, mo)), ge1 = th_pk_negmask_op(th_pk_sub(e1v, c1)), ge2 = th_pk_negmask_op(th_pk_sub(e2v, c2));
                    m = th_pk_max(m, mo); e1v = th_pk_max(e1v, c1); e2v = th_pk_max(e2v, c2);
                    PMV = th_sel(gm, TH_PK_DUP(o << TH_D16_PM_SHIFT), PMV); PE1V = th_sel(ge1, TH_PK_DUP(o << TH_D16_PE1_SHIFT), PE1V); PE2V = th_sel(ge2, TH_PK_DUP(o << TH_D16_PE2_SHIFT), PE2V);
                }
            }
        }
        const uint32_t hnf1 = th_pk_max(m, e1v), dg1 = th_pk_sub(m, e1v);           // sign of dg1: E1 strictly better
        const uint32_t hnf = th_pk_max(hnf1, e2v), dg2 = th_pk_sub(hnf1, e2v);
        // row maximum: the maximum of H is a maximum of hnf (an F value is always below the hnf it came from), and H == max only
        // where hnf == max.  Scanned here, next to the two F scans: three independent dependency chains fill each other's wait states
        const uint32_t hm2 = th_and_or(hnf, IM, NM);
        const int hmv = th_wave_incl_max(max(TH_PK_LO(hm2), TH_PK_HI(hm2)));
        // F1 / F2: max-plus prefix over the row in the domain value + column * extension
        const uint32_t JR = TH_PK(jr, jr + 1);
        const uint32_t JE1 = TH_PK(TH_PK_LO(JR) * E1, TH_PK_HI(JR) * E1), JE2 = TH_PK(TH_PK_LO(JR) * E2, TH_PK_HI(JR) * E2);
        const uint32_t a1 = th_and_or(th_pk_add(th_pk_sub(hnf, O1P), JE1), IM, NM);
        // (the default second gap open, 24, as an inline operand of the subtraction: as -24 of an addition it is a literal in a scalar register, moved there per trip)
#define TH_SUB_O2(x) ((__builtin_constant_p(O2P) && O2P == TH_PK_DUP(24)) ? TH_PK_SUB_IMM((x), 24) : th_pk_sub((x), O2P))
        const uint32_t a2 = th_and_or(th_pk_add(TH_SUB_O2(hnf), JE2), IM, NM);
        // lane totals are scanned as two 32-bit values: v_max_i32 takes the DPP operand directly (one instruction per
        // step), a packed max would need a move and an identity fill per step
        const int a1l = TH_PK_LO(a1), a2l = TH_PK_LO(a2);
        const int i1 = th_wave_incl_max(max(a1l, TH_PK_HI(a1))), i2 = th_wave_incl_max(max(a2l, TH_PK_HI(a2)));
        // best before this lane's pair.  (In the first trip the carries are minus infinity, the smallest value of the domain: nothing to take the
        // maximum with; the compiler peels that trip, so the test costs nothing.)
        const int b1 = t0 ? max(th_shr1(i1, carry1), carry1) : th_shr1(i1, -32768), b2 = t0 ? max(th_shr1(i2, carry2), carry2) : th_shr1(i2, -32768);
        i1p = i1; i2p = i2;
        const uint32_t x1 = TH_PK(b1, max(b1, a1l)), x2 = TH_PK(b2, max(b2, a2l));                 // (before lo, before hi)
        const uint32_t f1 = th_pk_sub(x1, JE1), f2 = th_pk_sub(x2, JE2);
        const uint32_t h3 = th_pk_max(hnf, f1), dg3 = th_pk_sub(hnf, f1);
        const uint32_t h = th_pk_max(h3, f2), dg4 = th_pk_sub(h3, f2);
        // direction word of the two columns: the sign bits of eight saturating differences, two differences per byte permute (its sign
        // selectors deliver 0xff / 0x00 per half), one bit position per pair (layout: th_k_poa.hip)
        const uint32_t hO1 = th_pk_sub(h, O1P), hO2 = TH_SUB_O2(h);
#undef TH_SUB_O2
        const uint32_t xf1 = th_pk_sub(hO1, f1), xf2 = th_pk_sub(hO2, f2);
        const uint32_t n1 = th_pk_sub(th_pk_max(hO1, e1v), E1P), n2 = th_pk_sub(th_pk_max(hO2, e2v), E2P);
        // column b-1 / e+1 of an odd edge land in the row's own padding.  drow is the word of column B0 (the row's even base): a
        // wave-uniform pointer + the lane's constant byte offset (+ the trip's): no 64-bit address arithmetic per trip on the scalar unit
        if (TH_POA_D8 && (D8 == 1 || (D8 == 2 && npk == 1))) {
            // one predecessor: no edge indices, eight bits per cell (layout: th_k_poa.hip) -- a difference's two signs land side by side
            // (bytes 0, 1 for the first of a pair, 2, 3 for the second), the halves are folded, the lane stores its two cells as one short
            uint32_t y = th_perm(dg1, dg2, TH_PERM_SIGNS8) & 0x01011010u;
            y = th_and_or(th_perm(dg3, dg4, TH_PERM_SIGNS8), 0x02022020u, y);
            y = th_and_or(th_perm(th_pk_sub(hO1, e1v), th_pk_sub(hO2, e2v), TH_PERM_SIGNS8), 0x04044040u, y);
            y = th_and_or(th_perm(xf1, xf2, TH_PERM_SIGNS8), 0x08088080u, y);
            y |= y >> 16;
            if (j0 <= e) TH_STORE_U16_SADDR(drow, (uint32_t)(2 * lane + t0), y);
        } else {
            uint32_t code = th_perm(dg1, dg2, TH_PERM_SIGNS) & 0x01010101u;
            code = th_and_or(th_perm(dg3, dg4, TH_PERM_SIGNS), 0x02020202u, code);
            code = th_and_or(th_perm(th_pk_sub(hO1, e1v), th_pk_sub(hO2, e2v), TH_PERM_SIGNS), 0x04040404u, code);
            code = th_and_or(th_perm(xf1, xf2, TH_PERM_SIGNS), 0x08080808u, code);
            if (npk > 1) code |= PMV | PE1V | PE2V;
            if (j0 <= e) TH_STORE_U32_SADDR(drow, (uint32_t)(4 * lane + 2 * t0), code);
        }
        if (t0 == 0 || wi < TH_POA_RS / 2) {     // (every lane of the first trip is inside the slot)
#if TH_POA_REL
            const uint32_t THRP = TH_PK_DUP(TH_POA_REL_THR);          // below the threshold: minus infinity, stored as exactly that
            wH[wi] = th_sel(IM & ~th_pk_negmask_op(th_pk_sub(h, THRP)), h, TH_PK_NEG);
            wE1[wi] = th_sel(IM & ~th_pk_negmask_op(th_pk_sub(n1, THRP)), n1, TH_PK_NEG);
            wE2[wi] = th_sel(IM & ~th_pk_negmask_op(th_pk_sub(n2, THRP)), n2, TH_PK_NEG);
#else
            wH[wi] = th_and_or(h, IM, NM); wE1[wi] = th_and_or(n1, IM, NM); wE2[wi] = th_and_or(n2, IM, NM);
#endif
        }
        if (FAR && full_row) {
            const int hl = TH_PK_LO(h), hh = TH_PK_HI(h), n1l = TH_PK_LO(n1), n1h = TH_PK_HI(n1), n2l = TH_PK_LO(n2), n2h = TH_PK_HI(n2);
            if (lo_ok) { int4 cv; cv.x = hl < neg_thr ? TH_NEG : hl + base; cv.y = n1l < neg_thr ? TH_NEG : n1l + base; cv.z = n2l < neg_thr ? TH_NEG : n2l + base; cv.w = 0; full_row[j0 - b] = cv; }
            if (hi_ok) { int4 cv; cv.x = hh < neg_thr ? TH_NEG : hh + base; cv.y = n1h < neg_thr ? TH_NEG : n1h + base; cv.z = n2h < neg_thr ? TH_NEG : n2h + base; cv.w = 0; full_row[j0 + 1 - b] = cv; }
        }
        // the row maximum's first / last column: the first / last lane with either half at the maximum, then which half.  (Some
        // in-band lane holds the maximum, so the masks are never both empty; the first trip always sets all three.)
        const int hm = TH_READLANE(hmv, 63);
        // (each half compared with the scalar maximum as a sign-extended operand: no packed copy of the maximum to build on the scalar unit)
        const unsigned long long mlo = __ballot(TH_PK_LO(hm2) == hm), mhi = __ballot(TH_PK_HI(hm2) == hm);
        const unsigned long long many = mlo | mhi;
        if (t0 == 0 || hm >= rmax) {               // (`many` is never empty: a trip runs only if it holds an in-band column, and that lane reaches the trip's maximum)
            const int fa = th_ctz64nz(many), la = th_msb64nz(many);
            const int fl = B0 + t0 + 2 * fa + (int)(~(mlo >> fa) & 1ull);
            const int lr = B0 + t0 + 2 * la + (int)((mhi >> la) & 1ull);
            if (t0 == 0 || hm > rmax) { rmax = hm; left = fl; right = lr; } else right = lr;
        }
    }
}

// end of a packed row taken by one of the two straight-line paths of the row loop: the row becomes "the previous row", its band /
// maximum columns / direction-word offset go into lane (row & 63) of the chunk registers, a complete chunk goes to the row tables
#if defined(TH_POA_PROF)
// (-DTH_POA_PROF=1, events only: the "cycles" field of a row category carries its BAND CELLS instead -- e - b + 1 per row: the DP cells of the bench line's secondary roofline)
#define TH_POA_PK_PROF(cat) TH_PROF_ROW(cat, TH_POA_PROF >= 2 ? TH_PROF_NOW() - row_t0 : (unsigned long long)(e - b + 1));
#else
#define TH_POA_PK_PROF(cat)
#endif
// -DTH_POA_DUP_ROW=cat (tools/poa_row_paths.sh; absolute-score variants only): every row the row loop takes on path `cat` (the TH_POA_PROF categories) is computed
// TWICE -- the loop's scalars restored in between, everything a row writes is written again with the same values -- so that the PMC totals of that build minus those of
// the build with cat = -2 (same code, no row repeated), over the rows of the path, are the path's instructions per row, further trips and set-up included.
#if defined(TH_POA_DUP_ROW) && !TH_POA_REL
#define TH_ROW_CAT(c) (row_cat = (c));
#else
#define TH_ROW_CAT(c)
#endif
#define TH_POA_PK_ROW_DONE(cat, FOFF) do { \
        L_b = b; L_e = e; L_l = TH_UNI(left); L_r = TH_UNI(right); L_c = TH_LC_ON; TH_POA_SET_RMAX(rmax); \
        TH_POA_PK_PROF(cat) TH_ROW_CAT(cat) \
        th_writelane6(m_beg, m_end, m_left, m_right, m_off, m_foff, b, e, L_l, L_r, my_off, (FOFF), sl); \
        if (i == flush_row) { \
            const int r_ = (i & ~63) + lane; \
            if (r_ <= i) { g.r_beg[r_] = m_beg; g.r_end[r_] = m_end; g.r_off[r_] = m_off; g.r_foff[r_] = m_foff; g.r_mpl[r_] = m_left; g.r_mpr[r_] = m_right; } \
        } \
        TH_WAVE_FENCE(); \
    } while (0)

// QCAP > 0: the query is staged in LDS (s_qs); RT = short: int16 ring with packed rows, RT = int: 32-bit ring (exact for any
// score range, one column per lane).  The int16 variant returns -5 when the alignment's scores may not fit.
// DEFSC: the scoring is abPOA's default as TideHunter sets it (match 2, mismatch 4, gaps 4 / 2 and 24 / 1: src/abpoa_cons.c:12-28) and is compiled in --
// the six values, their packed copies and the score table stop occupying scalar and vector registers of a loop that has none to spare (the packed
// constants become inline operands of the v_pk_* instructions); any other scoring (-M / -X / -O / -E) runs the generic instantiation
template <int QCAP, typename RT, bool DEFSC>
__device__ int th_poa_dp(ThPoaG &g, const uint8_t *__restrict__ q, const uint8_t *s_qs, RT (*s_ring)[3][TH_POA_RS], int (*s_rmeta)[8], int qlen, const ThDevPara &P, const bool noband)
{
    constexpr bool LQ = QCAP > 0, WIDE = sizeof(RT) == 4;
    int lane = TH_LANE();
    // the per-lane constants of the row loop (column offsets times the gap extensions, ring addresses ...) are derived from this copy:
    // they are rebuilt at every call instead of being hoisted out of the job's alignment loop, where they would stay live -- and
    // get spilled to scratch, to be reloaded INSIDE the row loop -- across the other phases
    TH_OPAQUE_V(lane);
    const int M = DEFSC ? 2 : P.match, X = DEFSC ? 4 : P.mismatch, O1 = DEFSC ? 4 : P.go1, E1 = DEFSC ? 2 : P.ge1, O2 = DEFSC ? 24 : P.go2, E2 = DEFSC ? 1 : P.ge2;
    qlen = TH_UNI(qlen);
    // noband (the rerun of a job whose adaptive band failed): a half-width beyond any column makes every row span 0..qlen through
    // the ordinary band arithmetic (a row's start is max(0, . - w) and never below a predecessor's, its end min(qlen, . + w))
    const int n = TH_UNI(g.n_node), w = noband ? qlen + 8 : TH_POA_WB + qlen / 100;
    uint32_t ncell = 0, nfull = 0;          // direction words / full cells handed out so far (32-bit: the region is clamped to 2^31 words)
    // what is left for direction words, in 2-byte units: the region minus the full cells handed out (8 units each).  ONE scalar for the capacity test of
    // every row -- the pair (nfull, g.cell_u2) was two spilled scalars read back with v_readlane per row (438.8 -> 433.0 ms); it goes negative when full
    // cells alone overrun the region
    int lim = (int)g.cell_u2;
    // Every real score lies within +-real_b: a path to (i, j) has at most i + j unit steps, the dearest of which is a
    // mismatch (X for two units) or a one-base gap (open + extend for one unit); E'/F carry one more open + extend.
    // Rows are cached as int16 (and packed rows are used) only if that, the upward drift of unreachable cells (M per
    // row) and the column offsets of the F scan (2 * ring width * extension) all fit below 32000.
    const int maxE = max(E1, E2), gap1 = min(O1 + E1, O2 + E2);
    const long long real_b = (long long)max(max(M, X), gap1) * (qlen + n) + O1 + O2 + 2 * maxE;
#if TH_POA_REL
    // relative scores: no bound on the alignment's length; the ring's dynamic range (see above: 3 x ring width x the dearest step) + the F
    // scan's column offsets must fit between the lowest row maximum the rebasing allows (-TH_POA_REL_HI) and the threshold
    const bool cache16_ok = M <= 31 && X <= 31 && (long long)max(max(M, X), gap1) * 3 * (TH_POA_RW + 64) + 2LL * (TH_POA_RW + 64) * maxE + O1 + O2 < -TH_POA_REL_THR - TH_POA_REL_HI;
    if (!cache16_ok) return -5;
    const int neg_thr = TH_POA_REL_THR; (void)real_b;
#else
    const bool cache16_ok = WIDE || (M <= 31 && X <= 31 && real_b + (long long)M * n + 2LL * (TH_POA_RW + 64) * maxE < 32000);
    if (!cache16_ok) return -5;
    const int neg_thr = -(int)real_b - 1;
#endif
    int base = 0, L_rmax = 0;         // TH_POA_REL: ring values are relative to base; L_rmax: the previous row's maximum as the ring holds it
    // packed constants
    uint32_t O1P = TH_PK_DUP(O1), O2P = TH_PK_DUP(O2), E1P = TH_PK_DUP(E1), E2P = TH_PK_DUP(E2);
    // the scalar register file of this kernel is full (the graph descriptor alone is ~35 pointer pairs), and a spilled scalar costs a
    // v_readlane -- a vector instruction -- at every use: the four constants every packed row uses several times live in VGPRs instead
    if (!DEFSC) { TH_OPAQUE_V(O1P); TH_OPAQUE_V(O2P); TH_OPAQUE_V(E1P); TH_OPAQUE_V(E2P); }
    const uint32_t TXB = (uint32_t)(32 - X) * 0x01010101u;           // score table of a packed row: one byte per query code A, C, G, T, biased by 32 ...
    // ... and with the match score in place for each node base: lane nb (0..4; 4 = N scores 0 against everything) of one register,
    // a row fetches its table with one v_readlane
    const uint32_t v_T = lane >= 4 ? 0x20202020u : ((TXB & ~(0xffu << (8 * lane))) | (uint32_t)(32 + M) << (8 * lane));
    int c_meta = 0, c_rem = 0, c_p0 = -1, c_p1 = -1, c_p2 = -1, c_p3 = -1, c_id = 0;   // row-table chunk cache
    // the previous row's band / row-maximum columns / cache state stay in wave-uniform registers (the common row has
    // exactly that predecessor); older ring slots keep theirs in s_rmeta[slot][]:
    //   0 row, 1 beg, 2 end, 3 left, 4 right, 5 cache state (0 none, 3 cached; every cached slot is clean), 6..7 cell offset
    int L_b = 0, L_e = -1, L_l = 0, L_r = 0, L_c = 0;
    // band, row-maximum columns and cell offset of the rows of the current 64-row chunk: lane (row & 63) of these registers
    // (one v_writelane each per row); they go to the row tables in HBM 64 rows at a time, so the common row issues no
    // metadata store at all.  A predecessor inside the chunk is read back with v_readlane, an older one from the tables.
    int m_beg = 0, m_end = 0, m_left = 0, m_right = 0, m_off = 0, m_foff = 0;
    int flush_row = 0;
#if TH_POA_TRIPS == 2
    int tail_dirty = (1 << TH_POA_RING) - 1;      // ring slots whose columns beyond the first 128 are not known to hold minus infinity (see th_poa_pk_row)
#else
    int tail_dirty = (1 << (2 * TH_POA_RING)) - 1;      // per ring slot (2 bits): the last trip whose columns are not known to hold minus infinity; all of them at first
#endif
    for (int x = lane; x < TH_POA_RING * 3 * TH_POA_PAD; x += TH_WAVE) s_ring[x / (3 * TH_POA_PAD)][(x / TH_POA_PAD) % 3][x % TH_POA_PAD] = (RT)(WIDE ? TH_NEG : -32768);
    TH_WAVE_FENCE();
    // ONE exit for the row loop (round 5).  A `return` inside it -- empty band, out of cells: eight sites -- makes a loop with nine exits, which LLVM unifies into one by way of a
    // dispatch word that every row then sets, copies, tests and branches on (~20 scalar instructions per row of the common path's ~110, in the gfx950 assembly).  A failing row
    // instead records its code and moves the row counter to the end -- through a value the compiler cannot see through, or it would thread the jump back into an exit.
    int dp_fail = 0;
#if defined(TH_POA_DUP_ROW) && !TH_POA_REL
    int dup_cat = TH_POA_DUP_ROW; TH_OPAQUE_S(dup_cat);
#endif
#define TH_POA_DP_FAIL(code) { dp_fail = (code); int stop_ = n; TH_OPAQUE_S(stop_); i = stop_; continue; }
    for (int i = 0; i + 1 < n; ++i) {
        if ((i & 63) == 0) {
            flush_row = th_smin(i | 63, n - 2);          // the row after which the chunk registers go to the tables: the chunk's last, or the last row of all (ONE compare per row
                                                         // instead of the two tests and their boolean algebra: round 5)
            const int r = i + lane;
            if (r < n) {
                c_meta = g.r_meta[r]; c_rem = g.r_rem[r]; c_id = g.order[r];
                c_p0 = g.r_p[r * 4]; c_p1 = g.r_p[r * 4 + 1]; c_p2 = g.r_p[r * 4 + 2]; c_p3 = g.r_p[r * 4 + 3];
            }
            TH_VM_WAIT0();      // here, once per 64 rows -- not at the first use below, where it would drain every row's stores
        }
#if defined(TH_POA_DUP_ROW) && !TH_POA_REL
        const uint32_t sv_ncell = ncell, sv_nfull = nfull; const int sv_lim = lim, sv_td = tail_dirty, sv_b = L_b, sv_e = L_e, sv_l = L_l, sv_r = L_r, sv_c = L_c;
        int row_cat = -1, row_rep = 0;
        row_again:
        do {
#endif
#if defined(TH_POA_PROF)
        const unsigned long long row_t0 = TH_PROF_NOW();
#endif
#if TH_POA_REL
        if (L_rmax > TH_POA_REL_HI || L_rmax < -TH_POA_REL_HI) {
            // rebase: the previous row's maximum becomes 0 -- every ring slot moves by it (minus infinity stays where it is)
            const int delta = L_rmax;
            const uint32_t DP = TH_PK_DUP(delta), THRP = TH_PK_DUP(TH_POA_REL_THR);
            uint32_t *w = (uint32_t *)s_ring;
            for (int x = lane; x < TH_POA_RING * 3 * TH_POA_RS / 2; x += TH_WAVE) {
                const uint32_t v = w[x];
                w[x] = th_sel(th_pk_negmask_op(th_pk_sub(v, THRP)), TH_PK_NEG, th_pk_sub(v, DP));
            }
            TH_WAVE_FENCE();
            base += delta; L_rmax = 0;
        }
#endif
        const int sl = i & 63;
        const int meta = TH_READLANE(c_meta, sl), rm = TH_READLANE(c_rem, sl);
        if (!WIDE && LQ) {
            // ---- the common row (58 % at the headline shape): one predecessor, the previous row, clean in its ring slot; no far row
            // will read this one; its band fits the ring.  Straight-line scalar set-up, then the packed row with one predecessor.
            // (bit 25 of the row's meta word, th_poa_rows: one predecessor, the row before, no far reader; L_c is that very bit when the previous row sits clean in
            // its ring slot, else 0: the test is one s_and_b32 and a branch on its condition code -- as two booleans it was seven scalar instructions, round 5)
            if (meta & L_c) {
                const int xd = qlen - rm;
                // (scalar max / min: hipcc makes a vector max3 + a vector compare of these -- th_hw.h)
                const int b = th_smax(th_smax0(th_smin(L_l + 1, xd) - w), L_b), e = th_smin(qlen, th_smax(L_r + 1, xd) + w);
                if (b > e) TH_POA_DP_FAIL(-2);
                if (e - b < TH_POA_RW) {
                    const int B0 = b & ~1, span = (e | 1) - B0 + 1, units = th_d_units(span, true);
                    if ((int)(ncell + (uint32_t)units) > lim) TH_POA_DP_FAIL(-3);
                    const int my_off = (int)ncell + (TH_POA_D8 ? 0 : b - B0);      // (8-bit words: the offset of the even base column)
                    ncell += units;
                    int rmax = TH_NEG - 0x10000000, left = b, right = b;
                    th_poa_pk_row<QCAP, RT, true, false, 1>(s_qs, s_ring, g.dcell + (my_off - (TH_POA_D8 ? 0 : b - B0)), lane, b, e, B0, i & 3, (uint32_t)TH_READLANE((int)v_T, meta & 0xff), 1, (i - 1) & 3, 0, 0, 0, L_b & ~1, 0, 0, 0,
                                            E1, E2, O1P, O2P, E1P, E2P, tail_dirty, rmax, left, right);
                    TH_POA_PK_ROW_DONE(0, 0);
                    continue;
                }
            }
        }
        const int p0 = TH_READLANE(c_p0, sl);          // (after the common row, which does not need it)
        if (!WIDE && LQ) {
            // ---- the branch / merge rows of bubbles (35 %): up to four predecessors, each one of the three rows before this one, in
            // this 64-row chunk (or the previous row) and clean in its ring slot; no far row will read this one.  Same thing, unrolled.
            const int inn_ = (meta >> 8) & 0xff;
            if ((meta >> 26) & 1) {
                int mpl = 0x7fffffff, mpr = 0, mb = 0x7fffffff, wmax = 0;
                int qs0 = 0, qs1 = 0, qs2 = 0, qs3 = 0, qb0 = 0, qb1 = 0, qb2 = 0, qb3 = 0;
                // (round 5: ONE code path per predecessor, all on the scalar unit.  The previous row's band sits in its lane of the chunk registers like any other
                // row's -- its end-of-row write-lane put it there -- so the `pi == i - 1` branch onto the L_* scalars is gone; "cached" is "its band fits the ring"
                // for every row that reaches this path, so the widest predecessor band, one running s_max, replaces the per-predecessor flag that hipcc kept in a
                // vector register: ~10 scalar + 2 vector instructions per predecessor, 600 M predecessors per launch)
#define TH_POA_RING_PRED(PI, QS, QB) do { \
                    const int pi_ = (PI), x_ = pi_ & 63; \
                    const int pb_ = TH_READLANE(m_beg, x_), pl_ = TH_READLANE(m_left, x_), pr_ = TH_READLANE(m_right, x_); \
                    wmax = th_smax(wmax, TH_READLANE(m_end, x_) - pb_);      /* (where the predecessor sits: bit 26 of the meta word) */ \
                    mpl = th_smin(mpl, pl_ + 1); mpr = th_smax(mpr, pr_ + 1); mb = th_smin(mb, pb_); QS = pi_ & 3; QB = pb_ & ~1; \
                } while (0)
                TH_POA_RING_PRED(p0, qs0, qb0);
                if (inn_ > 1) {
                    TH_POA_RING_PRED(TH_READLANE(c_p1, sl), qs1, qb1);
                    if (inn_ > 2) {
                        TH_POA_RING_PRED(TH_READLANE(c_p2, sl), qs2, qb2);
                        if (inn_ > 3) TH_POA_RING_PRED(TH_READLANE(c_p3, sl), qs3, qb3);
                    }
                }
                if (wmax < TH_POA_RW) {
                    const int xd = qlen - rm;
                    const int b = th_smax(th_smax0(th_smin(mpl, xd) - w), mb), e = th_smin(qlen, th_smax(mpr, xd) + w);
                    if (b > e) TH_POA_DP_FAIL(-2);
                    if (e - b < TH_POA_RW) {
                        const int B0 = b & ~1, span = (e | 1) - B0 + 1, units = th_d_units(span, inn_ == 1);
                        if ((int)(ncell + (uint32_t)units) > lim) TH_POA_DP_FAIL(-3);
                        const int my_off = (int)ncell + ((TH_POA_D8 && inn_ == 1) ? 0 : b - B0);
                        uint16_t *dbase = g.dcell + ncell;
                        ncell += units;
                        int rmax = TH_NEG - 0x10000000, left = b, right = b;
                        const uint32_t T_ = (uint32_t)TH_READLANE((int)v_T, meta & 0xff);
                        // (one or two predecessors as compile-time constants: the body's per-predecessor tests disappear)
                        if (inn_ == 1) th_poa_pk_row<QCAP, RT, false, false, 1>(s_qs, s_ring, dbase, lane, b, e, B0, i & 3, T_, 1, qs0, 0, 0, 0, qb0, 0, 0, 0, E1, E2, O1P, O2P, E1P, E2P, tail_dirty, rmax, left, right);
                        else if (inn_ == 2) th_poa_pk_row<QCAP, RT, false, false, 0>(s_qs, s_ring, dbase, lane, b, e, B0, i & 3, T_, 2, qs0, qs1, 0, 0, qb0, qb1, 0, 0, E1, E2, O1P, O2P, E1P, E2P, tail_dirty, rmax, left, right);
                        else th_poa_pk_row<QCAP, RT, false, false, 0>(s_qs, s_ring, dbase, lane, b, e, B0, i & 3, T_, inn_, qs0, qs1, qs2, qs3, qb0, qb1, qb2, qb3, E1, E2, O1P, O2P, E1P, E2P, tail_dirty, rmax, left, right);
                        TH_POA_PK_ROW_DONE(inn_ == 1 ? 1 : inn_ == 2 ? 6 : 7, 0);
                        continue;
                    }
                }
            }
            // ---- rows one step further out (~5 %): as above, but ONE predecessor may lie outside the ring (an edge over more than three
            // rows, or a predecessor too wide to be cached) -- it is read from its full cells in HBM -- and / or some far row will read
            // this one, which then also stores its cells in full.  Everything still inside this 64-row chunk; the rest goes on to the
            // general path below (one column per lane in 32 bits).
            if ((unsigned)(inn_ - 1) < 4u) {          // one to four predecessors (row 0, the source, has none): one compare
                int mpl = 0x7fffffff, mpr = 0, mb = 0x7fffffff, bad = 0;
                int qs0 = 0, qs1 = 0, qs2 = 0, qs3 = 0, qb0 = 0, qb1 = 0, qb2 = 0, qb3 = 0;
                int far_o = -1, far_pb = 0, far_pe = -1, far_po = 0;
#define TH_POA_FAR_PRED(O, PI, QS, QB) do { \
                    const int pi_ = (PI); int pb_ = 0, pl_ = 0, pr_ = 0; \
                    if (pi_ < i && i - pi_ < TH_WAVE) {      /* (its band is still in lane pi_ & 63 of the chunk registers: row pi_ + 64 has not been written yet) */ \
                        const int x_ = pi_ & 63; \
                        pb_ = TH_READLANE(m_beg, x_); pl_ = TH_READLANE(m_left, x_); pr_ = TH_READLANE(m_right, x_); \
                        const int pe_ = TH_READLANE(m_end, x_); \
                        if (i - pi_ >= TH_POA_RING || pe_ - pb_ >= TH_POA_RW) { bad |= far_o >= 0; far_o = (O); far_pb = pb_; far_pe = pe_; far_po = TH_READLANE(m_foff, x_); } \
                    } else bad = 1; \
                    mpl = min(mpl, pl_ + 1); mpr = max(mpr, pr_ + 1); mb = min(mb, pb_); QS = pi_ & 3; QB = pb_ & ~1; \
                } while (0)
                TH_POA_FAR_PRED(0, p0, qs0, qb0);
                if (inn_ > 1) {
                    TH_POA_FAR_PRED(1, TH_READLANE(c_p1, sl), qs1, qb1);
                    if (inn_ > 2) {
                        TH_POA_FAR_PRED(2, TH_READLANE(c_p2, sl), qs2, qb2);
                        if (inn_ > 3) TH_POA_FAR_PRED(3, TH_READLANE(c_p3, sl), qs3, qb3);
                    }
                }
                if (!bad) {
                    const int xd = qlen - rm;
                    const int b = max(max(0, min(mpl, xd) - w), mb), e = min(qlen, max(mpr, xd) + w);
                    if (b > e) TH_POA_DP_FAIL(-2);
                    if (e - b < TH_POA_RW) {
                        const int B0 = b & ~1, span = (e | 1) - B0 + 1, units = th_d_units(span, inn_ == 1);
                        const int full = (meta >> 24) & 1;
                        if (full) { nfull += span; lim -= 8 * span; }
                        if ((int)(ncell + (uint32_t)units) > lim) TH_POA_DP_FAIL(-3);
                        const int my_off = (int)ncell + ((TH_POA_D8 && inn_ == 1) ? 0 : b - B0), my_foff = full ? -(int)nfull + (b - B0) : 0;
                        uint16_t *dbase = g.dcell + ncell;
                        ncell += units;
                        int rmax = TH_NEG - 0x10000000, left = b, right = b;
                        th_poa_pk_row<QCAP, RT, false, true, 2>(s_qs, s_ring, dbase, lane, b, e, B0, i & 3, (uint32_t)TH_READLANE((int)v_T, meta & 0xff), inn_, qs0, qs1, qs2, qs3, qb0, qb1, qb2, qb3,
                                                E1, E2, O1P, O2P, E1P, E2P, tail_dirty, rmax, left, right, far_o, far_pb, far_pe, g.cell + far_po, full ? g.cell + my_foff : nullptr, neg_thr, base);
                        TH_POA_PK_ROW_DONE(9, my_foff);
                        continue;
                    }
                }
            }
        }
        const int nb = meta & 0xff, inn = (meta >> 8) & 0xff;
        const bool needfull = (meta >> 24) & 1;
        const bool fast = inn == 1 && p0 == i - 1 && L_c;        // one predecessor, the previous row, cached in LDS
        int p1 = -1, p2 = -1, p3 = -1, vid = 0;
        // predecessors as the packed row reads them: ring slot and even band base of each; ring_all: every predecessor is one of
        // the three rows before this one and sits (clean) in its ring slot
        int q_slot0 = (i - 1) & 3, q_slot1 = 0, q_slot2 = 0, q_slot3 = 0, q_pb0 = L_b & ~1, q_pb1 = 0, q_pb2 = 0, q_pb3 = 0;
        bool ring_all = fast && L_c != 0;
        // ---- band: pull the predecessors' band start and row-maximum columns ----
        int b = 0, e;
        if (fast) {
            b = max(max(0, min(L_l + 1, qlen - rm) - w), L_b);
            e = min(qlen, max(L_r + 1, qlen - rm) + w);
        } else {
            p1 = TH_READLANE(c_p1, sl); p2 = TH_READLANE(c_p2, sl); p3 = TH_READLANE(c_p3, sl); vid = TH_READLANE(c_id, sl);
            int mpl = 0x7fffffff, mpr = 0, mb = 0x7fffffff;
            ring_all = i > 0 && inn >= 1 && inn <= 4;
            for (int o = 0; o < inn; ++o) {
                const int pi = o == 0 ? p0 : o == 1 ? p1 : o == 2 ? p2 : o == 3 ? p3 : g.pos[th_poa_in(g, vid, o)];
                const int s = pi & 3;
                int pb, pl, pr, pe;
                if (pi == i - 1) { pb = L_b; pl = L_l; pr = L_r; pe = L_c != 0 ? L_b : L_b + TH_POA_RW; }
                else if (pi >= (i & ~63)) { pb = TH_READLANE(m_beg, pi & 63); pl = TH_READLANE(m_left, pi & 63); pr = TH_READLANE(m_right, pi & 63); pe = TH_READLANE(m_end, pi & 63); }
                else { pb = TH_UNI(g.r_beg[pi]); pl = TH_UNI(g.r_mpl[pi]); pr = TH_UNI(g.r_mpr[pi]); pe = pb + TH_POA_RW; }     // a row of the previous chunk: not worth a fourth load
                mpl = min(mpl, pl + 1); mpr = max(mpr, pr + 1); mb = min(mb, pb);
                ring_all = ring_all && i - pi >= 1 && i - pi < TH_POA_RING && pe - pb < TH_POA_RW;
                if (o == 0) { q_slot0 = s; q_pb0 = pb & ~1; } else if (o == 1) { q_slot1 = s; q_pb1 = pb & ~1; } else if (o == 2) { q_slot2 = s; q_pb2 = pb & ~1; } else if (o == 3) { q_slot3 = s; q_pb3 = pb & ~1; }
            }
            if (i > 0) { b = max(0, min(mpl, qlen - rm) - w); b = max(b, mb); }
            e = min(qlen, max(mpr, qlen - rm) + w);
        }
        b = TH_UNI(b); e = TH_UNI(e);
        if (b > e) TH_POA_DP_FAIL(-2);
        // a row owns the cells of columns (b & ~1) .. (e | 1): packed rows store column pairs
        const int B0 = b & ~1, span = (e | 1) - B0 + 1;
        const int myslot = i & 3, lslot = (i - 1) & 3;
        const bool cache_me = cache16_ok && e - b < TH_POA_RW;
        const bool store_full = needfull || !cache_me || inn > TH_D16_WIDE_DEG;     // otherwise only the 16-bit direction word leaves the CU
        // direction words (2 bytes per cell, every row) grow upwards from the start of the cell region; full cells (16 bytes, only the
        // rows that store them: g.cell points at the END of the region and their offsets are negative) grow downwards from its end.
        // A wave's scratch shrinks from ~18 to ~3-6 bytes per band cell, whatever the share of rows stored in full.
        const bool d8 = TH_POA_D8 && inn == 1;                   // one predecessor: 8-bit direction words (every reader goes by the in-degree)
        const int units = th_d_units(span, d8);
        if (store_full) { nfull += span; lim -= 8 * span; }
        if ((int)(ncell + (uint32_t)units) > lim) TH_POA_DP_FAIL(-3);
        const int my_off = (int)ncell + (d8 ? 0 : b - B0);       // 16-bit rows: offset of column b; 8-bit rows: of the even base column
        const int my_foff = store_full ? -(int)nfull + (b - B0) : 0;
        int4 *row = g.cell + my_foff; uint16_t *dbase = g.dcell + ncell; uint16_t *drow = dbase + (b - B0); uint8_t *drow8 = (uint8_t *)dbase + (b - B0);
        ncell += units;
        int rmax = TH_NEG - 0x10000000, left = b, right = b;
        const bool pk = !WIDE && LQ && ring_all && cache_me && !needfull;    // rows some far row will read take the 32-bit path (they store whole cells)
        const int npk = fast ? 1 : inn;
        if (pk) {
            th_poa_pk_row<QCAP, RT, false, false, 2>(s_qs, s_ring, dbase, lane, b, e, B0, myslot, (uint32_t)TH_READLANE((int)v_T, nb), npk, q_slot0, q_slot1, q_slot2, q_slot3, q_pb0, q_pb1, q_pb2, q_pb3,
                                    E1, E2, O1P, O2P, E1P, E2P, tail_dirty, rmax, left, right);
        } else {
        if (!WIDE && cache_me) {      // leave the slot clean for a packed successor: minus infinity wherever this row stores nothing
            uint32_t *wH = (uint32_t *)s_ring[myslot][0], *wE1 = (uint32_t *)s_ring[myslot][1], *wE2 = (uint32_t *)s_ring[myslot][2];
            for (int x = TH_POA_PAD / 2 + lane; x < TH_POA_RS / 2; x += TH_WAVE) { wH[x] = TH_PK_NEG; wE1[x] = TH_PK_NEG; wE2[x] = TH_PK_NEG; }
            TH_WAVE_FENCE();
#if TH_POA_TRIPS == 2
            tail_dirty |= 1 << myslot;            // the cells written below may reach beyond the slot's first 128 columns
#else
            TH_TD_SET(tail_dirty, myslot, TH_POA_TRIPS - 1);
#endif
        }
        int carry1 = TH_NEG + b * E1, carry2 = TH_NEG + b * E2;
        for (int c0 = b; c0 <= e; c0 += TH_WAVE) {
            const int j = c0 + lane;
            const bool in = j <= e;
            int m = TH_NEG, e1v = TH_NEG, e2v = TH_NEG, pm = 0, pe1 = 0, pe2 = 0;
            int sc = 0;
            if (in && j >= 1) { const int qb = LQ ? (int)((s_qs[j >> 1] >> ((j & 1) * 4)) & 15) : min((int)q[j - 1], 4); sc = (qb == 4 || nb == 4) ? 0 : (qb == nb ? M : -X); }
            if (fast) {
                if (in) {
                    if (j >= 1) { m = ((j - 1 >= L_b && j - 1 <= L_e) ? TH_RING_LD(s_ring[lslot][0][TH_RIDX(j - 1, L_b)]) : TH_NEG) + sc; }
                    if (j >= L_b && j <= L_e) { e1v = TH_RING_LD(s_ring[lslot][1][TH_RIDX(j, L_b)]); e2v = TH_RING_LD(s_ring[lslot][2][TH_RIDX(j, L_b)]); }
                }
            } else if (i > 0) {
                for (int o = 0; o < inn; ++o) {
                    const int pi = o == 0 ? p0 : o == 1 ? p1 : o == 2 ? p2 : o == 3 ? p3 : g.pos[th_poa_in(g, vid, o)];
                    const int s = pi & 3;
                    int hp = TH_NEG, c1 = TH_NEG, c2 = TH_NEG;
                    // the predecessor row's band and cell offset: chunk registers or row tables
                    int pb, pe, po;       // po: the predecessor's full cells (only read when it is not in the ring: such rows store them)
                    if (pi >= (i & ~63)) { pb = TH_READLANE(m_beg, pi & 63); pe = TH_READLANE(m_end, pi & 63); po = TH_READLANE(m_foff, pi & 63); }
                    else { pb = g.r_beg[pi]; pe = g.r_end[pi]; po = g.r_foff[pi]; }
                    // the three rows before this one are still in the LDS ring if their band fitted it (every cached slot is clean)
                    if (i - pi < TH_POA_RING && cache16_ok && pe - pb < TH_POA_RW) {           // LDS ring: no HBM round trip
                        if (in) {
                            if (j >= 1 && j - 1 >= pb && j - 1 <= pe) hp = TH_RING_LD(s_ring[s][0][TH_RIDX(j - 1, pb)]);
                            if (j >= pb && j <= pe) { c1 = TH_RING_LD(s_ring[s][1][TH_RIDX(j, pb)]); c2 = TH_RING_LD(s_ring[s][2][TH_RIDX(j, pb)]); }
                        }
                    } else {
                        const int4 *prow = g.cell + po;
                        if (in) {
                            if (j >= 1 && j - 1 >= pb && j - 1 <= pe) hp = prow[j - 1 - pb].x;
                            if (j >= pb && j <= pe) { const int4 pc = prow[j - pb]; c1 = pc.y; c2 = pc.z; }
                        }
                    }
                    if (in) {
                        if (j >= 1) { const int c = hp + sc; if (o == 0 || c > m) { m = c; pm = o; } }
                        if (o == 0 || c1 > e1v) { e1v = c1; pe1 = o; }
                        if (o == 0 || c2 > e2v) { e2v = c2; pe2 = o; }
                    }
                }
            } else if (in) {
                m = j == 0 ? 0 : -min(O1 + j * E1, O2 + j * E2);
            }
            int hnf = m, src = 0;
            if (e1v > hnf) { hnf = e1v; src = 1; }
            if (e2v > hnf) { hnf = e2v; src = 2; }
            const int a1 = in ? hnf - O1 + j * E1 : TH_NEG * 2, a2 = in ? hnf - O2 + j * E2 : TH_NEG * 2;
            const int i1 = th_wave_incl_max(a1), i2 = th_wave_incl_max(a2);
            const int x1 = max(th_shr1(i1, carry1), carry1), x2 = max(th_shr1(i2, carry2), carry2);
            carry1 = max(carry1, TH_READLANE(i1, 63)); carry2 = max(carry2, TH_READLANE(i2, 63));
            int f1 = x1 - j * E1, f2 = x2 - j * E2;
            if (i == 0) { f1 = TH_NEG; f2 = TH_NEG; }
            int h = hnf;
            if (f1 > h) { h = f1; src = 3; }
            if (f2 > h) { h = f2; src = 4; }
            if (in) {
                uint32_t d = 0;
                if (i > 0) {
                    d = (uint32_t)src | (uint32_t)pm << 3 | (uint32_t)pe1 << 11 | (uint32_t)pe2 << 19;
                    if (e1v > h - O1) d |= TH_D_XE1;
                    if (e2v > h - O2) d |= TH_D_XE2;
                    if (f1 > h - O1) d |= TH_D_XF1;
                    if (f2 > h - O2) d |= TH_D_XF2;
                }
                const int n1 = max(h - O1, e1v) - E1, n2 = max(h - O2, e2v) - E2;
                if (inn <= TH_D16_WIDE_DEG) {
                    const uint32_t w16 = th_d16_pack(src, d >> 27 & 1u, d >> 28 & 1u, d >> 29 & 1u, d >> 30 & 1u, (uint32_t)pm, (uint32_t)pe1, (uint32_t)pe2);
                    if (d8) drow8[j - b] = (uint8_t)th_d8_from16(w16); else drow[j - b] = (uint16_t)w16;
                }
                if (store_full) { int4 cv; cv.x = h; cv.y = n1; cv.z = n2; cv.w = (int)d; row[j - b] = cv; }
                if (cache_me) { s_ring[myslot][0][TH_RIDX(j, b)] = TH_RING_ST(h); s_ring[myslot][1][TH_RIDX(j, b)] = TH_RING_ST(n1); s_ring[myslot][2][TH_RIDX(j, b)] = TH_RING_ST(n2); }
            }
            const int hm = th_wave_max(in ? h : TH_NEG * 2);
            const unsigned long long mm = __ballot(in && h == hm);
            if (hm > rmax) { rmax = hm; left = c0 + th_ctz64(mm); right = c0 + th_msb64(mm); }
            else if (hm == rmax) right = c0 + th_msb64(mm);
        }
        }
        L_b = b; L_e = e; L_l = TH_UNI(left); L_r = TH_UNI(right); L_c = cache_me ? TH_LC_ON : 0;
        TH_POA_SET_RMAX(pk ? rmax : rmax - base);      // (the 32-bit path works with absolute scores)
#if defined(TH_POA_PROF)
        { const unsigned long long dt = TH_POA_PROF >= 2 ? TH_PROF_NOW() - row_t0 : (unsigned long long)(e - b + 1);
          const int cat = pk ? 1 : inn != 1 ? 2 : p0 != i - 1 ? 3 : needfull ? 4 : 5;     // (0: common packed row) / other packed / several predecessors / one far predecessor / stores full cells / other
          TH_PROF_ROW(cat, dt);
          if (!pk) {
              int why = 5;
              if (WIDE || !LQ) why = 6; else if (inn == 0) why = 0; else if (inn > 4) why = 1;
              else {
                  int nfar = 0, old = 0;
                  for (int o = 0; o < inn; ++o) {
                      const int pi = o == 0 ? p0 : o == 1 ? p1 : o == 2 ? p2 : p3;
                      if (fast) break;
                      if (i - pi >= TH_WAVE) old = 1;
                      else if (i - pi >= TH_POA_RING || TH_READLANE(m_end, pi & 63) - TH_READLANE(m_beg, pi & 63) >= TH_POA_RW) ++nfar;
                  }
                  why = old ? 2 : nfar >= 2 ? 3 : e - b >= TH_POA_RW ? 4 : 5;
              }
              if (TH_LANE() == 0) atomicAdd(&th_poa_prof_why[why], 1ull);
          } }
#endif
        TH_ROW_CAT(pk ? 1 : inn != 1 ? 2 : p0 != i - 1 ? 3 : needfull ? 4 : 5)
        th_writelane6(m_beg, m_end, m_left, m_right, m_off, m_foff, b, e, L_l, L_r, my_off, my_foff, sl);
        if (i == flush_row) {          // chunk complete (or last row): its rows' metadata to the tables, coalesced
            const int r = (i & ~63) + lane;
            if (r <= i) { g.r_beg[r] = m_beg; g.r_end[r] = m_end; g.r_off[r] = m_off; g.r_foff[r] = m_foff; g.r_mpl[r] = m_left; g.r_mpr[r] = m_right; }
        }
        TH_WAVE_FENCE();
#if defined(TH_POA_DUP_ROW) && !TH_POA_REL
        } while (0);
        { int again_ = (row_cat == dup_cat) & (row_rep == 0); TH_OPAQUE_S(again_);
          if (again_) { ncell = sv_ncell; nfull = sv_nfull; lim = sv_lim; tail_dirty = sv_td; L_b = sv_b; L_e = sv_e; L_l = sv_l; L_r = sv_r; L_c = sv_c; row_rep = 1; goto row_again; } }
#endif
    }
    if (dp_fail) return dp_fail;
    // global end: best predecessor row of the sink (row n-1) at column qlen, first in edge order
    int bi = -1, bs = 0;
    {
        const int sinn = g.inn[1];
        for (int o = 0; o < sinn; ++o) {
            const int pi = g.pos[th_poa_in(g, 1, o)];
            const int pb = g.r_beg[pi], pe = g.r_end[pi];
            const int c = (qlen >= pb && qlen <= pe) ? g.cell[g.r_foff[pi] + (qlen - pb)].x : TH_NEG;
            if (o == 0 || c > bs) { bs = c; bi = pi; }
        }
    }
    if (bs <= TH_NEG / 2) return -4;
    return bi;
}

// direction word of cell (row, column offset `o` from the row's band start) in the register layout of the TH_D_* macros
__device__ __forceinline__ uint32_t th_poa_dir(const ThPoaG &g, int row, int o)
{
    const int deg = (g.r_meta[row] >> 8) & 0xff;
    if (deg > TH_D16_WIDE_DEG) return (uint32_t)g.cell[g.r_foff[row] + o].w;
    if (TH_POA_D8 && deg == 1) return th_d8_expand(((const uint8_t *)(g.dcell + g.r_off[row]))[o + (g.r_beg[row] & 1)]);
    return th_d16_expand(g.dcell[g.r_off[row] + o]);
}

// ---- traceback: wave-uniform walk over tiles of 16 rows x 4 diagonals; the tile is DECODED by all lanes, WALKED on the scalar unit, its operations WRITTEN by all lanes ----
// Round 6.  Round 5's walk (below, -DTH_POA_TB=5) took one scalar trip per event -- tile load, run of plain diagonal steps, single step -- and stored every step's
// operation from lane 0 under an execution mask of its own: 45 scalar + 22 vector instructions per event, 500 M events per launch of 100 000 headline reads
// (profiles/r05_poa_phases.json).  Here a tile is lane l = 4 k + c <-> cell (row ti - k, column tj - k + TH_TB_C0 - c): the walk enters at lane TH_TB_C0 (column tj) and EVERY move goes to a
// higher lane -- a diagonal step to the row d above to l + 3 d + 1, a vertical one (deletion) to l + 3 d, a horizontal one (insertion) to l + 1 -- so
//   * what the walk needs of a cell is decoded ONCE per tile by the cell's own lane: lane masks of the plain diagonal steps, of the diagonal steps to any predecessor,
//     of the four "the gap goes on" bits, and the successor LANE of a diagonal step in a register (one v_readlane per step to another predecessor);
//   * the walk keeps a lane position and three visited masks (any / deletion / insertion) in scalar registers and stores nothing;
//   * when it leaves the tile the visited lanes write their operations together: a lane's rank among the visited ones is its place in the path.
// A gap state that leaves the tile is carried into the next one and tested against the entry cell's word there (round 5 fetched that one word from memory).
#ifndef TH_POA_TB
#define TH_POA_TB 6
#endif
#ifndef TH_TB_C0
#define TH_TB_C0 3        // the tile diagonal (c) the walk enters on: c + 1 .. 3 are reached by insertions, c - 1 .. 0 by deletions and by diagonal steps that skip rows.  3: no room on the
                          // insertion side -- a graph row in seven is a bubble's other branch that the path steps over, so the walk drifts towards c = 0 at ~0.2 per row: 91.3 -> 84.1 M tiles per launch, -1.5 ms (2: one diagonal either way)
#endif
#if TH_POA_TB >= 6
__device__ int th_poa_traceback(ThPoaG &g, int bi, int qlen)
{
    const int lane = TH_LANE();
    const int lk = lane >> 2, lc = lane & 3;
    int no = 0, ci = TH_UNI(bi), cj = TH_UNI(qlen), st = 0;
    constexpr unsigned long long C4 = 0x1111111111111111ull;
    while (ci > 0) {
        // the walk is wave-uniform; saying so at every tile keeps its state in scalar registers and its branches scalar
        ci = TH_UNI(ci); cj = TH_UNI(cj); st = TH_UNI(st); no = TH_UNI(no);
        const int ti = ci, tj = cj;
        TH_PROF_CNT(4, 1);
        // ---- the tile: every lane fetches and decodes its cell
        const int row = ti - lk, col = tj - lk + TH_TB_C0 - lc;
        // t_w: the cell's direction word as the DP stored it -- the raw 16-bit word (th_k_poa.hip), or bit 31 | the 32-bit word of a row with more than four predecessors;
        // 0 outside the row's band, as in round 5's walk.  The lanes decode the raw form (two instructions per mask); expanding it to the TH_D_* layout was 20 vector
        // instructions per tile for fields only the gap steps read -- those expand their one word on the scalar unit (TH_TB_D)
        int t_w = 0, t_p0 = -1, t_p1 = -1, t_p2 = -1, t_p3 = -1;
        if (row >= 1) {
            // everything indexed by the row is requested at once; the direction word is the only dependent load
            const int rb = g.r_beg[row], re = g.r_end[row], meta = g.r_meta[row], roff = g.r_off[row], rfoff = g.r_foff[row];
            const int4 pp = ((const int4 *)g.r_p)[row];
            t_p0 = pp.x; t_p1 = pp.y; t_p2 = pp.z; t_p3 = pp.w;
            TH_KEEP_V(rb); TH_KEEP_V(re); TH_KEEP_V(meta); TH_KEEP_V(roff); TH_KEEP_V(rfoff);      // (the four row words as ONE 16-byte table entry: measured, no difference -- 343.4 against 343.6 ms)
            if (col >= rb && col <= re) {
                const int deg_ = (meta >> 8) & 0xff;
                if (deg_ > TH_D16_WIDE_DEG) t_w = g.cell[rfoff + (col - rb)].w | (int)0x80000000;
                else if (TH_POA_D8 && deg_ == 1) { const uint32_t b_ = ((const uint8_t *)(g.dcell + roff))[col - (rb & ~1)]; t_w = (int)((b_ & 0x0fu) | (b_ & 0xf0u) << 4); }
                else t_w = (int)g.dcell[roff + (col - rb)];
            }
        }
        const uint32_t uw = (uint32_t)t_w;
        int wantm = (int)((uw >> TH_D16_PM_SHIFT) & 3u);
        bool mstep = row >= 1 && (uw & 0x0303u) == 0;                                             // no source flag set: the diagonal (a cell outside its band reads as word 0: a step to its first predecessor)
        bool xe1 = (uw & 0x400u) != 0, xe2 = (uw & 0x004u) != 0, xf1 = (uw & 0x800u) != 0, xf2 = (uw & 0x008u) != 0;
        if (__any(t_w < 0)) {                                                                     // rows with more than four predecessors in the tile (rare): the 32-bit layout
            if (t_w < 0) {
                const uint32_t d = uw & 0x7fffffffu;
                wantm = (int)TH_D_PM(d) & 3; mstep = TH_D_SRC(d) == 0 && TH_D_PM(d) < TH_POA_INL;
                xe1 = (d & TH_D_XE1) != 0; xe2 = (d & TH_D_XE2) != 0; xf1 = (d & TH_D_XF1) != 0; xf2 = (d & TH_D_XF2) != 0;
            }
        }
        const int t_pr_lo = (wantm & 1) ? t_p1 : t_p0, t_pr_hi = (wantm & 1) ? t_p3 : t_p2;
        const int t_pr = (wantm & 2) ? t_pr_hi : t_pr_lo;                                         // the row a diagonal step goes to (selects: as a chain of tests hipcc makes execution-mask blocks of it)
        const int dd = row - t_pr;
        // successor lane of the diagonal step, 64 if it leaves the tile (or reaches the source row: the walk ends there)
        const int t_nl = (mstep && t_pr >= 1 && lk + dd <= 15 && lc - dd + 1 >= 0) ? lane + 3 * dd + 1 : 64;
        const unsigned long long m_any = TH_BALLOT(mstep);
        const unsigned long long m_run = TH_BALLOT(mstep && dd == 1 && row >= 2);               // plain steps: match state, the row above
        const unsigned long long x_e1 = TH_BALLOT(xe1), x_e2 = TH_BALLOT(xe2), x_f1 = TH_BALLOT(xf1), x_f2 = TH_BALLOT(xf2);
#define TH_TB_D(p_) ({ const int w_ = TH_READLANE(t_w, (p_)); w_ < 0 ? (uint32_t)w_ & 0x7fffffffu : th_d16_expand((uint32_t)w_); })
#define TH_TB_XM(st_) ((st_) == 1 ? x_e1 : (st_) == 2 ? x_e2 : (st_) == 3 ? x_f1 : x_f2)
#define TH_TB_ROW(p_) (ti - ((p_) >> 2))
#define TH_TB_COL(p_) (tj - ((p_) >> 2) + TH_TB_C0 - ((p_) & 3))
        // ---- the walk inside the tile: scalar unit only (plus one lane read per irregular step).  ONE exit at the bottom (a loop with several `break`s gets a
        // dispatch word that every trip sets and tests -- DESIGN.md, "Control flow the scalar unit does not pay for"): a move that leaves the tile puts the lane position
        // beyond it -- 64 .. 67 when the position itself still says where the walk is (a run of plain steps off the tile's last row), 128 when xr / xc do
        int p = TH_TB_C0, xr = 0, xc = 0;
        unsigned long long vis = 0, vis_d = 0, vis_i = 0;
        if (st) { if (!((TH_TB_XM(st) >> TH_TB_C0) & 1)) st = 0; }       // a gap state carried in: does the entry cell extend it?
        do {
            const unsigned long long bit = 1ull << p;
            vis |= bit;                                                             // (every trip leaves lane p by exactly one operation)
            if (st == 0) {
                // a run of plain diagonal steps down this tile column is taken in one go (most of the path at 10 % error)
                const unsigned long long colb = (m_run >> p) & C4;                 // bit 4 x: step x of the run is plain (zeros beyond the tile's last row)
                const int run = th_ctz64(~colb & C4) >> 2;
                if (run > 0) {
                    vis |= (C4 >> (64 - 4 * run)) << p;
                    p += 4 * run;
                    TH_PROF_CNT(5, 1); TH_PROF_CNT(6, run);
                } else if (m_any & bit) {                                           // a diagonal step to another predecessor (the branch / merge rows of bubbles)
                    TH_PROF_CNT(7, 1);
                    const int nl = TH_READLANE(t_nl, p);
                    if (nl >= 64) { xr = TH_READLANE(t_pr, p); xc = TH_TB_COL(p) - 1; p = 128; } else p = nl;
                } else {
                    TH_PROF_CNT(7, 1);
                    const uint32_t d = TH_TB_D(p);
                    st = (int)TH_D_SRC(d);
                    if (st == 0) {                                                  // a diagonal step along an overflow edge (in-degree > 4)
                        xr = TH_UNI(g.pos[th_poa_in(g, g.order[TH_TB_ROW(p)], (int)TH_D_PM(d))]); xc = TH_TB_COL(p) - 1; p = 128;
                    } else vis &= ~bit;                                             // a gap starts here: nothing leaves the cell in this trip, the next one takes the gap step
                }
            } else if (st <= 2) {                                                   // deletion: to a predecessor row, same column
                TH_PROF_CNT(7, 1);
                vis_d |= bit;
                const uint32_t d = TH_TB_D(p);
                const int want = st == 1 ? (int)TH_D_PE1(d) : (int)TH_D_PE2(d);
                const int r_ = TH_TB_ROW(p);
                int pi;
                if (want < TH_POA_INL) { const int a0 = TH_READLANE(t_p0, p), a1 = TH_READLANE(t_p1, p), a2 = TH_READLANE(t_p2, p), a3 = TH_READLANE(t_p3, p); pi = want == 0 ? a0 : want == 1 ? a1 : want == 2 ? a2 : a3; }
                else pi = TH_UNI(g.pos[th_poa_in(g, g.order[r_], want)]);
                const int d_ = r_ - pi;
                if (pi >= 1 && (p >> 2) + d_ <= 15 && (p & 3) - d_ >= 0) { p += 3 * d_; if (!((TH_TB_XM(st) >> p) & 1)) st = 0; }
                else { xr = pi; xc = TH_TB_COL(p); p = 128; }
            } else {                                                                // insertion: same row, the column before
                TH_PROF_CNT(7, 1);
                vis_i |= bit;
                if ((p & 3) == 3) { xr = TH_TB_ROW(p); xc = TH_TB_COL(p) - 1; p = 128; }
                else { p += 1; if (!((TH_TB_XM(st) >> p) & 1)) st = 0; }
            }
        } while (p < 64);
        if (p < 128) { ci = TH_TB_ROW(p); cj = TH_TB_COL(p); } else { ci = xr; cj = xc; }
#undef TH_TB_XM
#undef TH_TB_D
#undef TH_TB_ROW
#undef TH_TB_COL
        // ---- the visited cells write their operations: rank among the visited lanes = place in the path
        if (TH_LANE_OF_MASK(vis, lane)) {
            const int k_ = no + th_mbcnt(vis);
            const bool is_d = TH_LANE_OF_MASK(vis_d, lane), is_i = TH_LANE_OF_MASK(vis_i, lane);
            int2 w;
            w.x = is_i ? 1 : ((row << 2) | (is_d ? 2 : 0));
            w.y = is_d ? -1 : col - 1;
            ((int2 *)g.ops)[k_] = w;
        }
        no += (int)__popcll(vis);
    }
    no = TH_UNI(no); cj = TH_UNI(cj);
    for (int c0 = cj; c0 > 0; c0 -= TH_WAVE) {      // the source row: what is left of the query is inserted
        const int j = c0 - lane;
        if (j > 0) { g.ops[2 * (no + lane)] = 1; g.ops[2 * (no + lane) + 1] = j - 1; }
        no += min(c0, TH_WAVE);
    }
    TH_WAVE_FENCE();
    return no;
}
#else
// ---- traceback: wave-uniform walk; direction words come from tiles of 16 rows x 4 diagonals ----
__device__ int th_poa_traceback(ThPoaG &g, int bi, int qlen)
{
    const int lane = TH_LANE();
    int no = 0, ci = TH_UNI(bi), cj = TH_UNI(qlen), st = 0;
    int ti = -1000000, tj = 0;              // tile anchor; lane l = 4*k + o holds cell (ti - k, tj - k + o - 1)
    int t_d = 0, t_p0 = -1, t_p1 = -1, t_p2 = -1, t_p3 = -1, t_ok = 0;
    unsigned long long t_run = 0;           // tile cells that are plain diagonal steps: match state, predecessor = the row above
    unsigned long long t_any = 0; int t_pr = 0;     // ... diagonal steps in match state to ANY predecessor row (t_pr: that row): one short trip each
    unsigned long long t_okm = 0;          // tile cells that lie inside their row's band (t_ok as a lane mask)
    while (ci > 0) {
        // the walk is wave-uniform; saying so at every trip keeps its state in scalar registers and its branches scalar (the compiler
        // otherwise treats the loop-carried state as divergent and compiles every step under exec masks)
        ci = TH_UNI(ci); cj = TH_UNI(cj); st = TH_UNI(st); no = TH_UNI(no); ti = TH_UNI(ti); tj = TH_UNI(tj);
        int k = ti - ci, o = cj - (tj - k) + 1;
        // (round 5: 0 <= k <= 15 and 0 <= o <= 3 as ONE test -- arithmetic shifts leave 0 only inside the ranges -- and the cell's "inside its row's band" flag from a lane
        // mask taken once per tile instead of a lane read per step: ~8 scalar and one vector instruction off each of the ~500 M steps per launch)
        const int out0 = (k >> 4) | (o >> 2);
        if (out0 != 0 || !((t_okm >> ((k * 4 + o) & 63)) & 1)) {
            ti = ci; tj = cj;
            TH_PROF_CNT(4, 1);
            const int row = ti - (lane >> 2), col = tj - (lane >> 2) + (lane & 3) - 1;
            t_ok = 0; t_d = 0;
            if (row >= 1) {
                // everything indexed by the row is requested at once; the direction word is the only dependent load (the walk waits
                // for the whole chain at every tile: it was four round trips long)
                const int rb = g.r_beg[row], re = g.r_end[row], meta = g.r_meta[row], roff = g.r_off[row], rfoff = g.r_foff[row];
                t_p0 = g.r_p[row * 4]; t_p1 = g.r_p[row * 4 + 1]; t_p2 = g.r_p[row * 4 + 2]; t_p3 = g.r_p[row * 4 + 3];
                TH_KEEP_V(rb); TH_KEEP_V(re); TH_KEEP_V(meta); TH_KEEP_V(roff); TH_KEEP_V(rfoff);     // (all five in flight together: the compiler would sink each load behind the test that needs it)
                if (col >= rb && col <= re) {
                    const int deg_ = (meta >> 8) & 0xff;
                    t_d = deg_ > TH_D16_WIDE_DEG ? g.cell[rfoff + (col - rb)].w
                        : (TH_POA_D8 && deg_ == 1) ? (int)th_d8_expand(((const uint8_t *)(g.dcell + roff))[col - (rb & ~1)]) : (int)th_d16_expand(g.dcell[roff + (col - rb)]);
                    t_ok = 1;
                }
            }
            {
                const int want = (int)TH_D_PM((uint32_t)t_d);
                const int pr = want == 0 ? t_p0 : want == 1 ? t_p1 : want == 2 ? t_p2 : t_p3;
                t_okm = __ballot(t_ok != 0);
                t_any = __ballot(t_ok && TH_D_SRC((uint32_t)t_d) == 0 && want < TH_POA_INL);
                t_run = t_any & __ballot(pr == row - 1);
                t_pr = pr;
            }
            k = 0; o = 1;
        }
        const int tl = k * 4 + o;
        if (st == 0) {
            // a run of plain diagonal steps down this tile column is emitted in one go (most of the path at 10 % error)
            const unsigned long long col_bits = (t_run >> tl) & 0x1111111111111111ull;      // bit 4x: step x of the run is plain
            const int run = th_ctz64(~col_bits & 0x1111111111111111ull) >> 2;              // 16 when the column is plain to the tile's end
            const int lim = min(run, 16 - k);
            if (lim > 0) {
                if (lane < lim) { g.ops[2 * (no + lane)] = 0 | (ci - lane) << 2; g.ops[2 * (no + lane) + 1] = cj - 1 - lane; }
                no += lim; ci -= lim; cj -= lim;
                TH_PROF_CNT(5, 1); TH_PROF_CNT(6, lim);
                continue;
            }
            // a diagonal step to another predecessor (the branch / merge rows of bubbles: every run above ends at one): the same
            // step the general code below would take, without its five lane reads, state machine and second look-up
            if ((t_any >> tl) & 1) {
                const int pr = TH_READLANE(t_pr, tl);
                if (lane == 0) { g.ops[2 * no] = 0 | ci << 2; g.ops[2 * no + 1] = cj - 1; }
                ++no; ci = pr; --cj;
                TH_PROF_CNT(7, 1);
                continue;
            }
        }
        TH_PROF_CNT(7, 1);
        const uint32_t d = (uint32_t)TH_READLANE(t_d, tl);
        const int a0 = TH_READLANE(t_p0, tl), a1 = TH_READLANE(t_p1, tl), a2 = TH_READLANE(t_p2, tl), a3 = TH_READLANE(t_p3, tl);
        if (st == 0) st = TH_D_SRC(d);
        int moved_gap = 0;
        if (st <= 2) {
            const int want = st == 0 ? TH_D_PM(d) : st == 1 ? TH_D_PE1(d) : TH_D_PE2(d);
            int pi;
            if (want < TH_POA_INL) pi = want == 0 ? a0 : want == 1 ? a1 : want == 2 ? a2 : a3;
            else pi = TH_UNI(g.pos[th_poa_in(g, g.order[ci], want)]);
            if (st == 0) { if (lane == 0) { g.ops[2 * no] = 0 | ci << 2; g.ops[2 * no + 1] = cj - 1; } ++no; ci = pi; --cj; }
            else { if (lane == 0) { g.ops[2 * no] = 2 | ci << 2; g.ops[2 * no + 1] = -1; } ++no; ci = pi; moved_gap = 1; }
        } else { if (lane == 0) { g.ops[2 * no] = 1; g.ops[2 * no + 1] = cj - 1; } ++no; --cj; moved_gap = 1; }
        // a gap state continues only if the cell we arrived at says so (its word may sit in the tile)
        {
            const int k2 = ti - ci, o2 = cj - (tj - k2) + 1;
            const bool hit = ((k2 >> 4) | (o2 >> 2)) == 0;
            const int okv = (int)((t_okm >> ((k2 * 4 + o2) & 63)) & 1), dv = TH_READLANE(t_d, hit ? (k2 * 4 + o2) : 0);
            if (moved_gap) {
                if (ci == 0) st = 0;
                else {
                    uint32_t pd;
                    if (hit && okv) pd = (uint32_t)dv;
                    else pd = (uint32_t)TH_UNI((int)th_poa_dir(g, ci, cj - g.r_beg[ci]));
                    const uint32_t bit = st == 1 ? TH_D_XE1 : st == 2 ? TH_D_XE2 : st == 3 ? TH_D_XF1 : TH_D_XF2;
                    if (!(pd & bit)) st = 0;
                }
            }
        }
    }
    while (cj > 0) { if (lane == 0) { g.ops[2 * no] = 1; g.ops[2 * no + 1] = cj - 1; } ++no; --cj; }
    TH_WAVE_FENCE();
    return no;
}

#endif

// ---- merge the alignment (ops were written backwards; rows refer to the CURRENT order) ----
// returns 1 (graph abandoned) when the new nodes would not fit the job's node capacity
__device__ int th_poa_merge(ThPoaG &g, const uint8_t *__restrict__ q, int n_ops, int *__restrict__ upath /* this unit's slice of g.upath, or null */)
{
    const int lane = TH_LANE(), n_old = g.n_node;
    for (int p = lane; p < n_old; p += TH_WAVE) { g.cntB[p] = 0; g.headB[p] = 0x7fffffff; g.aft[p] = -1; }
    TH_WAVE_FENCE();
    // pass 1 (forward): choose / create the node of every column, ring splice, read counts
    int new_cnt = 0, path_cnt = 0;
    for (int f0 = 0; f0 < n_ops; f0 += TH_WAVE) {
        const int f = f0 + lane; const bool ok = f < n_ops;
        const int a = n_ops - 1 - f;
        const int w0 = ok ? g.ops[2 * a] : 2, op = w0 & 3;
        const int qpos = ok ? g.ops[2 * a + 1] : 0;
        int b = 4, v = -1, lo = -1, hi = -1, u = -1, kind = 0;      // kind: 1 new ring member, 2 new inserted node
        if (ok && op != 2) b = min((int)q[qpos], 4);
        if (ok && op == 0) {
            u = g.order[w0 >> 2];
            lo = hi = g.pos[u];
            int found = g.nbase[u] == b ? u : -1;
            for (int m = g.ring[u]; m != u; m = g.ring[m]) {
                const int pm = g.pos[m]; lo = min(lo, pm); hi = max(hi, pm);
                if (found < 0 && g.nbase[m] == b) found = m;
            }
            if (found >= 0) v = found; else kind = 1;
        } else if (ok && op == 1) kind = 2;
        const bool isnew = kind != 0;
        const unsigned long long nm = __ballot(isnew);
        if (n_old + new_cnt + (int)__popcll(nm) > g.cap) return 1;          // (wave-uniform)
        if (isnew) {
            v = n_old + new_cnt + th_mbcnt(nm);
            th_poa_init_node(g, v, b, 0);
            if (kind == 1) { g.ring[v] = g.ring[u]; g.ring[u] = v; g.aft[hi] = v; }
        }
        new_cnt += __popcll(nm);
        const bool isnode = ok && op != 2;
        if (isnode) g.nread[v] += 1;
        const unsigned long long pmk = __ballot(isnode);
        if (isnode) { const int k_ = path_cnt + th_mbcnt(pmk); g.pth[1 + k_] = v; if (upath) upath[k_] = v; }
        path_cnt += __popcll(pmk);
        if (ok) { g.pv[f] = kind == 2 ? v : -1; g.glo[f] = op == 0 ? lo : -1; }
    }
    if (lane == 0) { g.pth[0] = 0; g.pth[1 + path_cnt] = 1; }
    g.n_node = n_old + new_cnt;
    TH_WAVE_FENCE();
    // pass 2: edges along the path (each node is `from` once and `to` once -> no two lanes touch one list)
    for (int k0 = 0; k0 <= path_cnt; k0 += TH_WAVE) {
        const int k = k0 + lane;
        if (k <= path_cnt) th_poa_add_edge(g, g.pth[k], g.pth[k + 1]);
    }
    // pass 3 (backward over forward indices): inserted runs are anchored before the ring of the next aligned column
    int carry_lo = g.pos[1];                   // nothing aligned after them: before the sink
    for (int f0 = ((n_ops - 1) & ~63); f0 >= 0; f0 -= TH_WAVE) {
        const int f = f0 + lane; const bool ok = f < n_ops;
        const int lo = ok ? g.glo[f] : -1, vi = ok ? g.pv[f] : -1;
        const unsigned long long mM = __ballot(lo >= 0);
        // (the lanes above this one as a shift of the ballot, not as a per-lane 64-bit mask: the compiler hoists such a mask to the kernel's
        // entry, and it then lives -- spilled and reloaded inside the two-predecessor rows -- across the whole DP loop; th_mbcnt likewise)
        const unsigned long long higher = lane == 63 ? 0ull : (mM >> (lane + 1));
        const int srcl = higher ? lane + 1 + th_ctz64(higher) : 0;
        const int alo = __shfl(lo, srcl);
        const int anchor = higher ? alo : carry_lo;
        if (vi >= 0) { atomicAdd(&g.cntB[anchor], 1); atomicMin(&g.headB[anchor], vi); }
        const int first_lo = TH_READLANE(lo, mM ? th_ctz64(mM) : 0);
        if (mM) carry_lo = first_lo;
    }
    TH_WAVE_FENCE();
    // pass 4: new positions = old position + inserted nodes at or before it + ring members added before it
    int shift = 0;
    for (int p0 = 0; p0 < n_old; p0 += TH_WAVE) {
        const int p = p0 + lane; const bool ok = p < n_old;
        const int cb = ok ? TH_LOAD_L2(&g.cntB[p]) : 0, af = ok ? (g.aft[p] >= 0) : 0;     // cntB/headB were built with atomics
        const int incB = th_wave_incl_sum(cb), incA = th_wave_incl_sum(af);
        if (ok) {
            const int np = p + shift + incB + (incA - af);
            const int id = g.order[p];
            g.order2[np] = id; g.pos[id] = np;
            const int hb = TH_LOAD_L2(&g.headB[p]);
            for (int x = 0; x < cb; ++x) { g.order2[np - cb + x] = hb + x; g.pos[hb + x] = np - cb + x; }
            if (af) { const int m = g.aft[p]; g.order2[np + 1] = m; g.pos[m] = np + 1; }
        }
        shift += TH_READLANE(incB, 63) + TH_READLANE(incA, 63);
    }
    int *t = g.order; g.order = g.order2; g.order2 = t;
    TH_WAVE_FENCE();
    return 0;
}

// ---- row-column consensus: one column = one aligned ring (its members are adjacent in the order) ----
__device__ int th_poa_consensus(ThPoaG &g, int n_seqs, uint8_t *__restrict__ cons, int *__restrict__ cov, int cons_cap, bool want_cols)
{
    const int lane = TH_LANE(), n = g.n_node;
    int cl = 0, over = 0;
    for (int c0 = 0; c0 < n; c0 += TH_WAVE) {
        const int idx = c0 + lane;
        int bb = -1, best = 0, lead_u = -1;          // lead_u: this lane leads the column of node lead_u
        if (idx >= 1 && idx < n) {
            const int u = g.order[idx];
            if (u != 1) {
                int c_a = 0, c_c = 0, c_g = 0, c_t = 0, c_n = 0, tot = 0, lead = 1;
                int m = u;
                do {
                    const int nr = g.nread[m], bs = g.nbase[m];
                    if (g.pos[m] < idx) lead = 0;
                    c_a += bs == 0 ? nr : 0; c_c += bs == 1 ? nr : 0; c_g += bs == 2 ? nr : 0; c_t += bs == 3 ? nr : 0; c_n += bs == 4 ? nr : 0;
                    tot += nr; m = g.ring[m];
                } while (m != u);
                if (lead) {
                    if (c_a > best) { best = c_a; bb = 0; }
                    if (c_c > best) { best = c_c; bb = 1; }
                    if (c_g > best) { best = c_g; bb = 2; }
                    if (c_t > best) { best = c_t; bb = 3; }
                    if (c_n > best) { best = c_n; bb = 4; }
                    if (TH_POA_GAP_WINS_TIE ? n_seqs - tot >= best : n_seqs - tot > best) bb = -1;
                    lead_u = u;
                }
            }
        }
        const unsigned long long em = __ballot(bb >= 0);
        int oc = -1;
        if (bb >= 0) {
            const int o = cl + th_mbcnt(em);
            if (o < cons_cap) { cons[o] = (uint8_t)bb; cov[o] = best; oc = o; } else over = 1;
        }
        // round 5: every node of the column learns the column's place in the consensus (-1: the column was dropped) -- what th_poa_unit_bounds scores the units' paths with
        if (want_cols && lead_u >= 0) { int m = lead_u; do { g.ccol[m] = oc; m = g.ring[m]; } while (m != lead_u); }
        cl += __popcll(em);
    }
    return __any(over) ? -1 : cl;
}


// ---- round 5: a lower bound of every unit's ksw2 score against the consensus, for free ----
// The POA has aligned every unit to the graph the consensus was read from: the unit's path (g.upath) against the kept columns (g.ccol) IS a global
// alignment of the unit and the consensus -- a unit base in a kept column is a match / mismatch column, one in a dropped column an inserted base, a kept
// column the path skips a deleted one.  Its score under ksw2's scheme as TideHunter sets it (src/ksw2_align.c:12-17,68: 1 / -2, -1 against N,
// a gap of l bases 2 + l) bounds the optimum of ksw2_global (src/gen_cons.c:213) from below, and th_k_align sizes a band that is certified for any optimum
// >= that bound -- no probing pass for the long units of configs 4 / 5.  One pass per unit, 64 bases per trip: the previous kept column is an exclusive
// prefix maximum (columns grow along a path), gap openings are "this element dropped, the one before kept".  Its own function, like th_poa_rows.
__device__ TH_NOINLINE void th_poa_unit_bounds(int *my, int cap, int ocap, int opn, int upn, const ThUnit *__restrict__ U, int n_units, const uint8_t *__restrict__ rb,
                                               const uint8_t *__restrict__ cons, int cl, int *__restrict__ slb)
{
    my = TH_UNI_PTR(int, my);
    ThPoaG g;
    th_poa_carve(g, my, TH_UNI(cap), TH_UNI(ocap), TH_UNI(opn), TH_UNI(upn));
    const int lane = TH_LANE();
    cl = TH_UNI(cl); n_units = TH_UNI(n_units);
    int off = 0;
    for (int s = 0; s < n_units; ++s) {
        const int len = TH_UNI(U[s].len), which = TH_UNI(U[s].which);
        const uint8_t *q = rb + TH_UNI(U[s].start);
        const int *path = g.upath + off;
        int last = -1, prev_kept = 1, sum = 0;          // last kept column so far; was the element before this trip kept (the start counts: a leading insertion opens a gap)
        for (int j0 = 0; j0 < len; j0 += TH_WAVE) {
            const int j = j0 + lane; const bool ok = j < len;
            const int c = ok ? g.ccol[path[ok ? j : 0]] : -1;
            const bool kept = ok && c >= 0;
            const int ub = ok ? (int)q[j] : 0, cb = kept ? (int)cons[c] : 0;
            int sc = kept ? ((ub == 4 || cb == 4) ? -1 : (ub == cb ? 1 : -2)) : 0;
            const int inc = th_wave_incl_max(kept ? c : -1);
            const int prevc = max(th_shr1(inc, -1), last);                       // the kept column before this element
            const int dgap = kept ? c - prevc - 1 : 0;
            if (dgap > 0) sc -= 2 + dgap;
            const int pk = th_shr1((int)kept, prev_kept);
            if (ok && !kept) sc -= 1 + (pk ? 2 : 0);
            sum += th_wave_incl_sum(sc);                                         // (lane 63 holds the trip's total)
            last = max(last, TH_READLANE(inc, 63));
            prev_kept = TH_READLANE((int)kept, min(len - j0, TH_WAVE) - 1);
        }
        int total = TH_READLANE(sum, 63);
        if (cl - 1 - last > 0) total -= 2 + (cl - 1 - last);                    // the consensus columns after the unit's last kept one
        if (lane == 0) slb[which] = total;
        off += len;
    }
}

// status: 0 ok, 1 scratch too small for the graph, 2 band cells overflow, 3 empty band, 4 band lost the path,
//         5 in-degree > 255, 6 consensus longer than cons_cap, 7 scores may not fit the int16 ring (rerun on the 32-bit variant)
// QCAP: query bases staged in LDS (0: read from HBM); RT: element type of the LDS ring of recent rows (see th_poa_dp)
template <int QCAP, typename RT, bool DEFSC = false>
// 6 waves per SIMD: 80 VGPRs (21 dwords of scratch) and 6.5 KB of LDS per wave (query codes packed two per byte); measured per
// 65536 reads: 4 waves 588 ms, 5 waves 557 ms
__global__ void __launch_bounds__(TH_POA_BLOCK) TH_WAVES_PER_SIMD(TH_POA_OCC_NS)
th_k_poa(const ThTask *__restrict__ tasks, const int *__restrict__ list, int n_list, const ThUnit *__restrict__ units, const uint32_t *__restrict__ roff,
         const uint8_t *__restrict__ bseq, ThDevPara P, int *__restrict__ scratch, long long scratch_ints,
         uint8_t *__restrict__ cons, int *__restrict__ cov, int *__restrict__ cons_len, int *__restrict__ status, int *__restrict__ queue, int flags,
         int *__restrict__ slb /* per alignment item (T.aln_off + which): the units' score bounds, th_poa_unit_bounds -- or null */)
{
    // flags: bit 0 = no adaptive band (the rerun of a job whose band failed); bits 8.. = test knob TH_POA_TEST_FAIL: jobs whose
    // index is a multiple of it report status 4 on the banded pass, so the tests can drive the rerun path
    constexpr bool LQ = QCAP > 0;
    const bool noband = flags & 1; const int fexp = (flags >> 1) & 15, test_fail = flags >> 8;
    __shared__ __attribute__((aligned(4))) uint8_t s_qs[QCAP / 2 + 4]; // column j's query code (0..4, code of q[j-1]; 4 for j = 0 and j > qlen) in nibble j & 1 of byte j >> 1
    __shared__ __attribute__((aligned(4))) RT s_ring[TH_POA_RING][3][TH_POA_RS];
    __shared__ int s_rmeta[TH_POA_RING][8];
    const int lane = TH_LANE();
    const int wave = blockIdx.x;
    int *my = scratch + (long long)wave * scratch_ints;
#if defined(TH_POA_PROF)
    if (lane < 10) s_poa_prof[lane] = 0;
    TH_WAVE_FENCE();
    const unsigned long long prof_t_start = __builtin_amdgcn_s_memtime(); int prof_jobs = 0;
#endif
    for (;;) {
        // jobs differ a lot in length (units, copies): waves pull the next job from a shared counter
        const int qi = th_queue_pop(queue);
        if (qi >= n_list) break;
        const int t = list ? TH_UNI(list[qi]) : qi;       // the task list of this kernel variant (th_k_jobs_emit), or every task
        const ThTask T = tasks[t];
        const uint8_t *rb = bseq + roff[T.read];
        const ThUnit *U = units + T.unit_off;
        if (slb) for (int x = lane; x < T.pos_n - 1; x += TH_WAVE) slb[T.aln_off + x] = TH_SLB_NONE;      // (no bound unless the job gets that far -- every variant writes this: th_k_align reads the slot)
        if (T.n_units <= 2) {            // src/abpoa_cons.c:60-83: consensus := unit 0
            const int len = U[0].len;
            const int st = len > T.cons_cap ? 6 : 0;
            if (!st) for (int x = lane; x < len; x += TH_WAVE) { cons[T.cons_off + x] = rb[U[0].start + x]; cov[T.cons_off + x] = 0; }
            if (lane == 0) { cons_len[t] = st ? 0 : len; status[t] = st; }
            continue;
        }
        ThPoaG g;
        const ThPoaCaps K = th_poa_caps(T.n_units, T.sum_len, T.max_len, fexp, noband);      // (node capacity: an estimate, not the worst case -- th_k_poa.hip)
        // (flags bit 5, test knob TH_POA_TEST_SMALLCAP: the first attempt carves node arrays just above the longest unit, so that the tests drive the status-1 rerun)
        const long long cap = (flags & 32) ? min(K.cap, (long long)T.max_len + 8) : K.cap, ocap = K.ocap, opn = cap + T.max_len + 2;
        g.cap = (int)cap; g.n_node = 0;
        // (bounds only where th_k_align uses them: units long enough for a first band of three registers per lane or a probe -- consensus beyond ~2200 bases; the
        // headline shape's jobs skip the pass and its path stores)
        // (compiled out of the absolute-int16 variant: its units end where the bounds begin -- 6 (qlen + rows) < 32000, th_poa_dp -- and the dead code cost the
        // headline kernel 3 ms through register allocation)
        constexpr bool HAS_BOUNDS = TH_POA_REL || sizeof(RT) == 4;
        const bool want_b = HAS_BOUNDS && slb != nullptr && T.max_len >= TH_SLB_MIN_UNIT;
        const long long upn = want_b ? (long long)T.sum_len + 8 : 0;
        int *p = th_poa_carve(g, my, cap, ocap, opn, upn);
        const int *order0 = g.order;       // (th_poa_merge swaps order / order2)
        {
            const long long avail = (scratch_ints - (p - my)) & ~3LL;
            g.cell_u2 = (uint32_t)min(avail * 2, 0x7ff00000LL); g.dcell = (uint16_t *)p; g.cell = (int4 *)(p + avail);
        }
        int st = 0;
        TH_PROF_CNT(26, 1);
#if defined(TH_POA_PROF)
        ++prof_jobs;
#endif
        if (p - my > scratch_ints || g.cell_u2 < 2048 || U[0].len + 2 > cap) st = 1;
        if (!st) th_poa_seed(g, rb + U[0].start, U[0].len, want_b ? g.upath : nullptr);
        int up_off = U[0].len;
        for (int s = 1; s < T.n_units && !st; ++s) {
            const uint8_t *q = rb + U[s].start; const int qlen = U[s].len;
            if (LQ) for (int x = lane; 2 * x < qlen + 3; x += TH_WAVE) {
                const int j0 = 2 * x, j1 = j0 + 1;
                const int c0 = (j0 >= 1 && j0 <= qlen) ? min((int)q[j0 - 1], 4) : 4, c1 = j1 <= qlen ? min((int)q[j1 - 1], 4) : 4;
                s_qs[x] = (uint8_t)(c0 | c1 << 4);
            }
            TH_WAVE_FENCE();
            TH_PROF_T0();
            // TH_POA_DUP_PHASE (measurement only, tools/poa_phases.sh): phase k (0 row tables, 1 DP, 2 traceback, 3 consensus -- all idempotent) runs
            // TWICE, in a loop whose trip count the compiler cannot see (one copy of the code), so that the phase's instructions and time are the
            // difference of two counter passes
            int deg = 0, bi = 0, no = 0;
            for (int r_ = 0; r_ < TH_DUP_N(0); ++r_) deg = th_poa_rows(my, (int)cap, (int)ocap, (int)opn, g.n_node, g.order != order0);
            TH_PROF(0);
            if (deg > 255) { st = 5; break; }
            for (int r_ = 0; r_ < TH_DUP_N(1); ++r_) bi = th_poa_dp<QCAP, RT, DEFSC>(g, q, s_qs, s_ring, s_rmeta, qlen, P, noband);
            bi = TH_UNI(bi);         // (the same in every lane, but loaded: a divergent `break` below would put the other phases under exec masks)
            if (test_fail && !noband && t % test_fail == 0 && bi >= 0) bi = -4;
            TH_PROF(1);
            if (bi < 0) { st = bi == -3 ? 2 : bi == -5 ? 7 : bi == -2 ? 3 : 4; break; }
            for (int r_ = 0; r_ < TH_DUP_N(2); ++r_) no = th_poa_traceback(g, bi, qlen);
            TH_PROF(2);
            if (th_poa_merge(g, q, no, want_b ? g.upath + up_off : nullptr)) { st = 1; break; }          // the graph outgrew its node capacity: rerun with twice the factor
            up_off += qlen;
            TH_PROF(3);
        }
        int cl = 0;
        if (!st) {
            TH_PROF_T0();
            for (int r_ = 0; r_ < TH_DUP_N(3); ++r_) cl = th_poa_consensus(g, T.n_units, cons + T.cons_off, cov + T.cons_off, T.cons_cap, want_b);
            TH_PROF(24);
            if (cl < 0) { st = 6; cl = 0; }
            else if (want_b) { TH_WAVE_FENCE(); th_poa_unit_bounds(my, (int)cap, (int)ocap, (int)opn, (int)upn, U, T.n_units, rb, cons + T.cons_off, cl, slb + T.aln_off); }
        }
        if (lane == 0) { cons_len[t] = st ? 0 : cl; status[t] = st; }
        TH_WAVE_FENCE();
    }
#if defined(TH_POA_PROF)
    TH_WAVE_FENCE();
    // when the waves of the persistent grid run out of jobs: per wave, the time from its start to its last job's end: shortest / longest / mean (the tail of the dynamic queue; all waves of the grid are resident from the start)
    if (lane == 0 && prof_jobs > 0) { const unsigned long long te = __builtin_amdgcn_s_memtime() - prof_t_start;      /* (per wave that took a job: the counters of different XCDs share no base) */ atomicMin(&th_poa_prof_rows[5], te); atomicMax(&th_poa_prof_rows[6], te); atomicAdd(&th_poa_prof_rows[7], te >> 10); atomicAdd(&th_poa_prof_rows[4], 1ull); }
    if (lane < 8) { const unsigned long long v = s_poa_prof[lane]; atomicAdd(&th_poa_prof[8 + lane], v >> 40); atomicAdd(&th_poa_prof[16 + lane], v & ((1ull << 40) - 1)); }
    if (lane == 8) atomicAdd(&th_poa_prof[27], s_poa_prof[8] >> 40);
    if (lane == 9) { atomicAdd(&th_poa_prof[28], s_poa_prof[9] >> 40); atomicAdd(&th_poa_prof[29], s_poa_prof[9] & ((1ull << 40) - 1)); }
#endif
}
}  // namespace TH_POA_NS
#undef TH_RING_LD
#undef TH_RING_ST
#undef TH_POA_SET_RMAX
