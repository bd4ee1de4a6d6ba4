// th_k_poa.hip -- stage 4: partial-order-alignment consensus over the repeat units of one run (gfx950).
//
// Stands where the reference calls abpoa_msa() (src/abpoa_cons.c:30-120; abPOA itself is an absent
// submodule -- see DESIGN.md "POA parity").  The algorithm is the one oracle/tho_poa.c states:
// progressive alignment of units 1..n-1 onto a graph seeded by unit 0, adaptive band
// (w = 10 + qlen/100 around the row maxima of the predecessor rows and the expected diagonal),
// convex gap cost (two affine pieces), row-column consensus by most frequent symbol.
//
// One consensus task per wavefront, waves loop over the task list (persistent launch).
//   * band cells of a graph row are one lane each (64 columns per trip); the in-row gap recurrences
//     F1/F2 are an exclusive max-plus prefix scan across the wave;
//   * each cell stores {H, E1-next, E2-next, direction word} as one 16-byte vector;
//   * graph maintenance (topological order, traceback, merging the alignment) is sequential
//     pointer work done by lane 0 on HBM-resident arrays of the wave's private scratch.
#include "th_dev.h"

#define TH_POA_BLOCK 64
#define TH_POA_WB 10

#define TH_D_SRC(d)  ((d) & 7)
#define TH_D_PM(d)   (((d) >> 3) & 0xff)
#define TH_D_PE1(d)  (((d) >> 11) & 0xff)
#define TH_D_PE2(d)  (((d) >> 19) & 0xff)
#define TH_D_XE1 (1u << 27)
#define TH_D_XE2 (1u << 28)
#define TH_D_XF1 (1u << 29)
#define TH_D_XF2 (1u << 30)

#define TH_POA_INL 4     // adjacency entries kept inline per node; more go to the overflow pool

struct ThPoaG {
    int cap, n_node;            // node capacity / nodes in use (0 = source, 1 = sink)
    int *base, *nread, *ring, *inn, *outn, *in4, *out4, *in_oh, *in_ot, *out_oh, *out_ot;
    int *ovf_val, *ovf_next; int ovf_n, ovf_cap;
    int *order, *index, *remain, *rem, *mpl, *mpr, *beg, *end, *ops;
    long long *roff;            // cell offset of each row
    int4 *cell; long long cell_cap;
};

// ---- adjacency (insertion order preserved: inline slots first, then the overflow list) ----
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
__device__ int th_poa_new_node(ThPoaG &g, int b)
{
    const int v = g.n_node++;
    g.base[v] = b; g.nread[v] = 0; g.ring[v] = v; g.inn[v] = 0; g.outn[v] = 0;
    g.in_oh[v] = g.in_ot[v] = g.out_oh[v] = g.out_ot[v] = -1;
    return v;
}
__device__ void th_poa_add_edge(ThPoaG &g, int from, int to)
{
    const int on = g.outn[from];
    {   // already there?
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
        const int id = g.ovf_n++; g.ovf_val[id] = to; g.ovf_next[id] = -1;
        if (g.out_ot[from] < 0) g.out_oh[from] = id; else g.ovf_next[g.out_ot[from]] = id;
        g.out_ot[from] = id;
    }
    g.outn[from] = on + 1;
    const int in = g.inn[to];
    if (in < TH_POA_INL) g.in4[to * TH_POA_INL + in] = from;
    else {
        const int id = g.ovf_n++; g.ovf_val[id] = from; g.ovf_next[id] = -1;
        if (g.in_ot[to] < 0) g.in_oh[to] = id; else g.ovf_next[g.in_ot[to]] = id;
        g.in_ot[to] = id;
    }
    g.inn[to] = in + 1;
}

// Kahn/BFS order, FIFO, aligned rings released together; `order` doubles as the queue.
__device__ int th_poa_topo(ThPoaG &g)
{
    const int n = g.n_node;
    for (int i = 0; i < n; ++i) g.rem[i] = g.inn[i];
    int qh = 0, qt = 0;
    g.order[qt++] = 0;
    while (qh < qt) {
        const int u = g.order[qh];
        g.index[u] = qh++;
        const int on = g.outn[u];
        int e = g.out_oh[u];
        for (int x = 0; x < on; ++x) {
            int v;
            if (x < TH_POA_INL) v = g.out4[u * TH_POA_INL + x];
            else { v = g.ovf_val[e]; e = g.ovf_next[e]; }
            if (--g.rem[v] > 0) continue;
            bool ok = true;
            for (int m = g.ring[v]; m != v; m = g.ring[m]) if (g.rem[m] > 0) { ok = false; break; }
            if (!ok) continue;
            g.order[qt++] = v;
            for (int m = g.ring[v]; m != v; m = g.ring[m]) g.order[qt++] = m;
        }
    }
    if (qt != n) return -1;
    for (int i = n - 1; i >= 0; --i) {
        const int u = g.order[i];
        int r = -1;
        if (u != 1) {
            const int on = g.outn[u];
            int e = g.out_oh[u];
            for (int x = 0; x < on; ++x) {
                int v;
                if (x < TH_POA_INL) v = g.out4[u * TH_POA_INL + x];
                else { v = g.ovf_val[e]; e = g.ovf_next[e]; }
                r = max(r, g.remain[v] + 1);
            }
        }
        g.remain[u] = r;
    }
    return 0;
}

// banded DP of query q[0..qlen) against the ordered graph; wave-cooperative.  Returns the row index
// of the sink's best predecessor (>= 0) or a negative status.
__device__ int th_poa_dp(ThPoaG &g, const uint8_t *__restrict__ q, int qlen, const ThDevPara &P)
{
    const int lane = TH_LANE();
    const int M = P.match, X = P.mismatch, O1 = P.go1, E1 = P.ge1, O2 = P.go2, E2 = P.ge2;
    const int n = g.n_node, w = TH_POA_WB + qlen / 100;
    for (int i = lane; i < n; i += TH_WAVE) { g.mpl[i] = 0x7fffffff; g.mpr[i] = 0; }
    TH_WAVE_FENCE();
    long long ncell = 0;
    for (int i = 0; i + 1 < n; ++i) {
        const int v = g.order[i];
        const int inn = g.inn[v], rm = g.remain[v];
        int b = 0;
        if (i > 0) {
            b = max(0, min(g.mpl[v], qlen - rm) - w);
            int mb = 0x7fffffff;
            for (int o = 0; o < inn; ++o) mb = min(mb, g.beg[g.index[th_poa_in(g, v, o)]]);
            b = max(b, mb);
        }
        const int e = min(qlen, max(g.mpr[v], qlen - rm) + w);
        if (b > e) return -2;
        if (ncell + (e - b + 1) > g.cell_cap) return -3;
        if (lane == 0) { g.beg[i] = b; g.end[i] = e; g.roff[i] = ncell; }
        int4 *row = g.cell + ncell;
        ncell += e - b + 1;
        const int nb = g.base[v];
        int rmax = TH_NEG - 0x10000000, left = b, right = b;
        int carry1 = TH_NEG + b * E1, carry2 = TH_NEG + b * E2;      // F recurrences in the "+ j*E" domain
        for (int c0 = b; c0 <= e; c0 += TH_WAVE) {
            const int j = c0 + lane;
            const bool in = j <= e;
            int m = TH_NEG, e1v = TH_NEG, e2v = TH_NEG, pm = 0, pe1 = 0, pe2 = 0;
            if (i > 0) {
                int sc = 0;
                if (in && j >= 1) { const int qb = min((int)q[j - 1], 4); sc = (qb == 4 || nb == 4) ? 0 : (qb == nb ? M : -X); }
                for (int o = 0; o < inn; ++o) {
                    const int pi = g.index[th_poa_in(g, v, o)];
                    const int pb = g.beg[pi], pe = g.end[pi];
                    const int4 *prow = g.cell + g.roff[pi];
                    if (in) {
                        if (j >= 1) {
                            const int hp = (j - 1 >= pb && j - 1 <= pe) ? prow[j - 1 - pb].x : TH_NEG;
                            const int c = hp + sc;
                            if (o == 0 || c > m) { m = c; pm = o; }
                        }
                        int c1 = TH_NEG, c2 = TH_NEG;
                        if (j >= pb && j <= pe) { const int4 pc = prow[j - pb]; c1 = pc.y; c2 = pc.z; }
                        if (o == 0 || c1 > e1v) { e1v = c1; pe1 = o; }
                        if (o == 0 || c2 > e2v) { e2v = c2; pe2 = o; }
                    }
                }
            } else if (in) {
                m = j == 0 ? 0 : -min(O1 + j * E1, O2 + j * E2);       // source row: pure leading gap
            }
            // H without the in-row gap states
            int hnf = m, src = 0;
            if (e1v > hnf) { hnf = e1v; src = 1; }
            if (e2v > hnf) { hnf = e2v; src = 2; }
            // F1(j) = max_{k<j}(H(k) - O1 - (j-k)E1): exclusive prefix max of H(k) - O1 + k*E1
            const int a1 = in ? hnf - O1 + j * E1 : TH_NEG * 2, a2 = in ? hnf - O2 + j * E2 : TH_NEG * 2;
            const int i1 = th_wave_incl_max(a1), i2 = th_wave_incl_max(a2);
            int x1 = __shfl_up(i1, 1), x2 = __shfl_up(i2, 1);
            if (lane == 0) { x1 = carry1; x2 = carry2; } else { x1 = max(x1, carry1); x2 = max(x2, carry2); }
            carry1 = max(carry1, __shfl(i1, 63)); carry2 = max(carry2, __shfl(i2, 63));
            int f1 = x1 - j * E1, f2 = x2 - j * E2;
            if (i == 0) { f1 = TH_NEG; f2 = TH_NEG; }
            else if (j == b) { f1 = TH_NEG; f2 = TH_NEG; }
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
#ifdef TH_CPU_EMUL
                if (getenv("TH_DEBUG3") && i == 5) fprintf(stderr, "[th_w] lane %d j %d b %d e %d ncell %lld h %d\n", lane, j, b, e, ncell, h);
#endif
                int4 cv; cv.x = h; cv.y = max(h - O1, e1v) - E1; cv.z = max(h - O2, e2v) - E2; cv.w = (int)d;
                row[j - b] = cv;
            }
            // row maximum, leftmost / rightmost column
            const int hm = th_wave_max(in ? h : TH_NEG * 2);
            const unsigned long long mm = __ballot(in && h == hm);
            if (hm > rmax) { rmax = hm; left = c0 + th_ctz64(mm); right = c0 + th_msb64(mm); }
            else if (hm == rmax) right = c0 + th_msb64(mm);
        }
#ifdef TH_CPU_EMUL
        if (lane == 0 && getenv("TH_DEBUG2") && i < 9) { fprintf(stderr, "[th_H] i %d:", i); for (int j = b; j <= e; ++j) fprintf(stderr, " %d/%d/%d", row[j-b].x, row[j-b].y, row[j-b].z); fprintf(stderr, "\n"); }
        if (lane == 0 && getenv("TH_DEBUG2")) fprintf(stderr, "[th_row] i %d v %d b %d e %d rmax %d l %d r %d\n", i, v, b, e, rmax, left, right);
#endif
        if (lane == 0) {
            const int on = g.outn[v];
            for (int o = 0; o < on; ++o) {
                const int u = th_poa_out(g, v, o);
                if (right + 1 > g.mpr[u]) g.mpr[u] = right + 1;
                if (left + 1 < g.mpl[u]) g.mpl[u] = left + 1;
            }
        }
        TH_WAVE_FENCE();
    }
    // global end: best predecessor row of the sink at column qlen, first in edge order
    int bi = -1, bs = 0;
    const int sinn = g.inn[1];
    for (int o = 0; o < sinn; ++o) {
        const int pi = g.index[th_poa_in(g, 1, o)];
        const int c = (qlen >= g.beg[pi] && qlen <= g.end[pi]) ? g.cell[g.roff[pi] + (qlen - g.beg[pi])].x : TH_NEG;
        if (o == 0 || c > bs) { bs = c; bi = pi; }
    }
#ifdef TH_CPU_EMUL
    if (lane == 0 && getenv("TH_DEBUG")) fprintf(stderr, "[th_poa] n_node %d qlen %d best %d at row %d\n", n, qlen, bs, bi);
#endif
    if (bs <= TH_NEG / 2) return -4;
    return bi;
}

// traceback (lane 0): ops written backwards into g.ops as pairs (op | node<<2, qpos); returns count
__device__ int th_poa_traceback(ThPoaG &g, int bi, int qlen)
{
    int no = 0, i = bi, j = qlen, st = 0;
    while (i > 0) {
        const uint32_t d = (uint32_t)g.cell[g.roff[i] + (j - g.beg[i])].w;
        const int v = g.order[i];
        if (st == 0) st = TH_D_SRC(d);
        if (st <= 2) {
            const int want = st == 0 ? TH_D_PM(d) : st == 1 ? TH_D_PE1(d) : TH_D_PE2(d);
            const int pi = g.index[th_poa_in(g, v, want)];
            if (st == 0) { g.ops[2 * no] = 0 | v << 2; g.ops[2 * no + 1] = j - 1; ++no; i = pi; --j; }
            else {
                g.ops[2 * no] = 2 | v << 2; g.ops[2 * no + 1] = -1; ++no; i = pi;
                if (i == 0) st = 0;
                else {
                    const uint32_t pd = (uint32_t)g.cell[g.roff[i] + (j - g.beg[i])].w;
                    if (!(pd & (st == 1 ? TH_D_XE1 : TH_D_XE2))) st = 0;
                }
            }
        } else {
            g.ops[2 * no] = 1; g.ops[2 * no + 1] = j - 1; ++no; --j;
            const uint32_t pd = (uint32_t)g.cell[g.roff[i] + (j - g.beg[i])].w;
            if (!(pd & (st == 3 ? TH_D_XF1 : TH_D_XF2))) st = 0;
        }
    }
    while (j > 0) { g.ops[2 * no] = 1; g.ops[2 * no + 1] = j - 1; ++no; --j; }
    return no;
}

// merge (lane 0); ops == nullptr means "first sequence"
__device__ void th_poa_add(ThPoaG &g, const uint8_t *__restrict__ q, int qlen, int n_ops)
{
    int last = 0;
    if (n_ops < 0) {
        for (int j = 0; j < qlen; ++j) { const int v = th_poa_new_node(g, q[j] > 4 ? 4 : q[j]); g.nread[v] = 1; th_poa_add_edge(g, last, v); last = v; }
    } else {
        for (int a = n_ops - 1; a >= 0; --a) {          // ops were written backwards
            const int op = g.ops[2 * a] & 3;
            if (op == 2) continue;
            int b = q[g.ops[2 * a + 1]]; if (b > 4) b = 4;
            int v = -1;
            if (op == 0) {
                const int u = g.ops[2 * a] >> 2;
                if (g.base[u] == b) v = u;
                else {
                    for (int m = g.ring[u]; m != u; m = g.ring[m]) if (g.base[m] == b) { v = m; break; }
                    if (v < 0) { v = th_poa_new_node(g, b); g.ring[v] = g.ring[u]; g.ring[u] = v; }
                }
            } else v = th_poa_new_node(g, b);
            ++g.nread[v];
            th_poa_add_edge(g, last, v);
            last = v;
        }
    }
    th_poa_add_edge(g, last, 1);
}

// scratch layout helper: carve `cnt` ints
__device__ __forceinline__ int *th_carve(int *&p, long long cnt) { int *r = p; p += (cnt + 3) & ~3LL; return r; }

// status: 0 ok, 1 scratch too small for the graph, 2 band cells overflow, 3 topo failure, 4 band lost the path,
//         5 in-degree > 255, 6 consensus longer than cons_cap
__global__ void __launch_bounds__(TH_POA_BLOCK)
th_k_poa(const ThTask *__restrict__ tasks, int n_tasks, const ThUnit *__restrict__ units, const uint32_t *__restrict__ roff,
         const uint8_t *__restrict__ bseq, ThDevPara P, int *__restrict__ scratch, long long scratch_ints,
         uint8_t *__restrict__ cons, int *__restrict__ cov, int *__restrict__ cons_len, int *__restrict__ status)
{
    const int lane = TH_LANE();
    const int wave = blockIdx.x, nwave = gridDim.x;
    int *my = scratch + (long long)wave * scratch_ints;
    for (int t = wave; t < n_tasks; t += nwave) {
        const ThTask T = tasks[t];
        const uint8_t *rb = bseq + roff[T.read];
        const ThUnit *U = units + T.unit_off;
        if (T.n_units <= 2) {            // src/abpoa_cons.c:60-83: consensus := unit 0
            const int len = U[0].len;
            int st = len > T.cons_cap ? 6 : 0;
            if (!st) for (int x = lane; x < len; x += TH_WAVE) { cons[T.cons_off + x] = rb[U[0].start + x]; cov[T.cons_off + x] = 0; }
            if (lane == 0) { cons_len[t] = st ? 0 : len; status[t] = st; }
            continue;
        }
        // ---- carve the wave's scratch for this task ----
        ThPoaG g;
        const int cap = T.sum_len + 2, ocap = T.sum_len + T.n_units + 8;
        int *p = my;
        g.cap = cap; g.n_node = 0; g.ovf_n = 0; g.ovf_cap = ocap;
        g.base = th_carve(p, cap); g.nread = th_carve(p, cap); g.ring = th_carve(p, cap); g.inn = th_carve(p, cap); g.outn = th_carve(p, cap);
        g.in4 = th_carve(p, (long long)cap * TH_POA_INL); g.out4 = th_carve(p, (long long)cap * TH_POA_INL);
        g.in_oh = th_carve(p, cap); g.in_ot = th_carve(p, cap); g.out_oh = th_carve(p, cap); g.out_ot = th_carve(p, cap);
        g.ovf_val = th_carve(p, 2LL * ocap); g.ovf_next = th_carve(p, 2LL * ocap);
        g.order = th_carve(p, cap); g.index = th_carve(p, cap); g.remain = th_carve(p, cap); g.rem = th_carve(p, cap);
        g.mpl = th_carve(p, cap); g.mpr = th_carve(p, cap); g.beg = th_carve(p, cap); g.end = th_carve(p, cap);
        g.ops = th_carve(p, 2LL * (cap + T.max_len + 2));
        g.roff = (long long *)th_carve(p, 2LL * cap);
        g.cell = (int4 *)p;
        g.cell_cap = (scratch_ints - (p - my)) / 4;
        int st = 0;
        if (p - my > scratch_ints || g.cell_cap < 64) st = 1;
        int n_node = 0, ovf_n = 0;
        if (!st) {
            if (lane == 0) {
                th_poa_new_node(g, 4); th_poa_new_node(g, 4);
                th_poa_add(g, rb + U[0].start, U[0].len, -1);
            }
            TH_WAVE_FENCE();
            n_node = __shfl(g.n_node, 0); ovf_n = __shfl(g.ovf_n, 0);
        }
        for (int s = 1; s < T.n_units && !st; ++s) {
            const uint8_t *q = rb + U[s].start; const int qlen = U[s].len;
            g.n_node = n_node; g.ovf_n = ovf_n;
            int ts = 0;
            if (lane == 0) ts = th_poa_topo(g);
            TH_WAVE_FENCE();
            ts = __shfl(ts, 0);
            if (ts < 0) { st = 3; break; }
            int deg = 0;
            for (int i = lane; i < n_node; i += TH_WAVE) deg = max(deg, g.inn[i]);
            if (th_wave_max(deg) > 255) { st = 5; break; }
            const int bi = th_poa_dp(g, q, qlen, P);
            if (bi < 0) { st = bi == -3 ? 2 : 4; break; }
            if (lane == 0) {
                const int no = th_poa_traceback(g, bi, qlen);
                th_poa_add(g, q, qlen, no);
            }
            TH_WAVE_FENCE();
            n_node = __shfl(g.n_node, 0); ovf_n = __shfl(g.ovf_n, 0);
        }
        int cl = 0;
        if (!st) {
            g.n_node = n_node; g.ovf_n = ovf_n;
            if (lane == 0) {
                if (th_poa_topo(g) < 0) st = 3;
                else {
                    // row-column consensus: a column = one aligned ring (its members are adjacent in the order)
                    for (int i = 1; i + 1 < n_node;) {
                        const int v = g.order[i];
                        int members = 1;
                        for (int m = g.ring[v]; m != v; m = g.ring[m]) ++members;
                        int cnt[5] = { 0, 0, 0, 0, 0 }, tot = 0;
                        for (int x = i; x < i + members; ++x) { const int u = g.order[x]; cnt[g.base[u]] += g.nread[u]; tot += g.nread[u]; }
                        int best = 0, bb = -1;
                        for (int b = 0; b < 5; ++b) if (cnt[b] > best) { best = cnt[b]; bb = b; }
                        if (T.n_units - tot > best) bb = -1;
                        if (bb >= 0) {
                            if (cl >= T.cons_cap) { st = 6; break; }
                            cons[T.cons_off + cl] = (uint8_t)bb; cov[T.cons_off + cl] = best; ++cl;
                        }
                        i += members;
                    }
                }
            }
            st = __shfl(st, 0); cl = __shfl(cl, 0);
        }
        if (lane == 0) { cons_len[t] = st ? 0 : cl; status[t] = st; }
        TH_WAVE_FENCE();
    }
}
