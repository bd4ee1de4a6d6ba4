/*
 * oracle/tho_poa.c -- TEST INFRASTRUCTURE (see tho.h).
 *
 * PARITY UNPINNED.  The reference calls abpoa_msa() (src/abpoa_cons.c:86-87) from libabpoa, a
 * git submodule that is empty in /root/reference (.gitmodules:1-4; commit id unknown, no copy
 * anywhere in this image).  This file therefore restates the *published* abPOA algorithm
 * (Gao et al., Bioinformatics 2021) with the parameters TideHunter sets
 * (src/abpoa_cons.c:12-28: match 2, mismatch 4, convex gaps O1=4,E1=2,O2=24,E2=1,
 * cons_agrm = 1 "RC"), not abPOA's source.  It is the de-facto oracle for the HIP POA kernel;
 * nothing here may be cited as "equal to the reference" beyond trivially identical units.
 *
 * Algorithm, as restated:
 *   graph    : nodes 0 = source, 1 = sink, >=2 bases in creation order; weighted edges kept in
 *              insertion order; nodes that carry different bases of one MSA column form an
 *              "aligned ring".
 *   order    : the rows of the DP follow a topological order of the graph in which the members of an aligned ring
 *              are adjacent.  Which such order is used cannot change the alignment: a row's values depend only on
 *              its predecessors' rows, predecessor ties are broken by edge-insertion order (not by row order), and
 *              the band of a row is derived from its predecessors' bands and row maxima.  abPOA obtains its order
 *              with a breadth-first (Kahn) sort after every sequence; this file maintains one incrementally instead
 *              (the first unit is the chain source, bases, sink; a new ring member goes right after the last member
 *              of its ring; a run of inserted bases right before the ring of the next aligned node of the path, or
 *              before the sink).  The only place an order could show is the left-to-right order of two parallel
 *              insertion branches in the consensus; oracle/tho_poa_exact.c re-sorts with the Kahn sort the paper
 *              describes (rows breadth-first, consensus columns depth- or breadth-first) and the tests compare:
 *              with the band switched off this file equals the exact POA on every job of tests/test_poa_exact.py and
 *              of the at-scale GPU-box run (tests/test_gpu_scale.py, >= 2000 jobs per config shape).  The adaptive
 *              BAND is a different matter: it is part of the algorithm and does change a consensus now and then
 *              (1 of 2053 config-2 jobs at 10 % error; a few per cent of jobs at 25-30 % error).
 *              max_remain[v] = longest node count from v to the sink (sink = -1).
 *   DP       : global alignment of the query (columns 0..qlen) against the ordered nodes,
 *              convex gap = two affine pieces, 32-bit scores, score(a,b) = 0 if either is N(4),
 *              +match if equal, -mismatch otherwise.
 *              adaptive band (abPOA: b = 10, f = 0.01): w = 10 + qlen/100,
 *                beg = max(0, min(max_pos_left[v],  qlen - max_remain[v]) - w), not below the
 *                      smallest beg of v's predecessors,
 *                end = min(qlen, max(max_pos_right[v], qlen - max_remain[v]) + w),
 *              where max_pos_left/right[v] = 1 + leftmost/rightmost column of the row maximum
 *              of any predecessor row.  Cells outside a band read as NEG.
 *   ties     : H prefers M, then E1, E2 (node skipped), then F1, F2 (query base inserted);
 *              among predecessors the first in edge order; a gap is continued only when
 *              that is strictly better than re-opening it (same convention as ksw2).
 *   merge    : walk the alignment; a matched base re-uses the node (or a ring member with that
 *              base), a mismatched base becomes a new ring member, an inserted base a new node.
 *   consensus: per column (ring) in topological order, the most frequent symbol among the
 *              column's bases (count = reads through the node) and the gap
 *              (n_seq - sum of counts); symbols are scanned A,C,G,T,N,gap and a later one must be
 *              strictly more frequent to win (so a base beats an equally frequent gap and the
 *              smaller code wins among bases).  cov = count of the chosen base.
 *              (Sanity anchor, not a pin: on test_data/full_length.fa this rule gives a 201-base
 *              consensus against the 203 bases README.md:232 shows for an older partition of
 *              the same read; requiring a strict majority over the gap gives 198.)
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "tho.h"

#define NEG (-0x20000000)
#define WB 10

/* Named switches (tho.h).  THO_POA_GAP_WINS_TIE: what a consensus column does when the gap is exactly as frequent as
 * the best base -- 0: the base is kept (default; the scan order A,C,G,T,N,gap with a strict comparison), 1: the column
 * is dropped.  libabpoa's own rule cannot be read here (absent submodule); when its source is available this default is
 * the ONE line to pin (the HIP kernel has the same constant, TH_POA_GAP_WINS_TIE in th_k_poa.hip).
 * THO_POA_OPT_BAND = 0 disables the adaptive band (every row spans 0..qlen): used by the cross-check tests only. */
static int g_gap_wins_tie = THO_POA_GAP_WINS_TIE, g_band = 1;
static long *g_band_hist = NULL;     /* measurement hook (tools): histogram of band widths in bins of 16 columns */
void tho_poa_band_hist(long *h) { g_band_hist = h; }
void tho_poa_set_opt(int what, int value)
{
    if (what == THO_POA_OPT_GAP_TIE) g_gap_wins_tie = value ? 1 : 0;
    else if (what == THO_POA_OPT_BAND) g_band = value ? 1 : 0;
}

typedef struct {
    uint8_t base; int n_read;
    int in_head, in_tail, out_head, out_tail, in_n;
    int ring;                 /* next node of the aligned ring (self if alone) */
} node_t;
typedef struct { int from, to, w, next_in, next_out; } edge_t;

typedef struct {
    node_t *node; int n_node, m_node;
    edge_t *edge; int n_edge, m_edge;
    int *order, *index;       /* order[idx] = node id, index[node id] = idx */
    int *remain;              /* by node id */
} graph_t;

static int new_node(graph_t *g, uint8_t base)
{
    if (g->n_node == g->m_node) { g->m_node = g->m_node ? g->m_node * 2 : 256; g->node = (node_t *)realloc(g->node, sizeof(node_t) * g->m_node); }
    node_t *v = &g->node[g->n_node];
    v->base = base; v->n_read = 0; v->in_head = v->in_tail = v->out_head = v->out_tail = -1; v->in_n = 0;
    v->ring = g->n_node;
    return g->n_node++;
}

static void add_edge(graph_t *g, int from, int to)
{
    for (int e = g->node[from].out_head; e >= 0; e = g->edge[e].next_out)
        if (g->edge[e].to == to) { ++g->edge[e].w; return; }
    if (g->n_edge == g->m_edge) { g->m_edge = g->m_edge ? g->m_edge * 2 : 256; g->edge = (edge_t *)realloc(g->edge, sizeof(edge_t) * g->m_edge); }
    int id = g->n_edge++;
    edge_t *e = &g->edge[id];
    e->from = from; e->to = to; e->w = 1; e->next_in = e->next_out = -1;
    node_t *a = &g->node[from], *b = &g->node[to];
    if (a->out_tail < 0) a->out_head = id; else g->edge[a->out_tail].next_out = id;
    a->out_tail = id;
    if (b->in_tail < 0) b->in_head = id; else g->edge[b->in_tail].next_in = id;
    b->in_tail = id; ++b->in_n;
}

/* max_remain for every node: longest node count from v to the sink (sink = -1), reverse pass over the order */
static void set_remain(graph_t *g)
{
    int n = g->n_node;
    g->remain = (int *)realloc(g->remain, sizeof(int) * n);
    for (int i = n - 1; i >= 0; --i) {
        int u = g->order[i], r = -1;
        if (u != 1)
            for (int e = g->node[u].out_head; e >= 0; e = g->edge[e].next_out) {
                int x = g->remain[g->edge[e].to] + 1;
                if (x > r) r = x;
            }
        g->remain[u] = r;
    }
}

/* alignment ops, forward order */
typedef struct { int op, node, qpos; } aop_t;   /* op: 0 match/mismatch, 1 insertion (query only), 2 deletion (node only) */

#define D_SRC(d)  ((d) & 7)
#define D_PM(d)   (((d) >> 3) & 0xff)
#define D_PE1(d)  (((d) >> 11) & 0xff)
#define D_PE2(d)  (((d) >> 19) & 0xff)
#define D_XE1 (1u << 27)
#define D_XE2 (1u << 28)
#define D_XF1 (1u << 29)
#define D_XF2 (1u << 30)

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* returns the op count, -1 if the adaptive band came out empty or lost the global path, -2 if a node has more than 255 predecessors */
static int align_to_graph(graph_t *g, const uint8_t *q, int qlen, const tho_para_t *p, aop_t **ops_out, int band)
{
    const int M = p->match, X = p->mismatch, O1 = p->gap_open1, E1 = p->gap_ext1, O2 = p->gap_open2, E2 = p->gap_ext2;
    int n = g->n_node, w = WB + qlen / 100;
    int *beg = (int *)malloc(sizeof(int) * n * 4), *end = beg + n, *mpl = end + n, *mpr = mpl + n;
    size_t *roff = (size_t *)malloc(sizeof(size_t) * n), ncell = 0;
    /* band-only storage: row i holds columns beg[i]..end[i] at roff[i].  The four cell arrays (a few MB each) are per-thread
     * and grow-only: allocating and freeing them per alignment made hundreds of worker threads contend in the kernel's
     * mmap/munmap path (the CPU baseline of bench.py runs this with one thread per core). */
    static __thread int *tl_H, *tl_N1, *tl_N2; static __thread uint32_t *tl_D; static __thread size_t tl_mcell;
    size_t mcell = (size_t)n * (2 * w + 64);
    if (mcell > tl_mcell) {
        tl_mcell = mcell + mcell / 4;
        tl_H = (int *)realloc(tl_H, sizeof(int) * tl_mcell); tl_N1 = (int *)realloc(tl_N1, sizeof(int) * tl_mcell);
        tl_N2 = (int *)realloc(tl_N2, sizeof(int) * tl_mcell); tl_D = (uint32_t *)realloc(tl_D, sizeof(uint32_t) * tl_mcell);
    }
    mcell = tl_mcell;
    int *H = tl_H, *N1 = tl_N1, *N2 = tl_N2; uint32_t *D = tl_D;
    for (int i = 0; i < n; ++i) { mpl[i] = 0x7fffffff; mpr[i] = 0; }
#define AT(A, i, j) ((A)[roff[i] + (size_t)((j) - beg[i])])
#define RD(A, i, j) (((j) < beg[i] || (j) > end[i]) ? NEG : AT(A, i, j))
    for (int i = 0; i + 1 < n; ++i) {            /* every row except the sink's */
        int v = g->order[i];
        int b, e;
        if (i == 0) { b = 0; }
        else {
            b = imax(0, imin(mpl[v], qlen - g->remain[v]) - w);
            int mb = 0x7fffffff;
            for (int ed = g->node[v].in_head; ed >= 0; ed = g->edge[ed].next_in) mb = imin(mb, beg[g->index[g->edge[ed].from]]);
            if (b < mb) b = mb;
        }
        e = imin(qlen, imax(mpr[v], qlen - g->remain[v]) + w);
        if (!band) { b = 0; e = qlen; }
        if (b > e) { free(beg); free(roff); return -1; }
        beg[i] = b; end[i] = e; roff[i] = ncell; ncell += (size_t)(e - b + 1);
        if (g_band_hist) { int wdt = (e - b + 1) / 16; __atomic_fetch_add(&g_band_hist[wdt > 63 ? 63 : wdt], 1L, __ATOMIC_RELAXED); }
        if (ncell > mcell) {
            while (ncell > mcell) mcell *= 2;
            H = tl_H = (int *)realloc(H, sizeof(int) * mcell); N1 = tl_N1 = (int *)realloc(N1, sizeof(int) * mcell);
            N2 = tl_N2 = (int *)realloc(N2, sizeof(int) * mcell); D = tl_D = (uint32_t *)realloc(D, sizeof(uint32_t) * mcell);
            tl_mcell = mcell;
        }
        int f1 = NEG, f2 = NEG, rmax = NEG - 0x10000000, left = b, right = b;
        for (int j = b; j <= e; ++j) {
            int h, src = 0, pm = 0, pe1 = 0, pe2 = 0, e1v = NEG, e2v = NEG;
            if (i == 0) {
                h = j == 0 ? 0 : -imin(O1 + j * E1, O2 + j * E2);
                /* the row maximum of the source row is column 0 by construction */
            } else {
                int m = NEG; int ord = 0;
                for (int ed = g->node[v].in_head; ed >= 0; ed = g->edge[ed].next_in, ++ord) {
                    int pi = g->index[g->edge[ed].from];
                    if (j >= 1) {
                        uint8_t qb = q[j - 1], nb = g->node[v].base;
                        int s = (qb == 4 || nb == 4) ? 0 : (qb == nb ? M : -X);
                        int c = RD(H, pi, j - 1) + s;
                        if (ord == 0 || c > m) { m = c; pm = ord; }
                    }
                    int c1 = RD(N1, pi, j), c2 = RD(N2, pi, j);
                    if (ord == 0 || c1 > e1v) { e1v = c1; pe1 = ord; }
                    if (ord == 0 || c2 > e2v) { e2v = c2; pe2 = ord; }
                }
                if (ord > 255) { free(beg); free(roff); return -2; }
                h = m; src = 0;
                if (e1v > h) { h = e1v; src = 1; }
                if (e2v > h) { h = e2v; src = 2; }
                if (f1 > h) { h = f1; src = 3; }
                if (f2 > h) { h = f2; src = 4; }
            }
            uint32_t d = (uint32_t)src | (uint32_t)pm << 3 | (uint32_t)pe1 << 11 | (uint32_t)pe2 << 19;
            if (i == 0) { e1v = NEG; e2v = NEG; if (j > 0) { f1 = -(O1 + j * E1); f2 = -(O2 + j * E2); } }
            if (e1v > h - O1) d |= D_XE1;
            if (e2v > h - O2) d |= D_XE2;
            if (f1 > h - O1) d |= D_XF1;
            if (f2 > h - O2) d |= D_XF2;
            AT(H, i, j) = h; AT(D, i, j) = d;
            AT(N1, i, j) = imax(h - O1, e1v) - E1;
            AT(N2, i, j) = imax(h - O2, e2v) - E2;
            f1 = imax(h - O1, f1) - E1;
            f2 = imax(h - O2, f2) - E2;
            if (h > rmax) { rmax = h; left = right = j; } else if (h == rmax) right = j;
        }
        for (int ed = g->node[v].out_head; ed >= 0; ed = g->edge[ed].next_out) {
            int u = g->edge[ed].to;
            if (right + 1 > mpr[u]) mpr[u] = right + 1;
            if (left + 1 < mpl[u]) mpl[u] = left + 1;
        }
    }
    /* global end: best predecessor row of the sink at column qlen (first in edge order) */
    int bi = -1, bs = 0, ord = 0;
    for (int ed = g->node[1].in_head; ed >= 0; ed = g->edge[ed].next_in, ++ord) {
        int pi = g->index[g->edge[ed].from], c = RD(H, pi, qlen);
        if (ord == 0 || c > bs) { bs = c; bi = pi; }
    }
    if (bs <= NEG / 2) { free(beg); free(roff); return -1; }
    /* traceback */
    aop_t *ops = (aop_t *)malloc(sizeof(aop_t) * (size_t)(n + qlen + 2));
    int no = 0, i = bi, j = qlen, st = 0;
    while (i > 0) {
        uint32_t d = AT(D, i, j);
        int v = g->order[i];
        if (st == 0) st = D_SRC(d);
        if (st == 0 || st == 1 || st == 2) {
            int want = st == 0 ? D_PM(d) : st == 1 ? D_PE1(d) : D_PE2(d), ed = g->node[v].in_head;
            for (int x = 0; x < want; ++x) ed = g->edge[ed].next_in;
            int pi = g->index[g->edge[ed].from];
            if (st == 0) { aop_t o = { 0, v, j - 1 }; ops[no++] = o; i = pi; --j; }
            else {
                aop_t o = { 2, v, -1 }; ops[no++] = o; i = pi;
                uint32_t pd = AT(D, i, j);
                if (i == 0 || !(pd & (st == 1 ? D_XE1 : D_XE2))) st = 0;
            }
        } else {
            aop_t o = { 1, -1, j - 1 }; ops[no++] = o; --j;
            uint32_t pd = AT(D, i, j);
            if (!(pd & (st == 3 ? D_XF1 : D_XF2))) st = 0;
        }
    }
    while (j > 0) { aop_t o = { 1, -1, j - 1 }; ops[no++] = o; --j; }
    for (int a = 0, b2 = no - 1; a < b2; ++a, --b2) { aop_t t = ops[a]; ops[a] = ops[b2]; ops[b2] = t; }
#undef AT
#undef RD
    free(beg); free(roff);
    *ops_out = ops;
    return no;
}

static void add_alignment(graph_t *g, const uint8_t *q, int qlen, const aop_t *ops, int n_ops)
{
    int last = 0;
    if (ops == NULL) {                       /* first sequence: a plain chain; order = source, bases, sink */
        for (int j = 0; j < qlen; ++j) { int v = new_node(g, q[j]); ++g->node[v].n_read; add_edge(g, last, v); last = v; }
        add_edge(g, last, 1);
        int n = g->n_node;
        g->order = (int *)realloc(g->order, sizeof(int) * n); g->index = (int *)realloc(g->index, sizeof(int) * n);
        g->order[0] = 0; for (int j = 0; j < qlen; ++j) g->order[1 + j] = 2 + j; g->order[n - 1] = 1;
        for (int i = 0; i < n; ++i) g->index[g->order[i]] = i;
        return;
    }
    int n_old = g->n_node;
    /* per old order position: new nodes to emit before it (a run of insertions) / one new ring member after it */
    int *ins_head = (int *)malloc(sizeof(int) * n_old * 3), *ins_cnt = ins_head + n_old, *after = ins_cnt + n_old;
    for (int i = 0; i < n_old; ++i) { ins_head[i] = -1; ins_cnt[i] = 0; after[i] = -1; }
    int *pend = (int *)malloc(sizeof(int) * (size_t)(qlen + 1)), n_pend = 0;   /* inserted nodes waiting for their anchor */
    for (int a = 0; a < n_ops; ++a) {
        if (ops[a].op == 2) continue;
        uint8_t b = q[ops[a].qpos];
        int v = -1;
        if (ops[a].op == 0) {
            int u = ops[a].node;
            /* first / last order position of u's ring (ring members are adjacent) */
            int lo = g->index[u], hi = g->index[u];
            for (int m = g->node[u].ring; m != u; m = g->node[m].ring) { if (g->index[m] < lo) lo = g->index[m]; if (g->index[m] > hi) hi = g->index[m]; }
            if (g->node[u].base == b) v = u;
            else {
                for (int m = g->node[u].ring; m != u; m = g->node[m].ring) if (g->node[m].base == b) { v = m; break; }
                if (v < 0) {             /* new member, linked right after u, ordered after the ring's last member */
                    v = new_node(g, b);
                    g->node[v].ring = g->node[u].ring; g->node[u].ring = v;
                    after[hi] = v;
                }
            }
            if (n_pend) { ins_head[lo] = pend[0]; ins_cnt[lo] = n_pend; n_pend = 0; }   /* insertions go right before this ring */
        } else { v = new_node(g, b); pend[n_pend++] = v; }
        ++g->node[v].n_read;
        add_edge(g, last, v);
        last = v;
    }
    add_edge(g, last, 1);
    int sink_pos = g->index[1];
    if (n_pend) { ins_head[sink_pos] = pend[0]; ins_cnt[sink_pos] = n_pend; }
    /* inserted nodes of one run have consecutive ids (created in path order) */
    int n = g->n_node, *no = (int *)malloc(sizeof(int) * n), k = 0;
    for (int i = 0; i < n_old; ++i) {
        for (int x = 0; x < ins_cnt[i]; ++x) no[k++] = ins_head[i] + x;
        no[k++] = g->order[i];
        if (after[i] >= 0) no[k++] = after[i];
    }
    free(g->order); g->order = no;
    g->index = (int *)realloc(g->index, sizeof(int) * n);
    for (int i = 0; i < n; ++i) g->index[g->order[i]] = i;
    free(ins_head); free(pend);
}

/* abPOA has no failure mode; the restated band could in principle come out empty or lose the global path on a pathological
 * graph.  Such a job is redone from scratch with full rows (the HIP pipeline does the same: status 3 / 4 -> rerun without a
 * band); a node with more than 255 predecessors (the direction word keeps 8 bits per predecessor index) leaves the job
 * without a consensus on both sides, with a warning -- never a dead process. */
int tho_poa_consensus(int n_seqs, const uint8_t *const *seqs, const int *lens, const tho_para_t *p, uint8_t *cons, int *cov)
{
    graph_t g; uint8_t *buf = NULL; int mbuf = 0, band = g_band, failed = 0;
RESTART:
    memset(&g, 0, sizeof(g));
    new_node(&g, 4); new_node(&g, 4);        /* source, sink */
    for (int s = 0; s < n_seqs; ++s) {
        if (lens[s] > mbuf) { mbuf = lens[s]; buf = (uint8_t *)realloc(buf, mbuf); }
        for (int j = 0; j < lens[s]; ++j) buf[j] = seqs[s][j] > 4 ? 4 : seqs[s][j];
        if (s == 0) add_alignment(&g, buf, lens[s], NULL, 0);
        else {
            set_remain(&g);
            aop_t *ops; int n_ops = align_to_graph(&g, buf, lens[s], p, &ops, band);
            if (n_ops < 0) {
                free(g.node); free(g.edge); free(g.order); free(g.index); free(g.remain);
                if (n_ops == -1 && band) { band = 0; goto RESTART; }
                fprintf(stderr, "[tho_poa] warning: a consensus job could not be aligned (%s); reported without a consensus\n", n_ops == -2 ? "in-degree > 255" : "no global path");
                failed = 1; memset(&g, 0, sizeof(g));
                break;
            }
            add_alignment(&g, buf, lens[s], ops, n_ops);
            free(ops);
        }
    }
    if (failed) { free(buf); return 0; }
    int cl = 0;
    for (int i = 1; i + 1 < g.n_node;) {
        int v = g.order[i], cnt[5] = { 0, 0, 0, 0, 0 }, tot = 0, k = i;
        /* the ring's members are adjacent in the order */
        int members = 1;
        for (int m = g.node[v].ring; m != v; m = g.node[m].ring) ++members;
        for (k = i; k < i + members; ++k) { int u = g.order[k]; cnt[g.node[u].base] += g.node[u].n_read; tot += g.node[u].n_read; }
        int gap = n_seqs - tot, best = 0, bb = -1;
        for (int b = 0; b < 5; ++b) if (cnt[b] > best) { best = cnt[b]; bb = b; }
        if (g_gap_wins_tie ? gap >= best : gap > best) bb = -1;
        if (bb >= 0) { cons[cl] = (uint8_t)bb; if (cov) cov[cl] = best; ++cl; }
        i += members;
    }
    free(buf); free(g.node); free(g.edge); free(g.order); free(g.index); free(g.remain);
    return cl;
}
