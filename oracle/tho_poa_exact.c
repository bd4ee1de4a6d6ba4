/*
 * oracle/tho_poa_exact.c -- TEST INFRASTRUCTURE (see tho.h).
 *
 * A SECOND, INDEPENDENT partial-order alignment, written to cross-check oracle/tho_poa.c (and, through it,
 * the HIP kernel) -- not to be fast and not to share code or data structures with it:
 *   - FULL dynamic-programming matrix over (graph node, query column): no band at all;
 *   - the node order is recomputed from scratch after every sequence by a Kahn / breadth-first topological sort
 *     that releases a node only together with all members of its aligned group, which is how the abPOA paper
 *     (Gao et al. 2021, section 2.2) and abPOA's abpoa_BFS_set_node_index() order the graph [the latter from
 *     memory: libabpoa is absent from /root/reference, .gitmodules:1-4];
 *   - consensus columns are ranked by a second sweep, depth-first (stack) or breadth-first (queue) -- abPOA's
 *     abpoa_DFS_set_msa_rank() is recalled as the stack form -- so the tests can tell whether the left-to-right
 *     order of parallel branches ever changes a consensus;
 *   - adjacency is kept as growable per-node arrays of (node, weight), aligned groups as explicit id lists.
 * Scoring and tie rules are those stated for the reference's call (src/abpoa_cons.c:12-28: match 2, mismatch 4,
 * convex gap min(4+2l, 24+l), row-column consensus); ties: diagonal before graph gaps before query gaps, the first
 * predecessor in edge-insertion order, a gap is extended only when that is strictly better than opening one.
 * What the cross-check can show: tho_poa.c's adaptive band and incremental order never change the result
 * (tests/test_poa_exact.py; at scale tests/test_gpu_scale.py on the GPU box's host cores).  What it cannot show:
 * equality with libabpoa -- PARITY UNPINNED stays in force for stage 4.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "tho.h"

#define XNEG (-0x30000000)

typedef struct { int *id, *w; int n, m; } xlist_t;
typedef struct { uint8_t base; int reads; xlist_t in, out; int *aln; int n_aln, m_aln; } xnode_t;
typedef struct { xnode_t *v; int n, m; int *index_of, *node_at; } xgraph_t;   /* node 0 = source, 1 = sink */

static void xl_push(xlist_t *l, int id)
{
    if (l->n == l->m) { l->m = l->m ? l->m * 2 : 2; l->id = (int *)realloc(l->id, sizeof(int) * l->m); l->w = (int *)realloc(l->w, sizeof(int) * l->m); }
    l->id[l->n] = id; l->w[l->n] = 1; ++l->n;
}
static int xg_node(xgraph_t *g, uint8_t base)
{
    if (g->n == g->m) { g->m = g->m ? g->m * 2 : 1024; g->v = (xnode_t *)realloc(g->v, sizeof(xnode_t) * g->m); }
    memset(&g->v[g->n], 0, sizeof(xnode_t));
    g->v[g->n].base = base;
    return g->n++;
}
static void xg_edge(xgraph_t *g, int a, int b)
{
    xlist_t *o = &g->v[a].out;
    for (int i = 0; i < o->n; ++i) if (o->id[i] == b) { ++o->w[i]; xlist_t *in = &g->v[b].in; for (int k = 0; k < in->n; ++k) if (in->id[k] == a) ++in->w[k]; return; }
    xl_push(o, b); xl_push(&g->v[b].in, a);
}
static void xg_align(xgraph_t *g, int a, int b)      /* b joins a's group: everybody learns about b, b about everybody */
{
    xnode_t *A = &g->v[a];
    int n = A->n_aln, *grp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    memcpy(grp, A->aln, sizeof(int) * (size_t)n); grp[n] = a;
    for (int i = 0; i <= n; ++i) {
        xnode_t *M = &g->v[grp[i]];
        if (M->n_aln == M->m_aln) { M->m_aln = M->m_aln ? M->m_aln * 2 : 2; M->aln = (int *)realloc(M->aln, sizeof(int) * M->m_aln); }
        M->aln[M->n_aln++] = b;
        xnode_t *B = &g->v[b];
        if (B->n_aln == B->m_aln) { B->m_aln = B->m_aln ? B->m_aln * 2 : 2; B->aln = (int *)realloc(B->aln, sizeof(int) * B->m_aln); }
        B->aln[B->n_aln++] = grp[i];
    }
    free(grp);
}

/* Kahn sweep; a node is released only when every member of its aligned group has no pending predecessor, and the
 * group is released together.  lifo = 0: queue (rows of the DP), lifo = 1: stack (column ranks).  out[] gets the node
 * ids in release order; returns how many were released (must equal g->n). */
static int xg_sweep(const xgraph_t *g, int lifo, int *out)
{
    int n = g->n, *deg = (int *)malloc(sizeof(int) * (size_t)n), *q = (int *)malloc(sizeof(int) * (size_t)n * 2), head = 0, tail = 0, k = 0;
    for (int i = 0; i < n; ++i) deg[i] = g->v[i].in.n;
    q[tail++] = 0;
    while (head < tail) {
        int cur = lifo ? q[--tail] : q[head++];
        out[k++] = cur;
        if (cur == 1) break;
        const xlist_t *o = &g->v[cur].out;
        for (int i = 0; i < o->n; ++i) {
            int u = o->id[i];
            if (--deg[u] != 0) continue;
            int ready = 1;
            for (int a = 0; a < g->v[u].n_aln; ++a) if (deg[g->v[u].aln[a]] != 0) { ready = 0; break; }
            if (!ready) continue;
            q[tail++] = u;
            for (int a = 0; a < g->v[u].n_aln; ++a) q[tail++] = g->v[u].aln[a];
        }
    }
    free(deg); free(q);
    return k;
}

typedef struct { int op, node, qpos; } xop_t;     /* 0 aligned (match or mismatch), 1 query base inserted, 2 node skipped */

/* full-matrix global alignment of q against the graph; rows follow g->node_at[]; returns the op count (forward order) */
static int xg_align_seq(const xgraph_t *g, const uint8_t *q, int qlen, const tho_para_t *p, xop_t **ops_out)
{
    const int M = p->match, X = p->mismatch, O1 = p->gap_open1, E1 = p->gap_ext1, O2 = p->gap_open2, E2 = p->gap_ext2;
    const int n = g->n, W = qlen + 1;
    /* scores only for rows that some unprocessed successor still needs; direction bytes for every cell */
    int **H = (int **)calloc((size_t)n, sizeof(int *)), **G1 = (int **)calloc((size_t)n, sizeof(int *)), **G2 = (int **)calloc((size_t)n, sizeof(int *));
    int *pending = (int *)malloc(sizeof(int) * (size_t)n);
    /* per cell: dir = where H came from (0 diag, 1 G1, 2 G2, 3 F1, 4 F2), which predecessor (edge order) gave the diagonal /
     * the G1 / the G2 value, and four "the gap was extended, not opened" flags */
    uint8_t *dir = (uint8_t *)malloc((size_t)n * W), *ext = (uint8_t *)malloc((size_t)n * W);
    uint8_t *wm = (uint8_t *)malloc((size_t)n * W), *w1 = (uint8_t *)malloc((size_t)n * W), *w2 = (uint8_t *)malloc((size_t)n * W);
    for (int r = 0; r < n; ++r) pending[r] = g->v[g->node_at[r]].out.n;
    for (int r = 0; r + 1 < n; ++r) {
        const int v = g->node_at[r];
        if (v == 1) { fprintf(stderr, "[tho_poa_exact] sink is not last\n"); exit(1); }
        int *h = (int *)malloc(sizeof(int) * (size_t)W), *g1 = (int *)malloc(sizeof(int) * (size_t)W), *g2 = (int *)malloc(sizeof(int) * (size_t)W);
        H[r] = h; G1[r] = g1; G2[r] = g2;
        uint8_t *D = dir + (size_t)r * W, *Ex = ext + (size_t)r * W, *Wm = wm + (size_t)r * W, *W1 = w1 + (size_t)r * W, *W2 = w2 + (size_t)r * W;
        if (r == 0) {       /* source row: query gaps only; the traceback never reads its direction bytes (it stops at row 0) */
            for (int j = 0; j <= qlen; ++j) {
                const int a = O1 + j * E1, b = O2 + j * E2;
                h[j] = j ? -(a < b ? a : b) : 0;
                D[j] = 0; Ex[j] = 0; Wm[j] = W1[j] = W2[j] = 0;
                g1[j] = h[j] - O1 - E1; g2[j] = h[j] - O2 - E2;      /* a graph gap opened below this cell */
            }
            continue;
        }
        const xlist_t *in = &g->v[v].in;
        const uint8_t nb = g->v[v].base;
        int f1 = XNEG, f2 = XNEG;
        for (int j = 0; j <= qlen; ++j) {
            int m = XNEG, e1 = XNEG, e2 = XNEG, km = 0, k1 = 0, k2 = 0;
            int s = 0;
            if (j >= 1) { uint8_t qb = q[j - 1]; s = (qb == 4 || nb == 4) ? 0 : (qb == nb ? M : -X); }
            for (int k = 0; k < in->n; ++k) {
                const int pr = g->index_of[in->id[k]];
                if (j >= 1) { int c = H[pr][j - 1] + s; if (k == 0 || c > m) { m = c; km = k; } }
                if (k == 0 || G1[pr][j] > e1) { e1 = G1[pr][j]; k1 = k; }
                if (k == 0 || G2[pr][j] > e2) { e2 = G2[pr][j]; k2 = k; }
            }
            int best = m, d = 0;
            if (e1 > best) { best = e1; d = 1; }
            if (e2 > best) { best = e2; d = 2; }
            if (f1 > best) { best = f1; d = 3; }
            if (f2 > best) { best = f2; d = 4; }
            h[j] = best; D[j] = (uint8_t)d;
            if (in->n > 255) { fprintf(stderr, "[tho_poa_exact] in-degree %d > 255\n", in->n); exit(1); }
            Wm[j] = (uint8_t)km; W1[j] = (uint8_t)k1; W2[j] = (uint8_t)k2;
            Ex[j] = (uint8_t)((e1 > best - O1 ? 1 : 0) | (e2 > best - O2 ? 2 : 0) | (f1 > best - O1 ? 4 : 0) | (f2 > best - O2 ? 8 : 0));
            g1[j] = (best - O1 > e1 ? best - O1 : e1) - E1;
            g2[j] = (best - O2 > e2 ? best - O2 : e2) - E2;
            f1 = (best - O1 > f1 ? best - O1 : f1) - E1;
            f2 = (best - O2 > f2 ? best - O2 : f2) - E2;
        }
        for (int k = 0; k < in->n; ++k) {          /* a predecessor row no later row waits for can go (the sink never counts down: rows it reads stay) */
            const int pr = g->index_of[in->id[k]];
            if (--pending[pr] == 0) { free(H[pr]); free(G1[pr]); free(G2[pr]); H[pr] = G1[pr] = G2[pr] = NULL; }
        }
    }
    /* global end: the best predecessor of the sink at the last column, first in edge order */
    const xlist_t *sin = &g->v[1].in;
    int br = -1, bs = 0;
    for (int k = 0; k < sin->n; ++k) { int pr = g->index_of[sin->id[k]], c = H[pr][qlen]; if (k == 0 || c > bs) { bs = c; br = pr; } }
    xop_t *ops = (xop_t *)malloc(sizeof(xop_t) * (size_t)(n + qlen + 2));
    int no = 0, r = br, j = qlen, st = 0;
    while (r > 0) {
        const size_t c = (size_t)r * W + j;
        const int v = g->node_at[r];
        if (st == 0) st = dir[c];
        if (st <= 2) {
            const int k = st == 0 ? wm[c] : st == 1 ? w1[c] : w2[c];
            const int pr = g->index_of[g->v[v].in.id[k]];
            if (st == 0) { xop_t o = { 0, v, j - 1 }; ops[no++] = o; r = pr; --j; }
            else {
                xop_t o = { 2, v, -1 }; ops[no++] = o; r = pr;
                if (r == 0 || !(ext[(size_t)r * W + j] & (st == 1 ? 1 : 2))) st = 0;
            }
        } else {
            xop_t o = { 1, -1, j - 1 }; ops[no++] = o; --j;
            if (!(ext[(size_t)r * W + j] & (st == 3 ? 4 : 8))) st = 0;
        }
    }
    while (j > 0) { xop_t o = { 1, -1, j - 1 }; ops[no++] = o; --j; }
    for (int a = 0, b = no - 1; a < b; ++a, --b) { xop_t t = ops[a]; ops[a] = ops[b]; ops[b] = t; }
    for (int i = 0; i < n; ++i) { free(H[i]); free(G1[i]); free(G2[i]); }
    free(H); free(G1); free(G2); free(pending); free(dir); free(ext); free(wm); free(w1); free(w2);
    *ops_out = ops;
    return no;
}

static void xg_thread(xgraph_t *g, const uint8_t *q, int qlen, const xop_t *ops, int n_ops)
{
    int last = 0;
    if (!ops) {
        for (int j = 0; j < qlen; ++j) { int v = xg_node(g, q[j]); ++g->v[v].reads; xg_edge(g, last, v); last = v; }
        xg_edge(g, last, 1);
        return;
    }
    for (int a = 0; a < n_ops; ++a) {
        if (ops[a].op == 2) continue;
        const uint8_t b = q[ops[a].qpos];
        int v = -1;
        if (ops[a].op == 0) {
            const int u = ops[a].node;
            if (g->v[u].base == b) v = u;
            else {
                for (int k = 0; k < g->v[u].n_aln; ++k) if (g->v[g->v[u].aln[k]].base == b) { v = g->v[u].aln[k]; break; }
                if (v < 0) { v = xg_node(g, b); xg_align(g, u, v); }
            }
        } else v = xg_node(g, b);
        ++g->v[v].reads;
        xg_edge(g, last, v);
        last = v;
    }
    xg_edge(g, last, 1);
}

static void xg_reindex(xgraph_t *g)
{
    g->node_at = (int *)realloc(g->node_at, sizeof(int) * (size_t)g->n); g->index_of = (int *)realloc(g->index_of, sizeof(int) * (size_t)g->n);
    int k = xg_sweep(g, 0, g->node_at);
    if (k != g->n) { fprintf(stderr, "[tho_poa_exact] topological sweep released %d of %d nodes\n", k, g->n); exit(1); }
    for (int i = 0; i < g->n; ++i) g->index_of[g->node_at[i]] = i;
}

/* flags: bit 0 = rank the consensus columns with the breadth-first sweep (default: depth-first, as abPOA's msa rank is
 * recalled); bit 1 = a gap that is as frequent as the best base wins the column (default: the base wins, THO_POA_GAP_WINS_TIE). */
int tho_poa_exact(int n_seqs, const uint8_t *const *seqs, const int *lens, const tho_para_t *p, int flags, uint8_t *cons, int *cov)
{
    xgraph_t g; memset(&g, 0, sizeof(g));
    xg_node(&g, 4); xg_node(&g, 4);
    uint8_t *buf = NULL; int mbuf = 0;
    for (int s = 0; s < n_seqs; ++s) {
        if (lens[s] > mbuf) { mbuf = lens[s]; buf = (uint8_t *)realloc(buf, (size_t)mbuf); }
        for (int j = 0; j < lens[s]; ++j) buf[j] = seqs[s][j] > 4 ? 4 : seqs[s][j];
        if (s == 0) xg_thread(&g, buf, lens[s], NULL, 0);
        else {
            xg_reindex(&g);
            xop_t *ops; int n_ops = xg_align_seq(&g, buf, lens[s], p, &ops);
            xg_thread(&g, buf, lens[s], ops, n_ops);
            free(ops);
        }
    }
    int *col = (int *)malloc(sizeof(int) * (size_t)g.n), cl = 0;
    uint8_t *seen = (uint8_t *)calloc((size_t)g.n, 1);
    int k = xg_sweep(&g, (flags & 1) ? 0 : 1, col);
    if (k != g.n) { fprintf(stderr, "[tho_poa_exact] rank sweep released %d of %d nodes\n", k, g.n); exit(1); }
    const int gap_wins = (flags >> 1) & 1;
    for (int i = 0; i < g.n; ++i) {
        const int v = col[i];
        if (v < 2 || seen[v]) continue;
        int cnt[5] = { 0, 0, 0, 0, 0 }, tot = 0;
        cnt[g.v[v].base] += g.v[v].reads; tot += g.v[v].reads; seen[v] = 1;
        for (int a = 0; a < g.v[v].n_aln; ++a) { int u = g.v[v].aln[a]; cnt[g.v[u].base] += g.v[u].reads; tot += g.v[u].reads; seen[u] = 1; }
        int best = 0, bb = -1;
        for (int b = 0; b < 5; ++b) if (cnt[b] > best) { best = cnt[b]; bb = b; }
        const int gap = n_seqs - tot;
        if (gap_wins ? gap >= best : gap > best) bb = -1;
        if (bb >= 0) { cons[cl] = (uint8_t)bb; if (cov) cov[cl] = best; ++cl; }
    }
    for (int i = 0; i < g.n; ++i) { free(g.v[i].in.id); free(g.v[i].in.w); free(g.v[i].out.id); free(g.v[i].out.w); free(g.v[i].aln); }
    free(g.v); free(g.node_at); free(g.index_of); free(col); free(seen); free(buf);
    return cl;
}
