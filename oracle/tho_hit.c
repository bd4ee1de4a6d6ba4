/*
 * oracle/tho_hit.c -- TEST INFRASTRUCTURE (see tho.h).
 * Stage 0 (encoding) and stage 1 (k-mer self-hit collection).
 * Restates src/seq.c:15-32,77-95 and src/tandem_hit.c:37-56,97-157,171-237.
 */
#include <stdlib.h>
#include <string.h>
#include "tho.h"

/* src/seq.c:15-32: A/a 0, C/c 1, G/g 2, T/t 3, raw bytes 0..3 map to themselves,
 * '-' -> 5, everything else -> 4.  Bytes >= 0x80 index the reference's table with a
 * negative int (UB there); treated as 4 here. */
static uint8_t code_of(unsigned char c)
{
    switch (c) {
    case 0: case 'A': case 'a': return 0;
    case 1: case 'C': case 'c': return 1;
    case 2: case 'G': case 'g': return 2;
    case 3: case 'T': case 't': return 3;
    case '-': return 5;
    default: return 4;
    }
}

void tho_encode(const char *seq, int len, uint8_t *bseq)
{
    for (int i = 0; i < len; ++i) bseq[i] = code_of((unsigned char)seq[i]);
}

/* src/seq.c:34-51,88-95: complement table maps A<->T, C<->G, everything else (incl '-') to N */
char *tho_rc_seq(const char *seq, int len)
{
    char *rc = (char *)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i) {
        uint8_t c = code_of((unsigned char)seq[i]);
        rc[len - 1 - i] = c < 4 ? "TGCA"[c] : 'N';
    }
    rc[len] = 0;
    return rc;
}

/* src/tandem_hit.c:37-56 direct_hash(): rolling 2-bit key, any code >= 4 resets the run,
 * entry = key<<32 | index of the k-mer's last base.  With -H a homopolymer run contributes
 * one symbol and pos is the last base of the run. */
static int direct_entries(const uint8_t *b, int len, int k, int hpc, uint64_t *h)
{
    uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * k)) - 1), key = 0;
    int run = 0, n = 0;
    for (int pos = 0; pos < len; ++pos) {
        int c = b[pos];
        if (c >= 4) { key = 0; run = 0; continue; }
        if (hpc) while (pos + 1 < len && b[pos + 1] == c) ++pos;
        key = (key << 2) | (uint32_t)c;
        if (++run >= k) {
            key &= mask;
            h[n++] = ((uint64_t)key << 32) | (uint32_t)pos;
        }
    }
    return n;
}

/* src/tandem_hit.c:97-157 minimizer_hash() (-w > 1): a sliding window over the last w k-mer slots
 * (one slot per loop trip: a trip consumes one base, or one whole homopolymer run with -H).  The
 * window's selected word is its smallest RAW key, the newest slot winning ties; a word is written
 * when it stops being the selection (displaced by a key <= it, or shifted out of the window),
 * together with every other slot of the window that carries the same key, and the last selection is
 * written at the end of the read.  The reference's gating is kept exactly:
 *   - a slot is empty unless at least k valid symbols precede the trip's end and (with -H) the
 *     k runs span fewer than 256 bases (:121-123); any code >= 4 empties the slot and restarts the
 *     count but does NOT clear the window (:124);
 *   - the "same key elsewhere" sweep of the first full window happens when count == w+k-1 (:127-132);
 *   - a displaced selection is written only if count >= w+k (:135), a shifted-out one and the
 *     sweep after re-selection only if count >= w+k-1 (:138,143).
 * Restrictions: 2 <= w <= 256 (the reference's window array has 256 slots and is not checked). */
typedef struct { uint32_t key, pos; } slot_t;
#define SLOT_NONE 0xffffffffu

static int minimizer_entries(const uint8_t *b, int len, int k, int w, int hpc, uint64_t *h)
{
    const uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * k)) - 1);
    slot_t win[256], sel = { SLOT_NONE, SLOT_NONE };
    int runs[32], r_head = 0, r_cnt = 0;    /* lengths of the last <= k runs (-H) */
    int span = 0, count = 0, at = 0, sel_at = 0, n = 0;
    uint32_t key = 0;
#define PUT(s) (h[n++] = ((uint64_t)(s).key << 32) | (s).pos)
#define SWEEP(lo, hi) for (int j_ = (lo); j_ < (hi); ++j_) \
        if (win[j_].key == sel.key && win[j_].pos != sel.pos) PUT(win[j_])
    for (int i = 0; i < len; ++i) {
        const int c = b[i];
        slot_t cur = { SLOT_NONE, SLOT_NONE };
        if (c < 4) {
            if (hpc) {
                int run = 1;
                while (i + run < len && b[i + run] == c) ++run;
                i += run - 1;
                runs[(r_head + r_cnt++) & 31] = run; span += run;
                if (r_cnt > k) { span -= runs[r_head]; r_head = (r_head + 1) & 31; --r_cnt; }
            } else span = count + 1 < k ? count + 1 : k;
            key = ((key << 2) | (uint32_t)c) & mask;
            ++count;
            if (count >= k && span < 256) { cur.key = key; cur.pos = (uint32_t)i; }
        } else { count = 0; r_head = r_cnt = 0; span = 0; key = 0; }
        win[at] = cur;
        if (count == w + k - 1 && sel.key != SLOT_NONE) { SWEEP(at + 1, w); SWEEP(0, at); }
        if (cur.key <= sel.key) {
            if (count >= w + k && sel.key != SLOT_NONE) PUT(sel);
            sel = cur; sel_at = at;
        } else if (at == sel_at) {
            const int open = count >= w + k - 1;
            if (open && sel.key != SLOT_NONE) PUT(sel);
            sel.key = SLOT_NONE;
            for (int j = at + 1; j < w; ++j) if (sel.key >= win[j].key) { sel = win[j]; sel_at = j; }
            for (int j = 0; j <= at; ++j) if (sel.key >= win[j].key) { sel = win[j]; sel_at = j; }
            if (open && sel.key != SLOT_NONE) { SWEEP(at + 1, w); SWEEP(0, at + 1); }
        }
        if (++at == w) at = 0;
    }
    if (sel.key != SLOT_NONE) PUT(sel);
#undef PUT
#undef SWEEP
    return n;
}

int tho_kmer_entries(const uint8_t *bseq, int len, const tho_para_t *p, uint64_t *h)
{
    if (p->w > 256) return -1;
    if (p->w > 1) return minimizer_entries(bseq, len, p->k, p->w, p->hpc, h);
    return direct_entries(bseq, len, p->k, p->hpc, h);
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* src/tandem_hit.c:171-225 collect_hash_hit(): after an ascending 64-bit sort (key major, pos
 * minor; src/ksort.h:101-151 -- any correct sort gives the same order because words are
 * unique or identical), member j>=1 of an equal-key group walks back j-1, j-2, ... to the
 * first member at distance >= min_p; if the walk exhausts, p is the distance to the
 * group's first member.  Kept iff min_p <= p <= max_p.  hit = end<<32 | period, then the hit
 * list is sorted ascending (by end, then period). */
int tho_collect_hits(const uint8_t *bseq, int len, const tho_para_t *p, uint64_t **hits)
{
    *hits = NULL;
    if (len - p->w <= 0) return 0;
    uint64_t *h = (uint64_t *)malloc((size_t)(len + 1) * sizeof(uint64_t));
    int hn = tho_kmer_entries(bseq, len, p, h);
    if (hn <= 0) { free(h); return 0; }
    qsort(h, hn, sizeof(uint64_t), cmp_u64);
    uint32_t min_p = (uint32_t)p->min_p, max_p = (uint32_t)p->max_p; /* src/tandem_hit.c:235 */
    uint64_t *out = (uint64_t *)malloc((size_t)hn * sizeof(uint64_t));
    int n = 0;
    for (int g0 = 0; g0 < hn;) {
        int g1 = g0 + 1;
        while (g1 < hn && (h[g1] >> 32) == (h[g0] >> 32)) ++g1;
        for (int j = g0 + 1; j < g1; ++j) {
            uint32_t per = 0;
            for (int q = j - 1; q >= g0; --q) {
                per = (uint32_t)(h[j] - h[q]);
                if (per >= min_p) break;
            }
            if (per >= min_p && per <= max_p)
                out[n++] = ((h[j] & 0xffffffffu) << 32) | per;
        }
        g0 = g1;
    }
    free(h);
    qsort(out, n, sizeof(uint64_t), cmp_u64);
    *hits = out;
    return n;
}
