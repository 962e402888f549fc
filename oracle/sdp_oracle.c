/*
 * sdp_oracle.c -- CPU restatement of the guide producer in front of the refine stage: exact
 * k-mer matches between a read and its reference window, chained by sparse dynamic programming
 * into the increasing list of gap-free blocks that becomes alignment->blocks, "THE GUIDE"
 * (/root/reference/iblasr/BlasrAlignImpl.hpp:980-987: SDPAlign(qAlignedSeq, tAlignedSeq,
 * distScoreFn, sdpTupleSize, params.sdpIns, params.sdpDel, params.indelRate*3, *alignment,
 * mappingBuffers, Local, ...); defaults sdpTupleSize 11, sdpIns 5, sdpDel 10,
 * /root/reference/iblasr/MappingParameters.h:220-222; the stand-alone form is
 * /root/reference/utils/SDPMatcher.cpp:139-148).
 *
 * TEST INFRASTRUCTURE ONLY: the checker of the HIP kernels k_sdp_* in blasr_amd/csrc/.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED.  SDPAlign itself is libcpp code (alignment/algorithms/alignment/SDPAlign*.hpp,
 * sdp/SDPFragment, SDPSet, SDPColumn) and libcpp is absent from /root/reference; none of the
 * reference's tests holds an SDP input/output pair.  What is restated is the published scheme:
 *   1. every position of the target is listed under its k-mer, every k-mer of the query is looked
 *      up, a (query position, target position) hit is a fragment of length k and weight k;
 *   2. fragments are chained: a chain is a sequence of fragments increasing in both coordinates,
 *      its score the sum of the fragment weights (scaled by the match reward) minus a penalty of
 *      sdpIns / sdpDel per base of drift between the diagonals of consecutive fragments; the
 *      best chain ending anywhere (Local) is traced back;
 *   3. consecutive fragments of the chain on one diagonal are merged, and where two consecutive
 *      fragments overlap in either sequence the earlier one is shortened by the overlap.
 * Two choices make that exact and order-independent, so that a parallel implementation can be
 * compared bit for bit:
 *   - step 1 and the merging of step 3 are done first: the fragments are the MAXIMAL exact
 *     matches of length >= k (a run of overlapping k-mer hits on one diagonal is one fragment),
 *     which is what step 3 makes of any chain through them;
 *   - step 2 is an exact optimum (every earlier fragment is considered as predecessor, the
 *     overlap trimming of step 3 is priced into the link: an overlap of o bases costs o bases of
 *     weight), not libcpp's sweep over an active set of columns, whose tie-breaks are unknown.
 * Letters outside ACGT match nothing.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t k;              /* tuple size, 4..15 */
    int32_t ins, del;       /* penalty per base of drift: query bases without target (ins), target bases without query (del) */
    int32_t match;          /* reward per anchored base (the reference's match score is -5: 5 here) */
    uint32_t max_fragments; /* more maximal matches than this, or more than 64 x this many k-mer hits: status SDP_TOO_MANY */
} sdo_params;

typedef struct { uint32_t q_pos, t_pos, length; } sdo_block;

enum { SDP_OK = 0, SDP_EMPTY = 1, SDP_TOO_LONG = 7, SDP_TOO_MANY = 8 };

static inline int code_of(uint8_t c)
{
    switch (c & 0xdf) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

typedef struct { uint32_t x, y, len; } frag;

static int cmp_frag(const void *a, const void *b)
{
    const frag *f = (const frag *)a, *g = (const frag *)b;
    if (f->x != g->x) return f->x < g->x ? -1 : 1;
    return f->y < g->y ? -1 : f->y > g->y;
}

/* valid k-mers of s as (kmer << 32 | position), in position order; returns the count */
static uint32_t kmers(const uint8_t *s, uint32_t n, int k, uint64_t *out)
{
    uint32_t cnt = 0, good = 0;
    uint64_t w = 0;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    for (uint32_t i = 0; i < n; i++) {
        const int c = code_of(s[i]);
        if (c > 3) { good = 0; w = 0; continue; }
        w = ((w << 2) | (uint64_t)c) & mask;
        if (++good >= (uint32_t)k) out[cnt++] = w << 32 | (i + 1 - (uint32_t)k);
    }
    return cnt;
}

static inline int same_base(const uint8_t *q, const uint8_t *t, uint32_t x, uint32_t y)
{
    const int a = code_of(q[x]);
    return a < 4 && a == code_of(t[y]);
}

/*
 * Chains the maximal exact matches of q and t.  Blocks come out in increasing order, relative to
 * the starts of q and t.  Returns the status; *n_blk the number of blocks of the chain (all of
 * them counted even when blk_cap is smaller), *chain_score its score, *n_frag the number of
 * maximal matches.
 */
int sdo_chain(const uint8_t *q, uint32_t ql, const uint8_t *t, uint32_t tl, const sdo_params *P,
              sdo_block *out, uint32_t blk_cap, uint32_t *n_blk, int32_t *chain_score, uint32_t *n_frag)
{
    *n_blk = 0; *chain_score = 0; *n_frag = 0;
    const int k = P->k;
    if (ql > 65535u || tl > 65535u) return SDP_TOO_LONG;
    if (ql < (uint32_t)k || tl < (uint32_t)k) return SDP_EMPTY;
    uint64_t *tk = (uint64_t *)malloc(sizeof(uint64_t) * (tl + 1)), *qk = (uint64_t *)malloc(sizeof(uint64_t) * (ql + 1));
    const uint32_t nt = kmers(t, tl, k, tk), nq = kmers(q, ql, k, qk);
    qsort(tk, nt, sizeof(uint64_t), cmp_u64);
    uint32_t cap = 1024, nf = 0;
    frag *F = (frag *)malloc(sizeof(frag) * cap);
    int status = SDP_OK;
    uint64_t hits = 0;
    for (uint32_t i = 0; i < nq && status == SDP_OK; i++) {
        const uint64_t km = qk[i] >> 32;
        const uint32_t x = (uint32_t)qk[i];
        /* first target entry with this k-mer */
        uint32_t lo = 0, hi = nt;
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if ((tk[mid] >> 32) < km) lo = mid + 1; else hi = mid; }
        for (; lo < nt && (tk[lo] >> 32) == km; lo++) {
            const uint32_t y = (uint32_t)tk[lo];
            if (++hits > 64ull * P->max_fragments) { status = SDP_TOO_MANY; break; }
            if (x > 0 && y > 0 && same_base(q, t, x - 1, y - 1)) continue;      /* not the start of its run */
            uint32_t len = (uint32_t)k;
            while (x + len < ql && y + len < tl && same_base(q, t, x + len, y + len)) len++;
            if (nf >= P->max_fragments) { status = SDP_TOO_MANY; break; }
            if (nf == cap) { cap *= 2; F = (frag *)realloc(F, sizeof(frag) * cap); }
            F[nf].x = x; F[nf].y = y; F[nf].len = len; nf++;
        }
    }
    free(tk); free(qk);
    *n_frag = nf;
    if (status != SDP_OK || nf == 0) { free(F); return status != SDP_OK ? status : SDP_EMPTY; }
    qsort(F, nf, sizeof(frag), cmp_frag);
    int32_t *score = (int32_t *)malloc(sizeof(int32_t) * nf);
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * nf);
    int32_t best_all = 0; uint32_t end = 0;
    for (uint32_t f = 0; f < nf; f++) {
        int32_t best = 0, bp = -1;
        const int32_t df = (int32_t)F[f].y - (int32_t)F[f].x;
        for (uint32_t p = 0; p < f; p++) {
            const int32_t ox = (int32_t)(F[p].x + F[p].len) - (int32_t)F[f].x, oy = (int32_t)(F[p].y + F[p].len) - (int32_t)F[f].y;
            int32_t o = ox > oy ? ox : oy;
            if (o < 0) o = 0;
            if ((uint32_t)o >= F[p].len) continue;                  /* nothing of p would be left in front of f */
            const int32_t d = df - ((int32_t)F[p].y - (int32_t)F[p].x);
            const int32_t gap = d >= 0 ? d * P->del : -d * P->ins;
            const int32_t g = score[p] - gap - o * P->match;
            if (g > best) { best = g; bp = (int32_t)p; }            /* ties: the earliest predecessor */
        }
        score[f] = (int32_t)F[f].len * P->match + best;
        prev[f] = bp;
        if (score[f] > best_all) { best_all = score[f]; end = f; }  /* ties: the earliest end */
    }
    /* trace back, then emit forwards with the overlap taken off the earlier fragment */
    uint32_t n = 0;
    for (int32_t f = (int32_t)end; f >= 0; f = prev[f]) n++;
    uint32_t *chain = (uint32_t *)malloc(sizeof(uint32_t) * n);
    { uint32_t i = n; for (int32_t f = (int32_t)end; f >= 0; f = prev[f]) chain[--i] = (uint32_t)f; }
    for (uint32_t i = 0; i < n; i++) {
        const frag *a = &F[chain[i]];
        uint32_t len = a->len;
        if (i + 1 < n) {
            const frag *b = &F[chain[i + 1]];
            const int32_t ox = (int32_t)(a->x + a->len) - (int32_t)b->x, oy = (int32_t)(a->y + a->len) - (int32_t)b->y;
            int32_t o = ox > oy ? ox : oy;
            if (o > 0) len -= (uint32_t)o;
        }
        if (i < blk_cap) { out[i].q_pos = a->x; out[i].t_pos = a->y; out[i].length = len; }
    }
    *n_blk = n; *chain_score = best_all;
    free(chain); free(score); free(prev); free(F);
    return SDP_OK;
}

/* Pairs [first, first + count) of a batch; blocks of pair i go to out + out_off[i] (at most caps[i]). */
int sdo_chain_batch(const uint8_t *q_arena, const uint64_t *q_off, const uint32_t *q_len,
                    const uint8_t *t_arena, const uint64_t *t_off, const uint32_t *t_len,
                    const sdo_params *P, uint32_t first, uint32_t count,
                    int32_t *status, uint32_t *n_blk, int32_t *chain_score, uint32_t *n_frag,
                    sdo_block *out, const uint64_t *out_off, const uint32_t *caps)
{
    for (uint32_t i = first; i < first + count; i++)
        status[i] = sdo_chain(q_arena + q_off[i], q_len[i], t_arena + t_off[i], t_len[i], P,
                              out + out_off[i], caps[i], &n_blk[i], &chain_score[i], &n_frag[i]);
    return 0;
}
