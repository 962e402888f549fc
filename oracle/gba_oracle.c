/*
 * gba_oracle.c -- CPU restatement of BLASR's guided banded alignment (refine stage).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP kernels in
 * blasr_amd/csrc/.  Nothing in the product path may link, import or call it;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED.  The arithmetic this file restates lives in the git
 * submodule `libcpp` (PacificBiosciences/blasr_libcpp, pinned commit unknown:
 * /root/reference/.gitmodules:1-3 carries only path+URL and /root/reference/libcpp
 * is empty).  No golden vector of the reference's own tests is present either
 * (all ctest inputs are on PacBio-internal mounts, /root/reference/ctest/setup.sh:3-6).
 * The functions below therefore follow
 *   - the call-site contract of RefineAlignment
 *     (/root/reference/iblasr/BlasrUtilsImpl.hpp:620-774),
 *   - the block coordinate convention (/root/reference/iblasr/BlasrAlignImpl.hpp:951-964),
 *   - the scoring convention "lower is better, match -5 / mismatch 6"
 *     (/root/reference/iblasr/RegisterBlasrOptions.h:350-360), and
 *   - the published algorithm of blasr_libcpp's GuidedAlign / AffineGuidedAlign /
 *     KBandAlign (alignment/algorithms/alignment/GuidedAlign*.hpp, KBandAlign.hpp):
 *     guide rows {q,t,tPre,tPost}, a "GetBufferIndex"-style validity test for the
 *     three predecessors, min-cost recurrence, arrow matrix, traceback from the
 *     global end cell, ArrowPathToAlignment block merging.
 * Every choice that cannot be verified here (tie-break order, band widening
 * rule, affine state machine) is a parameter of gbo_params.
 *
 * Coordinates used in this file are the reference's: guide row g = 0 is the
 * boundary row (query index qStart-1), guide row g >= 1 is query base
 * qStart+g-1; a column is a target index t in [tStart-1, tEnd-1], tStart-1
 * being the boundary column.
 *
 * Build: gcc -O3 -march=x86-64-v3 -shared -fPIC -pthread (see oracle/Makefile).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define GBO_INF 0x3fffffff

enum { GBO_OK = 0, GBO_EMPTY_GUIDE = 1, GBO_BAD_GUIDE = 2, GBO_UNREACHABLE = 3, GBO_NOSPACE = 4,
       GBO_END_OUT_OF_BAND = 5, GBO_NO_ALIGNMENT = 9 };

/* End rule of the guided DP.  GLOBAL: from the region's first cell to its last (RefineAlignment's calls,
 * /root/reference/iblasr/BlasrUtilsImpl.hpp:707-730).  LOCAL (the GuidedAlign(..., Local, ...) call of
 * AlignSubreadToAlignmentTarget, /root/reference/iblasr/BlasrAlignImpl.hpp:1418-1421): the alignment may begin
 * in any cell of the band at no cost and ends in the cheapest cell of the band -- Smith-Waterman's rule in
 * this file's lower-is-better scores.  What libcpp's GuidedAlign does with Local is not in the reference
 * tree: UNPINNED, stated here as: a cell takes the free start only when it is strictly cheaper than every
 * move into it; the end is the first cheapest cell in row-major order; an alignment needs a negative score. */
enum { GBO_ALIGN_GLOBAL = 0, GBO_ALIGN_LOCAL = 1 };

enum { GBO_MODE_LINEAR = 0, GBO_MODE_AFFINE = 1, GBO_MODE_KBAND = 2 };

/* arrows */
enum { A_NONE = 0, A_DIAG = 1, A_LEFT = 2, A_UP = 3, A_DELCLOSE = 4, A_INSCLOSE = 5, A_START = 6 };
enum { G_NONE = 0, G_OPEN = 1, G_EXTEND = 2 };

typedef struct { uint32_t q_pos, t_pos, length; } gbo_block;

typedef struct {
    int32_t sub[5][5];      /* [query code][target code], codes A,C,G,T,N = 0..4 */
    int32_t ins, del;       /* DP: ins = query base against a gap (Up), del = target base against a gap (Left) */
    int32_t affine_open, affine_extend;
    int32_t stats_ins, stats_del;          /* rescoring: distScoreFn2(indel, indel), BlasrUtilsImpl.hpp:633-634 */
    int32_t stats_affine_open, stats_affine_extend;
    int32_t band, max_band;
    int32_t mode;
    int32_t tie_break;      /* 0: Diagonal, Left, Up   1: Diagonal, Up, Left */
    int32_t kband_k;        /* GBO_MODE_KBAND only */
    int32_t stats_affine;   /* nonzero: rescoring charges gap runs open + L*extend */
    int32_t align_type;     /* GBO_ALIGN_*; LOCAL with linear gaps and guide blocks only */
} gbo_params;

typedef struct {
    int32_t status;
    int32_t dp_score;       /* optimum of the banded DP with (sub, ins, del [, affine]) */
    int32_t score;          /* ComputeAlignmentStats-style rescoring of the path */
    uint32_t q_pos, t_pos;  /* start of the DP region inside q / t (first guide block) */
    uint32_t n_blk;
    uint32_t n_match, n_mismatch, n_ins, n_del;
    uint32_t rows, cols;    /* Q, T of the DP region */
    int64_t ncells;         /* sum over guide rows of tPre+tPost+1 (unclamped) */
} gbo_result;

typedef struct { int32_t q, t, tPre, tPost; int64_t off; } guide_row;

/* ASCII -> {A,C,G,T,N}; lower case maps like upper case, everything else is N. */
int gbo_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* Per-thread work space: the score / arrow matrices, guide rows and path of the pair in hand,
 * grown to the largest pair seen and reused (the reference's per-thread MappingBuffers,
 * /root/reference/iblasr/MappingBuffers.hpp:20-49, Blasr.cpp:771). */
typedef struct { void *p; size_t cap; } gbo_buf;
static _Thread_local gbo_buf tls_buf[8];
static void *scratch(int which, size_t bytes)
{
    gbo_buf *b = &tls_buf[which];
    if (bytes > b->cap) {
        free(b->p);
        b->cap = bytes + bytes / 4 + 4096;
        b->p = malloc(b->cap);
        if (!b->p) { b->cap = 0; return NULL; }
    }
    return b->p;
}
enum { BUF_GUIDE = 0, BUF_S, BUF_A, BUF_SI, BUF_SD, BUF_AI, BUF_AD, BUF_PATH };

static int imin(int a, int b) { return a < b ? a : b; }
static int iabs(int a) { return a < 0 ? -a : a; }

/* Guide validation shared by every mode: blocks non-empty, each of positive
 * length, strictly ordered and non-overlapping in both sequences, inside the
 * sequences (RefineAlignment slices the sequences to end at the last block,
 * BlasrUtilsImpl.hpp:693-703). */
static int check_blocks(const gbo_block *b, int nb, uint32_t q_len, uint32_t t_len)
{
    if (nb <= 0) return GBO_EMPTY_GUIDE;
    for (int i = 0; i < nb; i++) {
        if (b[i].length == 0) return GBO_BAD_GUIDE;
        if ((uint64_t)b[i].q_pos + b[i].length > q_len) return GBO_BAD_GUIDE;
        if ((uint64_t)b[i].t_pos + b[i].length > t_len) return GBO_BAD_GUIDE;
        if (i > 0) {
            if (b[i].q_pos < b[i - 1].q_pos + b[i - 1].length) return GBO_BAD_GUIDE;
            if (b[i].t_pos < b[i - 1].t_pos + b[i - 1].length) return GBO_BAD_GUIDE;
        }
    }
    return GBO_OK;
}

static void add_row(guide_row *g, int gi, int q, int t, int w, int back)
{
    const guide_row *p = &g[gi - 1];
    int prevLo = p->t - p->tPre;
    g[gi].q = q;
    g[gi].t = t;
    g[gi].tPost = w;
    /* the band never reaches further back than the previous row's first column; `back` is
     * how far back it may reach at most (w, except on the first row of a block) */
    g[gi].tPre = imin(back, t - prevLo);
}

/* AlignmentToGuide: one row per query base from the first block's start to the
 * last block's end, plus the boundary row.  Inside a block the centre advances
 * one target base per row with half-width `band` (the block's first row reaches
 * back to the previous row's first column); between two blocks the common
 * part of the gap is walked diagonally and any remaining query bases repeat the
 * last target column, with half-width band + |drift| (capped at max_band). */
static guide_row *build_guide(const gbo_block *b, int nb, const gbo_params *P, int *n_rows)
{
    int qStart = (int)b[0].q_pos, tStart = (int)b[0].t_pos;
    int qEnd = (int)(b[nb - 1].q_pos + b[nb - 1].length);
    int Q = qEnd - qStart;
    int B = imin(P->band, P->max_band);
    guide_row *g = (guide_row *)scratch(BUF_GUIDE, sizeof(guide_row) * (size_t)(Q + 1));
    if (!g) return NULL;
    g[0].q = qStart - 1; g[0].t = tStart - 1; g[0].tPre = 0; g[0].tPost = B;
    int gi = 1;
    for (int i = 0; i < nb; i++) {
        /* the first row of a block reaches back to the previous row's first column (capped at
         * max_band), so that a target-only gap, which adds no rows of its own, stays connected */
        for (uint32_t p = 0; p < b[i].length; p++)
            add_row(g, gi++, (int)(b[i].q_pos + p), (int)(b[i].t_pos + p), B, p == 0 ? P->max_band : B);
        if (i + 1 < nb) {
            int q = (int)(b[i].q_pos + b[i].length), t = (int)(b[i].t_pos + b[i].length);
            int qGap = (int)b[i + 1].q_pos - q, tGap = (int)b[i + 1].t_pos - t;
            int drift = tGap - qGap;
            int w = imin(P->max_band, P->band + iabs(drift));
            int diag = imin(qGap, tGap);
            for (int x = 0; x < diag; x++) add_row(g, gi++, q++, t++, w, w);
            while (q < (int)b[i + 1].q_pos) add_row(g, gi++, q++, t - 1, w, w);
        }
    }
    *n_rows = gi;   /* == Q + 1 */
    return g;
}

/* KBandAlign geometry: a straight band of half-width k around the main
 * diagonal of the whole q x t rectangle (Global). */
static guide_row *build_kband(uint32_t q_len, uint32_t t_len, int k, int *n_rows)
{
    int Q = (int)q_len;
    guide_row *g = (guide_row *)scratch(BUF_GUIDE, sizeof(guide_row) * (size_t)(Q + 1));
    if (!g) return NULL;
    (void)t_len;
    for (int r = 0; r <= Q; r++) {
        g[r].q = r - 1; g[r].t = r - 1;
        g[r].tPost = k;
        g[r].tPre = imin(k, r);     /* column -1 is the boundary column */
    }
    *n_rows = Q + 1;
    return g;
}

static void store_offsets(guide_row *g, int n, int64_t *ncells)
{
    int64_t cur = 0;
    for (int r = 0; r < n; r++) { g[r].off = cur + g[r].tPre; cur += g[r].tPre + g[r].tPost + 1; }
    *ncells = cur;
}

/* GetBufferIndex: index of cell (guide row r, target column t) or -1. */
static inline int64_t cell_index(const guide_row *g, int n, int r, int t)
{
    if (r < 0 || r >= n) return -1;
    if (t <= g[r].t) { if (g[r].t - t <= g[r].tPre) return g[r].off - (g[r].t - t); return -1; }
    if (t - g[r].t <= g[r].tPost) return g[r].off + (t - g[r].t);
    return -1;
}

/* Guide rows in the clamped, 0-based DP coordinates the device uses:
 * row r covers columns j in [lo[r], hi[r]], j = t - tStart + 1 in [0, T]. */
int gbo_guide_rows(const gbo_block *b, int nb, uint32_t q_len, uint32_t t_len, const gbo_params *P,
                   int32_t *lo, int32_t *hi, int cap_rows, int32_t *rows_out, int64_t *ncells)
{
    guide_row *g; int n;
    int tStart, T;
    if (P->mode == GBO_MODE_KBAND) {
        g = build_kband(q_len, t_len, P->kband_k, &n); tStart = 0; T = (int)t_len;
    } else {
        int st = check_blocks(b, nb, q_len, t_len);
        if (st) return st;
        g = build_guide(b, nb, P, &n);
        tStart = (int)b[0].t_pos; T = (int)(b[nb - 1].t_pos + b[nb - 1].length) - tStart;
    }
    if (!g) return GBO_NOSPACE;
    store_offsets(g, n, ncells);
    *rows_out = n;
    if (n > cap_rows) return GBO_NOSPACE;
    for (int r = 0; r < n; r++) {
        int l = g[r].t - g[r].tPre - tStart + 1, h = g[r].t + g[r].tPost - tStart + 1;
        lo[r] = l < 0 ? 0 : l;
        hi[r] = h > T ? T : h;
    }
    return GBO_OK;
}

/* pick the arrow of the minimum under the configured priority */
static void pick3(int m, int l, int u, int tie, int *best, uint8_t *arrow)
{
    int mn = imin(m, imin(l, u));
    *best = mn;
    if (mn >= GBO_INF) { *best = GBO_INF; *arrow = A_NONE; return; }
    if (mn == m) { *arrow = A_DIAG; return; }
    if (tie == 0) *arrow = (mn == l) ? A_LEFT : A_UP;
    else          *arrow = (mn == u) ? A_UP : A_LEFT;
}

static int add_inf(int s, int c) { return s >= GBO_INF ? GBO_INF : s + c; }

/*
 * gbo_align: guide (or k-band) construction, DP fill, traceback, block merging
 * and statistics for one pair.  q/t are ASCII; blocks index into them.
 * blocks_out receives up to blk_cap blocks relative to (res->q_pos, res->t_pos).
 */
static int clamp_to(int v, int lim) { return v < -lim ? -lim : v > lim ? lim : v; }

/* prof (may be NULL): the costs of a quality-value scorer (IDSScoreFunction: GuidedAlign(..., idsScoreFn, ...),
 * /root/reference/iblasr/BlasrUtilsImpl.hpp:705-717) as 11 int16 per query base, indexed like q: Match against
 * target code A, C, G, T, N; Insertion of the base; Deletion of a target base of code A, C, G, T, N behind it.
 * Values are taken within +-127 (+-63 in the affine model, AffineGuidedAlign(..., idsScoreFn, ...), :707-710, where gap
 * open and extend stay P's constants); row 0 (before the first query base) deletes at P->del. */
int gbo_align_qv(const uint8_t *q, uint32_t q_len, const uint8_t *t, uint32_t t_len,
                 const gbo_block *b, int nb, const gbo_params *P, const int16_t *prof,
                 gbo_block *blocks_out, uint32_t blk_cap, gbo_result *res)
{
    memset(res, 0, sizeof(*res));
    guide_row *g; int n;
    int qStart, tStart, qEnd, tEnd;
    if (P->mode == GBO_MODE_KBAND) {
        qStart = 0; tStart = 0; qEnd = (int)q_len; tEnd = (int)t_len;
        int dd = (int)t_len - (int)q_len;
        if (iabs(dd) > P->kband_k) { res->status = GBO_END_OUT_OF_BAND; return res->status; }
        g = build_kband(q_len, t_len, P->kband_k, &n);
    } else {
        int st = check_blocks(b, nb, q_len, t_len);
        if (st) { res->status = st; return st; }
        qStart = (int)b[0].q_pos; tStart = (int)b[0].t_pos;
        qEnd = (int)(b[nb - 1].q_pos + b[nb - 1].length);
        tEnd = (int)(b[nb - 1].t_pos + b[nb - 1].length);
        g = build_guide(b, nb, P, &n);
    }
    if (!g) { res->status = GBO_NOSPACE; return res->status; }
    int64_t ncells;
    store_offsets(g, n, &ncells);
    res->ncells = ncells; res->q_pos = (uint32_t)qStart; res->t_pos = (uint32_t)tStart;
    res->rows = (uint32_t)(qEnd - qStart); res->cols = (uint32_t)(tEnd - tStart);

    const int affine = (P->mode == GBO_MODE_AFFINE);
    const int local = (P->align_type == GBO_ALIGN_LOCAL);
    if (local && P->mode != GBO_MODE_LINEAR) { res->status = GBO_BAD_GUIDE; return res->status; }
    int best_r = -1, best_t = 0, best_s = 0;        /* LOCAL: the cheapest cell so far */
    int32_t *S = (int32_t *)scratch(BUF_S, sizeof(int32_t) * (size_t)ncells);
    uint8_t *A = (uint8_t *)scratch(BUF_A, (size_t)ncells);
    int32_t *SI = NULL, *SD = NULL; uint8_t *AI = NULL, *AD = NULL;
    if (affine) {
        SI = (int32_t *)scratch(BUF_SI, sizeof(int32_t) * (size_t)ncells);
        SD = (int32_t *)scratch(BUF_SD, sizeof(int32_t) * (size_t)ncells);
        AI = (uint8_t *)scratch(BUF_AI, (size_t)ncells);
        AD = (uint8_t *)scratch(BUF_AD, (size_t)ncells);
    }
    if (!S || !A || (affine && (!SI || !SD || !AI || !AD))) { res->status = GBO_NOSPACE; return res->status; }
    memset(A, 0, (size_t)ncells);
    if (affine) { memset(AI, 0, (size_t)ncells); memset(AD, 0, (size_t)ncells); }
    for (int64_t i = 0; i < ncells; i++) { S[i] = GBO_INF; if (affine) { SI[i] = GBO_INF; SD[i] = GBO_INF; } }

    const int oe = P->affine_open + P->affine_extend, ex = P->affine_extend;
    for (int r = 0; r < n; r++) {
        int qi = qStart + r - 1;                    /* query base of this row (r >= 1) */
        for (int tt = g[r].t - g[r].tPre; tt <= g[r].t + g[r].tPost; tt++) {
            if (tt < tStart - 1) continue;
            if (tt > tEnd - 1) break;               /* past the end of the target slice */
            int64_t cur = cell_index(g, n, r, tt);
            if (r == 0 && tt == tStart - 1) { S[cur] = 0; A[cur] = local ? A_START : A_NONE; continue; }
            int64_t mi = (r > 0 && tt >= tStart) ? cell_index(g, n, r - 1, tt - 1) : -1;
            int64_t ui = (r > 0) ? cell_index(g, n, r - 1, tt) : -1;
            int64_t li = (tt >= tStart) ? cell_index(g, n, r, tt - 1) : -1;
            if (tt - 1 < tStart - 1) li = -1;
            int m = GBO_INF, u = GBO_INF, l = GBO_INF;
            const int16_t *pr = (prof && r > 0) ? prof + (size_t)qi * 11 : NULL;
            const int lim = affine ? 63 : 127;
            if (mi >= 0) m = add_inf(S[mi], pr ? clamp_to(pr[gbo_code(t[tt])], lim) : P->sub[gbo_code(q[qi])][gbo_code(t[tt])]);
            if (ui >= 0) u = add_inf(S[ui], pr ? clamp_to(pr[5], lim) : P->ins);
            if (li >= 0) l = add_inf(S[li], pr ? clamp_to(pr[6 + gbo_code(t[tt])], lim) : P->del);
            int best; uint8_t arrow;
            pick3(m, l, u, P->tie_break, &best, &arrow);
            if (affine) {
                /* insertion-gap state: vertical run, opened from the match state */
                if (ui >= 0) {
                    int o = add_inf(S[ui], oe), e = add_inf(SI[ui], ex);
                    if (o <= e) { SI[cur] = o; AI[cur] = (o >= GBO_INF) ? G_NONE : G_OPEN; }
                    else        { SI[cur] = e; AI[cur] = G_EXTEND; }
                }
                /* deletion-gap state: horizontal run */
                if (li >= 0) {
                    int o = add_inf(S[li], oe), e = add_inf(SD[li], ex);
                    if (o <= e) { SD[cur] = o; AD[cur] = (o >= GBO_INF) ? G_NONE : G_OPEN; }
                    else        { SD[cur] = e; AD[cur] = G_EXTEND; }
                }
                /* closing a gap state into the match state ranks after the three linear moves */
                if (SD[cur] < best) { best = SD[cur]; arrow = A_DELCLOSE; }
                if (SI[cur] < best) { best = SI[cur]; arrow = A_INSCLOSE; }
            }
            if (local) {
                if (0 < best) { best = 0; arrow = A_START; }
                if (best < best_s) { best_s = best; best_r = r; best_t = tt; }
            }
            S[cur] = best; A[cur] = arrow;
        }
    }

    /* traceback from the global end cell (last guide row, last target base) */
    int r = n - 1, tt = tEnd - 1;
    if (local) { r = best_r; tt = best_t; }
    int64_t endi = local && best_r < 0 ? -1 : cell_index(g, n, r, tt);
    int status = GBO_OK;
    if (local && best_r < 0) status = GBO_NO_ALIGNMENT;
    else if (endi < 0 || S[endi] >= GBO_INF) status = GBO_UNREACHABLE;
    uint32_t nblk = 0, nM = 0, nX = 0, nI = 0, nD = 0;
    int64_t score2 = 0;
    if (status == GBO_OK) {
        res->dp_score = S[endi];
        /* path is produced end->start; ops: 0 diag, 1 ins(up), 2 del(left) */
        size_t cap = (size_t)(qEnd - qStart) + (size_t)(tEnd - tStart) + 4, np = 0;
        uint8_t *path = (uint8_t *)scratch(BUF_PATH, cap);
        int state = 0;      /* 0 match, 1 ins-gap, 2 del-gap */
        while (!(r == 0 && tt == tStart - 1)) {
            int64_t ci = cell_index(g, n, r, tt);
            if (ci < 0 || np >= cap) { status = GBO_UNREACHABLE; break; }
            if (local && A[ci] == A_START) break;       /* the alignment begins behind this cell */
            if (state == 0) {
                uint8_t a = A[ci];
                if (a == A_DIAG) { path[np++] = 0; r--; tt--; }
                else if (a == A_UP) { path[np++] = 1; r--; }
                else if (a == A_LEFT) { path[np++] = 2; tt--; }
                else if (a == A_DELCLOSE) state = 2;
                else if (a == A_INSCLOSE) state = 1;
                else { status = GBO_UNREACHABLE; break; }
            } else if (state == 1) {
                uint8_t a = AI[ci];
                if (a == G_NONE) { status = GBO_UNREACHABLE; break; }
                path[np++] = 1; r--; if (a == G_OPEN) state = 0;
            } else {
                uint8_t a = AD[ci];
                if (a == G_NONE) { status = GBO_UNREACHABLE; break; }
                path[np++] = 2; tt--; if (a == G_OPEN) state = 0;
            }
        }
        if (status == GBO_OK) {
            /* ArrowPathToAlignment: merge consecutive diagonals into blocks; statistics
             * and rescoring as ComputeAlignmentStats (BlasrUtilsImpl.hpp:733-736). */
            /* (r, tt) is the cell the path begins behind: r query bases and tt - (tStart - 1) target bases of the
             * region lie in front of the alignment (both 0 for a global alignment) */
            uint32_t qp = (uint32_t)r, tp = (uint32_t)(tt - (tStart - 1)); int inblk = 0; int prev_op = 0;
            for (size_t k = np; k-- > 0;) {
                uint8_t op = path[k];
                if (op == 0) {
                    int qc = gbo_code(q[qStart + qp]), tc = gbo_code(t[tStart + tp]);
                    if (qc == tc) nM++; else nX++;
                    score2 += P->sub[qc][tc];
                    if (!inblk) {
                        if (nblk < blk_cap) { blocks_out[nblk].q_pos = qp; blocks_out[nblk].t_pos = tp; blocks_out[nblk].length = 0; }
                        nblk++; inblk = 1;
                    }
                    if (nblk <= blk_cap) blocks_out[nblk - 1].length++;
                    qp++; tp++;
                } else {
                    inblk = 0;
                    if (op == 1) { nI++; qp++; } else { nD++; tp++; }
                    if (P->stats_affine) {
                        if (prev_op != op) score2 += P->stats_affine_open;
                        score2 += P->stats_affine_extend;
                    } else {
                        score2 += (op == 1) ? P->stats_ins : P->stats_del;
                    }
                }
                prev_op = op;
            }
            if (nblk > blk_cap) status = GBO_NOSPACE;
        }
    }
    res->status = status; res->n_blk = nblk; res->score = (int32_t)score2;
    res->n_match = nM; res->n_mismatch = nX; res->n_ins = nI; res->n_del = nD;
    return status;
}

int gbo_align(const uint8_t *q, uint32_t q_len, const uint8_t *t, uint32_t t_len,
              const gbo_block *b, int nb, const gbo_params *P,
              gbo_block *blocks_out, uint32_t blk_cap, gbo_result *res)
{
    return gbo_align_qv(q, q_len, t, t_len, b, nb, P, NULL, blocks_out, blk_cap, res);
}

/*
 * Unbanded global alignment (full Needleman-Wunsch, linear gaps) used by the
 * property test "banded optimum == full-DP optimum when the band covers the
 * optimal path".  O(Q*T) memory-light (two rows), score only.
 */
int gbo_full_dp_score(const uint8_t *q, uint32_t q_len, const uint8_t *t, uint32_t t_len,
                      const gbo_params *P, int32_t *score)
{
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * (t_len + 1));
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (t_len + 1));
    if (!prev || !cur) { free(prev); free(cur); return GBO_NOSPACE; }
    for (uint32_t j = 0; j <= t_len; j++) prev[j] = (int32_t)j * P->del;
    for (uint32_t i = 1; i <= q_len; i++) {
        cur[0] = (int32_t)i * P->ins;
        int qc = gbo_code(q[i - 1]);
        for (uint32_t j = 1; j <= t_len; j++) {
            int m = prev[j - 1] + P->sub[qc][gbo_code(t[j - 1])];
            int u = prev[j] + P->ins, l = cur[j - 1] + P->del;
            cur[j] = imin(m, imin(u, l));
        }
        int32_t *tmp = prev; prev = cur; cur = tmp;
    }
    *score = prev[t_len];
    free(prev); free(cur);
    return GBO_OK;
}

/* Unbanded local alignment (Smith-Waterman in lower-is-better scores, linear gaps), score only: the cheapest
 * cell of the whole matrix when every cell may also start an alignment at cost 0.  Property test: the banded
 * LOCAL optimum equals it when the band holds the optimal local path. */
int gbo_full_local_score(const uint8_t *q, uint32_t q_len, const uint8_t *t, uint32_t t_len,
                         const gbo_params *P, int32_t *score)
{
    int32_t *prev = (int32_t *)calloc(t_len + 1, sizeof(int32_t));
    int32_t *cur = (int32_t *)calloc(t_len + 1, sizeof(int32_t));
    if (!prev || !cur) { free(prev); free(cur); return GBO_NOSPACE; }
    int32_t best = 0;
    for (uint32_t i = 1; i <= q_len; i++) {
        cur[0] = 0;
        int qc = gbo_code(q[i - 1]);
        for (uint32_t j = 1; j <= t_len; j++) {
            int m = prev[j - 1] + P->sub[qc][gbo_code(t[j - 1])];
            int u = prev[j] + P->ins, l = cur[j - 1] + P->del;
            int v = imin(m, imin(u, l));
            cur[j] = v < 0 ? v : 0;
            if (cur[j] < best) best = cur[j];
        }
        int32_t *tmp = prev; prev = cur; cur = tmp;
    }
    *score = best;
    free(prev); free(cur);
    return GBO_OK;
}

/* Batch driver used by bench.py's cpu_baseline leg and by the parity tests:
 * pairs [first, first+count) of an SoA batch, single thread. */
int gbo_align_batch(const uint8_t *q_arena, const uint64_t *q_off, const uint32_t *q_len,
                    const uint8_t *t_arena, const uint64_t *t_off, const uint32_t *t_len,
                    const gbo_block *blocks, const uint64_t *blk_off, const uint32_t *n_blk,
                    const gbo_params *P, uint32_t first, uint32_t count,
                    gbo_result *results, gbo_block *blocks_out, const uint64_t *out_off, const uint32_t *out_cap)
{
    for (uint32_t p = first; p < first + count; p++)
        gbo_align(q_arena + q_off[p], q_len[p], t_arena + t_off[p], t_len[p],
                  blocks + blk_off[p], (int)n_blk[p], P,
                  blocks_out + out_off[p], out_cap[p], &results[p]);
    return GBO_OK;
}

/*
 * The batch on all host cores, the way the reference runs it: `threads` workers, each with its
 * own work space (the per-thread MappingBuffers of /root/reference/Blasr.cpp:771), pulling the
 * next pair from a shared counter (the reader semaphore of
 * /root/reference/iblasr/BlasrMiscsImpl.hpp:15-44 hands out reads the same way).  Stops handing
 * out pairs after `seconds` (<= 0: never).  Used by bench.py's cpu_baseline leg.
 */
typedef struct {
    const uint8_t *q_arena, *t_arena; const uint64_t *q_off, *t_off, *blk_off, *out_off;
    const uint32_t *q_len, *t_len, *n_blk, *out_cap;
    const gbo_block *blocks; const gbo_params *P;
    gbo_result *results; gbo_block *blocks_out;
    uint32_t first, count;
    volatile uint32_t next;
    double deadline;
    uint64_t cells, pairs;
    pthread_mutex_t mu;
} mt_job;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    uint64_t cells = 0, pairs = 0;
    for (;;) {
        if (j->deadline > 0 && now_s() >= j->deadline) break;
        const uint32_t k = __atomic_fetch_add(&j->next, 1u, __ATOMIC_RELAXED);
        if (k >= j->count) break;
        const uint32_t p = j->first + k;
        gbo_align(j->q_arena + j->q_off[p], j->q_len[p], j->t_arena + j->t_off[p], j->t_len[p],
                  j->blocks + j->blk_off[p], (int)j->n_blk[p], j->P,
                  j->blocks_out + j->out_off[p], j->out_cap[p], &j->results[p]);
        if (j->results[p].status == GBO_OK) cells += (uint64_t)j->results[p].ncells;
        pairs++;
    }
    for (int x = 0; x < 8; x++) { free(tls_buf[x].p); tls_buf[x].p = NULL; tls_buf[x].cap = 0; }
    pthread_mutex_lock(&j->mu);
    j->cells += cells; j->pairs += pairs;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

int gbo_align_batch_mt(const uint8_t *q_arena, const uint64_t *q_off, const uint32_t *q_len,
                       const uint8_t *t_arena, const uint64_t *t_off, const uint32_t *t_len,
                       const gbo_block *blocks, const uint64_t *blk_off, const uint32_t *n_blk,
                       const gbo_params *P, uint32_t first, uint32_t count,
                       gbo_result *results, gbo_block *blocks_out, const uint64_t *out_off, const uint32_t *out_cap,
                       int threads, double seconds, uint64_t *pairs_done, uint64_t *cells_done, double *elapsed)
{
    mt_job j;
    memset(&j, 0, sizeof(j));
    j.q_arena = q_arena; j.t_arena = t_arena; j.q_off = q_off; j.t_off = t_off; j.blk_off = blk_off; j.out_off = out_off;
    j.q_len = q_len; j.t_len = t_len; j.n_blk = n_blk; j.out_cap = out_cap; j.blocks = blocks; j.P = P;
    j.results = results; j.blocks_out = blocks_out; j.first = first; j.count = count;
    pthread_mutex_init(&j.mu, NULL);
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    if (!th) return GBO_NOSPACE;
    const double t0 = now_s();
    j.deadline = seconds > 0 ? t0 + seconds : 0;
    int started = 0;
    for (int i = 0; i < threads; i++) if (pthread_create(&th[i], NULL, mt_worker, &j) == 0) th[started++] = th[i];
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    if (started == 0) mt_worker(&j);
    *elapsed = now_s() - t0;
    *pairs_done = j.pairs; *cells_done = j.cells;
    free(th);
    pthread_mutex_destroy(&j.mu);
    return GBO_OK;
}
