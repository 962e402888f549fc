/*
 * kband_oracle.c -- CPU restatement of BLASR's fixed-band aligners: KBandAlign (Global / Fit, with a
 * matrix scorer or a per-base quality-value scorer), AffineKBandAlign (homopolymer-insertion and
 * insertion gap states) and ExtendAlignmentForward / ExtendAlignmentReverse (banded extension that
 * stops after a number of rows without a new minimum).
 *
 * TEST INFRASTRUCTURE ONLY, like gba_oracle.c: the checker of the kernels in
 * blasr_amd/csrc/gba_kband.inc.  Only tests/, __graft_entry__.smoke() and bench.py may use it.
 *
 * PARITY UNPINNED.  These routines are libcpp code (alignment/algorithms/alignment/KBandAlign.hpp,
 * AffineKBandAlign.hpp, ExtendAlign.hpp) and /root/reference/libcpp is empty; no test of the
 * reference holds an input/output pair for them.  The file follows the call sites:
 *   - KBandAlign(qSeq, tSeq, SMRTDistanceMatrix, ins, del, k, scoreMat, pathMat, alignment, scoreFn, alignType)
 *     /root/reference/iblasr/BlasrUtilsImpl.hpp:657-661 (Global, k = drift), and
 *     /root/reference/iblasr/BlasrAlignImpl.hpp:1278-1294 (quality-value scorers, Fit);
 *   - AffineKBandAlign(qSeq, tSeq, SMRTDistanceMatrix, hpInsOpen, hpInsExtend, insOpen, insExtend, del,
 *     k, three score/path matrix pairs, alignment, Global)  BlasrAlignImpl.hpp:1259-1267;
 *   - ExtendAlignmentForward(readSuffix, 0, genomeSuffix, 0, extendBandSize, scoreMat, pathMat, out,
 *     distScoreFn, 1, maxExtendDropoff)  BlasrAlignImpl.hpp:1086-1099, ExtendAlignmentReverse :1144-1155,
 *     and what the caller does with the result (a negative score extends the alignment, :1102-1120);
 * and states every choice the call sites do not settle as a field of kbo_params.
 *
 * The model.  Cells (i, j), 0 <= i <= Q, 0 <= j <= T, |j - i| <= k.  Row i >= 1 is query base i - 1,
 * column j >= 1 target base j - 1.  Costs come from the matrix and two constants, or from a profile
 * of 11 int16 per query base (5 substitution costs by target code, the insertion cost of the base,
 * 5 deletion costs by target code for a target base skipped right after this query base): what
 * T_ScoreFn::Match / Insertion / Deletion return for that base (IDSScoreFunction,
 * QualityValueScoreFunction).  Lower is better.
 *   GLOBAL : M(0,0) = 0, row 0 by deletions, column 0 by insertions, end cell (Q, T).
 *   FIT    : the whole query against a part of the target: row 0 costs nothing, the end cell is the
 *            cheapest cell of row Q (the leftmost on ties).
 *   EXTEND : like GLOBAL from (0,0); rows are filled in order, the best cell so far is kept, and
 *            the fill stops after the row that is the (max_drops + 1)-th in a row without a new
 *            strict minimum of its own; end cell = the best cell; (0,0) itself means "no extension".
 * Gap models: LINEAR (one state) and AFFINE_HP (match state with linear deletions, an insertion
 * state and a homopolymer-insertion state; the first base of a run costs the open value, each
 * further base the extend value; a homopolymer insertion is allowed where the inserted base equals
 * the target base to its left (hp_rule 0) or the query base before it (hp_rule 1)).
 * `reverse`: both sequences are read back to front (ExtendAlignmentReverse aligns from the last
 * bases backwards); blocks are reported in forward coordinates.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KBO_INF 0x3fffffff

enum { KBO_OK = 0, KBO_NOSPACE = 4, KBO_END_OUT_OF_BAND = 5, KBO_TOO_LONG = 7 };
enum { KBO_GLOBAL = 0, KBO_FIT = 1, KBO_EXTEND = 2 };
enum { KBO_LINEAR = 0, KBO_AFFINE_HP = 1 };
enum { A_NONE = 0, A_DIAG = 1, A_LEFT = 2, A_UP = 3, A_INSCLOSE = 4, A_HPCLOSE = 5 };

typedef struct { uint32_t q_pos, t_pos, length; } kbo_block;

typedef struct {
    int32_t sub[5][5];
    int32_t ins, del;
    int32_t align_type, gap_model;
    int32_t hp_ins_open, hp_ins_extend, ins_open, ins_extend;
    int32_t hp_rule;
    int32_t tie_break;          /* 0: Diagonal, Left, Up (Ins, HpIns)   1: Diagonal, Up (Ins, HpIns), Left */
    int32_t max_drops;
    int32_t stats_ins, stats_del;
    int32_t stats_affine, stats_affine_open, stats_affine_extend;
} kbo_params;

/* same layout as gbo_result */
typedef struct {
    int32_t status, dp_score, score;
    uint32_t q_pos, t_pos, n_blk, n_match, n_mismatch, n_ins, n_del, rows, cols;
    int64_t ncells;
} kbo_result;

extern int gbo_code(uint8_t c);

static int add_inf(int s, int c) { return s >= KBO_INF ? KBO_INF : s + c; }

int kbo_align(const uint8_t *q, uint32_t ql, const uint8_t *t, uint32_t tl, int32_t k, int32_t reverse,
              const kbo_params *P, const int16_t *prof, kbo_block *blocks_out, uint32_t blk_cap, kbo_result *res)
{
    memset(res, 0, sizeof *res);
    const int Q = (int)ql, T = (int)tl;
    res->rows = ql; res->cols = tl;
    if (k < 0) k = 0;
    if (P->align_type == KBO_GLOBAL && abs(T - Q) > k) { res->status = KBO_END_OUT_OF_BAND; return res->status; }
    if (P->align_type == KBO_FIT && Q - T > k) { res->status = KBO_END_OUT_OF_BAND; return res->status; }   /* no cell in the last row */
    const int BW = 2 * k + 1;
    const int aff = P->gap_model == KBO_AFFINE_HP;
    const size_t ncell = (size_t)(Q + 1) * BW;
    int32_t *M = (int32_t *)malloc(ncell * 4), *I = NULL, *H = NULL;
    uint8_t *AM = (uint8_t *)calloc(ncell, 1), *AI = NULL, *AH = NULL;
    if (aff) {
        I = (int32_t *)malloc(ncell * 4); H = (int32_t *)malloc(ncell * 4);
        AI = (uint8_t *)calloc(ncell, 1); AH = (uint8_t *)calloc(ncell, 1);
    }
    for (size_t x = 0; x < ncell; x++) { M[x] = KBO_INF; if (aff) { I[x] = KBO_INF; H[x] = KBO_INF; } }
#define IDX(i, j) ((size_t)(i) * BW + (size_t)((j) - (i) + k))
#define INB(i, j) ((i) >= 0 && (j) >= 0 && (i) <= Q && (j) <= T && (j) - (i) <= k && (i) - (j) <= k)
#define QI(i) (reverse ? Q - (i) : (i) - 1)          /* index of the query base of row i */
#define TI(j) (reverse ? T - (j) : (j) - 1)
    int64_t ncells = 0;
    int bi = 0, bj = 0, gmin = 0, drops = 0, last_row = Q;
    for (int i = 0; i <= Q; i++) {
        const int lo = i - k < 0 ? 0 : i - k, hi = i + k > T ? T : i + k;
        int rm = KBO_INF, rmj = lo;
        const int qc = i >= 1 ? gbo_code(q[QI(i)]) : 4;
        const int16_t *pr = (prof && i >= 1) ? prof + (size_t)QI(i) * 11 : NULL;
        const int insc = pr ? pr[5] : P->ins;
        for (int j = lo; j <= hi; j++) {
            ncells++;
            const int tc = j >= 1 ? gbo_code(t[TI(j)]) : 4;
            int best = KBO_INF; uint8_t arrow = A_NONE;
            if (i == 0 && j == 0) best = 0;
            else if (i == 0) {
                if (P->align_type == KBO_FIT) best = 0;
                else { best = add_inf(M[IDX(0, j - 1)], P->del); arrow = A_LEFT; }
            } else {
                const int subc = pr ? pr[tc] : P->sub[qc][tc];
                const int delc = pr ? pr[6 + tc] : P->del;
                const int d = (j >= 1 && INB(i - 1, j - 1)) ? add_inf(M[IDX(i - 1, j - 1)], subc) : KBO_INF;
                const int l = (j >= 1 && INB(i, j - 1)) ? add_inf(M[IDX(i, j - 1)], delc) : KBO_INF;
                const int upok = INB(i - 1, j);
                if (!aff) {
                    const int u = upok ? add_inf(M[IDX(i - 1, j)], insc) : KBO_INF;
                    if (d < best) { best = d; arrow = A_DIAG; }
                    if (P->tie_break == 0) {
                        if (l < best) { best = l; arrow = A_LEFT; }
                        if (u < best) { best = u; arrow = A_UP; }
                    } else {
                        if (u < best) { best = u; arrow = A_UP; }
                        if (l < best) { best = l; arrow = A_LEFT; }
                    }
                } else {
                    int ig = KBO_INF, hg = KBO_INF; uint8_t ia = 0, ha = 0;      /* 1 open, 2 extend */
                    if (upok) {
                        const int io = add_inf(M[IDX(i - 1, j)], P->ins_open), ie = add_inf(I[IDX(i - 1, j)], P->ins_extend);
                        if (io <= ie) { ig = io; ia = 1; } else { ig = ie; ia = 2; }
                        if (ig >= KBO_INF) ia = 0;
                        int hp;
                        if (P->hp_rule == 0) hp = j >= 1 && qc == tc && qc != 4;
                        else hp = i >= 2 && qc == gbo_code(q[QI(i - 1)]) && qc != 4;
                        if (hp) {
                            const int ho = add_inf(M[IDX(i - 1, j)], P->hp_ins_open), he = add_inf(H[IDX(i - 1, j)], P->hp_ins_extend);
                            if (ho <= he) { hg = ho; ha = 1; } else { hg = he; ha = 2; }
                            if (hg >= KBO_INF) ha = 0;
                        }
                    }
                    I[IDX(i, j)] = ig; AI[IDX(i, j)] = ia;
                    H[IDX(i, j)] = hg; AH[IDX(i, j)] = ha;
                    if (d < best) { best = d; arrow = A_DIAG; }
                    if (P->tie_break == 0 && l < best) { best = l; arrow = A_LEFT; }
                    if (ig < best) { best = ig; arrow = A_INSCLOSE; }
                    if (hg < best) { best = hg; arrow = A_HPCLOSE; }
                    if (P->tie_break != 0 && l < best) { best = l; arrow = A_LEFT; }
                }
            }
            M[IDX(i, j)] = best; AM[IDX(i, j)] = arrow;
            if (best < rm) { rm = best; rmj = j; }
        }
        if (P->align_type == KBO_EXTEND && i >= 1) {
            if (rm < gmin) { gmin = rm; bi = i; bj = rmj; drops = 0; }
            else if (++drops > P->max_drops) { last_row = i; break; }
        }
        if (P->align_type == KBO_FIT && i == Q) { bi = Q; bj = rmj; }
    }
    res->ncells = ncells;
    (void)last_row;
    int ei, ej;
    if (P->align_type == KBO_GLOBAL) { ei = Q; ej = T; }
    else { ei = bi; ej = bj; }
    res->dp_score = M[IDX(ei, ej)];

    /* traceback: ops 0 diag, 1 ins (query base against a gap), 2 del; positions are those of the cell
     * the move ENTERS, i.e. the op consumes query base i and / or target base j of the cell it leaves */
    const size_t cap = (size_t)Q + (size_t)T + 4;
    uint8_t *path = (uint8_t *)malloc(cap);
    size_t np = 0;
    int i = ei, j = ej, state = 0, status = KBO_OK;
    while (!(i == 0 && (j == 0 || P->align_type == KBO_FIT)) && np < cap) {
        const size_t ci = IDX(i, j);
        if (state == 0) {
            const uint8_t a = AM[ci];
            if (a == A_DIAG) { path[np++] = 0; i--; j--; }
            else if (a == A_UP) { path[np++] = 1; i--; }
            else if (a == A_LEFT) { path[np++] = 2; j--; }
            else if (a == A_INSCLOSE) state = 1;
            else if (a == A_HPCLOSE) state = 2;
            else break;
        } else if (state == 1) {
            const uint8_t a = AI[ci];
            path[np++] = 1; i--; if (a != 2) state = 0;
        } else {
            const uint8_t a = AH[ci];
            path[np++] = 1; i--; if (a != 2) state = 0;
        }
    }
    /* forward order of the path: for a reversed pair the traceback order IS the forward order */
    uint32_t nblk = 0, nM = 0, nX = 0, nI = 0, nD = 0;
    int64_t score2 = 0;
    {
        /* forward coordinates of the first path column */
        uint32_t qp, tp;
        if (!reverse) { qp = (uint32_t)i; tp = (uint32_t)j; }              /* where the traceback ended */
        else { qp = (uint32_t)(Q - ei); tp = (uint32_t)(T - ej); }         /* where it started */
        uint32_t q0 = 0, t0 = 0; int have0 = 0, inblk = 0, prev_op = 0;
        for (size_t x = 0; x < np; x++) {
            const uint8_t op = reverse ? path[x] : path[np - 1 - x];
            if (op == 0) {
                const int qc = gbo_code(q[qp]), tc = gbo_code(t[tp]);
                if (qc == tc) nM++; else nX++;
                score2 += P->sub[qc][tc];
                if (!have0) { q0 = qp; t0 = tp; have0 = 1; }
                if (!inblk) {
                    if (nblk < blk_cap) { blocks_out[nblk].q_pos = qp - q0; blocks_out[nblk].t_pos = tp - t0; blocks_out[nblk].length = 0; }
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
        res->q_pos = q0; res->t_pos = t0;
        if (nblk > blk_cap) status = KBO_NOSPACE;
    }
    res->status = status; res->n_blk = nblk; res->score = (int32_t)score2;
    res->n_match = nM; res->n_mismatch = nX; res->n_ins = nI; res->n_del = nD;
    free(path); free(M); free(AM); free(I); free(H); free(AI); free(AH);
    return status;
#undef IDX
#undef INB
#undef QI
#undef TI
}

/* Unbanded reference for the property tests: cheapest score of the whole query against any part of
 * the target (FIT) or of the two whole sequences (GLOBAL), linear gaps, matrix scorer. */
int kbo_full_dp(const uint8_t *q, uint32_t ql, const uint8_t *t, uint32_t tl, const kbo_params *P, int32_t *score)
{
    int32_t *prev = (int32_t *)malloc(4 * ((size_t)tl + 1)), *cur = (int32_t *)malloc(4 * ((size_t)tl + 1));
    for (uint32_t j = 0; j <= tl; j++) prev[j] = P->align_type == KBO_FIT ? 0 : (int32_t)j * P->del;
    for (uint32_t i = 1; i <= ql; i++) {
        cur[0] = (int32_t)i * P->ins;
        const int qc = gbo_code(q[i - 1]);
        for (uint32_t j = 1; j <= tl; j++) {
            int m = prev[j - 1] + P->sub[qc][gbo_code(t[j - 1])], u = prev[j] + P->ins, l = cur[j - 1] + P->del;
            cur[j] = m < u ? (m < l ? m : l) : (u < l ? u : l);
        }
        int32_t *x = prev; prev = cur; cur = x;
    }
    int32_t best = prev[tl];
    if (P->align_type == KBO_FIT) for (uint32_t j = 0; j <= tl; j++) if (prev[j] < best) best = prev[j];
    *score = best;
    free(prev); free(cur);
    return 0;
}
