// gba.hip -- guided banded alignment for gfx950 (MI355X): kernels and the C ABI of include/gba.h.
//
// Replaces, for a whole batch of independent (query, target, guide) pairs, what
// RefineAlignment does per candidate through libcpp's GuidedAlign / KBandAlign
// (/root/reference/iblasr/BlasrUtilsImpl.hpp:676-774, :648-675): guide-row
// construction, the banded min-cost DP, traceback into gap-free blocks and the
// ComputeAlignmentStats rescoring.
//
// Pipeline of one batch (all on one stream):
//   k_meta      one lane per pair: validate the guide, derive the DP region (Q rows x T columns)
//   k_scan2     exclusive scans of rows / columns -> arena offsets, totals (read back: sizes scratch)
//   k_pack      ASCII -> base codes {A,C,G,T,N}=0..4 of the DP region, one block per pair, coalesced
//   k_guide     one lane per pair: guide blocks -> one u32 per DP row {width, shift of the first column}
//   k_plan      bucket pairs by the LDS slots their widest row needs; record offsets (read back: grids)
//   k_fill_*    the DP: one LANE per pair, 64 pairs per wave, previous row in LDS, 2-bit arrows to HBM
//   k_traceback one lane per pair: walk the arrows back, emit blocks + statistics
//   k_scan1 / k_emit   pack the blocks of all pairs into the caller's arena in input order
//
// DP cell encoding (fill kernels).  A cell word is (score << 5) | (arrow << 3) | tcode where
// tcode is the base code of the cell's COLUMN; stored words have arrow == 0.  The three
// candidates are built with the arrow code pre-added, so one v_min3_i32 yields the minimum
// score AND, among equal scores, the candidate with the smallest arrow code -- the reference's
// "first candidate in priority order wins" rule.  Substitution costs are looked up with one
// v_perm_b32 from a byte table of the query base's matrix row, biased by +128; the bias adds
// the same constant (4096 per row after the shift) to every cell of a DP row and so never
// changes a comparison; it is removed when the final score is read.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "../../include/gba.h"

namespace {

constexpr uint32_t INFW = 0x40000000u;       // "no path" cell word
constexpr int32_t UNREACH = 0x20000000;      // unbiased words at or above this are unreachable
constexpr int32_t BIASW = 128 << 5;          // per-row bias of the byte substitution table
constexpr uint32_t MAX_ROWS = 65535;         // keeps biased words inside int32 (see gba_set_scoring)
constexpr uint32_t MAX_COLS = 1u << 20;
// Classes of the lane-per-pair fill kernel: a pair's class is the smallest kernel template
// (unroll depth = LDS row slots = 16 x record dwords per row) that holds its widest
// "shift + width".  Each non-empty class is one launch.
constexpr int NCLS = 4;
constexpr int PAD_SLOTS = 8;                 // a lane touches up to 7 slots past its row end (see k_fill_linear)
__host__ __device__ constexpr uint32_t cls_slots(int c) { return 32u << c; }
__host__ __device__ constexpr uint32_t cls_rw(int c) { return 2u << c; }   // record dwords per row

struct Plan {                 // device -> host after k_scan2 / k_plan
    uint64_t total_rows, total_cols;
    uint64_t total_rec_dwords;
    uint64_t total_cells;
    uint32_t cls_count[NCLS + 1];
    uint32_t cls_base[NCLS + 1];
    uint32_t n_ok;
    uint32_t pad;
};

struct Dev {                  // per-batch device state, SoA per pair
    uint32_t n;
    // inputs (device pointers)
    const uint8_t *q_arena, *t_arena;
    const uint64_t *q_off, *t_off, *blk_off;
    const uint32_t *q_len, *t_len, *n_blk;
    const gba_block *blocks;
    // per pair
    int32_t *status;
    uint32_t *qS, *tS, *Q, *T, *rows, *cols, *span, *lo_last, *cls, *order;
    uint64_t *row_off, *col_off, *rec_off, *ncells;
    int32_t *dp_score;
    Plan *plan;
    // arenas
    uint32_t *guide;          // one u32 per DP row: width | shift << 16
    uint8_t *qcodes, *tcodes;
    uint32_t *rec;            // traceback records
    gba_block *blk_scratch;
};

struct FillConst {
    uint32_t mrow[10];        // per query code: 8 bytes = biased substitution costs against target codes 0..7
    int32_t cU, cL;           // (cost << 5) + (arrow code << 3) [+ BIASW for the vertical move]
    int32_t del;
    uint32_t codeL;           // arrow code of the Left move (row 0 records)
};

struct TbConst {
    int32_t sub[25];
    int32_t ins, del, aopen, aext, affine;
    uint32_t codeU, codeL;
};

__device__ __forceinline__ uint32_t base_code(uint8_t c)
{
    // A/a C/c G/g T/t -> 0..3, everything else N = 4
    uint32_t u = c & 0xdfu;
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}

__device__ __forceinline__ uint64_t align16(uint64_t x) { return (x + 15) & ~uint64_t(15); }
__device__ __forceinline__ uint64_t qc_offset(const Dev &d, uint32_t p) { return align16(d.row_off[p] + 32ull * p); }
__device__ __forceinline__ uint64_t tc_offset(const Dev &d, uint32_t p) { return align16(d.col_off[p] + 64ull * p); }
// guide words of a pair start 16-byte aligned and are followed by at least 29 spare words
__device__ __forceinline__ uint64_t guide_offset(const Dev &d, uint32_t p) { return (d.row_off[p] + 32ull * p + 3) & ~uint64_t(3); }

// ---------------------------------------------------------------- k_meta
// DP region of each pair.  Guided modes: from the first block's start to the last block's end
// (AlignmentToGuide: tStart/qStart = first block, tEnd/qEnd = last block end).  k-band: the
// whole q x t rectangle.
__global__ void k_meta(Dev d, int mode, int kband_k)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n) return;
    int st = GBA_PAIR_OK;
    uint32_t qS = 0, tS = 0, Q = 0, T = 0;
    const uint32_t ql = d.q_len[p], tl = d.t_len[p];
    if (mode == GBA_MODE_KBAND) {
        Q = ql; T = tl;
        int64_t dd = (int64_t)tl - (int64_t)ql;
        if (dd < 0) dd = -dd;
        if (dd > kband_k) st = GBA_PAIR_END_OUT_OF_BAND;
    } else {
        const uint32_t nb = d.n_blk[p];
        if (nb == 0) st = GBA_PAIR_EMPTY_GUIDE;
        else {
            const gba_block *bl = d.blocks + d.blk_off[p];
            uint64_t qe = 0, te = 0;
            for (uint32_t i = 0; i < nb; i++) {
                const uint32_t bq = bl[i].q_pos, bt = bl[i].t_pos, len = bl[i].length;
                if (len == 0 || (uint64_t)bq + len > ql || (uint64_t)bt + len > tl) st = GBA_PAIR_BAD_GUIDE;
                if (i == 0) { qS = bq; tS = bt; }
                else if (bq < qe || bt < te) st = GBA_PAIR_BAD_GUIDE;
                qe = (uint64_t)bq + len; te = (uint64_t)bt + len;
            }
            if (st == GBA_PAIR_OK) { Q = (uint32_t)qe - qS; T = (uint32_t)te - tS; }
        }
    }
    if (st == GBA_PAIR_OK && (Q > MAX_ROWS || T > MAX_COLS)) st = GBA_PAIR_TOO_LONG;
    d.status[p] = st; d.qS[p] = qS; d.tS[p] = tS; d.Q[p] = Q; d.T[p] = T;
    d.rows[p] = st == GBA_PAIR_OK ? Q + 1 : 0;
    d.cols[p] = st == GBA_PAIR_OK ? T + 1 : 0;
    d.span[p] = 0; d.ncells[p] = 0; d.lo_last[p] = 0; d.dp_score[p] = 0; d.cls[p] = NCLS;
}

// ---------------------------------------------------------------- block-wide scan helpers (1024 threads)
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t n = __shfl_up((unsigned long long)v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

// exclusive scan over the block; *total receives the block sum.  sm: 17 uint64 of LDS.
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t *sm, uint64_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint64_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint64_t x = lane < nw ? sm[lane] : 0;
        uint64_t xi = wave_incl_scan(x);
        if (lane < nw) sm[lane] = xi - x;
        if (lane == nw - 1) sm[16] = xi;
    }
    __syncthreads();
    *total = sm[16];
    return sm[w] + inc - v;
}

// rows -> row_off, cols -> col_off (single block)
__global__ __launch_bounds__(1024) void k_scan2(Dev d)
{
    __shared__ uint64_t sm[17];
    uint64_t carry_r = 0, carry_c = 0;
    for (uint32_t base = 0; base < d.n; base += blockDim.x) {
        uint32_t p = base + threadIdx.x;
        uint64_t r = p < d.n ? d.rows[p] : 0, c = p < d.n ? d.cols[p] : 0, tot;
        uint64_t er = block_excl_scan(r, sm, &tot);
        if (p < d.n) d.row_off[p] = carry_r + er;
        carry_r += tot;
        uint64_t ec = block_excl_scan(c, sm, &tot);
        if (p < d.n) d.col_off[p] = carry_c + ec;
        carry_c += tot;
    }
    if (threadIdx.x == 0) { d.plan->total_rows = carry_r; d.plan->total_cols = carry_c; }
}

// ---------------------------------------------------------------- k_pack
// qc[i] = code of the query base of DP row i (1..Q), tc[j] = code of the target base of DP
// column j (1..T); index 0 and the bytes past the end (16 for q, 48 for t) hold N so block loads
// never leave the pair.
__global__ __launch_bounds__(256) void k_pack(Dev d)
{
    const uint32_t p = blockIdx.x;
    if (d.status[p] != GBA_PAIR_OK) return;
    const uint32_t Q = d.Q[p], T = d.T[p];
    uint8_t *qc = d.qcodes + qc_offset(d, p);
    uint8_t *tc = d.tcodes + tc_offset(d, p);
    const uint8_t *q = d.q_arena + d.q_off[p] + d.qS[p];
    const uint8_t *t = d.t_arena + d.t_off[p] + d.tS[p];
    for (uint32_t i = threadIdx.x; i <= Q + 16; i += blockDim.x)
        qc[i] = (i >= 1 && i <= Q) ? (uint8_t)base_code(q[i - 1]) : (uint8_t)4;
    for (uint32_t j = threadIdx.x; j <= T + 48; j += blockDim.x)
        tc[j] = (j >= 1 && j <= T) ? (uint8_t)base_code(t[j - 1]) : (uint8_t)4;
}

// ---------------------------------------------------------------- k_guide
// Guide rows in DP coordinates: row r covers columns [lo_r, hi_r], 0 <= lo_r <= hi_r <= T, row 0
// is the boundary row.  A row has a centre column c and a half-width w:
//   tPost = w, tPre = min(w, c - lo_{r-1})     (the band never reaches back past the previous row;
//   on the first row of a block w is replaced by max_band here, so target-only gaps stay connected)
// Inside a block c advances by one per row with w = band; between blocks the common part of the
// gap is walked diagonally and leftover query bases repeat the column, with
// w = min(max_band, band + |drift|).  Emits one u32 per row:
//   width (13 bits) | code of the row's query base << 13 | (lo_r - lo_{r-1}) << 16.
__global__ void k_guide(Dev d, int mode, int band, int max_band, int kband_k)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n || d.status[p] != GBA_PAIR_OK) return;
    const uint32_t Q = d.Q[p], T = d.T[p], tS = d.tS[p];
    uint32_t *g = d.guide + guide_offset(d, p);
    const uint8_t *qc = d.qcodes + qc_offset(d, p);
    const int B = mode == GBA_MODE_KBAND ? kband_k : min(band, max_band);
    uint32_t W0 = min(T, (uint32_t)B) + 1;
    g[0] = W0;
    uint64_t ncells = (uint64_t)B + 1;
    uint32_t span = W0, lo_prev = 0;
    bool wide = false;
    // block state machine
    const gba_block *bl = d.blocks + (mode == GBA_MODE_KBAND ? 0 : d.blk_off[p]);
    const uint32_t nb = mode == GBA_MODE_KBAND ? 0 : d.n_blk[p];
    uint32_t b = 0, phase = 0, remaining = 0, c = 0, w = (uint32_t)B, step = 1;
    uint32_t bq = 0, bt = 0, blen = 0;          // current block
    if (mode != GBA_MODE_KBAND) {
        bq = bl[0].q_pos; bt = bl[0].t_pos; blen = bl[0].length;
        remaining = blen; c = bt - tS; step = 1;    // c is pre-incremented below: first row centre = bt - tS + 1
    }
    uint32_t ins_left = 0, wgap = 0;
    bool first = mode != GBA_MODE_KBAND;      // first row of a block: reaches back to the previous row's first column
    for (uint32_t r = 1; r <= Q; r++) {
        if (mode != GBA_MODE_KBAND) {
            while (remaining == 0) {
                if (phase == 0) {
                    // leaving block b: set up the gap towards block b+1
                    const uint32_t qe = bq + blen, te = bt + blen;
                    b++;
                    bq = bl[b].q_pos; bt = bl[b].t_pos; blen = bl[b].length;
                    const uint32_t qGap = bq - qe, tGap = bt - te;
                    const int drift = (int)tGap - (int)qGap;
                    wgap = (uint32_t)min(max_band, band + (drift < 0 ? -drift : drift));
                    const uint32_t diag = min(qGap, tGap);
                    ins_left = qGap - diag;
                    phase = 1; remaining = diag; w = wgap; step = 1;
                } else if (phase == 1) {
                    phase = 2; remaining = ins_left; w = wgap; step = 0;
                } else {
                    phase = 0; remaining = blen; w = (uint32_t)B; step = 1; first = true;
                    c = bt - tS;                 // jump: the next row's centre is the block's first column
                }
            }
            remaining--;
        } else {
            step = 1;
        }
        c += step;
        const uint32_t tPre = min(first ? max((uint32_t)max_band, w) : w, c - lo_prev);
        const uint32_t lo = c - tPre;
        const uint32_t hi = min(T, c + w);
        const uint32_t W = hi - lo + 1, dl = lo - lo_prev;
        ncells += (uint64_t)tPre + w + 1;
        span = max(span, dl + W);
        if (dl > 65535u || W > 8191u) wide = true;
        g[r] = (W & 0x1fffu) | ((uint32_t)qc[r] << 13) | (dl << 16);
        lo_prev = lo; first = false;
    }
    d.span[p] = span; d.ncells[p] = ncells; d.lo_last[p] = lo_prev;
    if (wide) d.status[p] = GBA_PAIR_TOO_WIDE;
}

// ---------------------------------------------------------------- k_plan
// Classes by LDS slots a pair needs (its largest shift + width): 32 / 64 / 128 / 256.  `order`
// lists the OK pairs grouped by class (stable); rec_off is each pair's record offset in dwords
// (a record row is slots/16 dwords).
__global__ __launch_bounds__(1024) void k_plan(Dev d)
{
    __shared__ uint64_t sm[17];
    __shared__ uint32_t cnt[NCLS + 1], base[NCLS + 1], run[NCLS + 1];
    if (threadIdx.x <= NCLS) { cnt[threadIdx.x] = 0; run[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < d.n; p += blockDim.x) {
        uint32_t c = NCLS;
        if (d.status[p] == GBA_PAIR_OK) {
            const uint32_t s = d.span[p];
            c = NCLS;
            for (int k = NCLS - 1; k >= 0; k--) if (s <= cls_slots(k)) c = (uint32_t)k;
            if (c == NCLS) d.status[p] = GBA_PAIR_TOO_WIDE;
            else atomicAdd(&cnt[c], 1u);
        }
        d.cls[p] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int c = 0; c < NCLS; c++) { base[c] = a; a += cnt[c]; }
        base[NCLS] = a;
        for (int c = 0; c <= NCLS; c++) { d.plan->cls_count[c] = c < NCLS ? cnt[c] : 0; d.plan->cls_base[c] = base[c]; }
        d.plan->n_ok = a;
    }
    __syncthreads();
    uint64_t carry = 0, cells = 0;
    for (uint32_t b0 = 0; b0 < d.n; b0 += blockDim.x) {
        const uint32_t p = b0 + threadIdx.x;
        const uint32_t c = p < d.n ? d.cls[p] : NCLS;
        uint64_t tot;
        for (int k = 0; k < NCLS; k++) {
            uint64_t e = block_excl_scan(c == (uint32_t)k ? 1 : 0, sm, &tot);
            if (c == (uint32_t)k) d.order[base[k] + run[k] + (uint32_t)e] = p;
            __syncthreads();
            if (threadIdx.x == 0) run[k] += (uint32_t)tot;
            __syncthreads();
        }
        const uint64_t rd = c < NCLS ? (uint64_t)d.rows[p] * cls_rw((int)c) : 0;
        uint64_t e = block_excl_scan(rd, sm, &tot);
        if (p < d.n) d.rec_off[p] = carry + e;
        carry += tot;
        block_excl_scan(c < NCLS ? d.ncells[p] : 0, sm, &tot);
        cells += tot;
    }
    if (threadIdx.x == 0) { d.plan->total_rec_dwords = carry; d.plan->total_cells = cells; }
}

// ---------------------------------------------------------------- k_fill_linear
// One lane per pair, 64 pairs per wave, rows in lock step.  LDS holds one DP row per lane,
// slot-major: word (slot s, lane l) at dword s*64 + l (conflict free: a wave access touches one
// dword per lane in 64 consecutive banks, whatever the per-lane slot).  Slots are RELATIVE to the
// row's first column: cell k of row i reads its vertical predecessor from slot k + shift (the
// previous row's layout) and overwrites slot k; the read of a slot always precedes the write
// that destroys it, so one row buffer is enough.  Columns that enter the band on the right get an
// INFW word carrying their target base code before the row starts.
//
// A lane whose row is shorter than the wave's widest row skips whole groups of 4 cells; inside a
// group it runs all 4 cells and prefetches the next group's 4 slots (up to 7 slots past its row
// end).  Those accesses stay below KMAX + PAD_SLOTS and only produce record bits nobody reads.
// One group of 4 cells per recursion step; the recursion is the (forced) full unroll of the
// row loop, so slot offsets and record bit positions are immediates.
template <int K4, int KMAX>
struct CellGroups {
    static __device__ __forceinline__ void run(const uint32_t *rd, uint32_t *wr, uint32_t W, uint32_t mlo, uint32_t mhi,
                                               int32_t cU, int32_t cL, uint32_t diag, uint32_t left,
                                               uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3, uint32_t *r)
    {
        if constexpr (K4 < KMAX) {
            const bool in = (uint32_t)K4 < W;
            if (__builtin_amdgcn_ballot_w64(in) == 0) return;
            uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
            if (in) {
                if constexpr (K4 + 4 < KMAX) {
                    n0 = rd[(K4 + 4) * 64]; n1 = rd[(K4 + 5) * 64]; n2 = rd[(K4 + 6) * 64]; n3 = rd[(K4 + 7) * 64];
                }
                const uint32_t ups[4] = {u0, u1, u2, u3};
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int k = K4 + kk;
                    const uint32_t up = ups[kk];
                    const uint32_t sel = (up & 7u) | 0x0c0c0c00u;
                    const uint32_t sb = __builtin_amdgcn_perm(mhi, mlo, sel);
                    const int32_t m = (int32_t)((sb << 5) + diag);
                    const int32_t u = (int32_t)up + cU;
                    const int32_t l = (int32_t)left + cL;
                    const uint32_t best = (uint32_t)min(m, min(u, l));
                    const uint32_t word = (up & 31u) | (best & ~31u);
                    wr[k * 64] = word;
                    // arrow = bits 3..4 of best -> record bits 2*(k%16)..
                    const int sh = (k & 15) * 2 - 3;
                    r[k >> 4] |= sh >= 0 ? (best & 0x18u) << (sh >= 0 ? sh : 0) : (best & 0x18u) >> (sh < 0 ? -sh : 0);
                    diag = up; left = word;
                }
            }
            CellGroups<K4 + 4, KMAX>::run(rd, wr, W, mlo, mhi, cU, cL, diag, left, n0, n1, n2, n3, r);
        }
    }
};

// Memory traffic is kept off the per-row critical path: guide words and target codes arrive in
// chunks (CH rows / 32 columns at a time), are requested one chunk ahead into registers and
// installed into small LDS rings at chunk boundaries, so the only vector-memory wait sits at a
// boundary and covers loads issued a whole chunk earlier.  Record rows are stored directly and
// never waited for.
constexpr int CH = 8;                         // rows per guide chunk
constexpr int LDS_TAB = 16, LDS_GBUF = 2 * CH * 64, LDS_TBUF = 16 * 64;

template <int KMAX>
__global__ __launch_bounds__(64) void k_fill_linear(Dev d, FillConst fc, uint32_t cls, uint32_t slots)
{
    extern __shared__ uint32_t lds[];
    constexpr int RW = KMAX / 16;
    const int lane = threadIdx.x;
    uint32_t *tab = lds + (slots + PAD_SLOTS) * 64;
    uint32_t *gbuf = tab + LDS_TAB;           // [2*CH rows][lane]: guide words of chunks c, c+1
    uint32_t *tbuf = gbuf + LDS_GBUF;         // [2 blocks][8 dwords][lane]: target codes, 32 columns per block
    if (lane < 10) tab[lane] = fc.mrow[lane];
    __syncthreads();
    const uint32_t cnt = d.plan->cls_count[cls], cbase = d.plan->cls_base[cls];
    const uint32_t idx = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= cnt) return;
    const bool live = idx < cnt;
    const uint32_t p = d.order[cbase + (live ? idx : cnt - 1)];
    const uint32_t Q = live ? d.Q[p] : 0, T = d.T[p];
    const uint32_t *g = d.guide + guide_offset(d, p);
    uint32_t *rec = d.rec + d.rec_off[p];
    const uint8_t *tc = d.tcodes + tc_offset(d, p);

    uint32_t maxQ = Q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxQ = max(maxQ, (uint32_t)__shfl_xor((int)maxQ, o, 64));

    // ---- target-code blocks: the ring holds blocks tblk-1 and tblk, tp* is block tblk+1 in flight
    auto tblock_ok = [&](uint32_t b) { return 32u * b <= T + 17u; };   // k_pack pads 48 codes past T
    uint32_t tblk = 1;
    uint4 tpa = make_uint4(0, 0, 0, 0), tpb = tpa;
    {
        const uint4 a = *(const uint4 *)(tc), b = *(const uint4 *)(tc + 16);
        uint4 c4 = tpa, d4 = tpa;
        if (tblock_ok(1)) { c4 = *(const uint4 *)(tc + 32); d4 = *(const uint4 *)(tc + 48); }
        if (tblock_ok(2)) { tpa = *(const uint4 *)(tc + 64); tpb = *(const uint4 *)(tc + 80); }
        uint32_t *t0 = tbuf + lane, *t1 = tbuf + 8 * 64 + lane;
        t0[0] = a.x; t0[64] = a.y; t0[128] = a.z; t0[192] = a.w; t0[256] = b.x; t0[320] = b.y; t0[384] = b.z; t0[448] = b.w;
        t1[0] = c4.x; t1[64] = c4.y; t1[128] = c4.z; t1[192] = c4.w; t1[256] = d4.x; t1[320] = d4.y; t1[384] = d4.z; t1[448] = d4.w;
    }
    auto tcode = [&](uint32_t c) -> uint32_t {
        const uint32_t b = c >> 5;
        if (b + 1u - tblk <= 1u)        // b == tblk-1 or b == tblk
            return ((const uint8_t *)tbuf)[(((b & 1u) * 8u + ((c & 31u) >> 2)) * 64u + lane) * 4u + (c & 3u)];
        return tc[c];                   // outside the ring (long jump or re-entry): direct load
    };

    // ---- guide chunks: LDS ring holds chunks c and c+1, gp* is chunk c+2 in flight
    uint4 gpa = make_uint4(0, 0, 0, 0), gpb = gpa;
    {
        uint4 a = gpa, b = gpa, c4 = gpa, d4 = gpa;
        if (live) { a = *(const uint4 *)(g); b = *(const uint4 *)(g + 4); }
        if (live && CH <= Q) { c4 = *(const uint4 *)(g + CH); d4 = *(const uint4 *)(g + CH + 4); }
        if (live && 2 * CH <= Q) { gpa = *(const uint4 *)(g + 2 * CH); gpb = *(const uint4 *)(g + 2 * CH + 4); }
        uint32_t *g0 = gbuf + lane, *g1 = gbuf + CH * 64 + lane;
        g0[0] = a.x; g0[64] = a.y; g0[128] = a.z; g0[192] = a.w; g0[256] = b.x; g0[320] = b.y; g0[384] = b.z; g0[448] = b.w;
        g1[0] = c4.x; g1[64] = c4.y; g1[128] = c4.z; g1[192] = c4.w; g1[256] = d4.x; g1[320] = d4.y; g1[384] = d4.z; g1[448] = d4.w;
    }

    // ---- row 0: cell j costs j deletions, arrows Left
    uint32_t Wp = live ? (gbuf[lane] & 0x1fffu) : 0, lo = 0;
    for (uint32_t k = 0; __any(k < Wp); k++)
        if (k < Wp) lds[k * 64 + lane] = ((uint32_t)((int32_t)k * fc.del) << 5) | tcode(k);
    if (live) {
        const uint32_t allL = fc.codeL * 0x55555555u;
        for (int w = 0; w < RW; w++) rec[w] = allL;
    }
    uint32_t final_word = INFW;
    if (live && Q == 0) final_word = lds[T * 64 + lane];

    // two-row software pipeline over the guide ring: g1 = word of the next row, m* = matrix row of
    // the current row
    uint32_t g0w = gbuf[1 * 64 + lane];
    uint32_t mlo = tab[((g0w >> 13) & 7u) * 2], mhi = tab[((g0w >> 13) & 7u) * 2 + 1];
    uint32_t g1w = gbuf[2 * 64 + lane];

    for (uint32_t i = 1; i <= maxQ; i++) {
        if ((i & (CH - 1)) == 0) {
            // chunk boundary (uniform): chunk c = i/CH starts; install chunk c+1, request chunk c+2
            const uint32_t c = i / CH;
            uint32_t *gi = gbuf + ((c + 1) & 1u) * (CH * 64) + lane;
            gi[0] = gpa.x; gi[64] = gpa.y; gi[128] = gpa.z; gi[192] = gpa.w;
            gi[256] = gpb.x; gi[320] = gpb.y; gi[384] = gpb.z; gi[448] = gpb.w;
            if (live && (c + 2) * CH <= Q) { gpa = *(const uint4 *)(g + (c + 2) * CH); gpb = *(const uint4 *)(g + (c + 2) * CH + 4); }
            // target codes: once the lane's next new column lies in block tblk, block tblk-1 is dead
            if (lo + Wp >= 32u * tblk) {
                uint32_t *ti = tbuf + ((tblk + 1u) & 1u) * (8 * 64) + lane;
                ti[0] = tpa.x; ti[64] = tpa.y; ti[128] = tpa.z; ti[192] = tpa.w;
                ti[256] = tpb.x; ti[320] = tpb.y; ti[384] = tpb.z; ti[448] = tpb.w;
                tblk++;
                if (tblock_ok(tblk + 1u)) {
                    tpa = *(const uint4 *)(tc + 32u * (tblk + 1u));
                    tpb = *(const uint4 *)(tc + 32u * (tblk + 1u) + 16u);
                }
            }
        }
        const bool act = i <= Q;
        const uint32_t gcur = g0w, cmlo = mlo, cmhi = mhi;
        // fetch for the next rows (results are consumed after this row's cells)
        g0w = g1w;
        mlo = tab[((g0w >> 13) & 7u) * 2]; mhi = tab[((g0w >> 13) & 7u) * 2 + 1];
        g1w = gbuf[((i + 2) & (2 * CH - 1)) * 64 + lane];     // row i+2 is always inside chunks c, c+1

        const uint32_t W = act ? (gcur & 0x1fffu) : 0u, dsh = act ? (gcur >> 16) : 0u;
        // columns entering on the right: previous-layout slots [Wp, dsh + W)
        {
            const uint32_t send = act ? dsh + W : 0u;
            for (uint32_t s = Wp; __any(s < send); s += 4) {
                if (s < send) {
                    const uint32_t c0 = tcode(lo + s), c1 = tcode(lo + s + 1), c2 = tcode(lo + s + 2), c3 = tcode(lo + s + 3);
                    uint32_t *w4 = lds + s * 64 + lane;      // slots up to send+2 < KMAX + PAD_SLOTS
                    w4[0] = INFW | c0; w4[64] = INFW | c1; w4[128] = INFW | c2; w4[192] = INFW | c3;
                }
            }
        }
        uint32_t diag = INFW, left = INFW;
        if (act && dsh >= 1) diag = lds[(dsh - 1) * 64 + lane];
        const uint32_t *rd = lds + dsh * 64 + lane;
        uint32_t *wr = lds + lane;
        uint32_t r[RW];
#pragma unroll
        for (int w = 0; w < RW; w++) r[w] = 0;
        // The vertical predecessors of a group are read one group ahead of the cells that use them
        // (a read of slot s always precedes, in program order, the write that overwrites slot s:
        // reads of group g+1 touch slots above every slot group g writes).
        uint32_t u0 = 0, u1 = 0, u2 = 0, u3 = 0;
        if (0 < W) { u0 = rd[0]; u1 = rd[64]; u2 = rd[128]; u3 = rd[192]; }
        CellGroups<0, KMAX>::run(rd, wr, W, cmlo, cmhi, fc.cU, fc.cL, diag, left, u0, u1, u2, u3, r);
        if (act) {
            uint32_t *rr = rec + (uint64_t)i * RW;
            if constexpr (RW == 2) *(uint2 *)rr = make_uint2(r[0], r[1]);
            else {
#pragma unroll
                for (int w = 0; w < RW; w += 4) *(uint4 *)(rr + w) = make_uint4(r[w], r[w + 1], r[w + 2], r[w + 3]);
            }
            lo += dsh; Wp = W;
            if (i == Q) final_word = lds[(T - lo) * 64 + lane];
        }
    }
    if (live) {
        const int32_t v = (int32_t)(final_word & ~31u) - (int32_t)Q * BIASW;
        if (v >= UNREACH) d.status[p] = GBA_PAIR_UNREACHABLE;
        d.dp_score[p] = v >> 5;
    }
}

// ---------------------------------------------------------------- k_traceback
// One lane per pair: from the end cell (Q, T) follow the arrows to (0, 0); merge diagonal runs
// into blocks (ArrowPathToAlignment), count matches / mismatches / gaps and rescore the path
// (ComputeAlignmentStats).  Blocks are produced last-to-first, so they are written from the end
// of the pair's scratch region backwards and come out in forward order.
__global__ __launch_bounds__(64) void k_traceback(Dev d, TbConst tc_, gba_pair_result *res)
{
    __shared__ int32_t sub[25];
    if (threadIdx.x < 25) sub[threadIdx.x] = tc_.sub[threadIdx.x];
    __syncthreads();
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d.plan->n_ok) return;
    const uint32_t p = d.order[idx];
    gba_pair_result &R = res[p];
    if (d.status[p] != GBA_PAIR_OK) return;      // fill found the end cell unreachable
    const uint32_t Q = d.Q[p], T = d.T[p];
    const uint32_t RW = cls_rw((int)d.cls[p]);
    const uint32_t *rec = d.rec + d.rec_off[p];
    const uint32_t *g = d.guide + guide_offset(d, p);
    const uint8_t *tc = d.tcodes + tc_offset(d, p);
    gba_block *out = d.blk_scratch + d.row_off[p];
    const uint32_t cap = Q + 1;
    uint32_t i = Q, j = T, lo = d.lo_last[p], nb = 0, run = 0;
    uint32_t nM = 0, nX = 0, nI = 0, nD = 0;
    int32_t score = 0;
    uint32_t gw = g[i], prev_op = 0;
    uint64_t steps = 0;
    const uint64_t limit = (uint64_t)Q + T + 2;
    int st = GBA_PAIR_OK;
    while (i > 0 || j > 0) {
        const uint32_t k = j - lo;
        if (++steps > limit || k >= (gw & 0x1fffu)) { st = GBA_PAIR_UNREACHABLE; break; }
        const uint32_t a = (rec[(uint64_t)i * RW + (k >> 4)] >> ((k & 15u) * 2u)) & 3u;
        if (a == 0) {
            if (i == 0 || j == 0) { st = GBA_PAIR_UNREACHABLE; break; }
            const uint32_t qb = (gw >> 13) & 7u, tb = tc[j];
            if (qb == tb) nM++; else nX++;
            score += sub[qb * 5 + tb];
            run++; i--; j--; lo -= gw >> 16; gw = g[i]; prev_op = 0;
        } else {
            if (run) { nb++; out[cap - nb] = gba_block{i, j, run}; run = 0; }
            uint32_t op;
            if (a == tc_.codeU) {
                if (i == 0) { st = GBA_PAIR_UNREACHABLE; break; }
                nI++; i--; lo -= gw >> 16; gw = g[i]; op = 1;
            } else if (a == tc_.codeL) {
                if (j == 0) { st = GBA_PAIR_UNREACHABLE; break; }
                nD++; j--; op = 2;
            } else { st = GBA_PAIR_UNREACHABLE; break; }
            if (tc_.affine) { if (prev_op != op) score += tc_.aopen; score += tc_.aext; }
            else score += op == 1 ? tc_.ins : tc_.del;
            prev_op = op;
        }
    }
    if (run) { nb++; out[cap - nb] = gba_block{i, j, run}; }
    if (st != GBA_PAIR_OK) { d.status[p] = st; nb = 0; }
    R.n_blk = nb; R.score = score;
    R.n_match = nM; R.n_mismatch = nX; R.n_ins = nI; R.n_del = nD;
}

// ---------------------------------------------------------------- results
__global__ void k_results_init(Dev d, gba_pair_result *res)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n) return;
    gba_pair_result r;
    memset(&r, 0, sizeof(r));
    r.status = d.status[p];
    res[p] = r;
}

// n_blk -> blk_off (single block); pairs whose blocks do not fit the arena get NOSPACE
__global__ __launch_bounds__(1024) void k_scan1(Dev d, gba_pair_result *res, uint64_t capacity, uint64_t *needed)
{
    __shared__ uint64_t sm[17];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < d.n; b0 += blockDim.x) {
        const uint32_t p = b0 + threadIdx.x;
        const bool ok = p < d.n && d.status[p] == GBA_PAIR_OK;
        uint64_t v = ok ? res[p].n_blk : 0, tot;
        uint64_t e = block_excl_scan(v, sm, &tot);
        if (p < d.n) {
            gba_pair_result &R = res[p];
            R.status = d.status[p];
            R.blk_off = carry + e;
            if (ok) {
                R.dp_score = d.dp_score[p];
                R.q_pos = d.qS[p]; R.t_pos = d.tS[p];
                R.rows = d.Q[p]; R.cols = d.T[p]; R.ncells = d.ncells[p];
                if (carry + e + v > capacity) R.status = GBA_PAIR_NOSPACE;
            }
        }
        carry += tot;
    }
    if (threadIdx.x == 0) *needed = carry;
}

__global__ __launch_bounds__(256) void k_emit(Dev d, const gba_pair_result *res, gba_block *out)
{
    const uint32_t p = blockIdx.x;
    if (res[p].status != GBA_PAIR_OK) return;
    const uint32_t nb = res[p].n_blk, cap = d.Q[p] + 1;
    const gba_block *src = d.blk_scratch + d.row_off[p] + (cap - nb);
    gba_block *dst = out + res[p].blk_off;
    for (uint32_t x = threadIdx.x; x < nb; x += blockDim.x) dst[x] = src[x];
}

// ================================================================ host side
struct Buf {
    void *ptr = nullptr;
    size_t cap = 0;
};

}  // namespace

constexpr int NAUX = 4;

struct gba_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip = 0;
    bool have_scoring = false;
    gba_scoring sc;
    FillConst fc;
    TbConst tbc;
    // device scratch (grow-only)
    Buf b_pair32[12], b_pair64[4], b_dp, b_plan, b_guide, b_qc, b_tc, b_rec, b_blk, b_needed;
    // staging for the host-pointer path
    Buf s_q, s_t, s_blk, s_qoff, s_toff, s_boff, s_qlen, s_tlen, s_nblk, s_res, s_out;
    Plan *h_plan = nullptr;          // pinned
    uint64_t *h_needed = nullptr;    // pinned
    bool in_flight = false;
    uint32_t flight_n = 0;
    int flight_rc = 0;
    uint64_t flight_needed = 0;
    hipEvent_t ev[6] = {};
    hipStream_t aux[4] = {};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {};
    gba_timing timing = {};
};

namespace {

#define HIPCHK(ctx, call)                                        \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) { (ctx)->last_hip = (int)e__; return e__ == hipErrorOutOfMemory ? GBA_E_NOMEM : GBA_E_HIP; } \
    } while (0)

int ensure(gba_ctx *c, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return GBA_OK;
    if (b.ptr) { HIPCHK(c, hipFree(b.ptr)); b.ptr = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.ptr, want));
    b.cap = want;
    return GBA_OK;
}

void release(Buf &b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr; b.cap = 0;
}

template <int KMAX>
int launch_fill(gba_ctx *c, const Dev &d, uint32_t cls, uint32_t count, hipStream_t st)
{
    const uint32_t slots = cls_slots((int)cls);
    const size_t lds = ((size_t)(slots + PAD_SLOTS) * 64 + LDS_TAB + LDS_GBUF + LDS_TBUF) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        const size_t mx = ((size_t)(KMAX + PAD_SLOTS) * 64 + LDS_TAB + LDS_GBUF + LDS_TBUF) * 4;
        HIPCHK(c, hipFuncSetAttribute((const void *)k_fill_linear<KMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_fill_linear<KMAX>, dim3((count + 63) / 64), dim3(64), lds, st, d, c->fc, cls, slots);
    HIPCHK(c, hipGetLastError());
    return GBA_OK;
}

// Runs one batch whose inputs and outputs are device pointers.  Synchronises the stream twice
// (arena sizes, launch grids) and once at the end.
int run_device(gba_ctx *c, const gba_batch *b, gba_results *out)
{
    const uint32_t n = b->n_pairs;
    out->blocks_needed = 0;
    if (n == 0) return GBA_OK;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 12; i++) { int rc = ensure(c, c->b_pair32[i], (size_t)n * 4); if (rc) return rc; }
    for (int i = 0; i < 4; i++) { int rc = ensure(c, c->b_pair64[i], (size_t)n * 8); if (rc) return rc; }
    { int rc = ensure(c, c->b_plan, sizeof(Plan)); if (rc) return rc; }
    { int rc = ensure(c, c->b_needed, 8); if (rc) return rc; }
    Dev d;
    memset(&d, 0, sizeof(d));
    d.n = n;
    d.q_arena = b->q_arena; d.t_arena = b->t_arena; d.blocks = b->blocks;
    d.q_off = b->q_off; d.t_off = b->t_off; d.blk_off = b->blk_off;
    d.q_len = b->q_len; d.t_len = b->t_len; d.n_blk = b->n_blk;
    d.status = (int32_t *)c->b_pair32[0].ptr;
    d.qS = (uint32_t *)c->b_pair32[1].ptr; d.tS = (uint32_t *)c->b_pair32[2].ptr;
    d.Q = (uint32_t *)c->b_pair32[3].ptr; d.T = (uint32_t *)c->b_pair32[4].ptr;
    d.rows = (uint32_t *)c->b_pair32[5].ptr; d.cols = (uint32_t *)c->b_pair32[6].ptr;
    d.span = (uint32_t *)c->b_pair32[7].ptr; d.lo_last = (uint32_t *)c->b_pair32[8].ptr;
    d.cls = (uint32_t *)c->b_pair32[9].ptr; d.order = (uint32_t *)c->b_pair32[10].ptr;
    d.dp_score = (int32_t *)c->b_pair32[11].ptr;
    d.row_off = (uint64_t *)c->b_pair64[0].ptr; d.col_off = (uint64_t *)c->b_pair64[1].ptr;
    d.rec_off = (uint64_t *)c->b_pair64[2].ptr; d.ncells = (uint64_t *)c->b_pair64[3].ptr;
    d.plan = (Plan *)c->b_plan.ptr;
    const gba_scoring &sc = c->sc;
    const bool timing = (c->flags & GBA_FLAG_TIMING) != 0;
    hipStream_t s = c->stream;
    const uint32_t g256 = (n + 255) / 256;

    if (timing) HIPCHK(c, hipEventRecord(c->ev[0], s));
    hipLaunchKernelGGL(k_meta, dim3(g256), dim3(256), 0, s, d, sc.mode, sc.kband_k);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(1024), 0, s, d);
    HIPCHK(c, hipMemcpyAsync(c->h_plan, d.plan, sizeof(Plan), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const uint64_t rows = c->h_plan->total_rows, cols = c->h_plan->total_cols;
    { int rc = ensure(c, c->b_guide, (size_t)(rows + 32ull * n + 64) * 4); if (rc) return rc; }
    { int rc = ensure(c, c->b_qc, (size_t)(rows + 32ull * n + 64)); if (rc) return rc; }
    { int rc = ensure(c, c->b_tc, (size_t)(cols + 64ull * n + 128)); if (rc) return rc; }
    { int rc = ensure(c, c->b_blk, (size_t)(rows + 4) * sizeof(gba_block)); if (rc) return rc; }
    d.guide = (uint32_t *)c->b_guide.ptr;
    d.qcodes = (uint8_t *)c->b_qc.ptr; d.tcodes = (uint8_t *)c->b_tc.ptr;
    d.blk_scratch = (gba_block *)c->b_blk.ptr;

    hipLaunchKernelGGL(k_pack, dim3(n), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_guide, dim3((n + 63) / 64), dim3(64), 0, s, d, sc.mode, sc.band, sc.max_band, sc.kband_k);
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, d);
    HIPCHK(c, hipMemcpyAsync(c->h_plan, d.plan, sizeof(Plan), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const Plan pl = *c->h_plan;
    { int rc = ensure(c, c->b_rec, (size_t)(pl.total_rec_dwords + 16) * 4); if (rc) return rc; }
    d.rec = (uint32_t *)c->b_rec.ptr;

    gba_pair_result *res = out->pairs;
    hipLaunchKernelGGL(k_results_init, dim3(g256), dim3(256), 0, s, d, res);
    if (timing) HIPCHK(c, hipEventRecord(c->ev[1], s));
    // One launch per non-empty class, spread over auxiliary streams so that classes with few
    // waves share the GPU instead of running back to back.
    uint32_t launches = 0;
    {
        HIPCHK(c, hipEventRecord(c->ev_fork, s));
        bool used[NAUX] = {};
        for (int k = 0; k < NCLS; k++) {
            const uint32_t cnt = pl.cls_count[k];
            if (!cnt) continue;
            const int a = (int)(launches % NAUX);
            hipStream_t st = c->aux[a];
            if (!used[a]) { HIPCHK(c, hipStreamWaitEvent(st, c->ev_fork, 0)); used[a] = true; }
            int rc;
            if (k == 0) rc = launch_fill<32>(c, d, 0, cnt, st);
            else if (k == 1) rc = launch_fill<64>(c, d, 1, cnt, st);
            else if (k == 2) rc = launch_fill<128>(c, d, 2, cnt, st);
            else rc = launch_fill<256>(c, d, 3, cnt, st);
            if (rc) return rc;
            launches++;
        }
        for (int a = 0; a < NAUX; a++) {
            if (!used[a]) continue;
            HIPCHK(c, hipEventRecord(c->ev_join[a], c->aux[a]));
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_join[a], 0));
        }
    }
    if (timing) HIPCHK(c, hipEventRecord(c->ev[2], s));
    if (pl.n_ok)
        hipLaunchKernelGGL(k_traceback, dim3((pl.n_ok + 63) / 64), dim3(64), 0, s, d, c->tbc, res);
    if (timing) HIPCHK(c, hipEventRecord(c->ev[3], s));
    hipLaunchKernelGGL(k_scan1, dim3(1), dim3(1024), 0, s, d, res, out->blocks_capacity, (uint64_t *)c->b_needed.ptr);
    hipLaunchKernelGGL(k_emit, dim3(n), dim3(256), 0, s, d, res, out->blocks);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_needed, c->b_needed.ptr, 8, hipMemcpyDeviceToHost, s));
    if (timing) HIPCHK(c, hipEventRecord(c->ev[4], s));
    HIPCHK(c, hipStreamSynchronize(s));
    out->blocks_needed = *c->h_needed;
    if (timing) {
        gba_timing &t = c->timing;
        HIPCHK(c, hipEventElapsedTime(&t.fill_ms, c->ev[1], c->ev[2]));
        HIPCHK(c, hipEventElapsedTime(&t.traceback_ms, c->ev[2], c->ev[3]));
        HIPCHK(c, hipEventElapsedTime(&t.total_ms, c->ev[0], c->ev[4]));
        t.fill_launches = launches;
        t.cells = pl.total_cells; t.rows = rows; t.record_bytes = pl.total_rec_dwords * 4;
    }
    return out->blocks_needed > out->blocks_capacity ? GBA_E_NOSPACE : GBA_OK;
}

}  // namespace

extern "C" {

int gba_version(void) { return GBA_VERSION; }

const char *gba_strerror(int code)
{
    switch (code) {
    case GBA_OK: return "ok";
    case GBA_E_INVALID: return "invalid argument";
    case GBA_E_NOMEM: return "out of memory";
    case GBA_E_HIP: return "HIP runtime error";
    case GBA_E_NOSPACE: return "block arena too small";
    case GBA_E_UNSUPPORTED: return "scoring not supported by the kernels";
    case GBA_E_STATE: return "call out of sequence";
    case GBA_PAIR_EMPTY_GUIDE: return "pair: empty guide";
    case GBA_PAIR_BAD_GUIDE: return "pair: malformed guide";
    case GBA_PAIR_UNREACHABLE: return "pair: end cell unreachable inside the band";
    case GBA_PAIR_NOSPACE: return "pair: blocks did not fit the arena";
    case GBA_PAIR_END_OUT_OF_BAND: return "pair: end cell outside the k-band";
    case GBA_PAIR_TOO_WIDE: return "pair: band row too wide";
    case GBA_PAIR_TOO_LONG: return "pair: too many rows or columns";
    default: return "unknown code";
    }
}

int gba_create(gba_ctx **out, const gba_config *cfg)
{
    if (!out || !cfg) return GBA_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return GBA_E_HIP;
    gba_ctx *c = new (std::nothrow) gba_ctx();
    if (!c) return GBA_E_NOMEM;
    c->device = cfg->device; c->flags = cfg->flags;
    int rc = GBA_OK;
    do {
        if (hipSetDevice(c->device) != hipSuccess) { rc = GBA_E_HIP; break; }
        if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
        else { if (hipStreamCreate(&c->stream) != hipSuccess) { rc = GBA_E_HIP; break; } c->own_stream = true; }
        if (hipHostMalloc((void **)&c->h_plan, sizeof(Plan)) != hipSuccess) { rc = GBA_E_NOMEM; break; }
        if (hipHostMalloc((void **)&c->h_needed, 8) != hipSuccess) { rc = GBA_E_NOMEM; break; }
        for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rc = GBA_E_HIP; break; }
        if (rc) break;
        if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) { rc = GBA_E_HIP; break; }
        for (int a = 0; a < NAUX; a++) {
            if (hipStreamCreateWithFlags(&c->aux[a], hipStreamNonBlocking) != hipSuccess) { rc = GBA_E_HIP; break; }
            if (hipEventCreateWithFlags(&c->ev_join[a], hipEventDisableTiming) != hipSuccess) { rc = GBA_E_HIP; break; }
        }
    } while (0);
    if (rc) { gba_destroy(c); return rc; }
    *out = c;
    return GBA_OK;
}

void gba_destroy(gba_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->b_pair32) release(b);
    for (auto &b : c->b_pair64) release(b);
    Buf *bs[] = {&c->b_dp, &c->b_plan, &c->b_guide, &c->b_qc, &c->b_tc, &c->b_rec, &c->b_blk, &c->b_needed,
                 &c->s_q, &c->s_t, &c->s_blk, &c->s_qoff, &c->s_toff, &c->s_boff, &c->s_qlen, &c->s_tlen,
                 &c->s_nblk, &c->s_res, &c->s_out};
    for (Buf *b : bs) release(*b);
    if (c->h_plan) (void)hipHostFree(c->h_plan);
    if (c->h_needed) (void)hipHostFree(c->h_needed);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int a = 0; a < NAUX; a++) {
        if (c->ev_join[a]) (void)hipEventDestroy(c->ev_join[a]);
        if (c->aux[a]) { (void)hipStreamSynchronize(c->aux[a]); (void)hipStreamDestroy(c->aux[a]); }
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gba_last_hip_error(gba_ctx *c) { return c ? c->last_hip : 0; }

int gba_set_scoring(gba_ctx *c, const gba_scoring *s)
{
    if (!c || !s) return GBA_E_INVALID;
    if (s->mode != GBA_MODE_LINEAR && s->mode != GBA_MODE_KBAND) return GBA_E_UNSUPPORTED;
    if (s->tie_break != GBA_TIE_DIAG_LEFT_UP && s->tie_break != GBA_TIE_DIAG_UP_LEFT) return GBA_E_INVALID;
    if (s->band < 0 || s->max_band < 0 || s->band > 30000 || s->max_band > 30000) return GBA_E_INVALID;
    if (s->mode == GBA_MODE_KBAND && (s->kband_k < 0 || s->kband_k > 30000)) return GBA_E_INVALID;
    // The fill kernels keep substitution costs in a +128-biased byte table and cell words in int32
    // with 5 low tag bits; |cost| <= 127 and at most 65535 rows keep every reachable word below
    // UNREACH and every word inside int32.
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            if (s->sub[a][b] < -128 || s->sub[a][b] > 127) return GBA_E_UNSUPPORTED;
    if (s->ins < -127 || s->ins > 127 || s->del < -127 || s->del > 127) return GBA_E_UNSUPPORTED;
    c->sc = *s;
    const uint32_t codeL = s->tie_break == GBA_TIE_DIAG_LEFT_UP ? 1u : 2u, codeU = 3u - codeL;
    FillConst &f = c->fc;
    for (int a = 0; a < 5; a++) {
        uint32_t lo = 0, hi = 0;
        for (int b = 0; b < 4; b++) lo |= (uint32_t)(uint8_t)(s->sub[a][b] + 128) << (8 * b);
        hi = (uint32_t)(uint8_t)(s->sub[a][4] + 128) * 0x01010101u;   // codes 4..7 all read the N column
        f.mrow[2 * a] = lo; f.mrow[2 * a + 1] = hi;
    }
    f.cU = s->ins * 32 + (int32_t)(codeU << 3) + BIASW;
    f.cL = s->del * 32 + (int32_t)(codeL << 3);
    f.del = s->del; f.codeL = codeL;
    TbConst &t = c->tbc;
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) t.sub[a * 5 + b] = s->sub[a][b];
    t.ins = s->stats_ins; t.del = s->stats_del; t.aopen = s->stats_affine_open; t.aext = s->stats_affine_extend;
    t.affine = s->stats_affine; t.codeU = codeU; t.codeL = codeL;
    c->have_scoring = true;
    return GBA_OK;
}

int gba_submit_device(gba_ctx *c, const gba_batch *b, gba_results *out)
{
    if (!c || !b || !out) return GBA_E_INVALID;
    if (!c->have_scoring || c->in_flight) return GBA_E_STATE;
    if (b->n_pairs && (!out->pairs || (!out->blocks && out->blocks_capacity))) return GBA_E_INVALID;
    return run_device(c, b, out);
}

int gba_submit(gba_ctx *c, const gba_batch *b)
{
    if (!c || !b) return GBA_E_INVALID;
    if (!c->have_scoring || c->in_flight) return GBA_E_STATE;
    const uint32_t n = b->n_pairs;
    HIPCHK(c, hipSetDevice(c->device));
    c->flight_n = n; c->flight_rc = GBA_OK; c->flight_needed = 0;
    if (n == 0) { c->in_flight = true; return GBA_OK; }
    int rc;
    struct { Buf *buf; const void *src; size_t bytes; } cp[] = {
        {&c->s_q, b->q_arena, (size_t)b->q_arena_bytes}, {&c->s_t, b->t_arena, (size_t)b->t_arena_bytes},
        {&c->s_blk, b->blocks, (size_t)b->n_blocks_total * sizeof(gba_block)},
        {&c->s_qoff, b->q_off, (size_t)n * 8}, {&c->s_toff, b->t_off, (size_t)n * 8}, {&c->s_boff, b->blk_off, (size_t)n * 8},
        {&c->s_qlen, b->q_len, (size_t)n * 4}, {&c->s_tlen, b->t_len, (size_t)n * 4}, {&c->s_nblk, b->n_blk, (size_t)n * 4}};
    for (auto &x : cp) {
        if ((rc = ensure(c, *x.buf, x.bytes + 16))) return rc;
        if (x.bytes) HIPCHK(c, hipMemcpyAsync(x.buf->ptr, x.src, x.bytes, hipMemcpyHostToDevice, c->stream));
    }
    // worst case one block per query base
    uint64_t cap = 0;
    for (uint32_t p = 0; p < n; p++) cap += (uint64_t)b->q_len[p] + 1;
    if ((rc = ensure(c, c->s_res, (size_t)n * sizeof(gba_pair_result)))) return rc;
    if ((rc = ensure(c, c->s_out, (size_t)cap * sizeof(gba_block)))) return rc;
    gba_batch db = *b;
    db.q_arena = (const uint8_t *)c->s_q.ptr; db.t_arena = (const uint8_t *)c->s_t.ptr;
    db.blocks = (const gba_block *)c->s_blk.ptr;
    db.q_off = (const uint64_t *)c->s_qoff.ptr; db.t_off = (const uint64_t *)c->s_toff.ptr;
    db.blk_off = (const uint64_t *)c->s_boff.ptr;
    db.q_len = (const uint32_t *)c->s_qlen.ptr; db.t_len = (const uint32_t *)c->s_tlen.ptr;
    db.n_blk = (const uint32_t *)c->s_nblk.ptr;
    gba_results dr;
    dr.pairs = (gba_pair_result *)c->s_res.ptr; dr.blocks = (gba_block *)c->s_out.ptr;
    dr.blocks_capacity = cap; dr.blocks_needed = 0;
    rc = run_device(c, &db, &dr);
    if (rc != GBA_OK && rc != GBA_E_NOSPACE) return rc;
    c->flight_rc = rc; c->flight_needed = dr.blocks_needed;
    c->in_flight = true;
    return GBA_OK;
}

int gba_wait(gba_ctx *c, gba_results *out)
{
    if (!c || !out) return GBA_E_INVALID;
    if (!c->in_flight) return GBA_E_STATE;
    c->in_flight = false;
    const uint32_t n = c->flight_n;
    out->blocks_needed = c->flight_needed;
    if (n == 0) return GBA_OK;
    if (!out->pairs) return GBA_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out->pairs, c->s_res.ptr, (size_t)n * sizeof(gba_pair_result), hipMemcpyDeviceToHost, c->stream));
    const uint64_t fit = c->flight_needed <= out->blocks_capacity ? c->flight_needed : out->blocks_capacity;
    if (fit && out->blocks)
        HIPCHK(c, hipMemcpyAsync(out->blocks, c->s_out.ptr, (size_t)fit * sizeof(gba_block), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->flight_needed > out->blocks_capacity) {
        for (uint32_t p = 0; p < n; p++) {
            gba_pair_result &R = out->pairs[p];
            if (R.status == GBA_PAIR_OK && R.blk_off + R.n_blk > out->blocks_capacity) R.status = GBA_PAIR_NOSPACE;
        }
        return GBA_E_NOSPACE;
    }
    return GBA_OK;
}

int gba_get_timing(gba_ctx *c, gba_timing *t)
{
    if (!c || !t) return GBA_E_INVALID;
    *t = c->timing;
    return GBA_OK;
}

}  // extern "C"
