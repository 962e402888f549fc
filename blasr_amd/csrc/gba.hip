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
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <algorithm>
#include <atomic>
#include <vector>

#include "../../include/gba.h"

namespace {

// vectors that are only dword aligned (per-lane positions in dword streams)
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32u1 __attribute__((aligned(1)));   // a dword at any byte address

constexpr uint32_t INFW = 0x40000000u;       // "no path" cell word
constexpr int32_t UNREACH = 0x20000000;      // unbiased words at or above this are unreachable
constexpr uint32_t MAX_ROWS = 65535;         // keeps biased words inside int32 (see gba_set_scoring)
constexpr uint32_t MAX_ROWS_AFFINE = 32767;  // affine words carry one more tag bit
constexpr uint32_t MAX_COLS = 1u << 20;
constexpr int N_PAIR32 = 18, N_PAIR64 = 6;   // per-pair scratch arrays of 4 and 8 bytes (gba_ctx::b_pair32 / b_pair64)
// Classes of the lane-per-pair fill kernel: a pair's class is the smallest LDS row buffer
// (32 << class slots) that holds its widest "shift + width".  Each non-empty class is one launch.
constexpr int NCLS = 5;                      // lane-per-pair classes of 32, 64, 128 slots, then the two wave-per-pair classes
constexpr int CLS_WIDE = NCLS - 1;           // pairs too wide for everything else: k_fill_wide_cols (k_fill_wide<4> with a cost profile) / k_fill_wide_affine<false>
constexpr int CLS_WNARROW = NCLS - 2;        // wave-per-pair with a small ring (shift + width <= WN_SPAN): pairs a little too wide for
                                             // the lane classes, and ALL pairs of a batch too small to fill the GPU one lane per pair
constexpr uint32_t WN_SPAN = 124;
static_assert(CLS_WNARROW == 3, "lane classes are 0..2");
constexpr int LANE_CLS_LINEAR = 3;           // linear gaps: lane classes of 32, 64, 128 slots, anything wider is cheaper wave-per-pair
                                             // (a 96-slot class was measured: the batch gets slower, not faster)
constexpr int LANE_CLS_AFFINE = 2;           // affine gaps: 32 and 64 slots; with two rings a wave of the 128-slot class takes 67 KB of LDS (two waves per CU), and
                                             // the narrow wave-per-pair kernel is faster than that at any batch size (measured on 20 kb pairs, band 16)
constexpr int PAD_SLOTS = 3;                 // spare slots behind a ring
constexpr int SPAN_MARGIN = 4;               // a pair fits a class if its widest shift + width <= slots - SPAN_MARGIN (k_fill touches up to 7 slots past a row end)
constexpr int CAP_LINEAR = 32, CAP_AFFINE = 40;   // cells per lane per step of k_fill (band 10 rows are 21+ cells, band 16 rows 33+)
// k_fill takes one word of a pair's guide stream per step, so a row of more cells than one step computes is followed in the
// stream by a continuation word (zero, never looked at) for every further step it needs.
__host__ __device__ constexpr uint32_t cont_words(uint32_t W, uint32_t cap) { return W > cap ? (W - 1u) / cap : 0u; }
__host__ __device__ constexpr uint32_t step_cap(int mode) { return mode == GBA_MODE_AFFINE ? (uint32_t)CAP_AFFINE : (uint32_t)CAP_LINEAR; }
constexpr size_t ARENA_SLACK = 16384 + 256;          // bytes in front of the record / target-code arenas: k_traceback's windows start before a pair's data
// slots of lane class c: 32, 64, and a widest class of `top` slots (128 for linear gaps, 96 for affine)
__host__ __device__ constexpr uint32_t cls_slots(int c, uint32_t top) { return c < 2 ? 32u << c : top; }

// Guide stream words (one stream per pair, built by k_guide, consumed by k_fill):
//   ROW  width[0:10) | query base code[10:13) | shift of the first column[13:23) | codes of the
//        first three columns entering the band on the right[23:32)
//   EXT  eight more entering-column codes, 3 bits each; a row has ceil((entering - 3) / 8) of them
constexpr uint32_t ROW_W_MASK = 1023u, ROW_META_MASK = 0x7fffffu;

struct Plan {                 // device -> host after k_scan3 / k_plan
    uint64_t total_rows, total_cols, total_words;
    uint64_t total_rec_dwords;
    uint64_t total_cells;
    uint32_t cls_count[NCLS + 1];
    uint32_t cls_base[NCLS + 1];
    uint32_t n_ok;
    uint32_t pool_fail;           // pairs whose blocks found no room (see blk_region_start)
    unsigned long long pool_used; // blocks handed out from the overflow pool
};

struct Dev {                  // per-batch device state, SoA per pair
    uint32_t n;
    // inputs (device pointers)
    const uint8_t *q_arena, *t_arena, *t_rc;
    uint64_t q_arena_bytes, t_arena_bytes, n_blocks_total;   // extents of the three arenas: every pair's slices are checked against them
    const uint64_t *q_off, *t_off, *blk_off;
    const uint32_t *q_len, *t_len, *n_blk;
    const gba_block *blocks;
    const int32_t *guide_status;   // guides made by the library itself: why a pair has none (null: the caller's guides)
    const int16_t *prof;           // costs of a quality-value scorer, 11 per byte of q_arena (null: matrix, ins, del)
    // Packed sources, 2 bits per base (base x = bits 2 (x & 15) .. of dword x >> 4; A, C, G, T = 0 .. 3):
    const uint32_t *q_pk;          // non-null: the query arena is packed (gba_batch.q_packed) and q_arena is not read
    const uint32_t *ref_pk, *ref_mask;   // non-null: the targets are windows of the resident reference, kept packed with one mask
                                   // bit per base (bit x & 31 of dword x >> 5) for the letters outside ACGT; t_arena is not read
    // per pair
    int32_t *status;
    uint32_t *qS, *tS, *Q, *T, *rows, *cols, *wcap, *span, *lo_last, *cls, *order, *recdw;
    uint64_t *row_off, *col_off, *word_off, *rec_off, *ncells;
    int32_t *dp_score;
    // GBA_ALIGN_LOCAL: the cheapest cell of each pair, written by the fill: its row, column, the first column of that row and
    // the position of that row's meta dword in the pair's records -- where the traceback begins instead of at (Q, T)
    uint32_t *end_i, *end_j, *end_lo, *end_ptr;
    Plan *plan;
    // arenas
    uint32_t *stream;         // guide stream words
    uint8_t *tcodes;
    uint32_t *rec;            // traceback records, per DP row: ceil(width/16) dwords of 2-bit arrows, then one meta dword
    gba_block *blk_scratch;
    uint64_t *blk_end;             // per pair: one past the last block of the region that holds its refined blocks
    uint64_t blk_pool_off, blk_pool_cap;    // the overflow pool behind the pairs' regions: first block, blocks
    uint32_t blk_div;              // a pair's region holds rows / blk_div + 31 blocks
};

struct FillConst {
    uint32_t mrow[10];        // per query code: 8 bytes = 0, then biased substitution costs against target codes A,C,G,T,N,N,N
    int32_t cU, cL;           // (cost << S) + (arrow code << 3) [+ row bias for the vertical move]; S = 5 linear, 6 affine
    int32_t del, ins;
    // affine only: gap-state constants and the row-0 diagonal that makes cell (0,0) cost 0
    int32_t cIO, cIE, cDO, cDE, diag0;
};

struct TbConst {
    int32_t sub[25];
    int32_t ins, del, aopen, aext, affine;
    uint32_t codeU, codeL;
    int32_t mode;
    int32_t n_extra;                       // further rescoring scorers
    int32_t xsub[GBA_MAX_EXTRA][25];
    int32_t xins[GBA_MAX_EXTRA], xdel[GBA_MAX_EXTRA], xaopen[GBA_MAX_EXTRA], xaext[GBA_MAX_EXTRA], xaffine[GBA_MAX_EXTRA];
};

// gap part of a rescoring: nI / nD gap bases in nopen runs
__device__ __forceinline__ int32_t gap_score(int32_t affine, int32_t ins, int32_t del, int32_t aopen, int32_t aext,
                                             uint32_t nI, uint32_t nD, uint32_t nopen)
{
    return affine ? (int32_t)nopen * aopen + (int32_t)(nI + nD) * aext : (int32_t)nI * ins + (int32_t)nD * del;
}

__device__ __forceinline__ uint32_t base_code(uint8_t c)
{
    // A/a C/c G/g T/t -> 0..3, everything else N = 4
    uint32_t u = c & 0xdfu;
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}

// code 0 .. 3 of base `pos` of a packed array
__device__ __forceinline__ uint32_t packed_code(const uint32_t *pk, uint64_t pos) { return (pk[pos >> 4] >> (2u * (uint32_t)(pos & 15u))) & 3u; }
__device__ __forceinline__ uint32_t mask_bit(const uint32_t *mk, uint64_t pos) { return (mk[pos >> 5] >> (uint32_t)(pos & 31u)) & 1u; }
// code 0 .. 4 of the query base at arena position `pos`: letters, or a packed arena
__device__ __forceinline__ uint32_t query_code(const uint8_t *q_arena, const uint32_t *q_pk, uint64_t pos)
{
    return q_pk ? packed_code(q_pk, pos) : base_code(q_arena[pos]);
}

__device__ __forceinline__ uint64_t align16(uint64_t x) { return (x + 15) & ~uint64_t(15); }
// code arrays of a pair start 16-byte aligned and are followed by at least 33 spare bytes
__device__ __forceinline__ uint64_t tc_offset(const Dev &d, uint32_t p) { return align16(d.col_off[p] + 48ull * p); }

// Block scratch of the tracebacks.  A refined alignment of Q rows can have Q blocks, but has about one per eight rows
// on 15 %-error reads; a pair's own region therefore holds rows / blk_div + 31 blocks (regions follow from the scan of the
// rows: disjoint, no offsets of their own), written from its end backwards.  A pair that needs more takes a region of
// full size from the pool behind the regions (one atomic add), moves what it has written there and goes on; when the
// pool is used up the pair gives up (Plan::pool_fail) and the host runs the batch again with blk_div = 1 (regions of
// rows + 31 blocks, which always suffice).  Readers find a pair's blocks at blk_end[p] - n_blk.
__device__ __forceinline__ uint64_t blk_region_start(const Dev &d, uint32_t p) { return d.row_off[p] / d.blk_div + 32ull * p; }
__device__ __forceinline__ uint64_t blk_region_end(const Dev &d, uint32_t p) { return (d.row_off[p] + d.rows[p]) / d.blk_div + 32ull * p + 31ull; }

// The writer's side: `outp` points at the block written last (a region's end at first), `lowp` at the region's start.
struct BlkOut {
    uint32_t *outp, *lowp;
    uint64_t end;
    bool room;
};
__device__ __forceinline__ BlkOut blk_out_begin(const Dev &d, uint32_t p)
{
    BlkOut o;
    o.end = blk_region_end(d, p);
    o.outp = (uint32_t *)(d.blk_scratch + o.end);
    o.lowp = (uint32_t *)(d.blk_scratch + blk_region_start(d, p));
    o.room = true;
    return o;
}
// called by ONE lane when outp == lowp and another block is to be written: `have` blocks are in the region
__device__ __forceinline__ void blk_out_spill(gba_block *scratch, Plan *plan, uint64_t pool_off, uint64_t pool_cap, BlkOut &o, uint32_t rows, uint32_t have)
{
    const unsigned long long at = pool_cap ? atomicAdd(&plan->pool_used, (unsigned long long)rows) : 0ull;
    if (pool_cap && at + rows <= pool_cap) {
        const uint64_t nend = pool_off + at + rows;
        uint32_t *dst = (uint32_t *)(scratch + nend) - 3u * have;
        const uint32_t *src = o.lowp;
        for (uint32_t x = 0; x < 3u * have; x++) dst[x] = src[x];
        o.outp = dst; o.lowp = (uint32_t *)(scratch + (nend - rows)); o.end = nend;
    } else {
        o.room = false;
        atomicAdd(&plan->pool_fail, 1u);
    }
}

// ---------------------------------------------------------------- k_meta
// DP region of each pair.  Guided modes: from the first block's start to the last block's end
// (AlignmentToGuide: tStart/qStart = first block, tEnd/qEnd = last block end).  k-band: the
// whole q x t rectangle.  Also an upper bound on the pair's guide stream length.
//
// score_a / score_b bound what a cell word can hold: a path makes Q row moves, each adding at most
// 128 (the bias of the substitution table) + score_a, and up to T moves along a row, each adding at
// most score_b; the sum must stay below score_lim (the value from which words count as unreachable),
// otherwise the pair is TOO_LONG.
__global__ void k_meta(Dev d, int mode, int kband_k, uint32_t score_a, uint32_t score_b, uint32_t score_lim, int band, int max_band)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n) return;
    int st = GBA_PAIR_OK;
    uint32_t qS = 0, tS = 0, Q = 0, T = 0;
    uint64_t drift_sum = 0, cont_sum = 0;
    // continuation words: a row's width is at most w + wp + 1 with its two half-widths of k_guide (hi = c + w, lo >= c - wp)
    const uint32_t cap = step_cap(mode);
    const uint32_t gB = (uint32_t)min(band, max_band), gfirst = max((uint32_t)max_band, gB);
    const uint32_t ql = d.q_len[p], tl = d.t_len[p];
    uint32_t nb = 0;
    // slices that leave their arena (a window outside the reference, a corrupt offset) are never touched
    const uint64_t qo = d.q_off[p], to = d.t_off[p];
    if (to > d.t_arena_bytes || tl > d.t_arena_bytes - to || qo > d.q_arena_bytes || ql > d.q_arena_bytes - qo)
        st = GBA_PAIR_BAD_GUIDE;
    if (mode == GBA_MODE_KBAND) {
        Q = ql; T = tl;
        int64_t dd = (int64_t)tl - (int64_t)ql;
        if (dd < 0) dd = -dd;
        if (st == GBA_PAIR_OK && dd > kband_k) st = GBA_PAIR_END_OUT_OF_BAND;
    } else {
        nb = d.n_blk[p];
        const uint64_t bo = d.blk_off[p];
        if (nb == 0) st = d.guide_status && d.guide_status[p] != GBA_PAIR_OK ? d.guide_status[p] : GBA_PAIR_EMPTY_GUIDE;
        else if (bo > d.n_blocks_total || nb > d.n_blocks_total - bo) { st = GBA_PAIR_BAD_GUIDE; nb = 0; }
        else if (st != GBA_PAIR_OK) nb = 0;
        else {
            const gba_block *bl = d.blocks + d.blk_off[p];
            uint64_t qe = 0, te = 0;
            for (uint32_t i = 0; i < nb; i++) {
                const uint32_t bq = bl[i].q_pos, bt = bl[i].t_pos, len = bl[i].length;
                if (len == 0 || (uint64_t)bq + len > ql || (uint64_t)bt + len > tl) st = GBA_PAIR_BAD_GUIDE;
                uint64_t adr = 0, wprev = gB;
                if (i == 0) { qS = bq; tS = bt; }
                else if (bq < qe || bt < te) st = GBA_PAIR_BAD_GUIDE;
                else {
                    const int64_t dr = (int64_t)(bt - te) - (int64_t)(bq - qe);
                    adr = (uint64_t)(dr < 0 ? -dr : dr);
                    drift_sum += adr;
                    const uint64_t wg = min((uint64_t)max_band, (uint64_t)band + adr);       // the gap's rows
                    const uint64_t qGap = (uint64_t)bq - qe;
                    cont_sum += qGap * cont_words((uint32_t)min(2 * wg + 1, (uint64_t)MAX_COLS + 1), cap);
                    if (qGap) wprev = wg;
                }
                // the block's first row reaches back to where the row before it began (the centre moves on by at most
                // 1 + |drift|), and never further than its own lower half-width; its other rows are 2 B + 1 wide at most
                cont_sum += cont_words((uint32_t)min(min((uint64_t)gB + gfirst + 1, gB + wprev + 2 + adr), (uint64_t)MAX_COLS + 1), cap);
                cont_sum += (uint64_t)(len - 1u) * cont_words(2u * gB + 1u, cap);
                qe = (uint64_t)bq + len; te = (uint64_t)bt + len;
            }
            if (st == GBA_PAIR_OK) { Q = (uint32_t)qe - qS; T = (uint32_t)te - tS; }
        }
    }
    if (st == GBA_PAIR_OK && (Q > (mode == GBA_MODE_AFFINE ? MAX_ROWS_AFFINE : MAX_ROWS) || T > MAX_COLS)) st = GBA_PAIR_TOO_LONG;
    if (st == GBA_PAIR_OK && (uint64_t)Q * (128u + score_a) + (uint64_t)T * score_b >= score_lim) st = GBA_PAIR_TOO_LONG;
    // the lane-per-pair fill of a batch with a cost profile puts the bias on the columns instead of the rows
    if (st == GBA_PAIR_OK && d.prof && (uint64_t)Q * score_a + (uint64_t)T * (128u + score_b) >= score_lim) st = GBA_PAIR_TOO_LONG;
    d.status[p] = st; d.qS[p] = qS; d.tS[p] = tS; d.Q[p] = Q; d.T[p] = T;
    d.rows[p] = st == GBA_PAIR_OK ? Q + 1 : 0;
    d.cols[p] = st == GBA_PAIR_OK ? T + 1 : 0;
    // Every DP row has one ROW word.  Columns enter the band at most T + 1 + sum|drift| times in
    // total (the right edge only moves back when a widened gap ends), eight per EXT word, plus at
    // most one partly filled EXT word per row that needs any (the boundary row, the first row of
    // a gap, the first row of a block).  Rounded up to whole 64-byte sectors plus one sector that
    // k_fill may prefetch past the end.
    if (mode == GBA_MODE_KBAND) cont_sum = ((uint64_t)Q + 1) * cont_words((uint32_t)min(2ull * (uint64_t)kband_k + 1, (uint64_t)MAX_COLS + 1), cap);
    else cont_sum += cont_words(gB + 1u, cap);          // the boundary row
    uint64_t wc = (uint64_t)Q + 1 + ((uint64_t)T + 1 + drift_sum) / 8 + 2ull * nb + 4 + cont_sum;
    wc = ((wc + 15) & ~15ull) + 16;
    d.wcap[p] = st == GBA_PAIR_OK ? (uint32_t)wc : 0;
    d.span[p] = 0; d.ncells[p] = 0; d.lo_last[p] = 0; d.dp_score[p] = 0; d.cls[p] = NCLS; d.recdw[p] = 0;
}

// ---------------------------------------------------------------- block-wide scan helpers
// The single-block kernels (k_scan3, k_plan, k_scan1) run SCAN_T threads with SCAN_E consecutive
// pairs per thread, one wave per SIMD and well under 112 VGPRs: that is what is left on a SIMD
// that holds two k_fill waves of another part of the batch.  (As 1024-thread blocks they needed a
// CU with at most one fill wave on every SIMD and waited ~18 ms for one whenever a fill launch had
// just filled the GPU.)
constexpr int SCAN_T = 256, SCAN_E = 16, PLAN_E = 8;     // k_plan keeps more per pair in registers
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

// exclusive scan of in[0..n) into out[0..n) by one block; returns the total
__device__ __forceinline__ uint64_t scan_array(const uint32_t *in, uint64_t *out, uint32_t n, uint64_t *sm)
{
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n; base += SCAN_E * blockDim.x) {
        uint32_t v[SCAN_E];
        uint64_t lv = 0;
#pragma unroll
        for (int x = 0; x < SCAN_E; x++) {
            const uint32_t p = base + SCAN_E * threadIdx.x + x;
            v[x] = p < n ? in[p] : 0u;
            lv += v[x];
        }
        uint64_t tot;
        uint64_t e = carry + block_excl_scan(lv, sm, &tot);
#pragma unroll
        for (int x = 0; x < SCAN_E; x++) {
            const uint32_t p = base + SCAN_E * threadIdx.x + x;
            if (p < n) out[p] = e;
            e += v[x];
        }
        carry += tot;
    }
    return carry;
}

// rows -> row_off, cols -> col_off, wcap -> word_off (single block)
__global__ __launch_bounds__(SCAN_T) void k_scan3(Dev d)
{
    __shared__ uint64_t sm[17];
    const uint64_t tr = scan_array(d.rows, d.row_off, d.n, sm);
    const uint64_t tc = scan_array(d.cols, d.col_off, d.n, sm);
    const uint64_t tw = scan_array(d.wcap, d.word_off, d.n, sm);
    if (threadIdx.x == 0) { d.plan->total_rows = tr; d.plan->total_cols = tc; d.plan->total_words = tw; }
}

// ---------------------------------------------------------------- k_pack
// tc[j] = code + 1 (1..5) of the target base of DP column j (1..T), the form the fill kernels'
// byte-table lookup takes; index 0 and the 16 bytes past the end hold N.  (Query bases are coded
// where they are used, in k_guide.)
__global__ __launch_bounds__(256) void k_pack(Dev d)
{
    const uint32_t p = blockIdx.x;
    if (d.status[p] != GBA_PAIR_OK) return;
    const uint32_t T = d.T[p];
    uint8_t *tc = d.tcodes + tc_offset(d, p);
    const bool rc = d.t_rc != nullptr && d.t_rc[p] != 0;
    if (d.ref_pk) {
        // a window of the packed reference: column j is base t_off + tS + j - 1 of it, or, on the reverse strand, base
        // t_off + t_len - tS - j complemented; a masked base (a letter outside ACGT) is N
        const uint64_t first = d.t_off[p] + (rc ? (uint64_t)d.t_len[p] - d.tS[p] : (uint64_t)d.tS[p]);
        for (uint32_t j0 = 4 * threadIdx.x; j0 <= T + 16; j0 += 4 * blockDim.x) {
            uint32_t outw = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t j = j0 + x;
                uint32_t code = 4u;
                if (j >= 1 && j <= T) {
                    const uint64_t g = rc ? first - j : first + j - 1;
                    const uint32_t f = packed_code(d.ref_pk, g);
                    code = mask_bit(d.ref_mask, g) ? 4u : rc ? 3u - f : f;
                }
                outw |= (code + 1u) << (8 * x);
            }
            *(uint32_t *)(tc + j0) = outw;
        }
        return;
    }
    // forward: column j is slice byte tS + j - 1; reverse complement: the slice read backwards from
    // its last byte, bases complemented (A<->T, C<->G, N stays)
    const uint8_t *t = d.t_arena + d.t_off[p] + (rc ? (uint64_t)d.t_len[p] - d.tS[p] : (uint64_t)d.tS[p]);
    // four columns per thread: one (unaligned) dword in, one aligned dword out
    for (uint32_t j0 = 4 * threadIdx.x; j0 <= T + 16; j0 += 4 * blockDim.x) {
        uint32_t in = 0;
        if (j0 >= 1 && j0 + 3 <= T) {
            // forward: bytes t[j0-1 .. j0+2]; reverse: bytes t[-j0-3 .. -j0], last one first
            in = rc ? __builtin_bswap32(*(const u32u1 *)(t - j0 - 3)) : *(const u32u1 *)(t + j0 - 1);
        } else {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t j = j0 + x;
                if (j >= 1 && j <= T) in |= (uint32_t)(rc ? *(t - j) : t[j - 1]) << (8 * x);
            }
        }
        uint32_t outw = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const uint32_t j = j0 + x;
            uint32_t code = 4u;
            if (j >= 1 && j <= T) {
                const uint32_t f = base_code((uint8_t)(in >> (8 * x)));
                code = rc && f < 4u ? 3u - f : f;
            }
            outw |= (code + 1u) << (8 * x);
        }
        *(uint32_t *)(tc + j0) = outw;
    }
}

// ---------------------------------------------------------------- k_guide
// Guide rows in DP coordinates: row r covers columns [lo_r, hi_r], 0 <= lo_r <= hi_r <= T, row 0
// is the boundary row.  A row has a centre column c and a half-width w:
//   hi = min(T, c + w),  lo = max(lo_{r-1}, c - w)   (the band never reaches back past the previous
//   row; on the first row of a block w is replaced by max(max_band, w) in the lo term, so
//   target-only gaps stay connected)
// Inside a block c advances by one per row with w = band; between blocks the common part of the
// gap is walked diagonally and leftover query bases repeat the column, with
// w = min(max_band, band + |drift|).  Emits the pair's guide stream: per row one ROW word, then
// EXT words while more than three columns enter the band on the right.
//
// One WAVE per pair, one LANE per row, 64 rows per iteration.  Row r of a pair lies in "unit" u =
// the gap in front of guide block u plus block u; units tile the query, so a row finds its unit
// from marks the unit starts leave in LDS plus a running maximum across lanes, and its centre
// column follows in closed form from blocks u-1 and u.  lo is a running maximum of (c - w) across
// rows, the stream offset a running sum of words per row: both are wave scans with a carry.
constexpr int GD_WIN = 128;   // guide blocks held in LDS: blocks [wb - 1, wb + GD_WIN]

// Inclusive wave scans through DPP row shifts and row broadcasts (no LDS traffic).  Lanes whose
// source lies outside the row / the written rows receive the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int32_t dpp_from(int32_t identity, int32_t v)
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int32_t wave_scan_max(int32_t v)
{
    constexpr int32_t ID = (int32_t)0x80000000;
    v = max(v, dpp_from<0x111, 0xf>(ID, v));     // row_shr:1
    v = max(v, dpp_from<0x112, 0xf>(ID, v));     // row_shr:2
    v = max(v, dpp_from<0x114, 0xf>(ID, v));     // row_shr:4
    v = max(v, dpp_from<0x118, 0xf>(ID, v));     // row_shr:8
    v = max(v, dpp_from<0x142, 0xa>(ID, v));     // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_from<0x143, 0xc>(ID, v));     // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t wave_scan_sum(uint32_t u)
{
    int32_t v = (int32_t)u;
    v += dpp_from<0x111, 0xf>(0, v);
    v += dpp_from<0x112, 0xf>(0, v);
    v += dpp_from<0x114, 0xf>(0, v);
    v += dpp_from<0x118, 0xf>(0, v);
    v += dpp_from<0x142, 0xa>(0, v);
    v += dpp_from<0x143, 0xc>(0, v);
    return (uint32_t)v;
}

// value of the previous lane (lane 0 receives `first`)
__device__ __forceinline__ uint32_t wave_prev(uint32_t v, uint32_t first)
{
    // wave_shr:1 shifts across the whole wave
    return (uint32_t)__builtin_amdgcn_update_dpp((int32_t)first, (int32_t)v, 0x138, 0xf, 0xf, false);
}

__global__ __launch_bounds__(64) void k_guide(Dev d, int mode, int band, int max_band, int kband_k)
{
    __shared__ uint32_t bwin[(GD_WIN + 2) * 3];
    const uint32_t p = blockIdx.x;
    const int lane = threadIdx.x;
    if (d.status[p] != GBA_PAIR_OK) return;
    const bool guided = mode != GBA_MODE_KBAND;
    const uint32_t Q = d.Q[p], T = d.T[p], tS = d.tS[p], qS = d.qS[p];
    uint32_t *g = d.stream + d.word_off[p];
    const uint32_t wcap = d.wcap[p];
    const uint64_t qa0 = d.q_off[p] + qS;                   // query base of DP row r is base qa0 + r - 1 of the query arena
    const uint8_t *tc = d.tcodes + tc_offset(d, p);
    const uint32_t B = (uint32_t)(guided ? min(band, max_band) : kband_k);
    const uint32_t wfirst = max((uint32_t)max_band, B);
    const uint32_t nb = guided ? d.n_blk[p] : 0u;
    const uint32_t *bl = (const uint32_t *)(d.blocks + (guided ? d.blk_off[p] : 0));
    uint32_t ub = 0, wb = 0;                     // unit of the previous iteration's last row; window base
    bool loaded = false;
    int32_t carry_lo = 0;
    uint32_t carry_e = 0, nw = 0;                // previous row's hi + 1; words produced
    uint32_t ncells32 = 0;                       // per lane: at most 1024 rows of at most 2047 cells
    uint32_t span = 0, recdw = 0;
    bool wide = false;
    for (uint32_t r0 = 0; r0 <= Q; r0 += 64) {
        const uint32_t r = r0 + lane;
        const bool valid = r <= Q;
        uint32_t c = r, w = B, wp = B;           // k-band: the centre is the row itself
        if (guided) {
            if (!loaded || ub + 64u > wb + (uint32_t)GD_WIN) {
                __syncthreads();
                wb = ub; loaded = true;
                for (uint32_t k = lane; k < (uint32_t)GD_WIN + 2u; k += 64) {
                    const uint32_t bi = wb + k - 1u;         // wraps for the slot in front of block 0
                    if (bi < nb) { bwin[3 * k] = bl[3 * bi]; bwin[3 * k + 1] = bl[3 * bi + 1]; bwin[3 * k + 2] = bl[3 * bi + 2]; }
                }
                __syncthreads();
            }
            // unit u starts on the row after the last row of block u - 1; the starts that fall into
            // this iteration's rows sit in the first lanes (they increase with u): every row counts
            // how many of them it has passed
            uint32_t rs = 0xffffffffu;
            {
                const uint32_t u = ub + 1u + lane;
                if (u < nb) {
                    const uint32_t *pb = bwin + 3 * (u - wb);        // block u - 1
                    rs = pb[0] + pb[2] - qS + 1u;
                }
            }
            const uint32_t nstart = (uint32_t)__popcll(__ballot(rs - r0 < 64u));
            uint32_t unit = ub;
            for (uint32_t k = 0; k < nstart; k++)
                unit += r >= (uint32_t)__builtin_amdgcn_readlane((int32_t)rs, (int)k) ? 1u : 0u;
            ub += nstart;
            if (valid && r > 0) {
                const uint32_t x = r - 1u;
                const uint32_t *cb = bwin + 3 * (unit - wb + 1u);    // block `unit`
                const uint32_t bq = cb[0], bt = cb[1];
                const uint32_t xb = bq - qS;
                if (x >= xb) {
                    c = bt - tS + 1u + (x - xb);
                    if (x == xb) wp = wfirst;
                } else {
                    const uint32_t qe = cb[-3] + cb[-1], te = cb[-2] + cb[-1];   // end of block unit - 1 (unit >= 1 here)
                    const uint32_t qGap = bq - qe, tGap = bt - te;
                    const int drift = (int)tGap - (int)qGap;
                    w = wp = (uint32_t)min(max_band, band + (drift < 0 ? -drift : drift));
                    c = te - tS + min(x - (qe - qS) + 1u, min(qGap, tGap));
                }
            } else {
                c = 0;
            }
        }
        const int32_t cand = valid ? (int32_t)c - (int32_t)wp : (int32_t)0x80000000;
        const uint32_t lo = (uint32_t)max(wave_scan_max(cand), carry_lo);
        const uint32_t e = min(T, c + w) + 1u;   // hi + 1
        const uint32_t lo_prev = wave_prev(lo, (uint32_t)carry_lo), e_prev = wave_prev(e, carry_e);
        carry_lo = (int32_t)__builtin_amdgcn_readlane((int32_t)lo, 63); carry_e = (uint32_t)__builtin_amdgcn_readlane((int32_t)e, 63);
        const uint32_t W = e - lo, dl = lo - lo_prev;
        // columns entering on the right: [hi_prev + 1, hi]
        const uint32_t nnew = valid && e > e_prev ? e - e_prev : 0u, cnew = e_prev;
        const uint32_t ncont = valid ? cont_words(W, step_cap(mode)) : 0u;
        const uint32_t words = valid ? 1u + (nnew > 3u ? (nnew + 4u) >> 3 : 0u) + ncont : 0u;
        const uint32_t incl = wave_scan_sum(words);
        const uint32_t off = nw + incl - words;
        nw += (uint32_t)__builtin_amdgcn_readlane((int32_t)incl, 63);
        if (valid) {
            ncells32 += (c - lo) + w + 1u;
            span = max(span, dl + W);
            recdw += 1u + (mode == GBA_MODE_AFFINE ? (W + 3u) >> 2 : (W + 15u) >> 4);
            if (dl > 1023u || W > 1023u) wide = true;
            uint32_t word = (W & ROW_W_MASK) | ((r > 0u ? query_code(d.q_arena, d.q_pk, qa0 + r - 1u) : 4u) << 10) | ((dl & 1023u) << 13);
            if (nnew > 0u) {
                const uint32_t t4 = *(const u32u1 *)(tc + cnew);           // codes of columns cnew .. cnew + 3 (padding behind the last one)
                word |= (t4 & 7u) << 23;
                if (nnew > 1u) word |= ((t4 >> 8) & 7u) << 26;
                if (nnew > 2u) word |= ((t4 >> 16) & 7u) << 29;
            }
            if (off < wcap) g[off] = word;
            uint32_t o = off + 1u;
            for (uint32_t j = 3; j < nnew; j += 8, o++) {
                // eight codes (bytes below 8) to 3 bits each; the code array has spare bytes behind its end
                auto pack4 = [](uint32_t v) { return (v & 7u) | ((v >> 5) & 0x38u) | ((v >> 10) & 0x1c0u) | ((v >> 15) & 0xe00u); };
                const uint32_t lo4 = *(const u32u1 *)(tc + cnew + j), hi4 = *(const u32u1 *)(tc + cnew + j + 4);
                uint32_t ew = pack4(lo4) | (pack4(hi4) << 12);
                const uint32_t left = nnew - j;
                if (left < 8u) ew &= (1u << (3u * left)) - 1u;
                if (o < wcap) g[o] = ew;
            }
            for (uint32_t x = 0; x < ncont; x++, o++)
                if (o < wcap) g[o] = 0;
            if (r == Q) d.lo_last[p] = lo;
        }
    }
    // totals of the pair
    uint64_t ncells = ncells32;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ncells += __shfl_xor((unsigned long long)ncells, o, 64);
        recdw += __shfl_xor(recdw, o, 64);
        span = max(span, __shfl_xor(span, o, 64));
    }
    wide = __any(wide) || nw + 8u > wcap;        // the second cannot happen (see k_meta); never overrun the stream
    // k_fill fetches whole sectors; what it may pop past the last word is zero
    for (uint32_t x = nw + lane; x < wcap && x < nw + 24u; x += 64) g[x] = 0;
    if (lane == 0) {
        d.span[p] = span; d.ncells[p] = ncells; d.recdw[p] = recdw;
        if (wide) d.status[p] = GBA_PAIR_TOO_WIDE;
    }
}

// ---------------------------------------------------------------- k_plan
// `order` lists the OK pairs grouped by class (stable); rec_off is each pair's record offset in
// dwords (4 spare dwords precede every pair: the traceback reads 16-byte windows ending at a row;
// 12 follow it: the fill kernel stores whole steps and lets the next store overwrite the tail).  Pairs whose widest row needs more than the largest class are TOO_WIDE.
__global__ __launch_bounds__(SCAN_T) void k_plan(Dev d, int ncls, int has_wide, uint32_t top_slots)
{
    __shared__ uint64_t sm[17];
    __shared__ uint32_t cnt[NCLS + 1], base[NCLS + 1], cmap[NCLS + 1];
    if (threadIdx.x <= NCLS) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < d.n; p += blockDim.x) {
        uint32_t c = NCLS;
        if (d.status[p] == GBA_PAIR_OK) {
            const uint32_t s = d.span[p];
            for (int k = ncls - 1; k >= 0; k--) if (s + SPAN_MARGIN <= cls_slots(k, top_slots)) c = (uint32_t)k;
            if (c == NCLS && has_wide) c = s <= WN_SPAN ? CLS_WNARROW : CLS_WIDE;
            if (c == NCLS) d.status[p] = GBA_PAIR_TOO_WIDE;
            else atomicAdd(&cnt[c], 1u);
        }
        d.cls[p] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // At most two lane classes are launched (with the wave-per-pair class that makes three
        // concurrent launches, as many as there are hardware queues to run them side by side; a
        // class queued behind another one adds a whole launch of ~20 ms).  A small widest class
        // goes wave-per-pair (a lone lane-per-pair wave lasts as long as a full launch); then the
        // narrowest classes merge upwards until two are left.
        for (int c = 0; c <= NCLS; c++) cmap[c] = (uint32_t)c;
        auto nonempty = [&]() { int k = 0; for (int c = 0; c < CLS_WNARROW; c++) k += cnt[c] ? 1 : 0; return k; };
        if (has_wide && nonempty() > 2) {
            int top = CLS_WNARROW - 1;
            while (!cnt[top]) top--;
            if (cnt[top] <= 1024u) { cnt[CLS_WNARROW] += cnt[top]; cnt[top] = 0; cmap[top] = CLS_WNARROW; }
        }
        while (nonempty() > 2) {
            int lo = 0;
            while (!cnt[lo]) lo++;
            int up = lo + 1;
            while (!cnt[up]) up++;
            cnt[up] += cnt[lo]; cnt[lo] = 0;
            for (int c = 0; c < CLS_WNARROW; c++) if (cmap[c] == (uint32_t)lo) cmap[c] = (uint32_t)up;
        }
        uint32_t a = 0;
        for (int c = 0; c < NCLS; c++) { base[c] = a; a += cnt[c]; }
        base[NCLS] = a;
        for (int c = 0; c <= NCLS; c++) { d.plan->cls_count[c] = c < NCLS ? cnt[c] : 0; d.plan->cls_base[c] = base[c]; }
        d.plan->n_ok = a;
    }
    __syncthreads();
    // PLAN_E consecutive pairs per thread and packed counters: within one iteration (4096 pairs) one
    // scan carries the counts of the four lane classes (16 bits each), a second one the count of
    // the wide class; totals are carried unpacked.
    static_assert(NCLS == 5 && CLS_WIDE == 4 && SCAN_T * PLAN_E < 65536, "packing of the class counters");
    uint32_t runc[NCLS] = {0, 0, 0, 0, 0};
    uint64_t cells = 0;
    for (uint32_t b0 = 0; b0 < d.n; b0 += PLAN_E * blockDim.x) {
        uint32_t c[PLAN_E];
        uint64_t la = 0, lb = 0;
#pragma unroll
        for (int x = 0; x < PLAN_E; x++) {
            const uint32_t p = b0 + PLAN_E * threadIdx.x + x;
            c[x] = p < d.n ? cmap[d.cls[p]] : NCLS;
            if (c[x] < NCLS) cells += d.ncells[p];
            la += c[x] < CLS_WIDE ? 1ull << (16 * c[x]) : 0ull;
            lb += c[x] == CLS_WIDE ? 1ull : 0ull;
        }
        uint64_t totA, totB;
        uint64_t ea = block_excl_scan(la, sm, &totA);
        uint64_t eb = block_excl_scan(lb, sm, &totB);
#pragma unroll
        for (int x = 0; x < PLAN_E; x++) {
            const uint32_t p = b0 + PLAN_E * threadIdx.x + x;
            if (c[x] < CLS_WIDE) d.order[base[c[x]] + runc[c[x]] + (uint32_t)((ea >> (16 * c[x])) & 0xffffu)] = p;
            else if (c[x] == CLS_WIDE) d.order[base[CLS_WIDE] + runc[CLS_WIDE] + (uint32_t)eb] = p;
            ea += c[x] < CLS_WIDE ? 1ull << (16 * c[x]) : 0ull;
            eb += c[x] == CLS_WIDE ? 1ull : 0ull;
        }
#pragma unroll
        for (int k = 0; k < CLS_WIDE; k++) runc[k] += (uint32_t)((totA >> (16 * k)) & 0xffffu);
        runc[CLS_WIDE] += (uint32_t)totB;
    }
    __syncthreads();                             // `order` is complete and visible to the block
    // Record offsets follow `order`: the records of a class are contiguous and the 64 pairs of a
    // fill wave are neighbours, so a wave's record region is at most 64 pairs long (k_fill addresses
    // it with 32-bit byte offsets from the region's start).
    const uint32_t n_ok = base[NCLS];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_ok; b0 += PLAN_E * blockDim.x) {
        uint32_t q[PLAN_E], rd[PLAN_E];
        uint64_t lr = 0;
#pragma unroll
        for (int x = 0; x < PLAN_E; x++) {
            const uint32_t i = b0 + PLAN_E * threadIdx.x + x;
            q[x] = i < n_ok ? d.order[i] : 0xffffffffu;
            rd[x] = i < n_ok ? d.recdw[q[x]] + 16u : 0u;
            lr += rd[x];
        }
        uint64_t tot;
        uint64_t er = carry + block_excl_scan(lr, sm, &tot);
#pragma unroll
        for (int x = 0; x < PLAN_E; x++) {
            if (q[x] != 0xffffffffu) d.rec_off[q[x]] = er + 4u;
            er += rd[x];
        }
        carry += tot;
    }
    uint64_t tot;
    block_excl_scan(cells, sm, &tot);
    if (threadIdx.x == 0) { d.plan->total_rec_dwords = carry; d.plan->total_cells = tot; d.plan->pool_used = 0; d.plan->pool_fail = 0; }
}

// ---------------------------------------------------------------- k_fill
// One LANE per pair, 64 pairs per wave.  Lanes are NOT in row lock step: every lane walks its own
// guide stream.  One step of the (wave-uniform) loop lets each lane
//   1. if its current row is finished, take the step's word as the next ROW word: new width / shift / query base,
//      write the INFW words of three entering columns; or, if the row still has entering columns
//      pending, take it as an EXT word and write eight more; (otherwise it is a continuation word)
//   2. compute up to CAP cells of its current row (rows wider than CAP continue next step).
// A lane with a 21-cell row and a lane with a 60-cell row therefore cost 2 and 4 steps, instead of
// both paying for the widest row of the wave.
//
// LDS holds one DP row per lane, slot-major: word (slot s, lane l) at dword s*64 + l (conflict
// free whatever the per-lane slot).  Slots are RELATIVE to the row's first column: cell k of a row
// reads its vertical predecessor from slot k + shift (the previous row's layout) and overwrites
// slot k; the read of a slot always precedes the write that destroys it, so one row buffer is
// enough.  Slots at and right of shift + width hold nothing a valid cell reads: the entering-column
// writes and the cell groups run unmasked past the row end (whole groups of 8 cells, whole words
// of 3 / 8 entering columns), which is why a pair's widest shift + width must stay SPAN_MARGIN
// below the ring size.
//
// Guide words: a lane takes exactly ONE word of its pair's stream per step -- the ROW word of a new row, an EXT word of
// entering columns, or, while a row wider than CAP is still being computed, a continuation word that k_guide put there
// for the purpose (zero, never looked at).  Lanes are out of step in ROWS, but in step in stream WORDS: word number `it`
// of every lane's stream belongs to step `it`, so which register holds this step's word is the same for the whole wave
// and is fixed by the loop counter, not selected per lane.  The loop takes four steps per turn on one 16-byte group of
// the 64-byte sector held in R0..R3; the sector after it waits in S0..S3, requested by the whole wave at once a full
// sector (sixteen steps) before its first word is needed, with ordinary loads whose arrival the compiler tracks.  A
// sector is fetched from memory once and used up by its lane (with one 16-byte load per four words the L2 had to keep a
// 128-byte line per lane for 32 steps, which it cannot do for 130 k lanes; measured at two waves per SIMD everywhere: the
// guide loads alone made the launch 45 % longer).  (Before, a lane took a word only when it had a use for one: the unread
// words shifted through a group of four registers refilled from R0..R3 by per-lane selects, and a lane asked for its next
// sector when it had used one up, from an inline-asm block that narrowed EXEC: about 60 of 575 instructions per step.)
// Record dwords are stored directly.  No LDS is spent on the stream: LDS is what limits the waves per CU.
constexpr int LDS_TAB = 16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Affine variant (AFF): the match state M and the insertion-gap state I of the previous row both
// live in LDS (two rings), the deletion-gap state D travels along the row in a register.  Words
// are score << 6 | arrow << 3 | low bits; the gap states put their own open/extend bit in bit 2
// (I) and bit 1 (D).  Records hold one byte per cell: M arrow (bits 3..5), I bit, D bit.
//
// Quality-value variant (QV; linear gaps, and affine gaps with the constants of the gap states moved the same way): the
// costs of a row come from the cost profile of its query base (Dev.prof: five substitution costs, the insertion cost,
// five deletion costs, each taken within +-127, affine +-63).  A lane fetches the 22 bytes of its next row when it takes
// a row, and at the end of every step every lane turns what it has fetched into two byte tables for v_perm_b32
// (substitution and deletion cost by target code) and one addend (insertion), which the next row takes over.  A table
// byte cannot be negative, so the deletion costs carry a bias of 128 like the substitution costs: the bias of a cell is
// 128 x its COLUMN here (a diagonal move and a move along the row add it, a move to the next row does not), where the
// matrix scorer's words carry 128 x the row.
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int LIM>
__device__ __forceinline__ uint32_t qv_bytes(uint32_t two)       // two int16 costs -> each within +-LIM, plus 128
{
    s16x2 x = __builtin_bit_cast(s16x2, two);
    x = __builtin_elementwise_min(__builtin_elementwise_max(x, (s16x2)(short)-LIM), (s16x2)(short)LIM) + (s16x2)(short)128;
    return __builtin_bit_cast(uint32_t, x);
}

//
// LOCAL (linear gaps; gba_scoring.align_type GBA_ALIGN_LOCAL as in k_fill_wide): every cell may begin an
// alignment (a fourth candidate of score 0 with the arrow code 3, last in the tie-break), row 0 has no deletion chain,
// and the lane keeps the cheapest cell of its pair, the first in row-major order among equals: per step the minimum of
// score | cell number over the step's cells OF THE ROW (the groups run past a row's end, and what they compute there may
// be a real cell outside the band), compared with the best so far once the row bias is taken off.  The cell's row,
// column, first column of its row and the meta dword of that row go to Dev::end_* for the traceback.  With a cost profile
// (QV: the CCS call site runs the rule with the quality-value scorer) the bias is 128 x the cell's column, so the start
// candidate and the key are made per cell, relative to the step's first column.
template <bool AFF, bool QV = false, bool LOCAL = false>
__global__ __launch_bounds__(64) void k_fill(Dev d, FillConst fc, uint32_t cls, uint32_t slots)
{
    static_assert(!LOCAL || !AFF, "the Local end rule: linear gaps");
    extern __shared__ uint32_t smem[];
    // Claims 200 of a SIMD's 512 VGPRs per wave, so that at most two fill waves (of any class) share
    // a SIMD and a wave of another kernel still fits beside them.  LDS would admit a third fill wave
    // on some SIMDs while others hold one, and a launch lasts as long as its busiest SIMD: with the
    // claim the bulk launch of 106 496 10-kb pairs takes 24.6 ms instead of 30.5 ms.
    asm volatile("" ::: "v199");
    constexpr int SH = AFF ? 6 : 5;                     // score shift of a cell word
    constexpr uint32_t TAGS = (1u << SH) - 1u;
    constexpr int CAP = AFF ? CAP_AFFINE : CAP_LINEAR;
    constexpr int NAR = AFF ? CAP / 4 : CAP / 16;       // record dwords of one step
    const int lane = threadIdx.x;
    const uint32_t SL = slots + PAD_SLOTS;              // slots of one ring
    const uint32_t IOFF = SL * 64;                      // dword offset of the I ring
    uint32_t *tab = smem, *lds = smem + LDS_TAB;
    if (lane < 10) tab[lane] = fc.mrow[lane];
    __syncthreads();
    const uint32_t cnt = d.plan->cls_count[cls], cbase = d.plan->cls_base[cls];
    const uint32_t idx = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= cnt) return;
    const bool live = idx < cnt;
    const uint32_t p = d.order[cbase + (live ? idx : cnt - 1)];
    const uint32_t Q = d.Q[p];
    const uint32_t fslot = d.T[p] - d.lo_last[p];       // slot of the end cell (Q, T) in the last row
    const uint32_t *st = d.stream + d.word_off[p];
    const uint32_t wcap = d.wcap[p];
    // record stores: a wave-uniform base (the wave's first pair has the lowest offset, pairs of a
    // class are ordered by index) plus a 32-bit byte offset per lane
    const uint64_t rbase = d.rec_off[d.order[cbase + blockIdx.x * 64]];
    char *recb = (char *)(d.rec + rbase);
    uint32_t rob = (uint32_t)(d.rec_off[p] - rbase) * 4u;

    // stream words: R0..R3 hold the sector in use, S0..S3 the one after it, requested a whole sector (sixteen steps) before
    // its first word is looked at.  Streams start 64-byte aligned and are whole sectors long; a lane that has come to the
    // end of its stream (it has finished by then) reads its last sector again.
    u32x4 R0 = *(const u32x4 *)(st), R1 = *(const u32x4 *)(st + 4), R2 = *(const u32x4 *)(st + 8), R3 = *(const u32x4 *)(st + 12);
    uint32_t ws = min(16u, wcap - 16u);      // first word of the sector in S
    u32x4 S0 = *(const u32x4 *)(st + ws), S1 = *(const u32x4 *)(st + ws + 4), S2 = *(const u32x4 *)(st + ws + 8), S3 = *(const u32x4 *)(st + ws + 12);
    // No pair needs more steps than its stream has words; the loop stops there whatever the stream holds.
    uint32_t max_steps = wcap;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_steps = max(max_steps, (uint32_t)__shfl_xor((int)max_steps, o, 64));

    // QV: the profile of the pair's first DP row with a query base (row 1), the constants of row 0 (deletions in front of
    // the first base cost fc.del), the arrow codes without a cost
    const char *prow = QV ? (const char *)(d.prof + (d.q_off[p] + d.qS[p]) * 11) : nullptr;     // row r >= 1: 22 bytes at prow + 22 (r - 1)
    uint32_t pa = 0, pb = 0, pc = 0, pd = 0, pe = 0, pf = 0;
    if constexpr (QV) {
        if (Q >= 1u) {
            pa = *(const u32u1 *)(prow); pb = *(const u32u1 *)(prow + 4); pc = *(const u32u1 *)(prow + 8);
            pd = *(const u32u1 *)(prow + 12); pe = *(const u32u1 *)(prow + 16); pf = *(const uint16_t *)(prow + 20);
        }
    }
    const uint32_t del0 = (uint32_t)(fc.del + 128) & 255u;
    const int32_t codeU8 = fc.cU - ((fc.ins + 128) << SH), codeL8 = fc.cL - (fc.del << SH);
    // affine gaps: a step of the insertion-gap state is a move to the next row (no bias here), one of the deletion-gap state a move along the row
    const int32_t cIO = fc.cIO - (QV ? 128 << SH : 0), cIE = fc.cIE - (QV ? 128 << SH : 0), cDO = fc.cDO + (QV ? 128 << SH : 0), cDE = fc.cDE + (QV ? 128 << SH : 0);
    uint32_t dlo = 0, dhi = 0, ndlo = 0, ndhi = 0;      // deletion costs + 128 by target code, as mlo / mhi hold the substitution costs; the next row's
    int32_t cUq = 0, ncUq = 0;                          // insertion cost << SH | arrow code
    // LOCAL: the current row's number, first column and start candidate; the cheapest cell so far (score << 5 without the
    // row bias, row, column, first column of its row, meta dword of its row; fresh: the row of the cell has not ended yet)
    const uint32_t rob0 = rob;
    uint32_t rrow = 0, lo_abs = 0, bi = 0, bj = 0, blo = 0, bptr = 0;
    int32_t startw = 3 << 3, bsc = 0;
    bool fresh = false;
    uint32_t rows_left = live ? Q + 1 : 0;              // rows not yet started
    uint32_t W = 0, dsh = 0, k0 = 0, Wp = 0;            // current row: width, shift, cells done; previous width
    uint32_t ext_left = 0, newslot = 0;
    uint32_t diag = INFW, left = INFW, leftD = INFW, mlo = 0, mhi = 0, meta = 0;
    bool first = true, finished = !live;
    uint32_t final_word = INFW;
    // the matrix row of the next word's query base is fetched one step ahead
    uint32_t nmlo = 0, nmhi = 0;
    if constexpr (!QV) { nmlo = tab[((R0.x >> 10) & 7u) * 2]; nmhi = tab[((R0.x >> 10) & 7u) * 2 + 1]; }

    // One step: `word` is this step's word of the stream, `next` the one after it.
    auto step = [&](const uint32_t word, const uint32_t next) __attribute__((always_inline)) {
            // ---- 1. word pop
            const bool pop_ext = !finished && ext_left > 0u;
            const bool pop_row = !finished && ext_left == 0u && k0 >= W && rows_left > 0u;
            if (pop_row) {
                rows_left--;
                Wp = W;
                W = word & ROW_W_MASK; dsh = (word >> 13) & 1023u;
                if constexpr (LOCAL) { rrow = Q - rows_left; lo_abs += dsh; startw = (int32_t)(rrow * 4096u) + (3 << 3); }
                if constexpr (QV) {
                    const uint32_t r = Q - rows_left;           // the row taken now
                    mlo = nmlo; mhi = nmhi; dlo = ndlo; dhi = ndhi; cUq = ncUq;
                    if (r == 0u) {                              // (row 1's costs stay where they are)
                        dlo = del0 * 0x01010100u; dhi = del0 * 0x01010101u;
                        if constexpr (AFF) { mlo = fc.mrow[8]; mhi = fc.mrow[9]; }     // the boundary row's query code is N: cell (0, 0) costs 0
                    }
                    else if (r < Q) {                           // row r + 1 takes the costs of query base r
                        const char *nx = prow + 22u * r;
                        pa = *(const u32u1 *)(nx); pb = *(const u32u1 *)(nx + 4); pc = *(const u32u1 *)(nx + 8);
                        pd = *(const u32u1 *)(nx + 12); pe = *(const u32u1 *)(nx + 16); pf = *(const uint16_t *)(nx + 20);
                    }
                } else { mlo = nmlo; mhi = nmhi; }
                const uint32_t tot = dsh + W;
                const uint32_t nnew = tot > Wp ? tot - Wp : 0u;
                uint32_t *ns = lds + Wp * 64 + lane;
                ns[0] = INFW | ((word >> 23) & 7u); ns[64] = INFW | ((word >> 26) & 7u); ns[128] = INFW | (word >> 29);
                if constexpr (AFF) { ns[IOFF] = INFW; ns[IOFF + 64] = INFW; ns[IOFF + 128] = INFW; }
                ext_left = nnew > 3u ? (nnew + 4u) >> 3 : 0u;
                newslot = Wp + 3u;
                k0 = 0;
                left = INFW; leftD = INFW;
                meta = word & ROW_META_MASK;
            }
            if (pop_ext) {
                ext_left--;
                uint32_t *ns = lds + newslot * 64 + lane;
    #pragma unroll
                for (int j = 0; j < 8; j++) {
                    ns[j * 64] = INFW | ((word >> (3 * j)) & 7u);
                    if constexpr (AFF) ns[IOFF + j * 64] = INFW;
                }
                newslot += 8u;
            }
            // ---- 2. up to CAP cells of the current row, in groups of 8
            const bool did = !finished && ext_left == 0u && k0 < W;
            const uint32_t rem = W - k0;
            uint32_t ar[NAR];
    #pragma unroll
            for (int x = 0; x < NAR; x++) ar[x] = 0;
            int32_t stepmin = 0x7fffffff;               // LOCAL: cheapest cell of the row in this step, score | cell number within the step
            const int32_t sbase = (int32_t)((lo_abs + k0) * 4096u) + (3 << 3);      // LOCAL with QV: the start candidate of the step's first cell
            // One LDS round trip for everything the step's first cells need: the diagonal predecessor of
            // a new row's first cell and the vertical predecessors of the first 8 cells (lanes without
            // cells this step read slots 0..7, which always exist).
            const uint32_t *rd = lds + (did ? dsh + k0 : 0u) * 64 + lane;
            uint32_t up[2][8], ui[2][8];
            {
                const uint32_t dv = lds[(max(dsh, 1u) - 1u) * 64 + lane];
    #pragma unroll
                for (int x = 0; x < 8; x++) { up[0][x] = rd[x * 64]; if constexpr (AFF) ui[0][x] = rd[IOFF + x * 64]; }
                if (pop_row) {
                    // INFW when the diagonal predecessor lies outside the previous row
                    diag = dsh - 1u < Wp ? dv : INFW;
                    if (first) {
                        // row 0: linear starts from "one deletion before column 0" so that cell j costs j
                        // deletions; affine makes the diagonal candidate of cell (0,0) cost exactly 0
                        if constexpr (AFF) diag = (uint32_t)fc.diag0;
                        else if constexpr (!LOCAL) left = (uint32_t)(-(fc.del + (QV ? 128 : 0)) * 32);       // (LOCAL: row 0 has no deletion chain)
                    }
                    first = false;
                }
            }
            if (did) {
                uint32_t *wr = lds + k0 * 64 + lane;
    #pragma unroll
                for (int G = 0; G < CAP / 8; G++) {
                    if (rem > (uint32_t)(8 * G)) {
                        if (G + 1 < CAP / 8 && rem > (uint32_t)(8 * G + 8)) {
    #pragma unroll
                            for (int x = 0; x < 8; x++) {
                                up[(G + 1) & 1][x] = rd[(8 * G + 8 + x) * 64];
                                if constexpr (AFF) ui[(G + 1) & 1][x] = rd[IOFF + (8 * G + 8 + x) * 64];
                            }
                        }
    #pragma unroll
                        for (int x = 0; x < 8; x++) {
                            const int k = 8 * G + x;
                            const uint32_t u_ = up[G & 1][x];
                            const uint32_t tcd = u_ & 7u;
                            const uint32_t sb = __builtin_amdgcn_perm(mhi, mlo, tcd);   // selector bytes 1..3 = 0 read the zero byte
                            const int32_t m = (int32_t)((sb << SH) + diag);
                            const int32_t u = (int32_t)u_ + (QV ? cUq : fc.cU);
                            int32_t l;
                            if constexpr (QV) l = (int32_t)((__builtin_amdgcn_perm(dhi, dlo, tcd) << SH) + left) + codeL8;
                            else l = (int32_t)left + fc.cL;
                            if constexpr (!AFF) {
                                uint32_t best = (uint32_t)min(m, min(u, l));
                                if constexpr (LOCAL) {
                                    // QV: sbase carries the bias of the step's first column, cell k has k x 4096 more
                                    best = (uint32_t)min((int32_t)best, QV ? sbase + k * 4096 : startw);
                                    const int32_t key = (int32_t)((best & ~TAGS) | (uint32_t)k) - (QV ? k * 4096 : 0);
                                    stepmin = min(stepmin, (uint32_t)k < rem ? key : 0x7fffffff);
                                }
                                const uint32_t cw = (u_ & TAGS) | (best & ~TAGS);
                                wr[k * 64] = cw;
                                // arrow = bits 3..4 of best -> record bits 2k..2k+1: a field extract and a shift-or with
                                // inline constants (left to itself the compiler shifts first and masks with a 32-bit literal,
                                // which costs a third instruction for every second cell)
                                uint32_t a2;
                                asm("v_bfe_u32 %0, %1, 3, 2" : "=v"(a2) : "v"(best));
                                if ((k & 15) == 0) ar[k >> 4] = a2;
                                else asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(ar[k >> 4]) : "v"(a2), "s"((uint32_t)((k & 15) * 2)));
                                left = cw;
                            } else {
                                // gap states: open from the match state (preferred on ties) or extend
                                const int32_t io = (int32_t)(u_ & ~TAGS) + cIO, ie = (int32_t)ui[G & 1][x] + cIE;
                                const uint32_t ig = (uint32_t)min(io, ie);                 // bit 2: 1 = extend
                                const int32_t dop = (int32_t)left + cDO, de = (int32_t)leftD + cDE;
                                const uint32_t dg = (uint32_t)min(dop, de);                // bit 1: 1 = extend
                                const int32_t dc = (int32_t)dg + (3 << 3), ic = (int32_t)ig + (4 << 3);
                                const uint32_t best = (uint32_t)min(min(m, min(u, l)), min(dc, ic));
                                const uint32_t lc = best & ~TAGS;
                                wr[k * 64] = lc | tcd;
                                wr[IOFF + k * 64] = ig & ~TAGS;
                                // record byte: M arrow (bits 3..5) | I bit (bit 2) | D bit (bit 1)
                                const uint32_t rbyte = (best & 0x38u) | (ig & 4u) | (dg & 2u);
                                ar[k >> 2] |= rbyte << ((k & 3) * 8);
                                left = lc; leftD = dg & ~TAGS;
                            }
                            diag = u_;
                        }
                    }
                }
            }
            if constexpr (!QV) { nmlo = tab[((next >> 10) & 7u) * 2]; nmhi = tab[((next >> 10) & 7u) * 2 + 1]; }
            else {
                // The tables of the lane's next row from what it fetched when it took the current one -- here, by every lane
                // in every step (a lane that took no row this step gets what it had), and not where the next row is taken:
                // a use there would wait for this step's record store as well, which is younger than the fetch.
                constexpr int LIM = AFF ? 63 : 127;
                const uint32_t a = qv_bytes<LIM>(pa), b = qv_bytes<LIM>(pb), c = qv_bytes<LIM>(pc), e = qv_bytes<LIM>(pd), f = qv_bytes<LIM>(pe), g = qv_bytes<LIM>(pf);
                // byte 0 of the low word stays zero (the lookup's unused selector bytes read it); codes 5..7 read N's cost
                nmlo = __builtin_amdgcn_perm(b, a, 0x0402000cu); nmhi = __builtin_amdgcn_perm(c, b, 0x04040402u);
                ndlo = __builtin_amdgcn_perm(f, e, 0x0402000cu); ndhi = __builtin_amdgcn_perm(g, f, 0x04040402u);
                ncUq = (int32_t)((c >> 16) << SH) + (codeU8 - (128 << SH));
            }
            if (did) {
                // record: the arrow dwords of this step (stored whole; the tail past the row end is
                // overwritten by what follows), then the row's meta dword once the row is done
                if constexpr (AFF) {
    #pragma unroll
                    for (int x = 0; x + 4 <= NAR; x += 4) *(u32x4u *)(recb + rob + 4 * x) = u32x4u{ar[x], ar[x + 1], ar[x + 2], ar[x + 3]};
                    if constexpr (NAR % 4 == 2) *(u32x2u *)(recb + rob + 4 * (NAR - 2)) = u32x2u{ar[NAR - 2], ar[NAR - 1]};
                    rob += ((min(rem, (uint32_t)CAP) + 3u) >> 2) * 4u;
                }
                k0 += CAP;
                if constexpr (!AFF) {
                    // one store per step: a row that ends here appends its meta dword to the arrow dwords
                    // (the dword behind a one-dword row is overwritten by the next store)
                    // (a row that goes on stores a dword too many as well: its next step overwrites it)
                    const bool one = k0 >= W && rem <= 16u;
                    *(u32x3u *)(recb + rob) = u32x3u{ar[0], one ? meta : ar[1], meta};
                    rob += k0 >= W && rem > 16u ? 12u : 8u;
                }
                if constexpr (LOCAL) {
                    const int32_t sc = (int32_t)((uint32_t)stepmin & ~TAGS) - (int32_t)((QV ? lo_abs + (k0 - (uint32_t)CAP) : rrow) * 4096u);
                    if (stepmin != 0x7fffffff && sc < bsc) { bsc = sc; bi = rrow; bj = lo_abs + (k0 - (uint32_t)CAP) + ((uint32_t)stepmin & TAGS); blo = lo_abs; fresh = true; }
                    if (k0 >= W && fresh) { bptr = (rob - 4u - rob0) >> 2; fresh = false; }      // the row's meta dword is the last one stored
                }
                if (k0 >= W) {
                    if constexpr (AFF) {
                        *(uint32_t *)(recb + rob) = meta;
                        rob += 4u;
                    }
                    if (rows_left == 0u) {
                        final_word = lds[fslot * 64 + lane];
                        finished = true;
                    }
                }
            }
    };
    // Four steps per turn on one 16-byte group of the sector; which group is the same for every lane (the loop counter
    // says it), so the words are named registers, no lane selects anything and the whole wave asks for the next sector
    // at once, every sixteenth step.
    for (uint32_t it = 0; it < max_steps && __any(!finished); it += 4) {
        const uint32_t g = (it >> 2) & 3u;
        u32x4 A; uint32_t nf;
        if (g == 0u) { A = R0; nf = R1.x; } else if (g == 1u) { A = R1; nf = R2.x; } else if (g == 2u) { A = R2; nf = R3.x; } else { A = R3; nf = S0.x; }
        step(A.x, A.y); step(A.y, A.z); step(A.z, A.w); step(A.w, nf);
        if (g == 3u) {
            R0 = S0; R1 = S1; R2 = S2; R3 = S3;
            ws = min(ws + 16u, wcap - 16u);
            const uint32_t *src = st + ws;
            S0 = *(const u32x4 *)(src); S1 = *(const u32x4 *)(src + 4); S2 = *(const u32x4 *)(src + 8); S3 = *(const u32x4 *)(src + 12);
        }
    }
    if constexpr (LOCAL) {
        if (live) {
            if (!finished) d.status[p] = GBA_PAIR_UNREACHABLE;
            else if (bsc >= 0) { d.status[p] = GBA_PAIR_NO_ALIGNMENT; d.dp_score[p] = 0; }
            else { d.end_i[p] = bi; d.end_j[p] = bj; d.end_lo[p] = blo; d.end_ptr[p] = bptr; d.dp_score[p] = bsc >> SH; }
        }
        return;
    }
    if (live) {
        const int32_t v = (int32_t)(final_word & ~TAGS) - (int32_t)(QV ? d.T[p] : Q) * (128 << SH);
        if (v >= UNREACH || !finished) d.status[p] = GBA_PAIR_UNREACHABLE;
        d.dp_score[p] = v >> SH;
    }
}

// ---------------------------------------------------------------- k_fill_wide
// Wave-per-pair fill for wide bands (linear gaps): one LANE per COLUMN, 64 columns per chunk, the
// chunks of a row in sequence.  The previous row sits in an LDS ring of the wave (slot = column -
// first column of the row, as in k_fill); a cell's vertical predecessor is slot k + shift, its
// diagonal predecessor the value the lane to the left read.  The dependency along the row
// (deletions) is a min-plus scan: with S(j) the best score of column j over its diagonal and
// vertical moves, the left candidate of column k is  min over j < k of S(j) + (k - j) * del,
// i.e. an exclusive prefix minimum of S(j) - j * del across the lanes plus k * del, with the last
// column of the previous chunk carried in.  Cell words, tie-break (the arrow code rides in the
// compared word) and records are exactly those of k_fill.
//
// Guide words are staged through an LDS ring of the wave, refilled 64 words at a time.
// Two instances: NB = 4 chunks of 64 columns in flight and a ring for rows of any width the guide
// stream can describe (the widest class), and NB = 1 with a small ring for shift + width <= WN_SPAN
// (2.4 KB of LDS per wave instead of 11.5 KB: eight waves per SIMD instead of three; this is the
// instance that takes whole batches too small to fill the GPU one lane per pair).
constexpr int WIDE_NB = 4;                   // chunks of 64 columns in flight
constexpr int WIDE_RING = 2048 + 64 * WIDE_NB;          // shift + width <= 2046, lanes run up to 63 columns past the row end
constexpr int WIDE_SBUF = 512;                // guide words buffered per wave (a row has at most 1 + 128)
constexpr int WN_RING = 192, WN_SBUF = 128;   // the small instance: WN_SPAN + 63 slots; a row has at most 1 + 16 words

__device__ __forceinline__ int32_t wave_scan_min(int32_t v, int32_t identity)
{
    v = min(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));
    return v;
}

//
// QV: the costs come from a quality-value scorer's profile (Dev.prof: per query base the substitution cost against each
// target code, the insertion cost, the deletion cost by target code; taken within +-127) instead of matrix, ins, del:
// GuidedAlign(..., idsScoreFn, ...) of /root/reference/iblasr/BlasrUtilsImpl.hpp:705-717.  A row's eleven values are
// fetched one row ahead by eleven lanes and handed round through LDS.  The deletion cost now differs from column to
// column, so the cost of reaching column k from column j along the row is P(k) - P(j) with P the prefix sum of the
// row's deletion costs (one more wave scan per chunk), where the fixed cost had (k - j) * del.
//
// LOCAL (gba_scoring.align_type GBA_ALIGN_LOCAL: GuidedAlign(..., Local, ...), /root/reference/iblasr/BlasrAlignImpl.hpp:1418-1421):
// every cell may also begin the alignment at cost 0 -- one more candidate, a word of score 0 (plus the row's bias) with the
// arrow code 3, folded into the moves from the previous row before the scan, so that the deletions chained along the row
// may start from it too -- and the alignment ends in the cheapest cell of the band: every lane keeps the cheapest of
// its own cells (the earliest on ties), the wave picks the first of them in row-major order at the end and leaves its
// row, column and record position for the traceback.
template <int NB, bool QV = false, bool LOCAL = false>
__global__ __launch_bounds__(64) void k_fill_wide(Dev d, FillConst fc, uint32_t cls)
{
    __shared__ int32_t ptab[16];
    constexpr int WIDE_NB = NB;
    constexpr int WIDE_SBUF = NB == 1 ? WN_SBUF : ::WIDE_SBUF;
    __shared__ uint32_t ring[NB == 1 ? WN_RING : WIDE_RING];
    __shared__ uint32_t sbuf[WIDE_SBUF];
    __shared__ uint32_t tab[LDS_TAB];
    constexpr uint32_t TAGS = 31u;
    constexpr int32_t BIG = 0x60000000;                 // identity of the prefix minimum (no overflow when offsets are added)
    const int lane = threadIdx.x;
    if (lane < 10) tab[lane] = fc.mrow[lane];
    const uint32_t p = d.order[d.plan->cls_base[cls] + blockIdx.x];
    const uint32_t Q = d.Q[p];
    const uint32_t fslot = d.T[p] - d.lo_last[p];       // slot of the end cell (Q, T) in the last row
    const uint32_t *st = d.stream + d.word_off[p];
    const uint32_t wcap = d.wcap[p];
    uint32_t *rec = d.rec + d.rec_off[p];
    const int32_t D = fc.del * 32;                      // one deletion, in cell-word units
    const int32_t codeL = fc.cL - D;                    // arrow code of a deletion (<< 3)
    uint32_t spos = 0, sload = 0;                       // guide words consumed / loaded
    uint32_t W = 0, Wp = 0, dsh = 0, roff = 0;
    const int16_t *prow = QV ? d.prof + (d.q_off[p] + d.qS[p]) * 11 : nullptr;      // row r >= 1 uses prow[(r - 1) * 11 ..]
    int32_t pnext = QV && Q >= 1u && lane < 11 ? (int32_t)prow[lane] : 0;
    int32_t bsc = 0;                                    // LOCAL: this lane's cheapest cell: its score (<< 5, bias removed), ...
    uint32_t bi = 0, bj = 0, blo = 0, bptr = 0, lo = 0; // ... row, column, first column and meta dword of its row; first column of the current row
    __syncthreads();
    for (uint32_t r = 0; r <= Q; r++) {
        if constexpr (QV) {
            if (lane < 11) ptab[lane] = min(max(pnext, -127), 127);              // read after the barriers below
            if (r + 1u <= Q && r >= 1u && lane < 11) pnext = (int32_t)prow[(size_t)r * 11 + lane];
        }
        // keep at least half the buffer's worth of guide words buffered (or everything that is left)
        while (sload - spos < (uint32_t)WIDE_SBUF / 2 && sload < wcap) {
            const uint32_t x = sload + lane;
            const uint32_t v = x < wcap ? st[x] : 0u;
            sbuf[x & (WIDE_SBUF - 1)] = v;
            sload += 64;
        }
        __syncthreads();
        const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)sbuf[spos & (WIDE_SBUF - 1)]);
        Wp = W;
        W = word & ROW_W_MASK; dsh = (word >> 13) & 1023u;
        const uint32_t qcode = (word >> 10) & 7u;
        const uint32_t mlo = tab[qcode * 2], mhi = tab[qcode * 2 + 1];
        const uint32_t tot = dsh + W;
        const uint32_t nnew = tot > Wp ? tot - Wp : 0u;
        // entering columns: INFW words carrying the column's base code
        if (nnew <= 3u) {                               // nearly every row: the entering codes sit in the row word itself
            if ((uint32_t)lane < nnew) ring[Wp + lane] = INFW | ((word >> (23u + 3u * (uint32_t)lane)) & 7u);
            spos += 1u + cont_words(W, CAP_LINEAR);
        } else {
            for (uint32_t x = lane; x < nnew; x += 64) {
                uint32_t code;
                if (x < 3u) code = (word >> (23u + 3u * x)) & 7u;
                else {
                    const uint32_t e = (x - 3u) >> 3, rr = (x - 3u) & 7u;
                    code = (sbuf[(spos + 1u + e) & (WIDE_SBUF - 1)] >> (3u * rr)) & 7u;
                }
                ring[Wp + x] = INFW | code;
            }
            spos += 1u + ((nnew + 4u) >> 3) + cont_words(W, CAP_LINEAR);
        }
        __syncthreads();
        // carries into the first chunk: the diagonal predecessor of column 0 and "the cell left of it"
        uint32_t cdiag = dsh >= 1u && dsh - 1u < Wp ? ring[dsh - 1u] : INFW;
        int32_t hc = r == 0 && !LOCAL ? -D : (int32_t)INFW;       // row 0: column j costs j deletions (LOCAL: every cell of it begins an alignment)
        const uint32_t nd = (W + 15u) >> 4;
        lo += dsh;
        const int32_t rbias = (int32_t)(r * 4096u);     // what the substitution table's bias has added to every cell of this row
        const int32_t startw = rbias + (3 << 3);        // LOCAL: "the alignment begins here"
        // WIDE_NB chunks of 64 columns at a time: their loads, candidate words and local scans are
        // independent of each other; only the carries (scalar) link a chunk to the one before it
        for (uint32_t c0 = 0; c0 < W; c0 += 64 * WIDE_NB) {
            uint32_t up[WIDE_NB];
            int32_t xv[WIDE_NB], incl[WIDE_NB], pd[WIDE_NB];
#pragma unroll
            for (int b = 0; b < WIDE_NB; b++) up[b] = ring[c0 + 64 * b + lane + dsh];     // past the row end: slots that exist, values nobody uses
            __syncthreads();                            // every lane has read its slots before any is overwritten
#pragma unroll
            for (int b = 0; b < WIDE_NB; b++) {
                const uint32_t diag = (uint32_t)__builtin_amdgcn_update_dpp((int32_t)cdiag, (int32_t)up[b], 0x138, 0xf, 0xf, false);
                cdiag = (uint32_t)__builtin_amdgcn_readlane((int32_t)up[b], 63);
                const uint32_t tcd = up[b] & 7u;
                if constexpr (QV) {
                    const uint32_t ti = min(max(tcd, 1u), 5u) - 1u;
                    const int32_t subc = ptab[ti], insc = ptab[5], delc = r == 0u ? fc.del : ptab[6 + ti];
                    const int32_t m = (int32_t)(((uint32_t)(subc + 128) << 5) + diag);
                    const int32_t u = (int32_t)up[b] + fc.cU + (insc - fc.ins) * 32;
                    xv[b] = LOCAL ? min(min(m, u), startw) : min(m, u);
                    pd[b] = (int32_t)wave_scan_sum((uint32_t)(delc * 32));          // P(k): the row's deletion costs up to and including column k of the chunk
                    incl[b] = wave_scan_min((int32_t)((uint32_t)xv[b] & ~TAGS) - pd[b], BIG);
                } else {
                    const uint32_t sb = __builtin_amdgcn_perm(mhi, mlo, tcd);
                    const int32_t m = (int32_t)((sb << 5) + diag);
                    const int32_t u = (int32_t)up[b] + fc.cU;
                    xv[b] = LOCAL ? min(min(m, u), startw) : min(m, u);     // best of the two moves from the previous row
                    incl[b] = wave_scan_min((int32_t)((uint32_t)xv[b] & ~TAGS) - lane * D, BIG);
                }
            }
#pragma unroll
            for (int b = 0; b < WIDE_NB; b++) {
                if (c0 + 64 * b < W) {
                    const int32_t excl = __builtin_amdgcn_update_dpp(BIG, incl[b], 0x138, 0xf, 0xf, false);
                    const int32_t lc = (QV ? min(excl, hc) + pd[b] : min(excl + lane * D, hc + (lane + 1) * D)) + codeL;
                    const uint32_t best = (uint32_t)min(xv[b], lc);
                    ring[c0 + 64 * b + lane] = (up[b] & TAGS) | (best & ~TAGS);
                    if constexpr (LOCAL) {
                        const uint32_t col = c0 + 64u * (uint32_t)b + (uint32_t)lane;
                        const int32_t sc = (int32_t)(best & ~TAGS) - rbias;
                        if (col < W && sc < bsc) { bsc = sc; bi = r; bj = lo + col; blo = lo; bptr = roff + nd; }
                    }
                    // score of the chunk's last column: its own best over this chunk's columns, or the carry
                    if constexpr (QV) hc = min(__builtin_amdgcn_readlane(incl[b], 63), hc) + __builtin_amdgcn_readlane(pd[b], 63);
                    else hc = min(__builtin_amdgcn_readlane(incl[b], 63) + 63 * D, hc + 64 * D);
                    // arrows: 2 bits per column, 16 columns per dword
                    int32_t ab = (int32_t)(((best >> 3) & 3u) << (2 * (lane & 15)));
                    ab |= __builtin_amdgcn_update_dpp(0, ab, 0x111, 0xf, 0xf, false);
                    ab |= __builtin_amdgcn_update_dpp(0, ab, 0x112, 0xf, 0xf, false);
                    ab |= __builtin_amdgcn_update_dpp(0, ab, 0x114, 0xf, 0xf, false);
                    ab |= __builtin_amdgcn_update_dpp(0, ab, 0x118, 0xf, 0xf, false);
                    const uint32_t dwi = ((c0 + 64 * b) >> 4) + ((uint32_t)lane >> 4);
                    if ((lane & 15) == 15 && dwi < nd) rec[roff + dwi] = (uint32_t)ab;
                }
            }
        }
        if (lane == 0) rec[roff + nd] = word & ROW_META_MASK;
        roff += nd + 1u;
        __syncthreads();
    }
    if constexpr (LOCAL) {
        // the cheapest cell of the band, the first in row-major order among equals (every cell is some lane's)
        int32_t m = bsc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
        unsigned long long key = bsc == m ? (unsigned long long)bi << 32 | bj : ~0ull, kmin = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(kmin, o, 64); kmin = x < kmin ? x : kmin; }
        if (m >= 0) { if (lane == 0) { d.status[p] = GBA_PAIR_NO_ALIGNMENT; d.dp_score[p] = 0; } }
        else if (key == kmin) { d.end_i[p] = bi; d.end_j[p] = bj; d.end_lo[p] = blo; d.end_ptr[p] = bptr; d.dp_score[p] = m >> 5; }
        return;
    }
    if (lane == 0) {
        const int32_t v = (int32_t)(ring[fslot] & ~TAGS) - (int32_t)Q * (128 << 5);
        if (v >= UNREACH) d.status[p] = GBA_PAIR_UNREACHABLE;
        d.dp_score[p] = v >> 5;
    }
}

// ---------------------------------------------------------------- k_fill_wide_cols
// The widest class with linear gaps and matrix costs: k_fill_wide<4> with the 256 columns in flight dealt to the lanes
// four NEIGHBOURING columns each instead of four chunks of 64.  The deletion chain then needs one wave scan per 256
// columns (a running minimum over the lane's own four, the scan over the lanes' totals, and the exclusive prefix applied
// to the four) where the chunked form ran four; a column's diagonal predecessor is the lane's own previous value for
// three columns of four; and the four 2-bit arrows of a lane are one byte of a record dword that four lanes put
// together in two steps.  Rows of 201 cells (band 100) take 0.3 of the instructions.  Words, tie-break and records are
// those of k_fill_wide.
__global__ __launch_bounds__(64) void k_fill_wide_cols(Dev d, FillConst fc, uint32_t cls)
{
    constexpr int CPL = 4, SPAN = 64 * CPL;
    __shared__ uint32_t ring[WIDE_RING];
    __shared__ uint32_t sbuf[WIDE_SBUF];
    __shared__ uint32_t tab[LDS_TAB];
    static_assert(WIDE_RING >= 2048 + SPAN, "lanes run up to SPAN - 1 columns past the row end");
    constexpr uint32_t TAGS = 31u;
    constexpr int32_t BIG = 0x60000000;
    const int lane = threadIdx.x;
    if (lane < 10) tab[lane] = fc.mrow[lane];
    const uint32_t p = d.order[d.plan->cls_base[cls] + blockIdx.x];
    const uint32_t Q = d.Q[p];
    const uint32_t fslot = d.T[p] - d.lo_last[p];
    const uint32_t *st = d.stream + d.word_off[p];
    const uint32_t wcap = d.wcap[p];
    uint32_t *rec = d.rec + d.rec_off[p];
    const int32_t D = fc.del * 32;
    const int32_t codeL = fc.cL - D;
    uint32_t spos = 0, sload = 0;
    uint32_t W = 0, Wp = 0, dsh = 0, roff = 0;
    __syncthreads();
    for (uint32_t r = 0; r <= Q; r++) {
        while (sload - spos < (uint32_t)WIDE_SBUF / 2 && sload < wcap) {
            const uint32_t x = sload + lane;
            const uint32_t v = x < wcap ? st[x] : 0u;
            sbuf[x & (WIDE_SBUF - 1)] = v;
            sload += 64;
        }
        __syncthreads();
        const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)sbuf[spos & (WIDE_SBUF - 1)]);
        Wp = W;
        W = word & ROW_W_MASK; dsh = (word >> 13) & 1023u;
        const uint32_t qcode = (word >> 10) & 7u;
        const uint32_t mlo = tab[qcode * 2], mhi = tab[qcode * 2 + 1];
        const uint32_t tot = dsh + W;
        const uint32_t nnew = tot > Wp ? tot - Wp : 0u;
        if (nnew <= 3u) {                               // nearly every row: the entering codes sit in the row word itself
            if ((uint32_t)lane < nnew) ring[Wp + lane] = INFW | ((word >> (23u + 3u * (uint32_t)lane)) & 7u);
            spos += 1u + cont_words(W, CAP_LINEAR);
        } else {
            for (uint32_t x = lane; x < nnew; x += 64) {
                uint32_t code;
                if (x < 3u) code = (word >> (23u + 3u * x)) & 7u;
                else {
                    const uint32_t e = (x - 3u) >> 3, rr = (x - 3u) & 7u;
                    code = (sbuf[(spos + 1u + e) & (WIDE_SBUF - 1)] >> (3u * rr)) & 7u;
                }
                ring[Wp + x] = INFW | code;
            }
            spos += 1u + ((nnew + 4u) >> 3) + cont_words(W, CAP_LINEAR);
        }
        __syncthreads();
        uint32_t cdiag = dsh >= 1u && dsh - 1u < Wp ? ring[dsh - 1u] : INFW;
        int32_t hc = r == 0 ? -D : (int32_t)INFW;       // row 0: column j costs j deletions
        const uint32_t nd = (W + 15u) >> 4;
        const int32_t kk = CPL * lane;                  // this lane's first column inside the span
        for (uint32_t c0 = 0; c0 < W; c0 += SPAN) {
            const uint32_t k0 = c0 + (uint32_t)kk;
            uint32_t up[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e++) up[e] = ring[k0 + e + dsh];       // past the row end: slots that exist, values nobody uses
            __syncthreads();                            // every lane has read its slots before any is overwritten
            const uint32_t dleft = (uint32_t)__builtin_amdgcn_update_dpp((int32_t)cdiag, (int32_t)up[CPL - 1], 0x138, 0xf, 0xf, false);
            cdiag = (uint32_t)__builtin_amdgcn_readlane((int32_t)up[CPL - 1], 63);
            int32_t xv[CPL], run[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const uint32_t diag = e == 0 ? dleft : up[e - 1];
                const uint32_t sb = __builtin_amdgcn_perm(mhi, mlo, up[e] & 7u);
                const int32_t m = (int32_t)((sb << 5) + diag);
                const int32_t u = (int32_t)up[e] + fc.cU;
                xv[e] = min(m, u);                      // best of the two moves from the previous row
                const int32_t v = (int32_t)((uint32_t)xv[e] & ~TAGS) - (kk + e) * D;
                run[e] = e == 0 ? v : min(run[e - 1], v);
            }
            const int32_t incl = wave_scan_min(run[CPL - 1], BIG);
            const int32_t excl = __builtin_amdgcn_update_dpp(BIG, incl, 0x138, 0xf, 0xf, false);      // every column left of this lane's four
            uint32_t ab = 0;
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const int32_t before = e == 0 ? excl : min(excl, run[e - 1]);
                const int32_t lc = min(before + (kk + e) * D, hc + (kk + e + 1) * D) + codeL;
                const uint32_t best = (uint32_t)min(xv[e], lc);
                ring[k0 + e] = (up[e] & TAGS) | (best & ~TAGS);
                ab |= ((best >> 3) & 3u) << (2 * (CPL * (lane & 3) + e));
            }
            hc = min(__builtin_amdgcn_readlane(incl, 63) + (SPAN - 1) * D, hc + SPAN * D);
            // arrows: 2 bits per column, 16 columns = four lanes per dword
            ab |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)ab, 0x111, 0xf, 0xf, false);
            ab |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)ab, 0x112, 0xf, 0xf, false);
            const uint32_t dwi = (c0 >> 4) + ((uint32_t)lane >> 2);
            if ((lane & 3) == 3 && dwi < nd) rec[roff + dwi] = ab;
        }
        if (lane == 0) rec[roff + nd] = word & ROW_META_MASK;
        roff += nd + 1u;
        __syncthreads();
    }
    if (lane == 0) {
        const int32_t v = (int32_t)(ring[fslot] & ~TAGS) - (int32_t)Q * (128 << 5);
        if (v >= UNREACH) d.status[p] = GBA_PAIR_UNREACHABLE;
        d.dp_score[p] = v >> 5;
    }
}

// ---------------------------------------------------------------- k_fill_wide_affine
// Wave-per-pair fill for the affine model: lane per column as in k_fill_wide, the rings of the
// match state M and the insertion-gap state I in LDS.  Along the row two things travel: the linear
// deletion (M[k-1] + del) and the deletion-gap state D (opened from M[k-1], extended from D[k-1]).
// With a(j) the best score of column j over the moves that do not come from the left (diagonal,
// insertion, closing an insertion gap), the score of M[k] is
//     x(k) = min( a(k), min over j < k of a(j) + f(k - j) ),  f(n) = min( n * P, OE + (n - 1) * E )
// (P = min(del, open + extend), OE = open + extend, E = extend: n linear steps, or one gap of
// length n), and the score of D[k] is  y(k) = min over j < k of a(j) + OE + (k - j - 1) * min(P, E).
// Both are prefix minima of a(j) - j * slope for the two slopes P and E: two wave scans with scalar
// carries between chunks.  The words of the left neighbour are then rebuilt from x and y, and the
// arrow, the gap bits and the tie-breaks are decided per cell exactly as in k_fill<true>.
//
// QV (AffineGuidedAlign(..., idsScoreFn, ...), /root/reference/iblasr/BlasrUtilsImpl.hpp:707-710): substitution, insertion and
// the one-base deletion take their costs from the profile (within +-63 here: the affine cell word has one score bit fewer),
// gap open and extend stay constants.  The deletion cost now differs from column to column, so the two slopes no longer
// describe the row: the pair (M score, D score) of a column is a min-plus affine map of the pair of the column to its left,
//     x(k) = min( a(k), x(k-1) + min(del(k), OE), y(k-1) + E ),   y(k) = min( x(k-1) + OE, y(k-1) + E ),
// and the wave scan composes these maps (a 2 x 2 min-plus matrix and a 2-vector per lane) instead of taking two prefix minima.
struct RowMap { int32_t m00, m01, m10, m11, b0, b1; };      // x' = min(m00 + x, m01 + y, b0), y' = min(m10 + x, m11 + y, b1)

template <int CTRL, int ROWS>
__device__ __forceinline__ void rowmap_step(RowMap &g, int32_t big)
{
    // f: the map accumulated by the lanes below (applied first); lanes the pattern does not reach see the identity map
    RowMap f;
    f.m00 = __builtin_amdgcn_update_dpp(0, g.m00, CTRL, ROWS, 0xf, false);
    f.m01 = __builtin_amdgcn_update_dpp(big, g.m01, CTRL, ROWS, 0xf, false);
    f.m10 = __builtin_amdgcn_update_dpp(big, g.m10, CTRL, ROWS, 0xf, false);
    f.m11 = __builtin_amdgcn_update_dpp(0, g.m11, CTRL, ROWS, 0xf, false);
    f.b0 = __builtin_amdgcn_update_dpp(big, g.b0, CTRL, ROWS, 0xf, false);
    f.b1 = __builtin_amdgcn_update_dpp(big, g.b1, CTRL, ROWS, 0xf, false);
    RowMap h;
    h.m00 = min(g.m00 + f.m00, g.m01 + f.m10); h.m01 = min(g.m00 + f.m01, g.m01 + f.m11);
    h.m10 = min(g.m10 + f.m00, g.m11 + f.m10); h.m11 = min(g.m10 + f.m01, g.m11 + f.m11);
    h.b0 = min(min(g.m00 + f.b0, g.m01 + f.b1), g.b0);
    h.b1 = min(min(g.m10 + f.b0, g.m11 + f.b1), g.b1);
    // entries stay below 2^31: no entry of a composed map exceeds `big` plus the row's largest path cost
    h.m01 = min(h.m01, big); h.m10 = min(h.m10, big); h.b0 = min(h.b0, big + big / 4); h.b1 = min(h.b1, big + big / 4);
    g = h;
}

template <bool NARROW, bool QV = false>
__global__ __launch_bounds__(64) void k_fill_wide_affine(Dev d, FillConst fc, uint32_t cls)
{
    __shared__ int32_t ptab[16];
    constexpr int WIDE_SBUF = NARROW ? WN_SBUF : ::WIDE_SBUF;
    __shared__ uint32_t ringM[NARROW ? WN_RING : WIDE_RING];
    __shared__ uint32_t ringI[NARROW ? WN_RING : WIDE_RING];
    __shared__ uint32_t sbuf[WIDE_SBUF];
    __shared__ uint32_t tab[LDS_TAB];
    constexpr uint32_t TAGS = 63u;
    constexpr int32_t BIG = 0x60000000;
    const int lane = threadIdx.x;
    if (lane < 10) tab[lane] = fc.mrow[lane];
    const uint32_t p = d.order[d.plan->cls_base[cls] + blockIdx.x];
    const uint32_t Q = d.Q[p];
    const uint32_t fslot = d.T[p] - d.lo_last[p];
    const uint32_t *st = d.stream + d.word_off[p];
    const uint32_t wcap = d.wcap[p];
    uint32_t *rec = d.rec + d.rec_off[p];
    const int32_t OE = fc.cDO, E = fc.cDE - 2;          // open + extend, extend, in cell-word units
    const int32_t DL = fc.del * 64;
    const int32_t P = min(DL, OE);
    const bool y_by_e = E <= P;                         // slope of the cheapest way to lengthen a path that ends in the gap state
    uint32_t spos = 0, sload = 0;
    uint32_t W = 0, Wp = 0, dsh = 0, roff = 0;
    const int16_t *prow = QV ? d.prof + (d.q_off[p] + d.qS[p]) * 11 : nullptr;      // row r >= 1 uses prow[(r - 1) * 11 ..]
    int32_t pnext = QV && Q >= 1u && lane < 11 ? (int32_t)prow[lane] : 0;
    __syncthreads();
    for (uint32_t r = 0; r <= Q; r++) {
        if constexpr (QV) {
            if (lane < 11) ptab[lane] = min(max(pnext, -63), 63);                // read after the barriers below
            if (r + 1u <= Q && r >= 1u && lane < 11) pnext = (int32_t)prow[(size_t)r * 11 + lane];
        }
        while (sload - spos < (uint32_t)WIDE_SBUF / 2 && sload < wcap) {
            const uint32_t x = sload + lane;
            const uint32_t v = x < wcap ? st[x] : 0u;
            sbuf[x & (WIDE_SBUF - 1)] = v;
            sload += 64;
        }
        __syncthreads();
        const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)sbuf[spos & (WIDE_SBUF - 1)]);
        Wp = W;
        W = word & ROW_W_MASK; dsh = (word >> 13) & 1023u;
        const uint32_t qcode = (word >> 10) & 7u;
        const uint32_t mlo = tab[qcode * 2], mhi = tab[qcode * 2 + 1];
        const uint32_t tot = dsh + W;
        const uint32_t nnew = tot > Wp ? tot - Wp : 0u;
        if (nnew <= 3u) {                               // nearly every row: the entering codes sit in the row word itself
            if ((uint32_t)lane < nnew) {
                ringM[Wp + lane] = INFW | ((word >> (23u + 3u * (uint32_t)lane)) & 7u);
                ringI[Wp + lane] = INFW;
            }
            spos += 1u + cont_words(W, CAP_AFFINE);
        } else {
            for (uint32_t x = lane; x < nnew; x += 64) {
                uint32_t code;
                if (x < 3u) code = (word >> (23u + 3u * x)) & 7u;
                else {
                    const uint32_t e = (x - 3u) >> 3, rr = (x - 3u) & 7u;
                    code = (sbuf[(spos + 1u + e) & (WIDE_SBUF - 1)] >> (3u * rr)) & 7u;
                }
                ringM[Wp + x] = INFW | code;
                ringI[Wp + x] = INFW;
            }
            spos += 1u + ((nnew + 4u) >> 3) + cont_words(W, CAP_AFFINE);
        }
        __syncthreads();
        // row 0: the diagonal candidate of cell (0,0) costs exactly 0
        uint32_t cdiag = r == 0 ? (uint32_t)fc.diag0 : dsh >= 1u && dsh - 1u < Wp ? ringM[dsh - 1u] : INFW;
        int32_t carry1 = BIG, carry2 = BIG;             // prefix minima of the columns left of the chunk, as seen from its first column
        uint32_t xl = INFW, yl = INFW;                  // M and D score of the column left of the chunk
        const uint32_t nd = (W + 3u) >> 2;
        for (uint32_t c0 = 0; c0 < W; c0 += 64) {
            const uint32_t k = c0 + lane;
            const uint32_t up = ringM[k + dsh], upi = ringI[k + dsh];
            __syncthreads();                            // every lane has read its slots before any is overwritten
            const uint32_t diag = (uint32_t)__builtin_amdgcn_update_dpp((int32_t)cdiag, (int32_t)up, 0x138, 0xf, 0xf, false);
            cdiag = (uint32_t)__builtin_amdgcn_readlane((int32_t)up, 63);
            const uint32_t tcd = up & 7u;
            uint32_t sb = __builtin_amdgcn_perm(mhi, mlo, tcd);
            int32_t cU = fc.cU, cL = fc.cL;
            if constexpr (QV) {
                if (r != 0u) {                                         // row 0 (no query base) keeps the matrix and the fixed deletion
                    const uint32_t ti = min(max(tcd, 1u), 5u) - 1u;
                    sb = (uint32_t)(ptab[ti] + 128);
                    cU += (ptab[5] - fc.ins) * 64;
                    cL += (ptab[6 + ti] - fc.del) * 64;
                }
            }
            const int32_t m = (int32_t)((sb << 6) + diag);
            const int32_t u = (int32_t)up + cU;
            const int32_t io = (int32_t)(up & ~TAGS) + fc.cIO, ie = (int32_t)upi + fc.cIE;
            const uint32_t ig = (uint32_t)min(io, ie);                 // bit 2: 1 = extend
            const int32_t ic = (int32_t)ig + (4 << 3);
            const int32_t av = min(m, min(u, ic));                     // best of the moves that do not come from the left
            const int32_t a = (int32_t)((uint32_t)av & ~TAGS);
            int32_t x, y;
            if constexpr (QV) {
                RowMap g;
                g.m00 = min((int32_t)((uint32_t)cL & ~TAGS), OE); g.m01 = E; g.m10 = OE; g.m11 = E; g.b0 = a; g.b1 = BIG;
                rowmap_step<0x111, 0xf>(g, BIG); rowmap_step<0x112, 0xf>(g, BIG); rowmap_step<0x114, 0xf>(g, BIG);
                rowmap_step<0x118, 0xf>(g, BIG); rowmap_step<0x142, 0xa>(g, BIG); rowmap_step<0x143, 0xc>(g, BIG);
                const int32_t xin = min((int32_t)xl, BIG), yin = min((int32_t)yl, BIG);       // scores may be negative: signed
                x = min(min(g.m00 + xin, g.m01 + yin), g.b0);          // score of M[k]: the chunk's maps applied to the column left of it
                y = min(min(g.m10 + xin, g.m11 + yin), g.b1);          // score of D[k]
            } else {
                const int32_t inc1 = wave_scan_min(a - lane * P, BIG), inc2 = wave_scan_min(a - lane * E, BIG);
                const int32_t ex1 = __builtin_amdgcn_update_dpp(BIG, inc1, 0x138, 0xf, 0xf, false);
                const int32_t ex2 = __builtin_amdgcn_update_dpp(BIG, inc2, 0x138, 0xf, 0xf, false);
                const int32_t s1 = min(ex1, carry1) + lane * P;            // best a(j) + (k - j) * P over j < k
                const int32_t s2 = min(ex2, carry2) + lane * E;            // best a(j) + (k - j) * E over j < k
                carry1 = min(__builtin_amdgcn_readlane(inc1, 63), carry1) + 64 * P;
                carry2 = min(__builtin_amdgcn_readlane(inc2, 63), carry2) + 64 * E;
                x = min(a, min(s1, s2 + OE - E));                          // score of M[k]
                y = y_by_e ? s2 + OE - E : s1 + OE - P;                    // score of D[k]
            }
            // the words of the left neighbour, then the cell exactly as the lane-per-pair kernel computes it
            const uint32_t xp = (uint32_t)__builtin_amdgcn_update_dpp((int32_t)xl, x, 0x138, 0xf, 0xf, false);
            const uint32_t yp = (uint32_t)__builtin_amdgcn_update_dpp((int32_t)yl, y, 0x138, 0xf, 0xf, false);
            xl = (uint32_t)__builtin_amdgcn_readlane(x, 63); yl = (uint32_t)__builtin_amdgcn_readlane(y, 63);
            const int32_t l = (int32_t)xp + cL;
            const int32_t dop = (int32_t)xp + fc.cDO, de = (int32_t)yp + fc.cDE;
            const uint32_t dg = (uint32_t)min(dop, de);                // bit 1: 1 = extend
            const int32_t dc = (int32_t)dg + (3 << 3);
            const uint32_t best = (uint32_t)min(min(av, l), dc);
            ringM[k] = (best & ~TAGS) | tcd;
            ringI[k] = ig & ~TAGS;
            // record byte: M arrow (bits 3..5) | I bit (bit 2) | D bit (bit 1); 4 columns per dword
            int32_t ab = (int32_t)(((best & 0x38u) | (ig & 4u) | (dg & 2u)) << (8 * (lane & 3)));
            ab |= __builtin_amdgcn_update_dpp(0, ab, 0x111, 0xf, 0xf, false);
            ab |= __builtin_amdgcn_update_dpp(0, ab, 0x112, 0xf, 0xf, false);
            const uint32_t dwi = (c0 >> 2) + ((uint32_t)lane >> 2);
            if ((lane & 3) == 3 && dwi < nd) rec[roff + dwi] = (uint32_t)ab;
        }
        if (lane == 0) rec[roff + nd] = word & ROW_META_MASK;
        roff += nd + 1u;
        __syncthreads();
    }
    if (lane == 0) {
        const int32_t v = (int32_t)(ringM[fslot] & ~TAGS) - (int32_t)Q * (128 << 6);
        if (v >= UNREACH) d.status[p] = GBA_PAIR_UNREACHABLE;
        d.dp_score[p] = v >> 6;
    }
}

// ---------------------------------------------------------------- k_fill_quad_affine
// The narrow affine instance again with FOUR pairs per wave: rows of the narrow class are 33 to 64 cells wide, so a
// wave with one lane per column idles on up to half its lanes and pays the whole per-row bookkeeping for one pair.
// Here a pair has 16 lanes (one DPP row) and a lane four neighbouring columns: the 64 columns of a chunk as before,
// a quarter of the wave-instructions per pair and row for everything but the cell code, four-step scans that never
// leave the DPP row, and one record dword (four arrow bytes) per lane without any gather.  Recurrence, words,
// tie-breaks and records are exactly those of k_fill_wide_affine<true>; what was wave-uniform there (row word,
// widths, stream position) is uniform per 16 lanes here.  The four pairs of a wave run their rows in step; a pair that
// has fewer rows idles at the end (pairs of a class are neighbours in length).
template <int CTRL>
__device__ __forceinline__ int32_t row16_from(int32_t identity, int32_t v)
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ int32_t row16_scan_min(int32_t v, int32_t identity)
{
    v = min(v, row16_from<0x111>(identity, v));
    v = min(v, row16_from<0x112>(identity, v));
    v = min(v, row16_from<0x114>(identity, v));
    v = min(v, row16_from<0x118>(identity, v));
    return v;
}

__global__ __launch_bounds__(64) void k_fill_quad_affine(Dev d, FillConst fc, uint32_t cls, uint32_t count)
{
    constexpr int PPW = 4, CPL = 4;
    __shared__ uint32_t ringM[PPW][WN_RING];
    __shared__ uint32_t ringI[PPW][WN_RING];
    __shared__ uint32_t sbuf[PPW][WN_SBUF];
    __shared__ uint32_t tab[LDS_TAB];
    constexpr uint32_t TAGS = 63u;
    constexpr int32_t BIG = 0x60000000;
    const int lane = threadIdx.x, g = lane >> 4, sub = lane & 15;
    if (lane < 10) tab[lane] = fc.mrow[lane];
    const uint32_t idx = blockIdx.x * PPW + (uint32_t)g;
    const bool have = idx < count;
    const uint32_t p = d.order[d.plan->cls_base[cls] + (have ? idx : count - 1u)];
    const uint32_t Q = d.Q[p];
    const uint32_t fslot = d.T[p] - d.lo_last[p];
    const uint32_t *st = d.stream + d.word_off[p];
    const uint32_t wcap = d.wcap[p];
    uint32_t *rec = d.rec + d.rec_off[p];
    uint32_t *rM = ringM[g], *rI = ringI[g], *sb_ = sbuf[g];
    const int32_t OE = fc.cDO, E = fc.cDE - 2;          // open + extend, extend, in cell-word units
    const int32_t DL = fc.del * 64;
    const int32_t P = min(DL, OE);
    const bool y_by_e = E <= P;
    uint32_t spos = 0, sload = 0;
    uint32_t W = 0, Wp = 0, dsh = 0, roff = 0;
    uint32_t Qmax = have ? Q : 0u;                      // rows of the longest of the wave's pairs
    Qmax = max(Qmax, (uint32_t)__shfl_xor((int)Qmax, 16, 64));
    Qmax = max(Qmax, (uint32_t)__shfl_xor((int)Qmax, 32, 64));
    Qmax = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)Qmax);
    __syncthreads();
    for (uint32_t r = 0; r <= Qmax; r++) {
        const bool act = have && r <= Q;
        // guide words: sixteen at a time per pair, at least 64 buffered (a row of this class has at most 1 + 16)
        for (;;) {
            const bool need = act && sload - spos < (uint32_t)WN_SBUF / 2 && sload < wcap;
            if (!__any(need)) break;
            if (need) {
                const uint32_t x = sload + (uint32_t)sub;
                sb_[x & (WN_SBUF - 1)] = x < wcap ? st[x] : 0u;
                sload += 16;
            }
        }
        __syncthreads();
        uint32_t word = 0, nnew = 0;
        if (act) {
            word = sb_[spos & (WN_SBUF - 1)];
            Wp = W;
            W = word & ROW_W_MASK; dsh = (word >> 13) & 1023u;
            const uint32_t tot = dsh + W;
            nnew = tot > Wp ? tot - Wp : 0u;
        }
        const uint32_t qcode = (word >> 10) & 7u;
        const uint32_t mlo = tab[qcode * 2], mhi = tab[qcode * 2 + 1];
        // entering columns: INFW words carrying the column's base code
        if (!__any(nnew > 3u)) {                        // nearly every row: the entering codes sit in the row words themselves
            if ((uint32_t)sub < nnew) {
                rM[Wp + sub] = INFW | ((word >> (23u + 3u * (uint32_t)sub)) & 7u);
                rI[Wp + sub] = INFW;
            }
        } else {
            for (uint32_t x = (uint32_t)sub; __any(x < nnew); x += 16) {
                if (x < nnew) {
                    uint32_t code;
                    if (x < 3u) code = (word >> (23u + 3u * x)) & 7u;
                    else {
                        const uint32_t e = (x - 3u) >> 3, rr = (x - 3u) & 7u;
                        code = (sb_[(spos + 1u + e) & (WN_SBUF - 1)] >> (3u * rr)) & 7u;
                    }
                    rM[Wp + x] = INFW | code;
                    rI[Wp + x] = INFW;
                }
            }
        }
        if (act) spos += 1u + (nnew > 3u ? (nnew + 4u) >> 3 : 0u) + cont_words(W, CAP_AFFINE);
        __syncthreads();
        // row 0: the diagonal candidate of cell (0,0) costs exactly 0
        uint32_t cdiag = INFW;
        if (act) cdiag = r == 0 ? (uint32_t)fc.diag0 : dsh >= 1u && dsh - 1u < Wp ? rM[dsh - 1u] : INFW;
        int32_t carry1 = BIG, carry2 = BIG;
        uint32_t xl = INFW, yl = INFW;
        const uint32_t nd = (W + 3u) >> 2;
        for (uint32_t c0 = 0; __any(act && c0 < W); c0 += 64) {
            const bool cact = act && c0 < W;
            const uint32_t k0 = c0 + (uint32_t)(CPL * sub);
            uint32_t up[CPL], upi[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                up[e] = cact ? rM[k0 + e + dsh] : INFW;     // past the row end: slots that exist, values nobody uses
                upi[e] = cact ? rI[k0 + e + dsh] : INFW;
            }
            __syncthreads();                            // every lane has read its slots before any is overwritten
            const uint32_t dleft = (uint32_t)row16_from<0x111>((int32_t)cdiag, (int32_t)up[CPL - 1]);
            const bool more = __any(act && c0 + 64u < W);
            if (more) cdiag = (uint32_t)__shfl((int)up[CPL - 1], lane | 15, 64);
            int32_t av[CPL], a[CPL];
            uint32_t ig[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const uint32_t diag = e == 0 ? dleft : up[e - 1];
                const uint32_t tcd = up[e] & 7u;
                const uint32_t sb = __builtin_amdgcn_perm(mhi, mlo, tcd);
                const int32_t m = (int32_t)((sb << 6) + diag);
                const int32_t u = (int32_t)up[e] + fc.cU;
                const int32_t io = (int32_t)(up[e] & ~TAGS) + fc.cIO, ie = (int32_t)upi[e] + fc.cIE;
                ig[e] = (uint32_t)min(io, ie);                          // bit 2: 1 = extend
                const int32_t ic = (int32_t)ig[e] + (4 << 3);
                av[e] = min(m, min(u, ic));                              // best of the moves that do not come from the left
                a[e] = (int32_t)((uint32_t)av[e] & ~TAGS);
            }
            // prefix minima of a(j) - j * slope over the chunk's columns: inside the lane, then across the pair's 16 lanes
            const int32_t kc = CPL * sub;
            int32_t l1[CPL], l2[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const int32_t v1 = a[e] - (kc + e) * P, v2 = a[e] - (kc + e) * E;
                l1[e] = e == 0 ? v1 : min(l1[e - 1], v1);
                l2[e] = e == 0 ? v2 : min(l2[e - 1], v2);
            }
            const int32_t t1 = row16_scan_min(l1[CPL - 1], BIG), t2 = row16_scan_min(l2[CPL - 1], BIG);
            const int32_t x1 = min(row16_from<0x111>(BIG, t1), carry1), x2 = min(row16_from<0x111>(BIG, t2), carry2);   // all columns left of this lane's
            if (more) {
                carry1 = min(__shfl(t1, lane | 15, 64), carry1) + 64 * P;
                carry2 = min(__shfl(t2, lane | 15, 64), carry2) + 64 * E;
            }
            int32_t x[CPL], y[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const int32_t s1 = (e == 0 ? x1 : min(x1, l1[e - 1])) + (kc + e) * P;      // best a(j) + (k - j) * P over j < k
                const int32_t s2 = (e == 0 ? x2 : min(x2, l2[e - 1])) + (kc + e) * E;      // best a(j) + (k - j) * E over j < k
                x[e] = min(a[e], min(s1, s2 + OE - E));                  // score of M[k]
                y[e] = y_by_e ? s2 + OE - E : s1 + OE - P;               // score of D[k]
            }
            const uint32_t xleft = (uint32_t)row16_from<0x111>((int32_t)xl, x[CPL - 1]);
            const uint32_t yleft = (uint32_t)row16_from<0x111>((int32_t)yl, y[CPL - 1]);
            if (more) { xl = (uint32_t)__shfl(x[CPL - 1], lane | 15, 64); yl = (uint32_t)__shfl(y[CPL - 1], lane | 15, 64); }
            uint32_t ab = 0;
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const uint32_t xp = e == 0 ? xleft : (uint32_t)x[e - 1], yp = e == 0 ? yleft : (uint32_t)y[e - 1];
                const int32_t l = (int32_t)xp + fc.cL;
                const int32_t dop = (int32_t)xp + fc.cDO, de = (int32_t)yp + fc.cDE;
                const uint32_t dg = (uint32_t)min(dop, de);              // bit 1: 1 = extend
                const int32_t dc = (int32_t)dg + (3 << 3);
                const uint32_t best = (uint32_t)min(min(av[e], l), dc);
                if (cact) {
                    rM[k0 + e] = (best & ~TAGS) | (up[e] & 7u);
                    rI[k0 + e] = ig[e] & ~TAGS;
                }
                // record byte: M arrow (bits 3..5) | I bit (bit 2) | D bit (bit 1)
                ab |= ((best & 0x38u) | (ig[e] & 4u) | (dg & 2u)) << (8 * e);
            }
            const uint32_t dwi = (c0 >> 2) + (uint32_t)sub;
            if (cact && dwi < nd) rec[roff + dwi] = ab;
        }
        if (act) {
            if (sub == 0) rec[roff + nd] = word & ROW_META_MASK;
            roff += nd + 1u;
        }
        __syncthreads();
    }
    if (have && sub == 0) {
        const int32_t v = (int32_t)(rM[fslot] & ~TAGS) - (int32_t)Q * (128 << 6);
        if (v >= UNREACH) d.status[p] = GBA_PAIR_UNREACHABLE;
        d.dp_score[p] = v >> 6;
    }
}

// ---------------------------------------------------------------- k_traceback
// One lane per pair: from the end cell (Q, T) follow the arrows to (0, 0); merge diagonal runs
// into blocks (ArrowPathToAlignment), count matches / mismatches / gaps and rescore the path
// (ComputeAlignmentStats).  Blocks are produced last-to-first, so they are written from the end
// of the pair's scratch region backwards and come out in forward order.
//
// A row's record is [arrow dwords][meta]; rows are walked last to first, so a lane reads its
// record array backwards.  Every lane keeps TB_RW record dwords that reach up to its position and
// TB_TW dwords of target codes in an LDS window ([dword][lane]).  A dependent global load per row
// would cost a full memory round trip per row; instead ALL lanes of the wave refill their windows
// together whenever any lane is about to leave one, one round trip per ~20 rows.
// Linear records hold 2 bits per cell, affine records one byte per cell (M arrow in bits 3..5,
// insertion-gap bit 2, deletion-gap bit 1).
#ifndef GBA_TB_TSOON
#define GBA_TB_TSOON 6
#endif
constexpr int TB_RW = 64, TB_RW_FEW = 192, TB_TW = 12, TB_TSOON = GBA_TB_TSOON;
#ifndef GBA_TB_LANES
#define GBA_TB_LANES 64
#endif
constexpr int TB_LANES = GBA_TB_LANES;      // pairs per wave: the walk is a dependent chain, more waves hide more of it

// RW: record dwords per lane in the window.  An affine row of 33 cells is ten dwords, so 64 dwords are five rows between
// refills and a walk of few pairs (one wave per CU or two, nothing to hide the refill behind) spends half its time on
// them; such launches take 192 dwords (TB_RW_FEW), which at 52 KB of LDS a wave is no way to run a full batch.
// LOCAL (linear gaps): the walk begins in the cell the fill left in Dev::end_* and ends at the first cell that carries the
// arrow code 3 ("the alignment begins behind this cell"); gap bases are counted inside the aligned stretch only.
template <bool AFF, int RW = TB_RW, bool LOCAL = false>
__global__ __launch_bounds__(TB_LANES) void k_traceback(Dev d, TbConst tc_, gba_pair_result *res, uint32_t first, uint32_t count)
{
    static_assert(!LOCAL || !AFF, "the Local end rule: linear gaps");
    __shared__ int32_t sub[25 * (1 + GBA_MAX_EXTRA)];     // the stats scorer's matrix, then the extra scorers'
    __shared__ uint32_t rwin[RW * TB_LANES];
    __shared__ uint32_t twin[TB_TW * TB_LANES];
    if (threadIdx.x < 25) {
        sub[threadIdx.x] = tc_.sub[threadIdx.x];
        for (int x = 0; x < GBA_MAX_EXTRA; x++) sub[25 * (x + 1) + threadIdx.x] = tc_.xsub[x][threadIdx.x];
    }
    __syncthreads();
    const int nx = tc_.n_extra;
    const int lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * TB_LANES + threadIdx.x;    // serves order[first .. first + count)
    if (blockIdx.x * TB_LANES >= count) return;
    const uint32_t p = d.order[first + (idx < count ? idx : count - 1)];
    const bool live = idx < count && d.status[p] == GBA_PAIR_OK;   // else: fill found the end cell unreachable
    const uint32_t Q = d.Q[p], T = d.T[p];
    const uint32_t *rec = d.rec + d.rec_off[p];
    const uint32_t *tc4 = (const uint32_t *)(d.tcodes + tc_offset(d, p));   // 4 column codes per dword
    const uint8_t *twin8 = (const uint8_t *)twin;
    constexpr uint32_t CSHIFT = AFF ? 2u : 4u, CMASK = AFF ? 3u : 15u, BITS = AFF ? 8u : 2u;
    BlkOut bo = blk_out_begin(d, p);
    uint32_t i = LOCAL ? d.end_i[p] : Q, j = LOCAL ? d.end_j[p] : T, lo = LOCAL ? d.end_lo[p] : d.lo_last[p], nb = 0, run = 0;
    const uint32_t ei = i, ej = j;
    bool done = false;                            // LOCAL: the walk has come to the cell the alignment begins behind
    uint32_t nM = 0, ndiag = 0, nopen = 0;
    int32_t score = 0;
    uint32_t prev_op = 0;
    int32_t pend = 0;                             // substitution cost of the last diagonal move, added one step later
    int32_t xs0 = 0, xs1 = 0;                     // substitution sums of the extra scorers
    // affine paths spend an extra step on every gap-state change
    uint32_t budget = (Q + T + 2u) * (AFF ? 3u : 1u);
    bool ok = true;
    int32_t ptr = LOCAL ? (int32_t)d.end_ptr[p] : (int32_t)d.recdw[p] - 1;        // meta dword of the current row
    int32_t rb = 0x40000000, tb = 0x40000000;     // first record dword / code dword held in the windows
    uint32_t gw = live ? rec[ptr] : 0u;
    uint32_t W = gw & ROW_W_MASK, n = (W + CMASK) >> CSHIFT;   // width, arrow dwords of the current row
    uint32_t state = 0;                           // affine: 0 match state, 1 insertion gap, 2 deletion gap
    // arrow code -> move, four bits per code: 0 diagonal, codeU up, codeL left; affine codes 3 and 4
    // change the gap state without moving (3: to the deletion gap, 4: to the insertion gap); the rest are errors
    uint32_t mvtab = 0;
    for (uint32_t a = 1; a < (AFF ? 8u : 4u); a++)
        mvtab |= (a == tc_.codeU ? 1u : a == tc_.codeL ? 2u : (AFF && a <= 4u) ? 3u : 4u) << (4u * a);
    constexpr uint32_t S0TAB = (2u << 12) | (1u << 16);
    for (;;) {
        const bool active = live && ok && (i | j) != 0u && !(LOCAL && done);
        if (!__any(active)) break;
        {
            // linear pairs of the lane classes have rows of at most 8 arrow dwords: always inside the window
            const int32_t look = AFF ? (int32_t)min(n + 1u, (uint32_t)RW - 16u) : (int32_t)n + 1;
            const int32_t jd = (int32_t)(j >> 2) - 1;
            if (__any(active && (ptr - look < rb || jd < tb))) {
                // The walk is bound by these scattered reads (every lane in a record of its own,
                // 64-byte requests).  The record window starts on a 64-byte boundary, so a refill
                // is four whole requests (the window still reaches up to ptr: it starts at most 15
                // dwords higher than it had to); the code window is refilled along with it only
                // when a lane will leave it within the next ~24 columns.
                const bool codes = __any(active && jd - TB_TSOON < tb);
                if (active) {
                    rb = ptr - (RW - 1);
                    rb += (int32_t)((16u - (uint32_t)((uintptr_t)(rec + rb) >> 2)) & 15u);
                    const uint32_t *src = rec + rb;
#pragma unroll
                    for (int c = 0; c < RW / 4; c++) {
                        const u32x4 v = *(const u32x4 *)(src + 4 * c);
                        uint32_t *w = rwin + (4 * c) * TB_LANES + lane;
                        w[0] = v.x; w[TB_LANES] = v.y; w[2 * TB_LANES] = v.z; w[3 * TB_LANES] = v.w;
                    }
                    if (codes) {
                        tb = (int32_t)(j >> 2) - (TB_TW - 1);
                        const uint32_t *ts = tc4 + tb;
#pragma unroll
                        for (int c = 0; c < TB_TW / 4; c++) {
                            const u32x4u v = *(const u32x4u *)(ts + 4 * c);
                            uint32_t *w = twin + (4 * c) * TB_LANES + lane;
                            w[0] = v.x; w[TB_LANES] = v.y; w[2 * TB_LANES] = v.z; w[3 * TB_LANES] = v.w;
                        }
                    }
                }
            }
        }
        if (active) {
            // The arrow dword, the target code of column j and the meta dword of the row above are
            // fetched together (one LDS round trip); the last two are only used by some moves.
            const uint32_t k = j - lo;
            const bool inband = k < W && budget-- != 0u;
            const uint32_t kc = inband ? k : 0u;
            const int32_t at = ptr - (int32_t)(n - (kc >> CSHIFT));     // dword holding cell k
            const int32_t up = ptr - (int32_t)n - 1;                    // meta dword of row i - 1
            uint32_t aw, gup;
            if (AFF && at < rb) aw = rec[at]; else aw = rwin[(at - rb) * TB_LANES + lane];    // affine rows can be wider than the window
            if (AFF && up < rb) gup = rec[up]; else gup = rwin[(up - rb) * TB_LANES + lane];
            const uint32_t tbase = (uint32_t)twin8[(((j >> 2) - (uint32_t)tb) * TB_LANES + lane) * 4 + (j & 3u)] - 1u;
            score += pend; pend = 0;
            const uint32_t cell = aw >> ((kc & CMASK) * BITS);
            // move: 0 diagonal, 1 up (insertion), 2 left (deletion), 3 none (state change only), 4 error
            uint32_t mv;
            if constexpr (!AFF) {
                mv = (mvtab >> ((cell & 3u) * 4u)) & 7u;
                if constexpr (LOCAL) { if (inband && (cell & 3u) == 3u) { mv = 3u; done = true; } }      // no move, and the walk is over
            } else {
                const uint32_t a4 = ((cell >> 3) & 7u) * 4u;
                const uint32_t m0 = (mvtab >> a4) & 7u;
                const uint32_t s0 = (S0TAB >> a4) & 3u;
                const uint32_t st = state;
                mv = st == 0u ? m0 : st;                                // in a gap state the move is the gap's own
                state = st == 0u ? s0 : st == 1u ? ((cell & 4u) ? 1u : 0u) : ((cell & 2u) ? 2u : 0u);
            }
            // everything below is selects: one lane's move is another lane's no-op
            const bool isd = mv == 0u, isu = mv == 1u, isl = mv == 2u;
            // a diagonal or up move needs a row above, a diagonal or left move a column to the left
            const bool bad = (int)!inband | (int)(mv == 4u) | ((int)(i == 0u) & (int)(mv < 2u)) | ((int)(j == 0u) & (int)((mv & 1u) == 0u) & (int)(mv < 3u));
            const bool dmove = isd && !bad, gmove = (isu || isl) && !bad, rowmove = (isd || isu) && !bad;
            ok = !bad;
            const uint32_t qb = (gw >> 10) & 7u;
            nM += dmove && qb == tbase ? 1u : 0u;
            const uint32_t si = qb * 5 + min(tbase, 4u);
            pend = sub[si];                                               // only kept for a diagonal move
            pend = dmove ? pend : 0;
            if (nx > 0) { const int32_t v = sub[25 + si]; xs0 += dmove ? v : 0; }
            if (nx > 1) { const int32_t v = sub[50 + si]; xs1 += dmove ? v : 0; }
            ndiag += dmove ? 1u : 0u;
            if (gmove && run != 0u) {
                if (bo.outp == bo.lowp && bo.room) blk_out_spill(d.blk_scratch, d.plan, d.blk_pool_off, d.blk_pool_cap, bo, Q + 1u, nb);
                if (bo.room) { bo.outp -= 3; bo.outp[0] = i; bo.outp[1] = j; bo.outp[2] = run; }
                nb++;
            }
            run = dmove ? run + 1u : gmove ? 0u : run;
            nopen += gmove && prev_op != mv ? 1u : 0u;
            prev_op = dmove ? 0u : gmove ? mv : prev_op;
            j -= (isd || isl) && !bad ? 1u : 0u;
            i -= rowmove ? 1u : 0u;
            lo -= rowmove ? (gw >> 13) & 1023u : 0u;
            ptr = rowmove ? up : ptr;
            gw = rowmove ? gup : gw;
            W = gw & ROW_W_MASK; n = (W + CMASK) >> CSHIFT;
        }
    }
    score += pend;
    if (!live) return;
    gba_pair_result &R = res[p];
    if (run) {
        if (bo.outp == bo.lowp && bo.room) blk_out_spill(d.blk_scratch, d.plan, d.blk_pool_off, d.blk_pool_cap, bo, Q + 1u, nb);
        if (bo.room) { bo.outp -= 3; bo.outp[0] = i; bo.outp[1] = j; bo.outp[2] = run; }
        nb++;
    }
    d.blk_end[p] = bo.end;
    if (!ok) { d.status[p] = GBA_PAIR_UNREACHABLE; return; }
    // every row is consumed by a diagonal or an up move, every column by a diagonal or a left move
    const uint32_t nI = (LOCAL ? ei - i : Q) - ndiag, nD = (LOCAL ? ej - j : T) - ndiag;
    score += gap_score(tc_.affine, tc_.ins, tc_.del, tc_.aopen, tc_.aext, nI, nD, nopen);
    R.n_blk = nb; R.score = score;
    d.span[p] = bo.room ? nb : 0u;            // no room: nothing of this pair is read, the batch is run again
    R.n_match = nM; R.n_mismatch = ndiag - nM; R.n_ins = nI; R.n_del = nD;
    if (nx > 0) R.extra_score[0] = xs0 + gap_score(tc_.xaffine[0], tc_.xins[0], tc_.xdel[0], tc_.xaopen[0], tc_.xaext[0], nI, nD, nopen);
    if (nx > 1) R.extra_score[1] = xs1 + gap_score(tc_.xaffine[1], tc_.xins[1], tc_.xdel[1], tc_.xaopen[1], tc_.xaext[1], nI, nD, nopen);
}

// ---------------------------------------------------------------- k_traceback_wide
// Traceback of the wave-per-pair class: the rows of such pairs have records of up to 65 dwords, so
// a lane-private window would hold two or three rows.  Here a WAVE serves one pair: all lanes
// load the next WT_WIN record dwords (and WT_TC target codes) into LDS with coalesced 16-byte
// loads, then the wave walks the rows that window covers.  Same records, moves and statistics as
// k_traceback.
constexpr int WT_WIN = 4096;      // record dwords per refill
constexpr int WT_TC = 2048;       // target codes (bytes) per refill

// LOCAL: the walk begins in the cell the fill left in Dev::end_* and ends at the first cell that carries the arrow code 3
// ("the alignment begins behind this cell"); gap bases are counted inside the aligned stretch only.
template <bool AFF, bool LOCAL = false>
__global__ __launch_bounds__(64) void k_traceback_wide(Dev d, TbConst tc_, gba_pair_result *res, uint32_t cls)
{
    constexpr uint32_t CSHIFT = AFF ? 2u : 4u, CMASK = AFF ? 3u : 15u, BITS = AFF ? 8u : 2u;
    __shared__ int32_t sub[25 * (1 + GBA_MAX_EXTRA)];
    __shared__ __attribute__((aligned(16))) uint32_t win[WT_WIN];
    __shared__ uint32_t tcw[WT_TC / 4];
    if (threadIdx.x < 25) {
        sub[threadIdx.x] = tc_.sub[threadIdx.x];
        for (int x = 0; x < GBA_MAX_EXTRA; x++) sub[25 * (x + 1) + threadIdx.x] = tc_.xsub[x][threadIdx.x];
    }
    const int nx = tc_.n_extra;
    int32_t xs0 = 0, xs1 = 0, xp0 = 0, xp1 = 0;         // extra scorers: sums, and the lookups of the last diagonal move
    const int lane = threadIdx.x;
    const uint32_t p = d.order[d.plan->cls_base[cls] + blockIdx.x];
    if (d.status[p] != GBA_PAIR_OK) return;             // fill found the end cell unreachable
    const uint32_t Q = d.Q[p], T = d.T[p];
    const uint32_t *rec = d.rec + d.rec_off[p];
    const uint8_t *tc = d.tcodes + tc_offset(d, p);
    const uint8_t *tcw8 = (const uint8_t *)tcw;
    BlkOut bo = blk_out_begin(d, p);                    // only lane 0's copy is used
    // one more block (i, j, run): lane 0 stores it
    auto put = [&](uint32_t bi_, uint32_t bj_, uint32_t brun, uint32_t have) {
        if (lane == 0) {
            if (bo.outp == bo.lowp && bo.room) blk_out_spill(d.blk_scratch, d.plan, d.blk_pool_off, d.blk_pool_cap, bo, Q + 1u, have);
            if (bo.room) { bo.outp -= 3; bo.outp[0] = bi_; bo.outp[1] = bj_; bo.outp[2] = brun; }
        }
    };
    // The walk is wave-uniform: every lane follows the same path, LDS values are made scalar right
    // after they are read, so the whole state machine runs on the scalar unit; only lane 0 stores.
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)v); };
    uint32_t i = LOCAL ? uni(d.end_i[p]) : Q, j = LOCAL ? uni(d.end_j[p]) : T, lo = uni(LOCAL ? d.end_lo[p] : d.lo_last[p]), nb = 0, run = 0;
    const uint32_t ei = i, ej = j;
    uint32_t nM = 0, ndiag = 0, nopen = 0, prev_op = 0;
    int32_t score = 0;
    uint32_t budget = (Q + T + 2u) * (AFF ? 3u : 1u);   // affine paths spend an extra step on every gap-state change
    uint32_t state = 0;                                 // affine: 0 match state, 1 insertion gap, 2 deletion gap
    bool ok = true, done = false, look_ahead = true;
    int32_t pend_v = 0;
    int32_t lane_score = 0, lane_x0 = 0, lane_x1 = 0;   // substitution costs of the runs taken lanes side by side, summed over the lanes at the end
    int32_t ptr = LOCAL ? (int32_t)uni(d.end_ptr[p]) : (int32_t)uni(d.recdw[p]) - 1;         // meta dword of the current row
    while (!done) {
        // refill: records [rb, rb + WT_WIN) ending at ptr, target codes [jb, jb + WT_TC) ending at column j
        const int32_t rb = ptr - (WT_WIN - 1);
        const int32_t jb = ((int32_t)j | 3) - (WT_TC - 1);               // multiple of 4
        __syncthreads();
        {
            u32x4u v[WT_WIN / 256];
            uint32_t t[WT_TC / 256];
#pragma unroll
            for (int x = 0; x < WT_WIN / 256; x++) v[x] = *(const u32x4u *)(rec + rb + x * 256 + lane * 4);   // may start before the pair's records (arena slack)
#pragma unroll
            for (int x = 0; x < WT_TC / 256; x++) t[x] = *(const uint32_t *)(tc + jb + x * 256 + lane * 4);
#pragma unroll
            for (int x = 0; x < WT_WIN / 256; x++) *(uint4 *)(win + x * 256 + lane * 4) = make_uint4(v[x].x, v[x].y, v[x].z, v[x].w);
#pragma unroll
            for (int x = 0; x < WT_TC / 256; x++) tcw[x * 64 + lane] = t[x];
        }
        __syncthreads();
        uint32_t gw = uni(win[ptr - rb]);
        uint32_t W = gw & ROW_W_MASK, n = (W + CMASK) >> CSHIFT;
        for (;;) {
            if ((i | j) == 0u) { done = true; break; }
            if (ptr - (int32_t)n - 1 < rb || (int32_t)j - 1 < jb) break;          // next refill
            if (look_ahead && state == 0u) {
                // A run of diagonal moves in one go: lane t looks at the cell t rows up and t columns left, on the
                // assumption that the rows in between are as wide as this one (so their records are as long); every
                // lane checks its row's width, its own column range and its arrow, and the run ends at the first lane
                // that finds anything else.  Most of a path is such runs; single steps take the scalar code below.
                const uint32_t stride = n + 1u;
                const int32_t mt = ptr - (int32_t)((uint32_t)lane * stride);          // meta dword of row i - lane
                const bool inwin = (uint32_t)lane < i && (uint32_t)lane < j && mt - (int32_t)stride >= rb && (int32_t)j - lane - 1 >= jb;
                const uint32_t gw_t = inwin ? win[mt - rb] : 0u;
                const bool same = inwin && (gw_t & ROW_W_MASK) == W;
                const uint32_t dsh_t = same ? (gw_t >> 13) & 1023u : 0u;
                const uint32_t dsum = wave_scan_sum(dsh_t);
                const uint32_t lo_t = lo - (dsum - dsh_t);                            // first column of row i - lane
                const uint32_t k_t = j - (uint32_t)lane - lo_t;
                const bool incol = same && k_t < W;
                const uint32_t aw_t = incol ? win[mt - (int32_t)n + (int32_t)(k_t >> CSHIFT) - rb] : ~0u;
                const uint32_t cell_t = aw_t >> ((k_t & CMASK) * BITS);
                const bool diag_t = incol && (AFF ? ((cell_t >> 3) & 7u) == 0u : (cell_t & 3u) == 0u);
                const unsigned long long okm = __ballot(diag_t);
                uint32_t L = okm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okm);
                L = min(L, budget);
                // the move behind the run goes along if the lane that ended it sees a plain insertion or deletion arrow
                // (its row has this row's width and its column lies inside: the row above it is in the window)
                uint32_t gap = 0;                     // 0 none, 1 up, 2 left
                if (L < 64u && budget > L) {
                    const uint32_t a_t = AFF ? (cell_t >> 3) & 7u : cell_t & 3u;
                    const uint32_t mv_t = incol && !diag_t ? (a_t == tc_.codeU ? 1u : a_t == tc_.codeL ? 2u : 0u) : 0u;
                    gap = (uint32_t)__builtin_amdgcn_readlane((int32_t)mv_t, (int)L);
                }
                if (L > 0u || gap != 0u) {
                    if (L > 0u) {
                        const bool mine = (uint32_t)lane < L;
                        const uint32_t qb_t = (gw_t >> 10) & 7u;
                        const uint32_t tb_t = mine ? (uint32_t)tcw8[(int32_t)j - lane - jb] - 1u : 0u;
                        if (mine) {
                            lane_score += sub[qb_t * 5 + tb_t];
                            if (nx > 0) lane_x0 += sub[25 + qb_t * 5 + tb_t];
                            if (nx > 1) lane_x1 += sub[50 + qb_t * 5 + tb_t];
                        }
                        nM += (uint32_t)__popcll(__ballot(mine && qb_t == tb_t));
                        run += L; ndiag += L; prev_op = 0; budget -= L;
                        i -= L; j -= L;
                        // the first column after L rows: the exclusive prefix at lane L (lane 63's inclusive one for a full run)
                        lo = L < 64u ? (uint32_t)__builtin_amdgcn_readlane((int32_t)lo_t, (int)L)
                                     : (uint32_t)__builtin_amdgcn_readlane((int32_t)(lo_t - dsh_t), 63);
                        ptr -= (int32_t)(L * stride);
                    }
                    if (gap != 0u) {
                        if (run) {
                            put(i, j, run, nb);
                            nb++; run = 0;
                        }
                        nopen += prev_op != gap ? 1u : 0u;
                        prev_op = gap; budget--;
                        if (gap == 1u) {
                            i--;
                            lo -= (uint32_t)__builtin_amdgcn_readlane((int32_t)dsh_t, (int)L);
                            ptr -= (int32_t)stride;
                        } else j--;
                    }
                    gw = uni(win[ptr - rb]);
                    W = gw & ROW_W_MASK; n = (W + CMASK) >> CSHIFT;
                    look_ahead = L == 64u || gap != 0u;       // a run that ended at anything else ended at a move for the scalar code
                    continue;
                }
            }
            look_ahead = true;
            const uint32_t k = j - lo;
            if (k >= W || budget-- == 0u) { ok = false; done = true; break; }
            const int32_t up = ptr - (int32_t)n - 1;                 // meta dword of row i - 1
            // one LDS round trip for the arrow dword, the meta dword of the row above and the target
            // code of column j (the last two are only used by some moves)
            const uint32_t aw_v = win[up + 1 + (int32_t)(k >> CSHIFT) - rb], gup_v = win[up - rb];
            const uint32_t tb_v = tcw8[(int32_t)j - jb];
            const uint32_t aw = uni(aw_v), gup = uni(gup_v), tbase = uni(tb_v) - 1u;
            score += (int32_t)uni((uint32_t)pend_v); pend_v = 0;
            if (nx > 0) { xs0 += (int32_t)uni((uint32_t)xp0); xp0 = 0; }
            if (nx > 1) { xs1 += (int32_t)uni((uint32_t)xp1); xp1 = 0; }
            const uint32_t cell = aw >> ((k & CMASK) * BITS);
            // move: 0 diagonal, 1 up (insertion), 2 left (deletion), 3 none (state change only), 4 error
            uint32_t mv;
            if constexpr (!AFF) {
                const uint32_t a = cell & 3u;
                if (LOCAL && a == 3u) { done = true; break; }          // the alignment begins behind this cell
                mv = a == 0u ? 0u : a == tc_.codeU ? 1u : a == tc_.codeL ? 2u : 4u;
            } else {
                const uint32_t a = (cell >> 3) & 7u;
                const uint32_t m0 = a == 0u ? 0u : a == tc_.codeU ? 1u : a == tc_.codeL ? 2u : a <= 4u ? 3u : 4u;
                const uint32_t s0 = a == 3u ? 2u : a == 4u ? 1u : 0u;
                const uint32_t st = state;
                mv = st == 0u ? m0 : st;
                state = st == 0u ? s0 : st == 1u ? ((cell & 4u) ? 1u : 0u) : ((cell & 2u) ? 2u : 0u);
                if (mv == 3u) continue;
            }
            if (mv == 4u || (i == 0u && mv != 2u) || (j == 0u && mv != 1u)) { ok = false; done = true; break; }
            if (mv == 0u) {
                const uint32_t qb = (gw >> 10) & 7u;
                nM += qb == tbase ? 1u : 0u;
                pend_v = sub[qb * 5 + tbase];         // added one step later: keeps the LDS read off the critical path
                if (nx > 0) xp0 = sub[25 + qb * 5 + tbase];
                if (nx > 1) xp1 = sub[50 + qb * 5 + tbase];
                run++; ndiag++; prev_op = 0;
            } else {
                if (run) {
                    put(i, j, run, nb);
                    nb++; run = 0;
                }
                nopen += prev_op != mv ? 1u : 0u;
                prev_op = mv;
            }
            if (mv != 1u) j--;
            if (mv != 2u) {
                i--;
                lo -= (gw >> 13) & 1023u;
                ptr = up;
                gw = gup;
                W = gw & ROW_W_MASK; n = (W + CMASK) >> CSHIFT;
            }
        }
    }
    score += (int32_t)uni((uint32_t)pend_v);
    xs0 += (int32_t)uni((uint32_t)xp0); xs1 += (int32_t)uni((uint32_t)xp1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lane_score += __shfl_xor(lane_score, o, 64);
        lane_x0 += __shfl_xor(lane_x0, o, 64);
        lane_x1 += __shfl_xor(lane_x1, o, 64);
    }
    score += lane_score; xs0 += lane_x0; xs1 += lane_x1;
    if (run) { put(i, j, run, nb); nb++; }
    if (lane != 0) return;
    d.blk_end[p] = bo.end;
    gba_pair_result &R = res[p];
    if (!ok) { d.status[p] = GBA_PAIR_UNREACHABLE; return; }
    const uint32_t nI = (LOCAL ? ei - i : Q) - ndiag, nD = (LOCAL ? ej - j : T) - ndiag;
    score += gap_score(tc_.affine, tc_.ins, tc_.del, tc_.aopen, tc_.aext, nI, nD, nopen);
    R.n_blk = nb; R.score = score;
    d.span[p] = bo.room ? nb : 0u;            // no room: nothing of this pair is read, the batch is run again
    R.n_match = nM; R.n_mismatch = ndiag - nM; R.n_ins = nI; R.n_del = nD;
    if (nx > 0) R.extra_score[0] = xs0 + gap_score(tc_.xaffine[0], tc_.xins[0], tc_.xdel[0], tc_.xaopen[0], tc_.xaext[0], nI, nD, nopen);
    if (nx > 1) R.extra_score[1] = xs1 + gap_score(tc_.xaffine[1], tc_.xins[1], tc_.xdel[1], tc_.xaopen[1], tc_.xaext[1], nI, nD, nopen);
}

// ---------------------------------------------------------------- results
// The reference as the library keeps it: 2 bits per base and one mask bit per base for every letter outside ACGT (upper or
// lower case).  One thread per 32 letters of a chunk that starts at base `first` (a multiple of 32) of the reference.
__global__ __launch_bounds__(256) void k_ref_pack(const uint8_t *letters, uint64_t n, uint64_t first, uint32_t *pk, uint32_t *mask)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;       // 32-base group of the chunk
    if (i * 32 >= n) return;
    uint32_t lo = 0, hi = 0, m = 0;
    for (uint32_t x = 0; x < 32u && i * 32 + x < n; x++) {
        const uint32_t c = base_code(letters[i * 32 + x]);
        const uint32_t v = c & 3u;                                       // N packs as A, and is masked
        if (x < 16u) lo |= (c < 4u ? v : 0u) << (2u * x); else hi |= (c < 4u ? v : 0u) << (2u * (x - 16u));
        m |= (c >= 4u ? 1u : 0u) << x;
    }
    const uint64_t g = first / 32 + i;
    pk[2 * g] = lo; pk[2 * g + 1] = hi; mask[g] = m;
}

__global__ void k_results_init(Dev d, gba_pair_result *res)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n) return;
    gba_pair_result r;
    memset(&r, 0, sizeof(r));
    r.status = d.status[p];
    res[p] = r;
    d.span[p] = 0;             // from here on: the pair's block count (written by the tracebacks)
}

// n_blk -> offsets into the caller's block arena (single block).  The block counts come from
// d.span (free after k_plan: k_results_init zeroes it, the tracebacks store the counts there), the
// offsets go to d.word_off (free after the fill kernels).
// *needed holds the blocks of the parts of the batch scanned before this one and receives the sum.
__global__ __launch_bounds__(SCAN_T) void k_scan1(Dev d, uint64_t *needed)
{
    __shared__ uint64_t sm[17];
    uint64_t carry = *needed;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < d.n; b0 += SCAN_E * blockDim.x) {
        uint32_t v[SCAN_E];
        uint64_t lv = 0;
#pragma unroll
        for (int x = 0; x < SCAN_E; x++) {
            const uint32_t p = b0 + SCAN_E * threadIdx.x + x;
            v[x] = p < d.n && d.status[p] == GBA_PAIR_OK ? d.span[p] : 0u;
            lv += v[x];
        }
        uint64_t tot;
        uint64_t e = carry + block_excl_scan(lv, sm, &tot);
#pragma unroll
        for (int x = 0; x < SCAN_E; x++) {
            const uint32_t p = b0 + SCAN_E * threadIdx.x + x;
            if (p < d.n) d.word_off[p] = e;
            e += v[x];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) { *needed = carry; needed[2] += d.plan->pool_fail; }
}

// One block per pair: completes the pair's result record (the aligned interval included) and, when
// the caller wants the blocks, copies them into the caller's arena; pairs whose blocks do not fit
// get NOSPACE.
__global__ __launch_bounds__(256) void k_emit(Dev d, gba_pair_result *res, gba_block *out, uint64_t capacity, int want_blocks)
{
    const uint32_t p = blockIdx.x;
    const int st = d.status[p];
    const uint64_t off = d.word_off[p];
    const uint32_t nb = st == GBA_PAIR_OK ? d.span[p] : 0u;
    const bool fits = !want_blocks || off + nb <= capacity;
    const gba_block *src = d.blk_scratch + (nb ? d.blk_end[p] - nb : 0ull);
    if (threadIdx.x == 0) {
        gba_pair_result &R = res[p];
        R.status = st;
        R.blk_off = off;
        if (st == GBA_PAIR_OK) {
            R.dp_score = d.dp_score[p];
            const uint32_t qS = d.qS[p], tS = d.tS[p];
            R.q_pos = qS; R.t_pos = tS;
            R.rows = d.Q[p]; R.cols = d.T[p]; R.ncells = d.ncells[p];
            R.q_start = R.q_end = qS; R.t_start = R.t_end = tS;
            if (nb) {
                const gba_block a = src[0], z = src[nb - 1];
                R.q_start = qS + a.q_pos; R.t_start = tS + a.t_pos;
                R.q_end = qS + z.q_pos + z.length; R.t_end = tS + z.t_pos + z.length;
            }
            if (!fits) R.status = GBA_PAIR_NOSPACE;
        }
    }
    if (st != GBA_PAIR_OK || !fits || !want_blocks) return;
    gba_block *dst = out + off;
    for (uint32_t x = threadIdx.x; x < nb; x += blockDim.x) dst[x] = src[x];
}

// ---------------------------------------------------------------- k_cigar
// CIGAR operations of the refined pairs, in the BAM encoding.  One wave per pair, one lane per
// "segment" = a gap-free block plus the gap that follows it.  With adjacent indels merged (the
// default of the reference's writers) the part an insertion and a deletion have in common becomes
// aligned columns, which extends the block's diagonal run; a segment is then: run aligned columns,
// ni inserted query bases, nd deleted target bases.  A column starts a new operation when its
// operation differs from the one of the column before, whichever segment that column belongs to;
// the last operation of a segment follows from the segment alone, so a lane knows what precedes
// it from its neighbour.  Two passes of the same walk: <false> counts the operations of each pair
// (d.recdw, free after the tracebacks), and after a scan over the pairs <true> writes them:
// first (column where the operation starts) << 4 | op at every start, then every entry's column is
// replaced by the distance to the next start.
// Bases are compared by their codes, like everywhere in this library (case-insensitive; two letters
// outside ACGT count as the same).  The counting pass leaves the query's codes (code + 1, the form
// of the target codes) at the start of the pair's traceback records, which nobody reads any more:
// eight columns are then compared with two 8-byte loads and a handful of integer operations.
typedef uint64_t u64u1 __attribute__((aligned(1)));

// bit k: bytes k of a and b are equal, k < 8; all bytes are below 0x80
__device__ __forceinline__ uint32_t equal_bytes(uint64_t a, uint64_t b)
{
    const uint64_t d = a ^ b;
    // 0x80 - d keeps bit 7 exactly where d is zero, and no byte borrows from the next
    const uint64_t e = (0x8080808080808080ull - d) & 0x8080808080808080ull;
    const uint32_t lo = (uint32_t)e >> 7, hi = (uint32_t)(e >> 32) >> 7;      // bits 0, 8, 16, 24
    // the four bits of a half gathered into bits 24..27 by one multiplication (no two products share a bit)
    constexpr uint32_t G = 1u << 24 | 1u << 17 | 1u << 10 | 1u << 3;
    return (lo * G) >> 24 | ((hi * G) >> 24) << 4;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void k_cigar(Dev d, uint32_t flags, const uint64_t *off, uint32_t *ops, uint64_t capacity)
{
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    const uint32_t nb = d.status[p] == GBA_PAIR_OK ? d.span[p] : 0u;
    if (nb == 0) {
        if (!WRITE && lane == 0) d.recdw[p] = 0;
        return;
    }
    const uint32_t nops = WRITE ? d.recdw[p] : 0u;
    uint32_t *out = nullptr;
    if (WRITE) {
        const uint64_t o = off[p];
        if (o + nops > capacity) return;             // the caller's arena is too small: the entry point reports it
        out = ops + o;
    }
    const uint32_t Q = d.Q[p];
    const gba_block *bl = d.blk_scratch + (d.blk_end[p] - nb);
    const uint8_t *tc = d.tcodes + tc_offset(d, p) + 1;      // tc[x]: code + 1 of the target base x columns after t_pos
    uint8_t *qc = (uint8_t *)(d.rec + d.rec_off[p]);         // qc[x]: the same for the query base x rows after q_pos
    const bool seqm = (flags & GBA_CIGAR_SEQ_MATCH) != 0, adj = (flags & GBA_CIGAR_ADJACENT_INDELS) != 0;
    if (!WRITE && seqm) {
        // a row's record is at least two dwords, so Q + 8 bytes fit the pair's records
        const uint64_t q0 = d.q_off[p] + d.qS[p];
        for (uint32_t x0 = 4 * lane; x0 < Q + 8; x0 += 256) {
            uint32_t w = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) w |= (x0 + x < Q ? query_code(d.q_arena, d.q_pk, q0 + x0 + x) + 1u : 6u) << (8 * x);   // past the end: equal to no target code
            *(uint32_t *)(qc + x0) = w;
        }
        __syncthreads();
    }
    uint32_t carry_op = 0xffu, carry_cnt = 0, carry_col = 0;
    for (uint32_t s0 = 0; s0 < nb; s0 += 64) {
        const uint32_t s = s0 + lane;
        uint32_t bq = 0, bt = 0, run = 0, ni = 0, nd = 0, last = 0xffu;
        if (s < nb) {
            const gba_block b = bl[s];
            bq = b.q_pos; bt = b.t_pos; run = b.length;
            if (s + 1 < nb) {
                const gba_block nx = bl[s + 1];
                ni = nx.q_pos - (bq + run); nd = nx.t_pos - (bt + run);
                if (!adj && ni && nd) { const uint32_t cm = ni < nd ? ni : nd; run += cm; ni -= cm; nd -= cm; }
            }
            last = nd ? (uint32_t)GBA_CIGAR_D : ni ? (uint32_t)GBA_CIGAR_I : !seqm ? (uint32_t)GBA_CIGAR_M
                 : qc[bq + run - 1] == tc[bt + run - 1] ? (uint32_t)GBA_CIGAR_EQ : (uint32_t)GBA_CIGAR_X;
        }
        const uint32_t before = wave_prev(last, carry_op);
        // walk(starts, emit): starts(n) for n operations that start in aligned columns (counting only),
        // or emit(column within the segment, op) for every operation that starts in the segment
        auto walk = [&](auto &&emit, bool counting, uint32_t &count) {
            uint32_t prev = before;
            if (!seqm) {
                if (run && prev != GBA_CIGAR_M) { if (counting) count++; else emit(0u, (uint32_t)GBA_CIGAR_M); }
                if (run) prev = GBA_CIGAR_M;
            } else {
                for (uint32_t k = 0; k < run; k += 8) {
                    const uint32_t n = run - k < 8u ? run - k : 8u;
                    const uint32_t m = equal_bytes(*(const u64u1 *)(qc + bq + k), *(const u64u1 *)(tc + bt + k)) & ((1u << n) - 1u);
                    // a column starts an operation when it differs from the column before it
                    uint32_t st = (m ^ (m << 1 | (prev == GBA_CIGAR_EQ ? 1u : 0u))) & ((1u << n) - 1u);
                    if (prev != GBA_CIGAR_EQ && prev != GBA_CIGAR_X) st |= 1u;
                    if (counting) count += (uint32_t)__builtin_popcount(st);
                    else while (st) {
                        const uint32_t x = (uint32_t)__builtin_ctz(st);
                        emit(k + x, (m >> x & 1u) ? (uint32_t)GBA_CIGAR_EQ : (uint32_t)GBA_CIGAR_X);
                        st &= st - 1u;
                    }
                    prev = (m >> (n - 1u) & 1u) ? (uint32_t)GBA_CIGAR_EQ : (uint32_t)GBA_CIGAR_X;
                }
            }
            if (ni) { if (prev != GBA_CIGAR_I) { if (counting) count++; else emit(run, (uint32_t)GBA_CIGAR_I); } prev = GBA_CIGAR_I; }
            if (nd && prev != GBA_CIGAR_D) { if (counting) count++; else emit(run + ni, (uint32_t)GBA_CIGAR_D); }
        };
        uint32_t cnt = 0, unused = 0;
        walk([](uint32_t, uint32_t) {}, true, cnt);
        const uint32_t cnt_incl = wave_scan_sum(cnt);
        if (WRITE) {
            const uint32_t cols = run + ni + nd;
            const uint32_t col_incl = wave_scan_sum(cols);
            uint32_t at = carry_cnt + cnt_incl - cnt;
            const uint32_t col0 = carry_col + col_incl - cols;
            walk([&](uint32_t k, uint32_t op) { out[at++] = (col0 + k) << 4 | op; }, false, unused);
            carry_col += (uint32_t)__builtin_amdgcn_readlane((int32_t)col_incl, 63);
        }
        carry_cnt += (uint32_t)__builtin_amdgcn_readlane((int32_t)cnt_incl, 63);
        carry_op = (uint32_t)__builtin_amdgcn_readlane((int32_t)last, 63);
    }
    if (!WRITE) {
        if (lane == 0) d.recdw[p] = carry_cnt;
        return;
    }
    __syncthreads();                                 // the starts written by other lanes are visible
    for (uint32_t i0 = 0; i0 < nops; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t w = 0, next = carry_col;            // the last operation ends with the alignment
        if (i < nops) { w = out[i]; if (i + 1 < nops) next = out[i + 1] >> 4; }
        __syncthreads();                             // every lane has read its neighbour's entry before anyone rewrites one
        if (i < nops) out[i] = (next - (w >> 4)) << 4 | (w & 15u);
    }
}

// Operation counts (d.recdw) -> offsets into the caller's operation arena, for the pairs of one
// part (single block).  *carry holds the operations of the parts scanned before and receives the
// sum; off[d.n] is the offset one past the part (the next part's first pair rewrites the same value).
__global__ __launch_bounds__(SCAN_T) void k_scan_ops(Dev d, uint64_t *carry, uint64_t *off)
{
    __shared__ uint64_t sm[17];
    const uint64_t c0 = *carry;
    __syncthreads();
    const uint64_t tot = scan_array(d.recdw, off, d.n, sm);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < d.n; p += blockDim.x) off[p] += c0;
    if (threadIdx.x == 0) { off[d.n] = c0 + tot; *carry = c0 + tot; }
}

// ================================================================ guide production
// The step in front of the refine stage: exact k-mer matches between a read and its reference
// window, chained by sparse dynamic programming into the increasing list of gap-free blocks that
// the guided alignment follows (SDPAlign(..., Local) at iblasr/BlasrAlignImpl.hpp:980-987).
//   fragments  the maximal exact matches of length >= k between ACGT bases (a run of overlapping
//              k-mer hits on one diagonal is one fragment), ordered by (query start, target start)
//   chain      score(f) = match * len(f) + max(0, max over earlier fragments p of
//              score(p) - drift penalty - match * overlap), where drift = change of diagonal
//              (del per target base gained, ins per query base gained) and overlap = the bases
//              of p's tail that reach past f's start in either sequence (p is shortened by them;
//              something of p must remain); ties go to the earliest p; the best score anywhere
//              (earliest on ties) ends the chain
// One workgroup of 256 threads per pair.  Both sequences are packed into LDS (2 bits per base plus
// a bit pair that marks letters outside ACGT and positions past the end); the target is indexed
// SDP_TILE positions at a time in an LDS hash table keyed by k-mer whose entries head a linked
// list of the tile's positions with that k-mer, and every query position is looked up.  A hit
// that continues a match from the previous diagonal cell is dropped, the others are extended 16
// bases at a time and become fragments.  The fragments are sorted in LDS (bitonic), then one wave
// chains them: 64 fragments per round, every lane scanning all earlier fragments (uniform LDS
// reads), the 64 of the round settling among themselves lane by lane.
constexpr uint32_t SDP_THREADS = 512;       // eight waves per pair: with three pairs per CU (47 KB of LDS each) 24 waves hide the table's latency (256: 13.3 ms per 32 768 pairs of 10 kb, 512: 9.7, 1024: 13.3)
typedef uint32_t u32x4u1 __attribute__((ext_vector_type(4), aligned(1)));

struct SdpConst {
    uint32_t k, kmask;            // tuple size, mask of a k-mer (2k bits)
    int32_t ins, del, match;
    uint32_t max_fragments, cap;  // cap: max_fragments rounded up to a power of two (sort capacity)
    uint32_t hit_limit;           // more k-mer hits than this in one pair: too repetitive
    uint32_t tile, slots_log2;    // target positions indexed at a time; log2 of the hash slots (= 2 * tile)
    uint32_t qwords, twords;      // dwords of a packed sequence (including two pad words)
    uint32_t queue;               // hits queued per tile before they are worked on
    uint32_t stop;                // experiments: leave the kernel after this phase (0: run it all)
};

// What k_sdp_match needs of a batch besides its constants (all device pointers).
struct SdpIn {
    const uint32_t *q_pk, *ref_pk, *ref_mask;      // packed sources (see Dev); null: letters
    const uint8_t *q_arena, *t_arena, *t_rc;
    uint64_t q_arena_bytes, t_arena_bytes;
    const uint64_t *q_off, *t_off;
    const uint32_t *q_len, *t_len;
};

// status of a pair before any matching: slices inside their arenas, lengths the kernels can take
__device__ __forceinline__ int sdp_pair_status(const SdpIn &in, uint32_t p, uint32_t k)
{
    const uint64_t qo = in.q_off[p], to = in.t_off[p];
    const uint32_t ql = in.q_len[p], tl = in.t_len[p];
    if (to > in.t_arena_bytes || tl > in.t_arena_bytes - to || qo > in.q_arena_bytes || ql > in.q_arena_bytes - qo) return GBA_PAIR_BAD_GUIDE;
    if (ql > 65535u || tl > 65535u) return GBA_PAIR_TOO_LONG;
    if (ql < k || tl < k) return GBA_PAIR_EMPTY_GUIDE;
    return GBA_PAIR_OK;
}

// longest read and longest window among the pairs that will be matched (mx[0], mx[1]: zeroed by the caller)
__global__ void k_sdp_max(SdpIn in, uint32_t n, uint32_t k, uint32_t *mx)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || sdp_pair_status(in, p, k) != GBA_PAIR_OK) return;
    atomicMax(&mx[0], in.q_len[p]);
    atomicMax(&mx[1], in.t_len[p]);
}

// 16 bases (32 bits) of a packed array starting at base `pos`
__device__ __forceinline__ uint32_t window16(const uint32_t *pk, uint32_t pos)
{
    const uint32_t w = pos >> 4, b = (pos & 15u) * 2u;
    const uint64_t v = (uint64_t)pk[w] | (uint64_t)pk[w + 1] << 32;
    return (uint32_t)(v >> b);
}
// the same for a packed array in global memory indexed by a 64-bit base position (a dword of padding follows the array)
__device__ __forceinline__ uint32_t gwindow16(const uint32_t *pk, uint64_t pos, uint64_t ndw)
{
    const uint64_t w = pos >> 4;
    const uint32_t b = (uint32_t)(pos & 15u) * 2u;
    const uint64_t v = (uint64_t)pk[w] | (uint64_t)(w + 1 < ndw ? pk[w + 1] : 0u) << 32;     // ndw: dwords of the array
    return (uint32_t)(v >> b);
}
// 16 mask bits (one per base) starting at base `pos`, spread to the "01 per marked base" form of the packed marks
__device__ __forceinline__ uint32_t gmarks16(const uint32_t *mk, uint64_t pos)
{
    const uint64_t w = pos >> 5;
    const uint32_t b = (uint32_t)(pos & 31u);
    uint32_t m = (uint32_t)(((uint64_t)mk[w] | (uint64_t)mk[w + 1] << 32) >> b) & 0xffffu;
    m = (m | m << 8) & 0x00ff00ffu; m = (m | m << 4) & 0x0f0f0f0fu; m = (m | m << 2) & 0x33333333u; m = (m | m << 1) & 0x55555555u;
    return m;
}
// the sixteen 2-bit fields of a dword in reverse order
__device__ __forceinline__ uint32_t reverse_fields(uint32_t v)
{
    v = __builtin_bitreverse32(v);
    return ((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u);
}

// Matches of one pair: its fragments (query start << 16 | target start, length) go to frags + pair * cap in no
// particular order, their number to nfrag (above max_fragments: too many), the pair's status to out_status.
__global__ __launch_bounds__(SDP_THREADS) void k_sdp_match(SdpIn in, SdpConst sc, int32_t *out_status, uint2 *frags, uint32_t *nfrag)
{
    extern __shared__ uint32_t smem[];
    __shared__ uint32_t s_nfrag, s_hits, s_nq;
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t SLOTS = 1u << sc.slots_log2;
    // LDS: hash slots (SLOTS dwords: tag << 16 | head of the k-mer's list) | next (u16 x tile) | query bases, query marks,
    // target bases, target marks.  A k-mer is spread by an odd multiplier modulo 4^k; the slot index is the top of the
    // product, the tag its low 16 bits.  A tag match is always checked against the sequence: an entry that linear probing
    // has moved off its home slot is no longer identified by slot and tag.
    uint32_t *tab = smem;
    unsigned short *nxt = (unsigned short *)(smem + SLOTS);
    uint32_t *hq = smem + SLOTS + sc.tile / 2;                            // queue of hits: query position << 16 | list head
    uint32_t *qp = hq + sc.queue, *qs = qp + sc.qwords, *tp = qs + sc.qwords, *ts = tp + sc.twords;

    const int st = sdp_pair_status(in, p, sc.k);
    if (st != GBA_PAIR_OK) {                                              // uniform over the workgroup
        if (tid == 0) { out_status[p] = st; nfrag[p] = 0; }
        return;
    }
    const uint32_t Q = in.q_len[p], T = in.t_len[p];
    if (tid == 0) { s_nfrag = 0; s_hits = 0; }
    // ---- pack both sequences straight from the caller's letters: 16 bases per dword, and a dword of marks (01 per base
    // that is no ACGT base of the sequence).  A window on the reverse strand is read backwards and complemented.
    {
        const uint8_t *qa = in.q_arena + in.q_off[p], *ta = in.t_arena + in.t_off[p];
        const bool rc = in.t_rc != nullptr && in.t_rc[p] != 0;
        for (uint32_t w = tid; w < sc.qwords + sc.twords; w += SDP_THREADS) {
            const bool isq = w < sc.qwords;
            const uint32_t ww = isq ? w : w - sc.qwords, len = isq ? Q : T;
            const uint8_t *seq = isq ? qa : ta;
            const bool back = !isq && rc;
            uint32_t bases = 0, marks = 0x55555555u;
            const uint32_t *pk = isq ? in.q_pk : in.ref_pk;
            if (pk && ww * 16u < len) {
                // a packed source: sixteen bases are one funnel shift (read backwards: the sixteen bases that end where the
                // word begins, fields reversed and complemented); what lies beyond the sequence's end is marked
                const uint32_t x0 = ww * 16u, v = len - x0 < 16u ? len - x0 : 16u;
                const uint64_t base0 = isq ? in.q_off[p] : in.t_off[p];
                const uint64_t ndw = ((isq ? in.q_arena_bytes : in.t_arena_bytes) + 15u) >> 4;
                uint32_t mk = 0;
                if (!back) {
                    bases = gwindow16(pk, base0 + x0, ndw);
                    if (!isq) mk = gmarks16(in.ref_mask, base0 + x0);
                } else {
                    // bases len - 1 - x0 - x, x = 0 .. 15: the forward stretch [len - x0 - 16, len - x0) reversed; near the
                    // window's start the stretch begins before the window (v < 16): shift what is outside away
                    const uint64_t hi = base0 + len - x0;                       // one past the last forward base of the word
                    const uint64_t lo16 = hi >= 16u ? hi - 16u : 0u;
                    const uint32_t miss = (uint32_t)(16u - (hi - lo16));        // bases in front of the array's start (only for hi < 16)
                    bases = reverse_fields(gwindow16(pk, lo16, ndw) << (2u * miss)) ^ 0xffffffffu;
                    mk = reverse_fields(gmarks16(in.ref_mask, lo16) << (2u * miss));
                }
                if (v < 16u) { const uint32_t keep = (1u << (2u * v)) - 1u; bases &= keep; mk = (mk & keep) | (0x55555555u & ~keep); }
                marks = mk;
                bases &= ~(marks * 3u);                                         // no base: 0, marked
            } else if (!pk && ww * 16u < len) {
                const uint32_t x0 = ww * 16u;
                u32x4u1 raw = {0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu};     // 'N'
                // base x0 + x is byte x0 + x of the slice, or, read backwards, byte len - 1 - x0 - x
                if (x0 + 16u <= len) {
                    raw = *(const u32x4u1 *)(seq + (back ? len - x0 - 16u : x0));
                    if (back) {                      // last byte first
                        const u32x4u1 r = raw;
                        raw[0] = __builtin_bswap32(r[3]); raw[1] = __builtin_bswap32(r[2]); raw[2] = __builtin_bswap32(r[1]); raw[3] = __builtin_bswap32(r[0]);
                    }
                } else {
                    for (uint32_t x = 0; x0 + x < len; x++)
                        raw[x >> 2] = (raw[x >> 2] & ~(0xffu << (8 * (x & 3)))) | (uint32_t)seq[back ? len - 1u - x0 - x : x0 + x] << (8 * (x & 3));
                }
                bases = 0; marks = 0;
#pragma unroll
                for (int z = 0; z < 4; z++) {
                    // four letters at once: A 0x41, C 0x43, G 0x47, T 0x54 (case bit cleared) give code = bits 1..2 ^ bit 2,
                    // and a letter is one of the four when the code's own letter is the letter
                    const uint32_t u = raw[z] & 0xdfdfdfdfu;
                    uint32_t code = ((u >> 1) & 0x03030303u) ^ ((u >> 2) & 0x01010101u);
                    const uint32_t diff = __builtin_amdgcn_perm(0u, 0x54474341u, code) ^ u;                 // zero byte: a base
                    const uint32_t bad = ((((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u) >> 7;   // bits 0, 8, 16, 24
                    if (back) code ^= 0x03030303u;                                                             // complement
                    code &= ~(bad * 3u);                                                                       // no base: 0, marked
                    // the four 2-bit fields at bits 0, 8, 16, 24 side by side in one byte
                    uint32_t c2 = code | code >> 6, m2 = bad | bad >> 6;
                    c2 = (c2 | c2 >> 12) & 0xffu; m2 = (m2 | m2 >> 12) & 0x55u;
                    bases |= c2 << (8 * z); marks |= m2 << (8 * z);
                }
            }
            (isq ? qp : tp)[ww] = bases;
            (isq ? qs : ts)[ww] = marks;
        }
    }
    if (sc.stop == 1) return;
    const uint32_t kbits = 2u * sc.k;
    const uint32_t kmark = (uint32_t)((1ull << kbits) - 1ull) & 0x55555555u;
    const uint32_t hshift = kbits - sc.slots_log2;                        // (the tuple has at least log2(SLOTS) bits: run_sdp sees to it)
    uint2 *myfrags = frags + (uint64_t)p * sc.cap;
    // hits of the query position x with the list of target positions that starts at v (tile-relative, + 1)
    auto process = [&](uint32_t x, uint32_t v, uint32_t tile0) {
        uint32_t hits = 0;
        for (; v; v = nxt[v - 1u]) {
            const uint32_t y = tile0 + v - 1u;
            hits++;
            // the start of its run? (the cell one step back on the diagonal is no match)
            if (x > 0u && y > 0u) {
                const uint32_t a = window16(qp, x - 1u), b = window16(tp, y - 1u), m = window16(qs, x - 1u) | window16(ts, y - 1u);
                if ((((a ^ b) | m) & 3u) == 0u) continue;
            }
            uint32_t len = sc.k;
            for (;;) {
                const uint32_t a = window16(qp, x + len), b = window16(tp, y + len), m = window16(qs, x + len) | window16(ts, y + len);
                const uint32_t df = a ^ b;
                const uint32_t bad = ((df | df >> 1) | m) & 0x55555555u;
                if (bad) { len += (uint32_t)__builtin_ctz(bad) >> 1; break; }
                len += 16u;
            }
            const uint32_t at = atomicAdd(&s_nfrag, 1u);
            if (at < sc.cap) myfrags[at] = make_uint2(x << 16 | y, len);
        }
        atomicAdd(&s_hits, hits);
    };
    // ---- one tile of the target at a time
    // (Measured alternatives, all slower on 10-kb reads: k-mers rolled through registers along per-thread stretches, lanes
    // of a wave probing out of step, four independent positions per thread in flight.  16-bit slots without a tag (every
    // occupied slot probed is checked against the sequence) with eight slots per position: 5.7 ms against 6.0 for the
    // match of 32 768 pairs, not worth a compare-and-swap on half words.)
    for (uint32_t tile0 = 0; tile0 + sc.k <= T; tile0 += sc.tile) {
        __syncthreads();
        for (uint32_t i = tid; i < SLOTS; i += SDP_THREADS) tab[i] = 0;
        if (tid == 0) s_nq = 0;
        __syncthreads();
        const uint32_t tend = min(tile0 + sc.tile, T - sc.k + 1u);        // k-mers start in [tile0, tend)
        for (uint32_t y = tile0 + tid; y < tend; y += SDP_THREADS) {
            if (window16(ts, y) & kmark) continue;                        // a letter outside ACGT in the k-mer
            const uint32_t km = window16(tp, y) & sc.kmask;
            const uint32_t yl = y - tile0;
            const uint32_t mix = (km * 0x9E3779B1u) & sc.kmask;
            uint32_t h = mix >> hshift;
            const uint32_t mine = mix << 16 | (yl + 1u);
            for (;;) {
                const uint32_t v = *(volatile uint32_t *)&tab[h];
                if (v == 0u) {
                    nxt[yl] = 0;
                    if (atomicCAS(&tab[h], 0u, mine) == 0u) break;
                    continue;                                             // somebody took the slot: look at it again
                }
                if ((v ^ mine) >> 16 == 0u && (window16(tp, tile0 + (v & 0xffffu) - 1u) & sc.kmask) == km) {
                    nxt[yl] = (unsigned short)v;                          // the list of this k-mer: push in front
                    if (atomicCAS(&tab[h], v, mine) == v) break;
                    continue;
                }
                h = (h + 1u) & (SLOTS - 1u);
            }
        }
        __syncthreads();
        if (sc.stop == 5) continue;
        for (uint32_t x = tid; x + sc.k <= Q; x += SDP_THREADS) {
            if ((x & (15u * SDP_THREADS)) == 0u && *(volatile uint32_t *)&s_hits > sc.hit_limit) break;     // too repetitive: give up
            if (window16(qs, x) & kmark) continue;
            const uint32_t km = window16(qp, x) & sc.kmask;
            const uint32_t mix = (km * 0x9E3779B1u) & sc.kmask;
            uint32_t h = mix >> hshift, v;
            for (;;) {
                v = tab[h];
                if (v == 0u) break;
                if ((v >> 16 ^ mix) << 16 == 0u && (window16(tp, tile0 + (v & 0xffffu) - 1u) & sc.kmask) == km) { v &= 0xffffu; break; }
                h = (h + 1u) & (SLOTS - 1u);
            }
            // a hit is queued and worked on after the loop, one hit per thread (in the loop a wave would run the whole of
            // the hit code at nearly every position for the few lanes that have one); a full queue: at once
            if (v) {
                const uint32_t qi = atomicAdd(&s_nq, 1u);
                if (qi < sc.queue) hq[qi] = x << 16 | v;
                else process(x, v, tile0);
            }
        }
        __syncthreads();
        const uint32_t nq = min(*(volatile uint32_t *)&s_nq, sc.queue);
        for (uint32_t i = tid; i < nq; i += SDP_THREADS) process(hq[i] >> 16, hq[i] & 0xffffu, tile0);
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t n = s_nfrag;
        const bool many = n > sc.max_fragments || s_hits > sc.hit_limit;
        out_status[p] = many ? GBA_PAIR_TOO_MANY : n == 0u ? GBA_PAIR_EMPTY_GUIDE : GBA_PAIR_OK;
        nfrag[p] = many ? 0u : n;
    }
}

// Chain of one pair: one wave.  Sorts the pair's fragments in LDS, settles them 64 at a time, traces the best chain
// back and writes its blocks to chains + pair * max_fragments.
// Launched twice: for the pairs with at most SDP_SMALL fragments with LDS for that many (more waves per CU: most pairs of
// 10-kb reads have ~300), and for the others with LDS for the whole budget; a wave leaves at once when its pair is not of its launch.
constexpr uint32_t SDP_SMALL = 512;
__global__ __launch_bounds__(64) void k_sdp_chain(SdpConst sc, const uint2 *frags, const uint32_t *nfrag, int32_t *out_score,
                                                   uint32_t *out_nblk, gba_block *chains, uint32_t n_above, uint32_t n_upto)
{
    extern __shared__ uint32_t smem[];
    uint4 *F = (uint4 *)(((uintptr_t)smem + 15u) & ~(uintptr_t)15u);      // .x = query start << 16 | target start, .y = length, .z = score, .w = predecessor
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = nfrag[p];
    if (n == 0u) {
        if (tid == 0 && n_above == 0u) { out_score[p] = 0; out_nblk[p] = 0; }
        return;
    }
    if (n <= n_above || n > n_upto) return;
    // ---- sort by (query start, target start); the keys are distinct
    const uint2 *src = frags + (uint64_t)p * sc.cap;
    if (n <= SDP_SMALL) {
        // by rank: a lane keeps its (at most 8) fragments in registers, counts for each the keys below it with uniform LDS
        // reads of all keys, and puts it at that place: a third of the instructions of the bitonic network at ~300 fragments
        constexpr int OWN = SDP_SMALL / 64;
        uint2 own[OWN];
        uint32_t rank[OWN];
#pragma unroll
        for (int z = 0; z < OWN; z++) {
            const uint32_t i = tid + 64u * (uint32_t)z;
            own[z] = i < n ? src[i] : make_uint2(0xffffffffu, 0u);
            rank[z] = 0;
            if (i < n) F[i].x = own[z].x;
        }
        __syncthreads();
        const uint32_t n8 = (n + 7u) & ~7u;                   // (the keys behind the last fragment: set below, never smaller than a key)
        for (uint32_t i = n + tid; i < n8; i += 64) F[i].x = 0xffffffffu;
        __syncthreads();
        for (uint32_t q0 = 0; q0 < n8; q0 += 8) {
            uint32_t key[8];
#pragma unroll
            for (int y = 0; y < 8; y++) key[y] = F[q0 + y].x;
#pragma unroll
            for (int z = 0; z < OWN; z++)
#pragma unroll
                for (int y = 0; y < 8; y++) rank[z] += key[y] < own[z].x ? 1u : 0u;
        }
        __syncthreads();
        if (sc.stop == 2) return;
#pragma unroll
        for (int z = 0; z < OWN; z++)
            if (tid + 64u * (uint32_t)z < n) F[rank[z]] = make_uint4(own[z].x, own[z].y, 0u, 0xffffu);
        __syncthreads();
    } else {
        uint32_t P2 = 1;
        while (P2 < n) P2 <<= 1;
        for (uint32_t i = tid; i < P2; i += 64) {
            const uint2 f = i < n ? src[i] : make_uint2(0xffffffffu, 0u);
            F[i] = make_uint4(f.x, f.y, 0u, 0xffffu);
        }
        __syncthreads();
        if (sc.stop == 2) return;
        // bitonic, one compare-exchange per lane and step
        for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1)
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t e = tid; e < P2 / 2; e += 64) {
                    const uint32_t i = ((e & ~(j - 1u)) << 1) | (e & (j - 1u)), o = i | j;       // the pair (i, i + j) number e
                    const uint4 a = F[i], b = F[o];
                    if ((a.x > b.x) == ((i & k2) == 0u)) { F[i] = b; F[o] = a; }
                }
                __syncthreads();
            }
    }
    if (sc.stop == 3) return;
    // ---- chain
    const int32_t M = sc.match, INS = sc.ins, DEL = sc.del;
    int32_t best_all = 0; uint32_t end = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += 64) {
        const uint32_t f = b0 + tid;
        const bool live = f < n;
        const uint4 me = live ? F[f] : make_uint4(0u, 0u, 0u, 0u);
        const int32_t x = (int32_t)(me.x >> 16), y = (int32_t)(me.x & 0xffffu), diag = y - x;
        int32_t best = 0; uint32_t bp = 0xffffu;
        auto relax = [&](uint4 pr, uint32_t pidx, bool allowed) {
            const int32_t px = (int32_t)(pr.x >> 16), py = (int32_t)(pr.x & 0xffffu), pl = (int32_t)pr.y;
            const int32_t o = max(max(px + pl - x, py + pl - y), 0);
            const int32_t dd = diag - (py - px);
            const int32_t g = (int32_t)pr.z - max(dd * DEL, -dd * INS) - o * M;
            if (allowed && o < pl && g > best) { best = g; bp = pidx; }
        };
        // (b0 is a multiple of 64) eight earlier fragments per turn: their LDS reads travel together
        for (uint32_t q0 = 0; q0 < b0; q0 += 8) {
            uint4 pr[8];
#pragma unroll
            for (int z = 0; z < 8; z++) pr[z] = F[q0 + z];
#pragma unroll
            for (int z = 0; z < 8; z++) relax(pr[z], q0 + z, true);
        }
        // the fragments of this round among themselves: lane j is final once lanes 0..j-1 have been offered to it
        const uint32_t lim = min(64u, n - b0);
        for (uint32_t j = 0; j < lim; j++) {
            const int32_t mine = (int32_t)me.y * M + best;
            uint4 pr;
            pr.x = (uint32_t)__builtin_amdgcn_readlane((int32_t)me.x, j);
            pr.y = (uint32_t)__builtin_amdgcn_readlane((int32_t)me.y, j);
            pr.z = (uint32_t)__builtin_amdgcn_readlane(mine, j);
            pr.w = 0;
            relax(pr, b0 + j, tid > j);
        }
        if (live) {
            const int32_t sc_f = (int32_t)me.y * M + best;
            F[f].z = (uint32_t)sc_f; F[f].w = bp;
        }
        // best end so far: the highest score, the earliest fragment on ties
        int32_t v = live ? (int32_t)me.y * M + best : 0;
        uint32_t vi = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t ov = __shfl_xor(v, o, 64);
            const uint32_t oi = (uint32_t)__shfl_xor((int)vi, o, 64);
            if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
        }
        if (v > best_all) { best_all = v; end = vi; }
        __syncthreads();
    }
    if (sc.stop == 4) return;
    // trace back: lane 0 counts the chain and writes its fragment numbers where the scores were, last first
    uint32_t nc = 0;
    if (tid == 0) {
        for (uint32_t f = end; f != 0xffffu; f = F[f].w) nc++;
        uint32_t i = nc;
        for (uint32_t f = end; f != 0xffffu;) { const uint32_t nf = F[f].w; F[--i].z = f; f = nf; }
        out_score[p] = best_all; out_nblk[p] = nc;
    }
    nc = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)nc);
    __syncthreads();
    gba_block *outb = chains + (uint64_t)p * sc.max_fragments;
    for (uint32_t i = tid; i < nc; i += 64) {
        const uint4 a = F[F[i].z];
        uint32_t len = a.y;
        if (i + 1 < nc) {
            const uint4 b = F[F[i + 1].z];
            const int32_t ox = (int32_t)((a.x >> 16) + a.y) - (int32_t)(b.x >> 16), oy = (int32_t)((a.x & 0xffffu) + a.y) - (int32_t)(b.x & 0xffffu);
            const int32_t o = max(max(ox, oy), 0);
            len -= (uint32_t)o;
        }
        gba_block bl;
        bl.q_pos = a.x >> 16; bl.t_pos = a.x & 0xffffu; bl.length = len;
        outb[i] = bl;
    }
}

// counts -> offsets (single block), carried from part to part like k_scan1
__global__ __launch_bounds__(SCAN_T) void k_scan_cnt(const uint32_t *cnt, uint64_t *off, uint32_t n, uint64_t *carry)
{
    __shared__ uint64_t sm[17];
    const uint64_t c0 = *carry;
    __syncthreads();
    const uint64_t tot = scan_array(cnt, off, n, sm);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) off[p] += c0;
    if (threadIdx.x == 0) *carry = c0 + tot;
}

// the chains of a part packed into the caller's arena
__global__ __launch_bounds__(256) void k_sdp_emit(const gba_block *chains, uint32_t stride, const uint32_t *nblk, const uint64_t *off,
                                                  gba_block *out, uint64_t capacity, int32_t *status)
{
    const uint32_t p = blockIdx.x, nb = nblk[p];
    const uint64_t o = off[p];
    if (nb == 0) return;
    if (o + nb > capacity) { if (threadIdx.x == 0) status[p] = GBA_PAIR_NOSPACE; return; }
    const gba_block *src = chains + (uint64_t)p * stride;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) out[o + i] = src[i];
}

#define GBA_KBAND_KERNELS
#include "gba_kband.inc"
#undef GBA_KBAND_KERNELS

// ================================================================ host side
struct Buf {
    void *ptr = nullptr;
    size_t cap = 0;
};

// Scratch of gba_submit_device_noalloc: every buffer that entry point needs beyond the caller's
// inputs and outputs is carved from this array, one copy per GPU, held until the process ends.
// Host code never uses the array's name as an address: gba_create asks the runtime for the
// address on its device (hipGetSymbolAddress) and everything below works with that pointer.
static __device__ __align__(256) unsigned char scratch[GBA_NOALLOC_SCRATCH_BYTES];

}  // namespace

#ifndef GBA_NSLOT
#define GBA_NSLOT 3
#endif
constexpr int NSLOT = GBA_NSLOT;      // a large batch is run in parts that take the slots in turn, staggered, so that one part's guide
                              // construction and traceback run under the DP fill of the parts around it.  With two slots a part's
                              // target packing and guide wait for the traceback of the part two before it (the same slot's stream
                              // and scratch) and the GPU has no fill to run in between (10 - 20 ms per part of 123 k pairs of
                              // 10 kb); with three the next part's guide is ready long before a fill ends.
constexpr int NAUXS = 2;      // auxiliary streams per slot (wider classes run beside the bulk class)

// Everything one (half) batch in flight needs of its own.
struct Slot {
    Buf b_plan, b_guide, b_tc, b_rec, b_blk;     // device scratch (grow-only)
    Buf b_out;                                   // host-buffer batches: the packed blocks of the part in this slot, on their way to the host
    Buf b_cig;                                   // the same for its CIGAR operations
    Plan *h_plan = nullptr;                      // pinned
    uint64_t *h_cnt = nullptr;                   // pinned: blocks [0] and CIGAR operations [1] of the batch up to and including the part in this slot
    hipStream_t s = nullptr, aux[NAUXS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[NAUXS] = {}, ev_done = nullptr, ev_scan = nullptr;
    hipEvent_t ev_cnt = nullptr, ev_emit = nullptr, ev_d2h = nullptr;   // host-buffer batches: count read back; blocks packed; blocks copied out
    bool d2h_pending = false;                    // ev_d2h has been recorded for a copy out of b_out
    Dev dev;                                     // the part in this slot (kept for its deferred block emission)
    uint32_t first = 0;                          // its first pair
    hipEvent_t t_fill0 = nullptr, t_fill1 = nullptr, t_tb1 = nullptr;
    Plan pl;                                     // host copy of the half batch's plan
    uint64_t rows = 0;
    uint32_t launches = 0;
};

struct gba_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip = 0;
    bool have_scoring = false;
    bool lds_attr_set = false;       // the fill kernels' dynamic-LDS limit has been raised on this device
    bool sdp_attr_set = false;       // the same for k_sdp
    // k_sdp_match: target positions indexed at a time, at most (env GBA_SDP_TILE; power of two, 1024..16384), and hash slots
    // per indexed position (env GBA_SDP_SLOTS: 2 or 4).  Measured on 32 768 pairs of 10 kb, whole guide production, 512
    // threads per pair: 2048 x 4: 9.7 ms; 4096 x 2: 13.1; 4096 x 4: 14.1; 8192 x 2: 16.3
    // block scratch of the tracebacks (blk_region_start): a pair's own region holds rows / blk_div + 31 blocks (env GBA_BLK_DIV,
    // 1 = room for the worst case, no pool), the overflow pool total rows / blk_pool_div blocks plus one pair of the largest
    // size (env GBA_BLK_POOL_DIV); blk_div_now is what the running batch uses (1 when it is being repeated)
    uint32_t blk_div = 3, blk_pool_div = 8, blk_div_now = 3;
    // slots of the part pipeline a batch uses (each holds the scratch of one part): batches whose buffers are on the
    // device (env GBA_SLOTS_DEVICE) and host-buffer batches (env GBA_SLOTS_HOST); 2 .. NSLOT
    uint32_t slots_device = NSLOT, slots_host = 2;
    bool shape_parts = true;          // a context that has the GPU to itself shapes the parts of a device-resident batch for its head and tail (run_device_once; env GBA_EDGE_PARTS=0: equal parts always)
    std::vector<uint32_t> part_shape;   // env GBA_PART_SHAPE (experiments): relative sizes of the parts of a device-resident batch, "1,2,2,2,1"
    bool alone_on_device = true;     // no other context of the process had a batch in flight on the device when this one began
    bool fill_chain = false;         // a part's fill launches wait for the end of the bulk fill of the part before (env GBA_FILL_CHAIN)
    uint32_t scratch_retries = 0;    // batches repeated with full-size block scratch
    uint32_t sdp_queue = 1024;       // hits queued per tile (env GBA_SDP_QUEUE, 64..16384)
    uint32_t sdp_tile_max = 2048;
    uint32_t sdp_slots_per_pos = 4;
    gba_sdp_params guide_sp;         // gba_set_guide_producer: batches without guide blocks get their guides from gba_sdp's chains
    bool have_guide_sp = false;
    const int32_t *cur_guide_status = nullptr;   // set around run_device for such a batch
    gba_scoring sc;
    FillConst fc;
    TbConst tbc;
    uint32_t affine_top_slots = 0;   // affine gaps: a third lane class of this many slots (env GBA_AFFINE_TOP_SLOTS; 0: none).  Measured on 98 304 pairs
                                     // of 20 kb, band 16 (99 % of them fit 96 slots): none 221 GCUPS, 96 slots 212, 80 slots 190, 128 slots 187: three
                                     // lane-per-pair waves per CU are no faster than the narrow wave-per-pair kernel
    uint32_t tb_few_max = 32768;         // affine lane-per-pair tracebacks of at most this many pairs (two waves per CU) use the large record window (env GBA_TB_FEW_MAX)
    uint32_t tb_wave_max = 6144;         // a batch that ran on the wave kernels altogether with at most this many narrow-class pairs is walked
                                         // back one wave per pair too (measured crossover: 6 600 pairs of 10 kb, 10 600 of 20 kb affine; env GBA_TB_WAVE_MAX)
    uint32_t quad_affine_min = 10240;   // narrow affine wave class: this many pairs or more run four pairs per wave (fewer leave SIMDs
                                        // without a wave that way; measured crossover between 8 192 and 12 288 pairs of 20 kb; env GBA_QUAD_AFFINE_MIN)
    uint32_t wave_pairs_affine_prof = 24576;   // the same for affine batches with a cost profile (GBA_WAVE_PAIRS_AFFINE sets both)
    uint32_t wave_pairs = 20480, wave_pairs_affine = 57344;   // batches (parts) smaller than this run wave-per-pair (env GBA_WAVE_PAIRS, GBA_WAVE_PAIRS_AFFINE)
    uint32_t score_a = 0, score_b = 0;   // largest cost of a row move / of a move along a row (k_meta's score-range check)
    uint32_t part_pairs = 122880;    // a batch of more than ~1.07 x this many pairs is run in staggered parts of about this
                                     // size (env GBA_PART_PAIRS, 0 = never).  Two lane-per-pair waves per SIMD is what the GPU
                                     // can use (1920 waves here); measured on 10-kb pairs: 245 760 pairs in one piece 658 GCUPS,
                                     // in two staggered parts 734, a single batch of 122 880 pairs 714
    // device scratch (grow-only)
    Buf b_ref, b_refm;               // reference for batches without a target arena (gba_set_reference): 2 bits per base, and a mask bit per base
    uint64_t ref_len = 0;
    Buf s_trc;
    Buf b_pair32[N_PAIR32], b_pair64[N_PAIR64], b_needed;
    Slot slot[NSLOT];
    // host-buffer batches: device copies of the caller's batch, the copy streams, the worker thread
    Buf s_q, s_t, s_blk, s_qoff, s_toff, s_boff, s_qlen, s_tlen, s_nblk, s_res, s_cigoff, s_gout;
    Buf s_prof;                      // host-buffer batches with a quality-value profile: its device copy
    Buf kb[9];                       // gba_kband: band widths, strand flags, cost profile, arrow offsets, matrix, arrows, block slots, counts, offsets
    hipStream_t s_in = nullptr, s_out = nullptr;     // host-to-device and device-to-host copies, beside the kernels
    std::vector<hipEvent_t> ev_in;                   // per part: its inputs have arrived
    std::vector<hipEvent_t> ev_sdp;                  // per part of the guide producer: its sequences have arrived
    std::vector<void *> host_allocs;                 // gba_host_alloc
    uint64_t *h_needed = nullptr;    // pinned
    bool in_flight = false;          // a host-buffer batch has been submitted and not yet waited for
    std::thread worker;              // runs host-buffer batches, so that gba_submit returns at once
    std::mutex mu;
    std::condition_variable cv;
    bool job_ready = false, job_done = false, quit = false;
    gba_batch job_batch = {};
    gba_results *job_out = nullptr;
    int job_rc = 0;
    hipEvent_t ev[2] = {};           // start and end of a batch
    gba_timing timing = {};
    // gba_submit_device_noalloc: address of this device's copy of `scratch` (null: the lookup failed)
    // and, while such a call runs, the bytes already handed out (fixed_on: ensure() carves, never allocates)
    unsigned char *fixed_base = nullptr;
    size_t fixed_used = 0;
    bool fixed_on = false;
};

namespace {

#define HIPCHK(ctx, call)                                        \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) { (ctx)->last_hip = (int)e__; return e__ == hipErrorOutOfMemory ? GBA_E_NOMEM : GBA_E_HIP; } \
    } while (0)

int ensure(gba_ctx *c, Buf &b, size_t bytes)
{
    if (c->fixed_on) {
        // no-allocation mode: a fresh 256-byte aligned piece of the fixed scratch per request
        if (bytes <= b.cap) return GBA_OK;
        const size_t at = (c->fixed_used + 255) & ~size_t(255);
        if (bytes > GBA_NOALLOC_SCRATCH_BYTES || at > GBA_NOALLOC_SCRATCH_BYTES - bytes) return GBA_E_TOOBIG;
        b.ptr = c->fixed_base + at; b.cap = bytes;
        c->fixed_used = at + bytes;
        return GBA_OK;
    }
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

// Slots of the widest lane class.  Affine gaps: a third class only where asked for (gba_ctx::affine_top_slots), and for batches
// with a cost profile, whose wave-per-pair kernel composes a min-plus map per column and is four times slower than the plain one.
uint32_t top_lane_slots(const gba_ctx *c, const Dev &d)
{
    if (c->sc.mode != GBA_MODE_AFFINE) return 128u;
    return c->affine_top_slots ? c->affine_top_slots : d.prof ? 128u : 0u;
}

int launch_fill(gba_ctx *c, const Dev &d, uint32_t cls, uint32_t count, hipStream_t st)
{
    const bool aff = c->sc.mode == GBA_MODE_AFFINE;
    const uint32_t slots = cls_slots((int)cls, top_lane_slots(c, d));
    const size_t rings = aff ? 2 : 1;
    const size_t lds = (rings * (slots + PAD_SLOTS) * 64 + LDS_TAB) * 4;
    if (!c->lds_attr_set) {                     // per device, and a ctx lives on one device
        HIPCHK(c, hipFuncSetAttribute((const void *)k_fill<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(c, hipFuncSetAttribute((const void *)k_fill<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(c, hipFuncSetAttribute((const void *)(k_fill<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(c, hipFuncSetAttribute((const void *)(k_fill<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(c, hipFuncSetAttribute((const void *)(k_fill<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(c, hipFuncSetAttribute((const void *)(k_fill<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        c->lds_attr_set = true;
    }
    const dim3 grid((count + 63) / 64), block(64);
    if (aff && d.prof) hipLaunchKernelGGL((k_fill<true, true>), grid, block, lds, st, d, c->fc, cls, slots);
    else if (aff) hipLaunchKernelGGL(k_fill<true>, grid, block, lds, st, d, c->fc, cls, slots);
    else if (d.prof && c->sc.align_type == GBA_ALIGN_LOCAL) hipLaunchKernelGGL((k_fill<false, true, true>), grid, block, lds, st, d, c->fc, cls, slots);
    else if (d.prof) hipLaunchKernelGGL((k_fill<false, true>), grid, block, lds, st, d, c->fc, cls, slots);
    else if (c->sc.align_type == GBA_ALIGN_LOCAL) hipLaunchKernelGGL((k_fill<false, false, true>), grid, block, lds, st, d, c->fc, cls, slots);
    else hipLaunchKernelGGL(k_fill<false>, grid, block, lds, st, d, c->fc, cls, slots);
    HIPCHK(c, hipGetLastError());
    return GBA_OK;
}

// The pair range [p0, p0 + cnt) of a batch as a batch of its own: every per-pair array shifted.
Dev sub_dev(const Dev &g, uint32_t p0, uint32_t cnt, Slot &sl)
{
    Dev d = g;
    d.n = cnt;
    d.q_off += p0; d.t_off += p0; d.blk_off += p0; d.q_len += p0; d.t_len += p0; d.n_blk += p0;
    if (d.t_rc) d.t_rc += p0;
    if (d.guide_status) d.guide_status += p0;
    d.status += p0; d.qS += p0; d.tS += p0; d.Q += p0; d.T += p0; d.rows += p0; d.cols += p0; d.wcap += p0;
    d.span += p0; d.lo_last += p0; d.cls += p0; d.order += p0; d.recdw += p0;
    d.end_i += p0; d.end_j += p0; d.end_lo += p0; d.end_ptr += p0;
    d.row_off += p0; d.col_off += p0; d.word_off += p0; d.rec_off += p0; d.ncells += p0; d.dp_score += p0; d.blk_end += p0;
    d.plan = (Plan *)sl.b_plan.ptr;
    return d;
}

// k_meta, arena offsets; the totals travel to the host
int phase_sizes(gba_ctx *c, Slot &sl, const Dev &d)
{
    const gba_scoring &sc = c->sc;
    // a batch with a cost profile takes its per-base costs from there, each within +-127 (+-63 with affine gaps, where open +
    // extend, already in score_a / score_b, may be larger than that): the bound is the profile's range, not the matrix's
    uint32_t sa = c->score_a, sb = c->score_b;
    if (d.prof) {
        const uint32_t pmax = sc.mode == GBA_MODE_AFFINE ? 63u : 127u;
        sa = sa > pmax ? sa : pmax; sb = sb > pmax ? sb : pmax;
    }
    hipLaunchKernelGGL(k_meta, dim3((d.n + 255) / 256), dim3(256), 0, sl.s, d, sc.mode, sc.kband_k, sa, sb,
                       (uint32_t)(UNREACH >> (sc.mode == GBA_MODE_AFFINE ? 6 : 5)), sc.band, sc.max_band);
    hipLaunchKernelGGL(k_scan3, dim3(1), dim3(SCAN_T), 0, sl.s, d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(sl.h_plan, d.plan, sizeof(Plan), hipMemcpyDeviceToHost, sl.s));
    return GBA_OK;
}

// after the sizes have arrived: arenas, then target codes, guide stream, classes; the plan travels to the host
int phase_guide(gba_ctx *c, Slot &sl, Dev &d)
{
    const gba_scoring &sc = c->sc;
    HIPCHK(c, hipStreamSynchronize(sl.s));
    sl.rows = sl.h_plan->total_rows;
    const uint64_t cols = sl.h_plan->total_cols;
    { int rc = ensure(c, sl.b_guide, (size_t)(sl.h_plan->total_words + 16) * 4); if (rc) return rc; }
    { int rc = ensure(c, sl.b_tc, (size_t)(cols + 48ull * d.n + 64 + 2 * ARENA_SLACK)); if (rc) return rc; }
    d.blk_div = c->blk_div_now;
    d.blk_pool_off = sl.rows / d.blk_div + 32ull * d.n + 32;
    d.blk_pool_cap = d.blk_div > 1u ? sl.rows / c->blk_pool_div + MAX_ROWS + 64 : 0;
    { int rc = ensure(c, sl.b_blk, (size_t)(d.blk_pool_off + d.blk_pool_cap + 4) * sizeof(gba_block)); if (rc) return rc; }
    d.stream = (uint32_t *)sl.b_guide.ptr;
    d.tcodes = (uint8_t *)sl.b_tc.ptr + ARENA_SLACK;
    d.blk_scratch = (gba_block *)sl.b_blk.ptr;
    hipLaunchKernelGGL(k_pack, dim3(d.n), dim3(256), 0, sl.s, d);
    hipLaunchKernelGGL(k_guide, dim3(d.n), dim3(64), 0, sl.s, d, sc.mode, sc.band, sc.max_band, sc.kband_k);
    // A lane-per-pair launch lasts as long as its longest pair whatever the number of pairs, and needs
    // two waves on every SIMD (131 072 pairs) to fill the GPU; the wave-per-pair kernels take several
    // times the instructions per cell but fill the GPU from a few thousand pairs on.  Below wave_pairs
    // pairs the whole batch runs one wave per pair.
    const bool affine = sc.mode == GBA_MODE_AFFINE;
    // (affine gaps with a cost profile: the wave-per-pair kernel composes a min-plus map per column, and the lane classes win from
    // about 24 000 pairs of 10 kb on: 32 768 pairs 146 against 91 GCUPS, 16 384 pairs 76 against 83)
    const uint32_t lanes_from = affine ? (d.prof ? c->wave_pairs_affine_prof : c->wave_pairs_affine) : c->wave_pairs;
    const int lane_classes = d.n < lanes_from ? 0
                           : affine ? (top_lane_slots(c, d) ? LANE_CLS_AFFINE + 1 : LANE_CLS_AFFINE) : LANE_CLS_LINEAR;
    hipLaunchKernelGGL(k_plan, dim3(1), dim3(SCAN_T), 0, sl.s, d, lane_classes, 1, top_lane_slots(c, d));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(sl.h_plan, d.plan, sizeof(Plan), hipMemcpyDeviceToHost, sl.s));
    return GBA_OK;
}

// after the plan has arrived: records arena, then one fill launch per non-empty class, widest
// class first (its few waves run longest): the widest classes go to the slot's auxiliary streams
// and run beside the bulk on the slot's main stream.  The lane classes share ONE traceback launch
// on the main stream after all their fills (the pairs of a class are neighbours in `order`, and
// the traceback does not care about the class); the wave-per-pair class has its own traceback
// behind its fill.  An auxiliary stream therefore carries fills only, and the next part's wide
// class, which shares a hardware queue with this one's, does not wait behind a traceback.
int phase_fill(gba_ctx *c, Slot &sl, Dev &d, gba_pair_result *res, uint32_t max_aux, hipEvent_t after = nullptr)
{
    const gba_scoring &sc = c->sc;
    HIPCHK(c, hipStreamSynchronize(sl.s));
    sl.pl = *sl.h_plan;
    const Plan &pl = sl.pl;
    { int rc = ensure(c, sl.b_rec, (size_t)(pl.total_rec_dwords + 16) * 4 + ARENA_SLACK); if (rc) return rc; }
    d.rec = (uint32_t *)((uint8_t *)sl.b_rec.ptr + ARENA_SLACK);
    hipStream_t s = sl.s;
    hipLaunchKernelGGL(k_results_init, dim3((d.n + 255) / 256), dim3(256), 0, s, d, res);
    if (after) HIPCHK(c, hipStreamWaitEvent(s, after, 0));
    HIPCHK(c, hipEventRecord(sl.t_fill0, s));
    const bool aff = sc.mode == GBA_MODE_AFFINE;
    const bool local = sc.align_type == GBA_ALIGN_LOCAL;    // linear gaps only (gba_set_scoring)
    auto launch_tb = [&](int k, hipStream_t st) {          // a wave-per-pair class
        const uint32_t cnt = pl.cls_count[k];
        if (aff) hipLaunchKernelGGL(k_traceback_wide<true>, dim3(cnt), dim3(64), 0, st, d, c->tbc, res, (uint32_t)k);
        else if (local) hipLaunchKernelGGL((k_traceback_wide<false, true>), dim3(cnt), dim3(64), 0, st, d, c->tbc, res, (uint32_t)k);
        else hipLaunchKernelGGL(k_traceback_wide<false>, dim3(cnt), dim3(64), 0, st, d, c->tbc, res, (uint32_t)k);
    };
    // The lane-per-pair traceback serves the lane classes AND the narrow wave-per-pair class (order[0 .. first pair of the
    // widest class)): a wave that walks one pair on the scalar unit pays off for rows of hundreds of cells, not for these.
    auto launch_tb_lanes = [&](hipStream_t st) {
        const uint32_t cnt = pl.cls_base[CLS_WIDE];
        const dim3 grid((cnt + TB_LANES - 1) / TB_LANES), block(TB_LANES);
        if (local) hipLaunchKernelGGL((k_traceback<false, TB_RW, true>), grid, block, 0, st, d, c->tbc, res, 0u, cnt);
        else if (aff && cnt <= c->tb_few_max) hipLaunchKernelGGL((k_traceback<true, TB_RW_FEW>), grid, block, 0, st, d, c->tbc, res, 0u, cnt);
        else if (aff) hipLaunchKernelGGL(k_traceback<true>, grid, block, 0, st, d, c->tbc, res, 0u, cnt);
        else hipLaunchKernelGGL(k_traceback<false>, grid, block, 0, st, d, c->tbc, res, 0u, cnt);
    };
    HIPCHK(c, hipEventRecord(sl.ev_fork, s));
    bool used[NAUXS] = {}, wave_on_aux[NAUXS] = {}, wave_on_main[NCLS] = {};
    int nonempty = 0;
    for (int k = 0; k < NCLS; k++) nonempty += pl.cls_count[k] ? 1 : 0;
    uint32_t n_aux = nonempty > 0 ? (uint32_t)nonempty - 1u : 0u;       // the narrowest class stays on the main stream
    if (n_aux > max_aux) n_aux = max_aux;
    sl.launches = 0;
    for (int k = NCLS - 1; k >= 0; k--) {
        const uint32_t cnt = pl.cls_count[k];
        if (!cnt) continue;
        hipStream_t st = s;
        const bool aux = sl.launches < n_aux, wave = k >= CLS_WNARROW;
        const int a = (int)sl.launches;
        if (aux) {
            st = sl.aux[a];
            HIPCHK(c, hipStreamWaitEvent(st, sl.ev_fork, 0));
            used[a] = true;
        }
        if (wave) {
            if (k == CLS_WIDE) {
                if (aff && d.prof) hipLaunchKernelGGL((k_fill_wide_affine<false, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (aff) hipLaunchKernelGGL((k_fill_wide_affine<false>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (local && d.prof) hipLaunchKernelGGL((k_fill_wide<4, true, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (local) hipLaunchKernelGGL((k_fill_wide<4, false, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (d.prof) hipLaunchKernelGGL((k_fill_wide<4, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else hipLaunchKernelGGL(k_fill_wide_cols, dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
            } else {
                if (aff && d.prof) hipLaunchKernelGGL((k_fill_wide_affine<true, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (aff && cnt >= c->quad_affine_min) hipLaunchKernelGGL(k_fill_quad_affine, dim3((cnt + 3) / 4), dim3(64), 0, st, d, c->fc, (uint32_t)k, cnt);
                else if (aff) hipLaunchKernelGGL((k_fill_wide_affine<true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (local && d.prof) hipLaunchKernelGGL((k_fill_wide<1, true, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (local) hipLaunchKernelGGL((k_fill_wide<1, false, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else if (d.prof) hipLaunchKernelGGL((k_fill_wide<1, true>), dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
                else hipLaunchKernelGGL(k_fill_wide<1>, dim3(cnt), dim3(64), 0, st, d, c->fc, (uint32_t)k);
            }
            HIPCHK(c, hipGetLastError());
        } else {
            int rc = launch_fill(c, d, (uint32_t)k, cnt, st);
            if (rc) return rc;
        }
        if (aux) {
            if (k == CLS_WIDE) { launch_tb(k, st); wave_on_aux[a] = true; }     // the widest class's traceback follows its fill on its own stream
            HIPCHK(c, hipEventRecord(sl.ev_join[a], st));
        } else if (k == CLS_WIDE) wave_on_main[k] = true;
        sl.launches++;
    }
    HIPCHK(c, hipEventRecord(sl.t_fill1, s));
    for (int a = 0; a < NAUXS; a++)
        if (used[a] && !wave_on_aux[a]) HIPCHK(c, hipStreamWaitEvent(s, sl.ev_join[a], 0));
    if (wave_on_main[CLS_WIDE]) launch_tb(CLS_WIDE, s);
    // a batch that ran on the wave kernels altogether and leaves the lane-per-pair traceback with too few waves to hide
    // its walk (one dependent chain per pair, however few pairs): the narrow wave class is walked one wave per pair too
    if (pl.cls_base[CLS_WNARROW] == 0 && pl.cls_count[CLS_WNARROW] > 0 && (local || pl.cls_count[CLS_WNARROW] <= c->tb_wave_max)) launch_tb(CLS_WNARROW, s);
    else if (pl.cls_base[CLS_WIDE] > 0) launch_tb_lanes(s);
    HIPCHK(c, hipGetLastError());
    for (int a = 0; a < NAUXS; a++)
        if (used[a] && wave_on_aux[a]) HIPCHK(c, hipStreamWaitEvent(s, sl.ev_join[a], 0));
    HIPCHK(c, hipEventRecord(sl.t_tb1, s));
    HIPCHK(c, hipEventRecord(sl.ev_done, s));
    return GBA_OK;
}

// A batch with pairs must bring every array the kernels index (the target arena may be absent:
// windows of the ctx reference; the strand flags are optional).
bool batch_complete(const gba_ctx *c, const gba_batch *b)
{
    if (b->n_pairs == 0) return true;
    if (!b->q_arena || !b->q_off || !b->t_off || !b->q_len || !b->t_len) return false;
    if (!b->blocks) return c->have_guide_sp;       // the guides are made on the device
    return b->blk_off && b->n_blk;
}

// What the timing of a batch is made of, folded in part by part.
struct TimeAcc {
    float f0 = 1e30f, f1 = 0, tb1 = 0, fill_sum = 0;
    std::vector<std::pair<float, float>> fills;       // the parts' bulk fill launches, ms from the batch's first event
    uint32_t launches = 0, wide = 0;
    uint64_t cells = 0, rows = 0, rec = 0;
};

// The part that ran last in `sl` has completed: add its events and counts.
int fold_timing(gba_ctx *c, Slot &sl, TimeAcc &a)
{
    float x, e, t;
    HIPCHK(c, hipEventElapsedTime(&x, c->ev[0], sl.t_fill0));
    HIPCHK(c, hipEventElapsedTime(&e, c->ev[0], sl.t_fill1));
    HIPCHK(c, hipEventElapsedTime(&t, c->ev[0], sl.t_tb1));
    a.f0 = x < a.f0 ? x : a.f0; a.f1 = e > a.f1 ? e : a.f1; a.tb1 = t > a.tb1 ? t : a.tb1;
    a.fill_sum += e - x;
    a.fills.emplace_back(x, e);
    a.wide += sl.pl.cls_count[CLS_WIDE] + sl.pl.cls_count[CLS_WNARROW];       // pairs filled one wave per pair
    a.launches += sl.launches; a.cells += sl.pl.total_cells; a.rows += sl.rows; a.rec += sl.pl.total_rec_dwords * 4;
    return GBA_OK;
}

// Host-buffer batches: what run_device needs to move a batch in and its results out part by part.
struct HostIO {
    const gba_batch *hb;          // the caller's batch (host pointers)
    gba_results *hout;            // the caller's result buffers (host pointers)
    gba_pair_result *d_res;       // device copy of the pair results
    bool all_ahead = false;       // every part's inputs are queued at once (page-locked arenas, a hardware queue to spare)
    uint64_t need_before = 0;     // blocks of the parts emitted so far
    uint64_t ops_before = 0;      // CIGAR operations of the parts emitted so far
    uint64_t *d_cigoff = nullptr; // device copy of the operation offsets (n_pairs + 1), null: no CIGARs wanted
    // byte / element ranges of the three arenas that are on the device already
    uint64_t have[3][2] = {{0, 0}, {0, 0}, {0, 0}};
};

// Copies what part [first, first + cnt) of a host batch needs and the device does not have yet:
// the per-pair arrays of the part and the stretch of each arena its pairs point into (arenas are
// normally laid out in pair order, so consecutive parts need consecutive stretches; whatever lies
// between the stretch on the device and the new one is copied too, so one interval stays valid).
int stage_part(gba_ctx *c, HostIO &io, uint32_t first, uint32_t cnt, hipEvent_t done, bool arenas_only = false)
{
    const gba_batch &b = *io.hb;
    hipStream_t st = c->s_in;
    // The copies are the runtime's (copy engines): kernels that read or write host memory over the link
    // themselves were measured and take far more from the DP kernels running beside them than they save.
    auto move = [&](void *dst, const void *src_host, size_t bytes) -> int {
        if (bytes) HIPCHK(c, hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, st));
        return GBA_OK;
    };
    struct { Buf *buf; const void *src; size_t esz; } per_pair[] = {
        {&c->s_qoff, b.q_off, 8}, {&c->s_toff, b.t_off, 8}, {&c->s_boff, b.blk_off, 8},
        {&c->s_qlen, b.q_len, 4}, {&c->s_tlen, b.t_len, 4}, {&c->s_nblk, b.n_blk, 4}, {&c->s_trc, b.t_rc, 1}};
    for (int k = 0; k < 7 && !arenas_only; k++) {
        auto &x = per_pair[k];
        if (!x.src) continue;
        int rc = move((char *)x.buf->ptr + (size_t)first * x.esz, (const char *)x.src + (size_t)first * x.esz, (size_t)cnt * x.esz);
        if (rc) return rc;
    }
    struct { int which; Buf *buf; const uint8_t *src; uint64_t extent; size_t esz; const uint64_t *off; const uint32_t *len; } arena[] = {
        {0, &c->s_q, b.q_arena, b.q_arena_bytes, 1, b.q_off, b.q_len},
        {1, &c->s_t, b.t_arena, b.t_arena_bytes, 1, b.t_off, b.t_len},
        {2, &c->s_blk, (const uint8_t *)b.blocks, b.n_blocks_total, sizeof(gba_block), b.blk_off, b.n_blk}};
    for (auto &a : arena) {
        if (!a.src) continue;                        // no target arena: windows of the ctx reference
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t p = first; p < first + cnt; p++) {
            const uint64_t o = a.off[p], l = a.len[p];
            if (l == 0 || o > a.extent || l > a.extent - o) continue;      // the kernels refuse such a pair without touching it
            lo = o < lo ? o : lo; hi = o + l > hi ? o + l : hi;
        }
        if (lo >= hi) continue;
        const bool packed = a.which == 0 && b.q_packed;
        if (packed) {       // packed reads move and are spelt out 16 bases at a time
            const uint64_t top = (a.extent + 15) & ~15ull;
            lo &= ~15ull; hi = (hi + 15) & ~15ull; hi = hi < top ? hi : top;
        }
        uint64_t *have = io.have[a.which];
        uint64_t pieces[2][2]; int np = 0;
        if (have[0] >= have[1]) { pieces[np][0] = lo; pieces[np][1] = hi; np++; have[0] = lo; have[1] = hi; }
        else {
            if (lo < have[0]) { pieces[np][0] = lo; pieces[np][1] = have[0]; np++; have[0] = lo; }
            if (hi > have[1]) { pieces[np][0] = have[1]; pieces[np][1] = hi; np++; have[1] = hi; }
        }
        for (int k = 0; k < np; k++) {
            if (packed) {           // the packed bytes go where the kernels read them: nothing is spelt out
                const uint64_t f16 = pieces[k][0] / 16, n16 = (pieces[k][1] - pieces[k][0]) / 16;
                int rc = move((char *)a.buf->ptr + f16 * 4, a.src + f16 * 4, (size_t)n16 * 4);
                if (rc) return rc;
                continue;
            }
            int rc = move((char *)a.buf->ptr + pieces[k][0] * a.esz, a.src + pieces[k][0] * a.esz,
                          (size_t)(pieces[k][1] - pieces[k][0]) * a.esz);
            if (rc) return rc;
        }
    }
    HIPCHK(c, hipEventRecord(done, st));
    return GBA_OK;
}

// Deferred end of the part in `sl` (host-buffer batches): its block count has been read back, so
// its blocks are packed into a device buffer of exactly that size and both the pair results and
// the blocks go to the caller's buffers on the device-to-host stream.
int finish_part(gba_ctx *c, HostIO &io, Slot &sl)
{
    HIPCHK(c, hipEventSynchronize(sl.ev_cnt));
    const uint64_t after = sl.h_cnt[0], before = io.need_before, cnt = after - before;
    const bool want = io.hout->blocks != nullptr;      // otherwise only the pair results (and CIGARs) travel
    const uint64_t cap = io.hout->blocks_capacity;
    const uint64_t fit = !want || before >= cap ? 0 : (after <= cap ? cnt : cap - before);     // blocks of this part the caller's arena takes
    const bool cig = io.d_cigoff != nullptr;
    const uint64_t ops_after = cig ? sl.h_cnt[1] : 0, ops_before = io.ops_before, ops_cnt = ops_after - ops_before;
    const uint64_t ops_cap = io.hout->cigar_capacity;
    const uint64_t ops_fit = ops_before >= ops_cap ? 0 : (ops_after <= ops_cap ? ops_cnt : ops_cap - ops_before);
    if (sl.d2h_pending) {
        // the previous part of this slot may still be copying out of b_out / b_cig
        if ((want && (size_t)cnt * sizeof(gba_block) > sl.b_out.cap) || (size_t)ops_cnt * 4 > sl.b_cig.cap) HIPCHK(c, hipEventSynchronize(sl.ev_d2h));
        else HIPCHK(c, hipStreamWaitEvent(sl.s, sl.ev_d2h, 0));
    }
    if (want) { int rc = ensure(c, sl.b_out, (size_t)cnt * sizeof(gba_block) + 16); if (rc) return rc; }
    if (cig) { int rc = ensure(c, sl.b_cig, (size_t)ops_cnt * 4 + 16); if (rc) return rc; }
    // offsets are those of the whole batch (that is what the caller's arena is indexed by), so the
    // part's buffer is addressed as if it began `before` blocks earlier
    hipLaunchKernelGGL(k_emit, dim3(sl.dev.n), dim3(256), 0, sl.s, sl.dev, io.d_res + sl.first,
                       want ? (gba_block *)sl.b_out.ptr - before : (gba_block *)nullptr, cap, want ? 1 : 0);
    if (cig)
        hipLaunchKernelGGL(k_cigar<true>, dim3(sl.dev.n), dim3(64), 0, sl.s, sl.dev, io.hout->cigar_flags, io.d_cigoff + sl.first,
                           (uint32_t *)sl.b_cig.ptr - ops_before, ops_cap);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(sl.ev_emit, sl.s));
    HIPCHK(c, hipEventRecord(sl.ev_done, sl.s));
    HIPCHK(c, hipStreamWaitEvent(c->s_out, sl.ev_emit, 0));
    HIPCHK(c, hipMemcpyAsync(io.hout->pairs + sl.first, io.d_res + sl.first, (size_t)sl.dev.n * sizeof(gba_pair_result),
                             hipMemcpyDeviceToHost, c->s_out));
    if (fit)
        HIPCHK(c, hipMemcpyAsync(io.hout->blocks + before, sl.b_out.ptr, (size_t)fit * sizeof(gba_block), hipMemcpyDeviceToHost, c->s_out));
    if (cig) {
        // the entry one past the part is the next part's first (the same value) or the end of the batch
        HIPCHK(c, hipMemcpyAsync(io.hout->cigar_off + sl.first, io.d_cigoff + sl.first, ((size_t)sl.dev.n + 1) * 8, hipMemcpyDeviceToHost, c->s_out));
        if (ops_fit)
            HIPCHK(c, hipMemcpyAsync(io.hout->cigar_ops + ops_before, sl.b_cig.ptr, (size_t)ops_fit * 4, hipMemcpyDeviceToHost, c->s_out));
    }
    HIPCHK(c, hipEventRecord(sl.ev_d2h, c->s_out));
    sl.d2h_pending = true;
    io.need_before = after;
    io.ops_before = ops_after;
    return GBA_OK;
}

// Runs one batch whose inputs and outputs are device pointers.  A batch much larger than
// gba_ctx::part_pairs is run in parts of about that size which alternate between the two slots:
// a part's target packing and guide construction are enqueued while the part before it fills, and
// its traceback and block copy run under the fill of the part after it; a slot's scratch is reused
// by every second part.  The host waits twice per part (arena sizes, launch grids) and once at the end.
//
// With `io` the batch comes from and goes to host buffers (b and out then hold the device copies'
// addresses): the inputs of part h + 1 are copied in while part h is being launched, and a part's
// blocks are packed and copied out one part later, once its block count has been read back, so
// copies in both directions run beside the kernels of the parts around them.
constexpr int RETRY_FULL_SCRATCH = 1 << 20;     // internal: never leaves run_device

int run_device_once(gba_ctx *c, const gba_batch *b, gba_results *out, HostIO *io);

// Batches in flight on each device, over all contexts of the process: a context that has the GPU to itself shapes the parts
// of a batch differently from one whose batch runs beside another context's (run_device_once, edge parts).
std::atomic<int> g_batches_on_device[64];
struct BatchInFlight {
    int dev, others;
    explicit BatchInFlight(int d) : dev(d & 63), others(g_batches_on_device[d & 63].fetch_add(1)) {}
    ~BatchInFlight() { g_batches_on_device[dev].fetch_sub(1); }
};

int run_device(gba_ctx *c, const gba_batch *b, gba_results *out, HostIO *io = nullptr)
{
    BatchInFlight in_flight(c->device);
    c->alone_on_device = in_flight.others == 0;
    HostIO fresh;
    if (io) fresh = *io;
    c->blk_div_now = c->blk_div;
    int rc = run_device_once(c, b, out, io);
    if (rc == RETRY_FULL_SCRATCH) {
        // rare: alignments with far more blocks than rows / blk_div, more of them than the pool takes
        if (io) *io = fresh;
        c->blk_div_now = 1;
        c->scratch_retries++;
        rc = run_device_once(c, b, out, io);
        c->blk_div_now = c->blk_div;
        if (rc == RETRY_FULL_SCRATCH) rc = GBA_E_HIP;    // cannot happen: full-size regions take every alignment
    }
    return rc;
}

int run_device_once(gba_ctx *c, const gba_batch *b, gba_results *out, HostIO *io)
{
    const uint32_t n = b->n_pairs;
    const bool cig = io ? io->d_cigoff != nullptr : out->cigar_ops != nullptr;
    uint64_t *cig_off = io ? io->d_cigoff : out->cigar_off;
    const bool want_blocks = (io ? io->hout->blocks : out->blocks) != nullptr;
    out->blocks_needed = 0; out->cigar_needed = 0;
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) {
        if (cig && io) io->hout->cigar_off[0] = 0;
        else if (cig) { HIPCHK(c, hipMemsetAsync(cig_off, 0, 8, c->slot[0].s)); HIPCHK(c, hipStreamSynchronize(c->slot[0].s)); }
        return GBA_OK;
    }
    for (int i = 0; i < N_PAIR32; i++) { int rc = ensure(c, c->b_pair32[i], (size_t)n * 4); if (rc) return rc; }
    for (int i = 0; i < N_PAIR64; i++) { int rc = ensure(c, c->b_pair64[i], (size_t)n * 8); if (rc) return rc; }
    { int rc = ensure(c, c->b_needed, 32); if (rc) return rc; }
    for (auto &sl : c->slot) { int rc = ensure(c, sl.b_plan, sizeof(Plan)); if (rc) return rc; }
    Dev g;
    memset(&g, 0, sizeof(g));
    g.n = n;
    g.q_arena = b->q_arena; g.t_arena = b->t_arena; g.blocks = b->blocks;
    g.t_rc = b->t_rc;
    g.q_arena_bytes = b->q_arena_bytes; g.t_arena_bytes = b->t_arena_bytes; g.n_blocks_total = b->n_blocks_total;
    if (!g.t_arena) {            // targets are windows of the ctx reference (kept packed)
        if (!c->b_ref.ptr) return GBA_E_STATE;
        g.ref_pk = (const uint32_t *)c->b_ref.ptr; g.ref_mask = (const uint32_t *)c->b_refm.ptr; g.t_arena_bytes = c->ref_len;
    }
    if (b->q_packed) g.q_pk = (const uint32_t *)b->q_arena;       // the reads stay 2 bits per base in device memory
    g.q_off = b->q_off; g.t_off = b->t_off; g.blk_off = b->blk_off;
    g.q_len = b->q_len; g.t_len = b->t_len; g.n_blk = b->n_blk;
    g.guide_status = c->cur_guide_status;
    g.prof = b->q_profile;
    g.status = (int32_t *)c->b_pair32[0].ptr;
    g.qS = (uint32_t *)c->b_pair32[1].ptr; g.tS = (uint32_t *)c->b_pair32[2].ptr;
    g.Q = (uint32_t *)c->b_pair32[3].ptr; g.T = (uint32_t *)c->b_pair32[4].ptr;
    g.rows = (uint32_t *)c->b_pair32[5].ptr; g.cols = (uint32_t *)c->b_pair32[6].ptr;
    g.span = (uint32_t *)c->b_pair32[7].ptr; g.lo_last = (uint32_t *)c->b_pair32[8].ptr;
    g.cls = (uint32_t *)c->b_pair32[9].ptr; g.order = (uint32_t *)c->b_pair32[10].ptr;
    g.dp_score = (int32_t *)c->b_pair32[11].ptr;
    g.wcap = (uint32_t *)c->b_pair32[12].ptr;
    g.recdw = (uint32_t *)c->b_pair32[13].ptr;
    g.end_i = (uint32_t *)c->b_pair32[14].ptr; g.end_j = (uint32_t *)c->b_pair32[15].ptr;
    g.end_lo = (uint32_t *)c->b_pair32[16].ptr; g.end_ptr = (uint32_t *)c->b_pair32[17].ptr;
    g.row_off = (uint64_t *)c->b_pair64[0].ptr; g.col_off = (uint64_t *)c->b_pair64[1].ptr;
    g.rec_off = (uint64_t *)c->b_pair64[2].ptr; g.ncells = (uint64_t *)c->b_pair64[3].ptr;
    g.word_off = (uint64_t *)c->b_pair64[4].ptr;
    g.blk_end = (uint64_t *)c->b_pair64[5].ptr;
    g.blk_div = c->blk_div_now;
    const bool timing = (c->flags & GBA_FLAG_TIMING) != 0;
    uint32_t parts = 1;
    if (!c->fixed_on && c->part_pairs && n > c->part_pairs + c->part_pairs / 15) {
        parts = (uint32_t)(((uint64_t)n + c->part_pairs / 2) / c->part_pairs);
        if (parts < 2) parts = 2;
    }
    const uint32_t per = (uint32_t)((((uint64_t)n + parts - 1) / parts + 63) & ~63ull);     // whole waves
    parts = (n + per - 1) / per;
    // part h is pairs [bound[h], bound[h + 1]).  Batches from host buffers, and device-resident batches of a context that shares
    // the GPU with another context's batch (a caller that streams batches through two contexts), run in equal parts of about
    // part_pairs: the other batch's kernels fill this one's head (the first part's packing and guide, nothing else to run) and
    // tail (the last part's traceback).  A context that has the GPU to itself shapes a device-resident batch for its own head
    // and tail (shape_parts): up to 1.7 x part_pairs two halves, beyond that the fewest parts P >= 3 in the proportions
    // 3 : 4 : ... : 4 : 3 whose middle parts stay below 1.04 x part_pairs.  The first fill starts sooner and shares the SIMDs
    // with the second part's as soon as that is ready (parts a little below two waves per SIMD leave room for the next one's
    // first waves), and the last traceback has fewer waves.  Measured on pairs of 10 kb, GCUPS with these parts / with equal
    // parts of about part_pairs (tools/run_shape2.sh): 245 760 pairs 796 / 707, 307 200: 813 / 796, 368 640: 821-842 / 788,
    // 491 520: 856-875 / 823-850, 614 400: 881 / 848; parts below 90 000 pairs lose (fewer than 1.4 waves per SIMD).
    // (on the stack for up to 31 parts: the entry point that allocates nothing runs one part and must not touch the heap either)
    struct Bounds {
        uint32_t small[32];
        std::vector<uint32_t> big;
        uint32_t count = 0;
        void push_back(uint32_t x)
        {
            if (big.empty() && count < 32u) small[count] = x;
            else { if (big.empty()) big.assign(small, small + count); big.push_back(x); }
            count++;
        }
        uint32_t operator[](uint32_t i) const { return big.empty() ? small[i] : big[i]; }
        uint32_t back() const { return (*this)[count - 1u]; }
        uint32_t size() const { return count; }
    } bound;
    {
        bound.push_back(0);
        std::vector<uint32_t> shape;
        if (io && !c->part_shape.empty() && getenv("GBA_PART_SHAPE_HOST")) shape = c->part_shape;       // experiments
        // A host-buffer batch of one part has nothing to move its copies under: from about half a part on it runs as two halves
        // (as long as the halves stay on the lane classes), so that the second half's inputs and the first half's results
        // travel under the other half's kernels.  Two contexts in turn on pairs of 10 kb, GCUPS as halves / in one part:
        // 49 152 pairs 411 / 393, 65 536: 447 / 390, 98 304: 538 / 452, 122 880: 539 / 398 (32 768: 213 / 328).
        if (io && shape.empty() && parts == 1 && c->part_pairs && n >= c->part_pairs * 2 / 5) {
            const bool aff = c->sc.mode == GBA_MODE_AFFINE;
            const uint32_t lanes_from = aff ? (b->q_profile ? c->wave_pairs_affine_prof : c->wave_pairs_affine) : c->wave_pairs;
            if (n / 2 >= lanes_from) shape = {1, 1};
        }
        if (!io && !c->fixed_on && c->part_pairs) {
            if (!c->part_shape.empty()) shape = c->part_shape;       // experiments: parts in the proportions of GBA_PART_SHAPE
            else if (c->shape_parts && c->alone_on_device && parts >= 2) {
                if ((uint64_t)n * 10 <= (uint64_t)c->part_pairs * 17) shape = {1, 1};
                else {
                    uint32_t P = 3;
                    while ((uint64_t)n * 2 * 24 > (uint64_t)c->part_pairs * 25 * (2 * P - 1)) P++;      // n / (P - 1/2) <= 25/24 part_pairs
                    shape.assign(P, 4);
                    shape.front() = shape.back() = 3;
                }
            }
        }
        if (!shape.empty()) {
            uint64_t total = 0, cum = 0;
            for (uint32_t w : shape) total += w;
            for (size_t i = 0; i + 1 < shape.size(); i++) {
                cum += shape[i];
                const uint32_t x = (uint32_t)(((uint64_t)n * cum / total + 63) & ~63ull);
                if (x > bound.back() && x < n) bound.push_back(x);
            }
        } else
            for (uint64_t x = per; x < n; x += per) bound.push_back((uint32_t)x);
        bound.push_back(n);
        parts = (uint32_t)bound.size() - 1;
    }
    gba_pair_result *res = out->pairs;
    uint64_t *needed = (uint64_t *)c->b_needed.ptr;
    hipStream_t s = c->slot[0].s;
    TimeAcc acc;
    bool used[NSLOT] = {};
    // slots in use: host-buffer batches overlap with those of a second context (HostPipeline) and get by with two
    const uint32_t ns = io ? c->slots_host : c->slots_device;

    if (io) {
        while (c->ev_in.size() < parts) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->ev_in.push_back(e);
        }
        for (auto &sl : c->slot) sl.d2h_pending = false;
        // all parts now, back to back beside the kernels, or one part ahead of the kernels
        for (uint32_t h = 0; h < (io->all_ahead ? parts : 1u); h++) {
            int rc = stage_part(c, *io, bound[h], bound[h + 1] - bound[h], c->ev_in[h]); if (rc) return rc;
        }
    }
    if (timing) HIPCHK(c, hipEventRecord(c->ev[0], s));
    HIPCHK(c, hipMemsetAsync(needed, 0, 32, s));
    if (parts > 1) {                                   // the other slots' streams start after the batch's inputs are ready
        HIPCHK(c, hipEventRecord(c->slot[0].ev_fork, s));
        for (int x = 1; x < NSLOT; x++) HIPCHK(c, hipStreamWaitEvent(c->slot[x].s, c->slot[0].ev_fork, 0));
    }
    for (uint32_t h = 0; h < parts; h++) {
        Slot &sl = c->slot[h % ns];
        const uint32_t first = bound[h], cnt = bound[h + 1] - first;
        Dev d = sub_dev(g, first, cnt, sl);
        if (io) HIPCHK(c, hipStreamWaitEvent(sl.s, c->ev_in[h], 0));
        int rc = phase_sizes(c, sl, d); if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(sl.s));         // the sizes are here, and the slot's previous part has completed
        if (timing && used[h % ns]) { rc = fold_timing(c, sl, acc); if (rc) return rc; }
        used[h % ns] = true;
        rc = phase_guide(c, sl, d); if (rc) return rc;
        rc = phase_fill(c, sl, d, res + first, (uint32_t)NAUXS, c->fill_chain && h > 0 ? c->slot[(h - 1) % ns].t_fill1 : nullptr); if (rc) return rc;
        if (io && !io->all_ahead && h + 1 < parts) {       // the next part's inputs travel while this part runs
            rc = stage_part(c, *io, bound[h + 1], bound[h + 2] - bound[h + 1], c->ev_in[h + 1]); if (rc) return rc;
        }
        if (cig) hipLaunchKernelGGL(k_cigar<false>, dim3(d.n), dim3(64), 0, sl.s, d, out->cigar_flags, (const uint64_t *)nullptr, (uint32_t *)nullptr, 0ull);
        // block (and operation) offsets continue where the part before stopped
        if (h > 0) HIPCHK(c, hipStreamWaitEvent(sl.s, c->slot[(h - 1) % ns].ev_scan, 0));
        hipLaunchKernelGGL(k_scan1, dim3(1), dim3(SCAN_T), 0, sl.s, d, needed);
        if (cig) hipLaunchKernelGGL(k_scan_ops, dim3(1), dim3(SCAN_T), 0, sl.s, d, needed + 1, cig_off + first);
        HIPCHK(c, hipEventRecord(sl.ev_scan, sl.s));
        if (io) {
            sl.dev = d; sl.first = first;
            HIPCHK(c, hipMemcpyAsync(sl.h_cnt, needed, 16, hipMemcpyDeviceToHost, sl.s));
            HIPCHK(c, hipEventRecord(sl.ev_cnt, sl.s));
            HIPCHK(c, hipEventRecord(sl.ev_done, sl.s));
            if (h > 0) { rc = finish_part(c, *io, c->slot[(h - 1) % ns]); if (rc) return rc; }
            continue;
        }
        hipLaunchKernelGGL(k_emit, dim3(d.n), dim3(256), 0, sl.s, d, res + first, out->blocks, out->blocks_capacity, want_blocks ? 1 : 0);
        if (cig)
            hipLaunchKernelGGL(k_cigar<true>, dim3(d.n), dim3(64), 0, sl.s, d, out->cigar_flags, (const uint64_t *)(cig_off + first),
                               out->cigar_ops, out->cigar_capacity);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(sl.ev_done, sl.s));
    }
    if (io) { int rc = finish_part(c, *io, c->slot[(parts - 1) % ns]); if (rc) return rc; }
    for (int x = 1; x < NSLOT; x++)
        if (used[x]) HIPCHK(c, hipStreamWaitEvent(s, c->slot[x].ev_done, 0));
    HIPCHK(c, hipMemcpyAsync(c->h_needed, needed, 24, hipMemcpyDeviceToHost, s));
    if (timing) HIPCHK(c, hipEventRecord(c->ev[1], s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (io) HIPCHK(c, hipStreamSynchronize(c->s_out));
    out->blocks_needed = c->h_needed[0];
    out->cigar_needed = cig ? c->h_needed[1] : 0;
    if (c->h_needed[2]) return RETRY_FULL_SCRATCH;      // some pair's blocks found no room: the caller repeats the batch
    if (timing) {
        // fill: the bulk fill launches of the parts, one after the other (what else runs on the GPU
        // during a launch makes it longer); traceback: from the last fill's end to the last traceback's
        for (int x = 0; x < NSLOT; x++)
            if (used[x]) { int rc = fold_timing(c, c->slot[x], acc); if (rc) return rc; }
        gba_timing &t = c->timing;
        t.fill_launches = acc.launches; t.cells = acc.cells; t.rows = acc.rows; t.record_bytes = acc.rec; t.wide_pairs = acc.wide;
        t.fill_ms = acc.fill_sum; t.traceback_ms = acc.tb1 > acc.f1 ? acc.tb1 - acc.f1 : 0;
        // the union of the launches' intervals: consecutive parts' launches overlap (the next one's waves take the SIMDs the
        // retiring one's leave), and the sum counts such a stretch once per launch
        std::sort(acc.fills.begin(), acc.fills.end());
        float busy = 0, hi = -1e30f;
        for (const auto &iv : acc.fills) {
            const float from = iv.first > hi ? iv.first : hi;
            if (iv.second > from) { busy += iv.second - from; hi = iv.second; }
        }
        t.fill_busy_ms = busy;
        HIPCHK(c, hipEventElapsedTime(&t.total_ms, c->ev[0], c->ev[1]));
    }
    return (want_blocks && out->blocks_needed > out->blocks_capacity) || out->cigar_needed > out->cigar_capacity ? GBA_E_NOSPACE : GBA_OK;
}

// Guide production for a batch whose inputs and outputs are device pointers, in parts of at most
// SDP_PART pairs on the ctx's main stream (the chains of a part wait in device scratch, max_fragments
// blocks per pair, until the part's block offsets are known).
constexpr uint32_t SDP_PART = 32768;

// io: the sequences are still in host memory (the per-pair arrays are on the device): once the longest
// read and window are known, the arena stretch of every part is queued on the copy stream, and a part's
// kernels wait only for their own stretch, so the sequences travel under the kernels of the parts before.
uint32_t sdp_part_size(uint32_t n) { return n < SDP_PART ? n : SDP_PART; }

int run_sdp(gba_ctx *c, const gba_batch *b, const gba_sdp_params *sp, gba_guides *out, HostIO *io = nullptr)
{
    if (sp->tuple_size < 4 || sp->tuple_size > 15 || sp->match < 1 || sp->match > 127 || sp->ins < 0 || sp->ins > 127 ||
        sp->del < 0 || sp->del > 127 || sp->max_fragments < 1 || sp->max_fragments > 4096) return GBA_E_UNSUPPORTED;      // sdp_params_ok
    const uint32_t n = b->n_pairs;
    out->blocks_needed = 0;
    if (n == 0) return GBA_OK;
    HIPCHK(c, hipSetDevice(c->device));
    Slot &sl = c->slot[0];
    hipStream_t s = sl.s;
    const uint32_t per = sdp_part_size(n);
    { int rc = ensure(c, c->b_pair32[0], (size_t)per * 4); if (rc) return rc; }
    { int rc = ensure(c, c->b_needed, 32); if (rc) return rc; }
    uint64_t *needed = (uint64_t *)c->b_needed.ptr;
    HIPCHK(c, hipMemsetAsync(needed, 0, 16, s));
    SdpConst k;
    k.k = (uint32_t)sp->tuple_size; k.kmask = (uint32_t)((1ull << (2 * sp->tuple_size)) - 1ull);
    k.ins = sp->ins; k.del = sp->del; k.match = sp->match;
    k.max_fragments = sp->max_fragments;
    k.cap = 64; while (k.cap < k.max_fragments) k.cap <<= 1;
    k.hit_limit = 64u * k.max_fragments;
    k.stop = 0;
    if (const char *e = getenv("GBA_SDP_STOP")) k.stop = (uint32_t)atoi(e);
    if (!c->sdp_attr_set) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k_sdp_match, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));   // the kernel has a few static words too
        HIPCHK(c, hipFuncSetAttribute((const void *)k_sdp_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        c->sdp_attr_set = true;
    }
    SdpIn in;
    in.q_arena = b->q_arena; in.t_arena = b->t_arena; in.t_rc = b->t_rc;
    in.q_arena_bytes = b->q_arena_bytes; in.t_arena_bytes = b->t_arena_bytes;
    in.q_pk = b->q_packed ? (const uint32_t *)b->q_arena : nullptr;
    in.ref_pk = in.ref_mask = nullptr;
    if (!in.t_arena) { in.ref_pk = (const uint32_t *)c->b_ref.ptr; in.ref_mask = (const uint32_t *)c->b_refm.ptr; in.t_arena_bytes = c->ref_len; }
    in.q_off = b->q_off; in.t_off = b->t_off; in.q_len = b->q_len; in.t_len = b->t_len;
    // the one thing the host must know of the batch: its longest read and window (they size the kernel's LDS)
    hipLaunchKernelGGL(k_sdp_max, dim3((n + 255) / 256), dim3(256), 0, s, in, n, k.k, (uint32_t *)(needed + 1));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_needed + 1, needed + 1, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const uint32_t maxQ = ((const uint32_t *)(c->h_needed + 1))[0], maxT = ((const uint32_t *)(c->h_needed + 1))[1];
    // the target is indexed a tile at a time: tile positions x slots per position, sized to the longest window
    k.tile = 256; while (k.tile < maxT && k.tile < c->sdp_tile_max) k.tile <<= 1;
    k.slots_log2 = 1; while ((1u << k.slots_log2) < c->sdp_slots_per_pos * k.tile) k.slots_log2++;
    while (k.slots_log2 > 2u * k.k) k.slots_log2--;                 // short tuples: no more slots than k-mers exist
    k.qwords = maxQ / 16 + 3; k.twords = maxT / 16 + 3;
    size_t lds;
    k.queue = c->sdp_queue;
    for (;;) {
        lds = ((size_t)(1u << k.slots_log2) + k.tile / 2 + k.queue + 2 * (size_t)(k.qwords + k.twords)) * 4;
        if (lds <= 160 * 1024 - 256 || k.tile <= 256) break;
        k.tile >>= 1; k.slots_log2--;                                   // long sequences leave less room for the table
    }
    if (lds > 160 * 1024 - 256) return GBA_E_UNSUPPORTED;
    { int rc = ensure(c, sl.b_blk, (size_t)per * k.max_fragments * sizeof(gba_block) + 16); if (rc) return rc; }
    { int rc = ensure(c, sl.b_guide, (size_t)per * k.cap * sizeof(uint2) + 16); if (rc) return rc; }
    uint32_t *nfrag = (uint32_t *)c->b_pair32[0].ptr;
    if (io) {
        const uint32_t nparts = (n + per - 1) / per;
        while (c->ev_sdp.size() < nparts) {
            hipEvent_t e;
            HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->ev_sdp.push_back(e);
        }
        for (uint32_t h = 0; h < nparts; h++) {
            const uint32_t f = h * per;
            int rc = stage_part(c, *io, f, n - f < per ? n - f : per, c->ev_sdp[h], true); if (rc) return rc;
        }
    }
    // the parts follow each other on the stream without the host in between; they share the scratch
    for (uint32_t first = 0; first < n; first += per) {
        const uint32_t cnt = n - first < per ? n - first : per;
        SdpIn pin = in;
        pin.q_off += first; pin.t_off += first; pin.q_len += first; pin.t_len += first;
        if (pin.t_rc) pin.t_rc += first;
        if (io) HIPCHK(c, hipStreamWaitEvent(s, c->ev_sdp[first / per], 0));
        hipLaunchKernelGGL(k_sdp_match, dim3(cnt), dim3(SDP_THREADS), lds, s, pin, k, out->status + first, (uint2 *)sl.b_guide.ptr, nfrag);
        {
            const uint32_t small = k.cap < SDP_SMALL ? k.cap : SDP_SMALL;
            hipLaunchKernelGGL(k_sdp_chain, dim3(cnt), dim3(64), (size_t)small * 16 + 16, s, k, (const uint2 *)sl.b_guide.ptr, (const uint32_t *)nfrag,
                               out->chain_score + first, out->n_blk + first, (gba_block *)sl.b_blk.ptr, 0u, small);
            if (k.cap > small)
                hipLaunchKernelGGL(k_sdp_chain, dim3(cnt), dim3(64), (size_t)k.cap * 16 + 16, s, k, (const uint2 *)sl.b_guide.ptr, (const uint32_t *)nfrag,
                                   out->chain_score + first, out->n_blk + first, (gba_block *)sl.b_blk.ptr, small, k.cap);
        }
        hipLaunchKernelGGL(k_scan_cnt, dim3(1), dim3(SCAN_T), 0, s, (const uint32_t *)(out->n_blk + first), out->blk_off + first, cnt, needed);
        hipLaunchKernelGGL(k_sdp_emit, dim3(cnt), dim3(256), 0, s, (const gba_block *)sl.b_blk.ptr, k.max_fragments,
                           (const uint32_t *)(out->n_blk + first), (const uint64_t *)(out->blk_off + first), out->blocks, out->blocks_capacity, out->status + first);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(c->h_needed, needed, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    out->blocks_needed = c->h_needed[0];
    return out->blocks_needed > out->blocks_capacity ? GBA_E_NOSPACE : GBA_OK;
}

bool sdp_params_ok(const gba_sdp_params *sp)
{
    return !(sp->tuple_size < 4 || sp->tuple_size > 15 || sp->match < 1 || sp->match > 127 || sp->ins < 0 || sp->ins > 127 ||
             sp->del < 0 || sp->del > 127 || sp->max_fragments < 1 || sp->max_fragments > 4096);
}

// Guides for a device batch that brings none (gba_set_guide_producer): the chains of run_sdp go to
// buffers of the ctx and *db is completed with them.  `hint`: blocks to make room for at first (0: the
// room of the batch before); when that is too little the producer says how many there are and runs again.
int make_guides(gba_ctx *c, gba_batch *db, uint64_t hint, HostIO *io = nullptr)
{
    const uint32_t n = db->n_pairs;
    int rc;
    if ((rc = ensure(c, c->s_gout, (size_t)n * 8 + 16))) return rc;
    if ((rc = ensure(c, c->s_nblk, (size_t)n * 4 + 16))) return rc;
    if ((rc = ensure(c, c->s_boff, (size_t)n * 8 + 16))) return rc;
    gba_guides dg;
    dg.status = (int32_t *)c->s_gout.ptr; dg.chain_score = dg.status + n;
    dg.n_blk = (uint32_t *)c->s_nblk.ptr; dg.blk_off = (uint64_t *)c->s_boff.ptr;
    uint64_t cap = c->s_blk.cap / sizeof(gba_block);
    if (cap < hint) cap = hint;
    if (cap < (uint64_t)n * 16) cap = (uint64_t)n * 16;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = ensure(c, c->s_blk, (size_t)cap * sizeof(gba_block) + 16))) return rc;
        dg.blocks = (gba_block *)c->s_blk.ptr; dg.blocks_capacity = cap; dg.blocks_needed = 0;
        rc = run_sdp(c, db, &c->guide_sp, &dg, io);
        if (rc != GBA_E_NOSPACE) break;
        cap = dg.blocks_needed + dg.blocks_needed / 16;
    }
    if (rc != GBA_OK) return rc;
    db->blocks = dg.blocks; db->blk_off = dg.blk_off; db->n_blk = dg.n_blk; db->n_blocks_total = dg.blocks_needed;
    c->cur_guide_status = dg.status;
    return GBA_OK;
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
    case GBA_E_TOOBIG: return "batch needs more scratch than the fixed no-allocation buffer holds";
    case GBA_PAIR_EMPTY_GUIDE: return "pair: empty guide";
    case GBA_PAIR_BAD_GUIDE: return "pair: malformed guide";
    case GBA_PAIR_UNREACHABLE: return "pair: end cell unreachable inside the band";
    case GBA_PAIR_NOSPACE: return "pair: blocks did not fit the arena";
    case GBA_PAIR_END_OUT_OF_BAND: return "pair: end cell outside the k-band";
    case GBA_PAIR_TOO_WIDE: return "pair: band row too wide";
    case GBA_PAIR_TOO_LONG: return "pair: too many rows or columns";
    case GBA_PAIR_TOO_MANY: return "pair: too many k-mer matches to chain";
    case GBA_PAIR_NO_ALIGNMENT: return "pair: nothing aligns (local end rule)";
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
        if (hipHostMalloc((void **)&c->h_needed, 32) != hipSuccess) { rc = GBA_E_NOMEM; break; }
        for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rc = GBA_E_HIP; break; }
        if (rc) break;
        {   // this device's copy of the no-allocation scratch (the ctx stays usable without it)
            void *sp = nullptr;
            if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) == hipSuccess) c->fixed_base = (unsigned char *)sp;
            else (void)hipGetLastError();
        }
        if (const char *e = getenv("GBA_PART_PAIRS")) c->part_pairs = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBA_WAVE_PAIRS")) c->wave_pairs = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBA_TB_WAVE_MAX")) c->tb_wave_max = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBA_TB_FEW_MAX")) c->tb_few_max = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBA_QUAD_AFFINE_MIN")) c->quad_affine_min = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBA_WAVE_PAIRS_AFFINE")) c->wave_pairs_affine = c->wave_pairs_affine_prof = (uint32_t)strtoul(e, nullptr, 10);
        if (const char *e = getenv("GBA_AFFINE_TOP_SLOTS")) { const uint32_t v = (uint32_t)strtoul(e, nullptr, 10); if (v == 0 || (v > 64 && v <= 128)) c->affine_top_slots = v; }
        if (const char *e = getenv("GBA_SDP_TILE")) {
            const uint32_t v = (uint32_t)strtoul(e, nullptr, 10);
            if (v >= 256 && v <= 16384 && (v & (v - 1)) == 0) c->sdp_tile_max = v;
        }
        if (const char *e = getenv("GBA_SLOTS_DEVICE")) { const int v = atoi(e); if (v >= 2 && v <= NSLOT) c->slots_device = (uint32_t)v; }
        if (const char *e = getenv("GBA_FILL_CHAIN")) c->fill_chain = atoi(e) != 0;
        if (const char *e = getenv("GBA_PART_SHAPE")) {
            for (const char *q = e; *q;) {
                char *end = nullptr;
                const unsigned long v = strtoul(q, &end, 10);
                if (end == q) break;
                if (v > 0 && v < 1000) c->part_shape.push_back((uint32_t)v);
                q = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') break;
            }
        }
        if (const char *e = getenv("GBA_EDGE_PARTS")) c->shape_parts = atoi(e) != 0;
        if (const char *e = getenv("GBA_SLOTS_HOST")) { const int v = atoi(e); if (v >= 2 && v <= NSLOT) c->slots_host = (uint32_t)v; }
        if (const char *e = getenv("GBA_BLK_DIV")) { const int v = atoi(e); if (v >= 1 && v <= 1024) c->blk_div = (uint32_t)v; }
        if (const char *e = getenv("GBA_BLK_POOL_DIV")) { const int v = atoi(e); if (v >= 1 && v <= (1 << 20)) c->blk_pool_div = (uint32_t)v; }
        if (const char *e = getenv("GBA_SDP_QUEUE")) { const int v = atoi(e); if (v >= 64 && v <= 16384) c->sdp_queue = (uint32_t)v; }
        if (const char *e = getenv("GBA_SDP_SLOTS")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16) c->sdp_slots_per_pos = (uint32_t)v; }
        for (int h = 0; h < NSLOT && !rc; h++) {
            Slot &sl = c->slot[h];
            if (h == 0) sl.s = c->stream;
            else if (hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking) != hipSuccess) { rc = GBA_E_HIP; break; }
            if (hipHostMalloc((void **)&sl.h_plan, sizeof(Plan)) != hipSuccess) { rc = GBA_E_NOMEM; break; }
            if (hipHostMalloc((void **)&sl.h_cnt, 16) != hipSuccess) { rc = GBA_E_NOMEM; break; }
            hipEvent_t *plain[] = {&sl.ev_fork, &sl.ev_done, &sl.ev_scan, &sl.ev_join[0], &sl.ev_join[1], &sl.ev_cnt, &sl.ev_emit, &sl.ev_d2h};
            for (hipEvent_t *e : plain) if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = GBA_E_HIP;
            hipEvent_t *timed[] = {&sl.t_fill0, &sl.t_fill1, &sl.t_tb1};
            for (hipEvent_t *e : timed) if (hipEventCreate(e) != hipSuccess) rc = GBA_E_HIP;
            for (int a = 0; a < NAUXS; a++)
                if (hipStreamCreateWithFlags(&sl.aux[a], hipStreamNonBlocking) != hipSuccess) rc = GBA_E_HIP;
        }
        // The copy streams come last: the runtime deals streams to its hardware queues (GPU_MAX_HW_QUEUES, 4 by
        // default) in creation order, and which kernel streams share a queue decides how well the parts of a batch
        // overlap.  With 8 queues every stream of a ctx has its own.
        if (!rc && (hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking) != hipSuccess ||
                    hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking) != hipSuccess)) rc = GBA_E_HIP;
    } while (0);
    if (rc) { gba_destroy(c); return rc; }
    *out = c;
    return GBA_OK;
}

void gba_destroy(gba_ctx *c)
{
    if (!c) return;
    if (c->worker.joinable()) {
        {
            std::unique_lock<std::mutex> lk(c->mu);
            if (c->in_flight) c->cv.wait(lk, [c] { return c->job_done; });
            c->quit = true;
            c->cv.notify_all();
        }
        c->worker.join();
    }
    (void)hipSetDevice(c->device);
    if (c->s_in) { (void)hipStreamSynchronize(c->s_in); (void)hipStreamDestroy(c->s_in); }
    if (c->s_out) { (void)hipStreamSynchronize(c->s_out); (void)hipStreamDestroy(c->s_out); }
    for (hipEvent_t e : c->ev_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_sdp) (void)hipEventDestroy(e);
    for (void *p : c->host_allocs) (void)hipHostFree(p);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->b_pair32) release(b);
    for (auto &b : c->b_pair64) release(b);
    Buf *bs[] = {&c->b_ref, &c->b_refm, &c->s_trc, &c->b_needed,
                 &c->s_q, &c->s_t, &c->s_blk, &c->s_qoff, &c->s_toff, &c->s_boff, &c->s_qlen, &c->s_tlen,
                 &c->s_nblk, &c->s_res, &c->s_cigoff, &c->s_gout};
    for (Buf *b : bs) release(*b);
    for (auto &b : c->kb) release(b);
    release(c->s_prof);
    if (c->h_needed) (void)hipHostFree(c->h_needed);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (int h = 0; h < NSLOT; h++) {
        Slot &sl = c->slot[h];
        for (int a = 0; a < NAUXS; a++) if (sl.aux[a]) { (void)hipStreamSynchronize(sl.aux[a]); (void)hipStreamDestroy(sl.aux[a]); }
        if (h > 0 && sl.s) { (void)hipStreamSynchronize(sl.s); (void)hipStreamDestroy(sl.s); }
        Buf *sb[] = {&sl.b_plan, &sl.b_guide, &sl.b_tc, &sl.b_rec, &sl.b_blk, &sl.b_out, &sl.b_cig};
        for (Buf *b : sb) release(*b);
        if (sl.h_plan) (void)hipHostFree(sl.h_plan);
        if (sl.h_cnt) (void)hipHostFree(sl.h_cnt);
        hipEvent_t evs[] = {sl.ev_fork, sl.ev_done, sl.ev_scan, sl.ev_join[0], sl.ev_join[1], sl.t_fill0, sl.t_fill1, sl.t_tb1,
                            sl.ev_cnt, sl.ev_emit, sl.ev_d2h};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gba_last_hip_error(gba_ctx *c) { return c ? c->last_hip : 0; }

int gba_set_scoring(gba_ctx *c, const gba_scoring *s)
{
    if (!c || !s) return GBA_E_INVALID;
    if (s->mode != GBA_MODE_LINEAR && s->mode != GBA_MODE_KBAND && s->mode != GBA_MODE_AFFINE) return GBA_E_INVALID;
    if (s->tie_break != GBA_TIE_DIAG_LEFT_UP && s->tie_break != GBA_TIE_DIAG_UP_LEFT) return GBA_E_INVALID;
    if (s->n_extra < 0 || s->n_extra > GBA_MAX_EXTRA) return GBA_E_INVALID;
    if (s->band < 0 || s->max_band < 0 || s->band > 30000 || s->max_band > 30000) return GBA_E_INVALID;
    if (s->mode == GBA_MODE_KBAND && (s->kband_k < 0 || s->kband_k > 30000)) return GBA_E_INVALID;
    if (s->align_type != GBA_ALIGN_GLOBAL && s->align_type != GBA_ALIGN_LOCAL) return GBA_E_INVALID;
    if (s->align_type == GBA_ALIGN_LOCAL && s->mode != GBA_MODE_LINEAR) return GBA_E_UNSUPPORTED;      // the reference's one Local call is GuidedAlign
    // The fill kernels keep substitution costs in a +128-biased byte table and cell words in int32
    // with 5 low tag bits (affine: 6): |cost| <= 127 keeps the table in bytes.  Whether a pair's
    // scores fit the word is a per-pair matter (rows x row-move cost + columns x deletion cost,
    // k_meta): at the default costs every pair up to the row limit fits; at |cost| = 127 a pair
    // with many columns per row may not, and is handed back TOO_LONG.
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            if (s->sub[a][b] < -128 || s->sub[a][b] > 127) return GBA_E_UNSUPPORTED;
    if (s->ins < -127 || s->ins > 127 || s->del < -127 || s->del > 127) return GBA_E_UNSUPPORTED;
    if (s->mode == GBA_MODE_AFFINE) {
        // affine words keep one more tag bit: half the cost range (and k_meta half the row count)
        const int32_t oe = s->affine_open + s->affine_extend;
        for (int a = 0; a < 5; a++)
            for (int b = 0; b < 5; b++)
                if (s->sub[a][b] < -63 || s->sub[a][b] > 63) return GBA_E_UNSUPPORTED;
        if (s->affine_extend < 0 || s->affine_extend > 63 || oe < 0 || oe > 63) return GBA_E_UNSUPPORTED;
    }
    // nothing of the ctx changes before this point: a refused scoring leaves the previous one in force
    c->sc = *s;
    {
        auto mag = [](int32_t v) { return (uint32_t)(v < 0 ? -v : v); };
        uint32_t a = mag(s->ins);
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) a = a > mag(s->sub[x][y]) ? a : mag(s->sub[x][y]);
        uint32_t b = mag(s->del);
        if (s->mode == GBA_MODE_AFFINE) {
            const uint32_t oe = (uint32_t)(s->affine_open + s->affine_extend);       // checked: 0..63
            a = a > oe ? a : oe; b = b > oe ? b : oe;
        }
        c->score_a = a; c->score_b = b;
    }
    const uint32_t codeL = s->tie_break == GBA_TIE_DIAG_LEFT_UP ? 1u : 2u, codeU = 3u - codeL;
    FillConst &f = c->fc;
    for (int a = 0; a < 5; a++) {
        uint32_t lo = 0, hi = 0;
        // byte 0 is zero: the lookup's unused selector bytes read it; bytes 1..5 = A,C,G,T,N
        for (int b = 0; b < 3; b++) lo |= (uint32_t)(uint8_t)(s->sub[a][b] + 128) << (8 * (b + 1));
        hi = (uint32_t)(uint8_t)(s->sub[a][3] + 128) | ((uint32_t)(uint8_t)(s->sub[a][4] + 128) * 0x01010100u);
        f.mrow[2 * a] = lo; f.mrow[2 * a + 1] = hi;
    }
    const int SH = s->mode == GBA_MODE_AFFINE ? 6 : 5;
    const int32_t bias = 128 << SH;
    if (s->mode == GBA_MODE_AFFINE) {
        const int32_t oe = s->affine_open + s->affine_extend;
        f.cIO = oe * 64 + bias; f.cIE = s->affine_extend * 64 + bias + 4;
        f.cDO = oe * 64; f.cDE = s->affine_extend * 64 + 2;
        f.diag0 = -((s->sub[4][4] + 128) * 64);
    } else {
        f.cIO = f.cIE = f.cDO = f.cDE = f.diag0 = 0;
    }
    f.cU = s->ins * (1 << SH) + (int32_t)(codeU << 3) + bias;
    f.cL = s->del * (1 << SH) + (int32_t)(codeL << 3);
    f.del = s->del; f.ins = s->ins;
    TbConst &t = c->tbc;
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) t.sub[a * 5 + b] = s->sub[a][b];
    t.ins = s->stats_ins; t.del = s->stats_del; t.aopen = s->stats_affine_open; t.aext = s->stats_affine_extend;
    t.affine = s->stats_affine; t.codeU = codeU; t.codeL = codeL; t.mode = s->mode;
    t.n_extra = s->n_extra;
    for (int x = 0; x < GBA_MAX_EXTRA; x++) {
        const gba_rescorer &r = s->extra[x];
        const bool on = x < s->n_extra;
        for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) t.xsub[x][a * 5 + b] = on ? r.sub[a][b] : 0;
        t.xins[x] = on ? r.ins : 0; t.xdel[x] = on ? r.del : 0;
        t.xaopen[x] = on ? r.affine_open : 0; t.xaext[x] = on ? r.affine_extend : 0; t.xaffine[x] = on ? r.affine : 0;
    }
    c->have_scoring = true;
    return GBA_OK;
}

int gba_submit_device(gba_ctx *c, const gba_batch *b, gba_results *out)
{
    if (!c || !b || !out || (b->q_packed && ((uintptr_t)b->q_arena & 3u))) return GBA_E_INVALID;
    if (!c->have_scoring || c->in_flight) return GBA_E_STATE;
    if (b->n_pairs && (!out->pairs || (!out->blocks && out->blocks_capacity))) return GBA_E_INVALID;
    if (out->cigar_ops && !out->cigar_off) return GBA_E_INVALID;
    if (!batch_complete(c, b)) return GBA_E_INVALID;
    if (!b->t_arena && !c->b_ref.ptr) return GBA_E_STATE;
    if (b->n_pairs && !b->blocks) {                  // guides made here first (gba_set_guide_producer)
        gba_batch db = *b;
        int rc = make_guides(c, &db, 0);
        if (rc == GBA_OK) rc = run_device(c, &db, out);
        c->cur_guide_status = nullptr;
        return rc;
    }
    return run_device(c, b, out);
}

int gba_submit_device_noalloc(gba_ctx *c, const gba_batch *b, gba_results *out)
{
    if (!c || !b || !out || (b->q_packed && ((uintptr_t)b->q_arena & 3u))) return GBA_E_INVALID;
    if (!c->have_scoring || c->in_flight) return GBA_E_STATE;
    if (b->n_pairs && (!out->pairs || (!out->blocks && out->blocks_capacity))) return GBA_E_INVALID;
    if (out->cigar_ops && !out->cigar_off) return GBA_E_INVALID;
    if (!batch_complete(c, b)) return GBA_E_INVALID;
    if (!b->t_arena && !c->b_ref.ptr) return GBA_E_STATE;
    if (b->n_pairs && !b->blocks) return GBA_E_INVALID;     // guide production allocates: not for this entry point
    if (!c->fixed_base) return GBA_E_HIP;             // hipGetSymbolAddress failed in gba_create
    // The ctx's own (allocated, grow-only) buffers are set aside for the duration of the call and the
    // same pipeline runs on buffers carved from the fixed scratch, as one part on slot 0.
    Buf *all[N_PAIR32 + N_PAIR64 + 1 + 5 * NSLOT];
    Buf saved[N_PAIR32 + N_PAIR64 + 1 + 5 * NSLOT];
    int k = 0;
    for (auto &x : c->b_pair32) all[k++] = &x;
    for (auto &x : c->b_pair64) all[k++] = &x;
    all[k++] = &c->b_needed;
    for (auto &sl : c->slot) { all[k++] = &sl.b_plan; all[k++] = &sl.b_guide; all[k++] = &sl.b_tc; all[k++] = &sl.b_rec; all[k++] = &sl.b_blk; }
    for (int i = 0; i < k; i++) { saved[i] = *all[i]; *all[i] = Buf(); }
    c->fixed_on = true; c->fixed_used = 0;
    const int rc = run_device(c, b, out);
    c->fixed_on = false;
    for (int i = 0; i < k; i++) *all[i] = saved[i];
    return rc;
}

// Body of a host-buffer batch, on the ctx's worker thread: device copies of the caller's arrays
// (grow-only), then the pipeline of run_device with the copies in both directions inside it.
static int run_host_batch(gba_ctx *c, const gba_batch *b, gba_results *out)
{
    const uint32_t n = b->n_pairs;
    const bool cig = out->cigar_ops != nullptr;
    out->blocks_needed = 0; out->cigar_needed = 0;
    if (n == 0) { if (cig) out->cigar_off[0] = 0; return GBA_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    struct { Buf *buf; size_t bytes; } need[] = {
        {&c->s_q, b->q_packed ? (size_t)b->q_arena_bytes / 4 + 32 : (size_t)b->q_arena_bytes}, {&c->s_t, b->t_arena ? (size_t)b->t_arena_bytes : 0},
        {&c->s_blk, (size_t)b->n_blocks_total * sizeof(gba_block)},
        {&c->s_qoff, (size_t)n * 8}, {&c->s_toff, (size_t)n * 8}, {&c->s_boff, (size_t)n * 8},
        {&c->s_qlen, (size_t)n * 4}, {&c->s_tlen, (size_t)n * 4}, {&c->s_nblk, (size_t)n * 4},
        {&c->s_trc, b->t_rc ? (size_t)n : 0}, {&c->s_res, (size_t)n * sizeof(gba_pair_result)},
        {&c->s_cigoff, cig ? ((size_t)n + 1) * 8 : 0},
        {&c->s_prof, b->q_profile ? (size_t)b->q_arena_bytes * 22 : 0}};
    for (auto &x : need) if ((rc = ensure(c, *x.buf, x.bytes + 16))) return rc;
    gba_batch db = *b;
    db.q_arena = (const uint8_t *)c->s_q.ptr; db.t_arena = b->t_arena ? (const uint8_t *)c->s_t.ptr : nullptr;
    db.t_rc = b->t_rc ? (const uint8_t *)c->s_trc.ptr : nullptr;
    db.blocks = (const gba_block *)c->s_blk.ptr;
    if (b->q_profile) {
        // the profile goes first on the stream of the input copies: every part's "inputs have arrived" event follows it
        HIPCHK(c, hipMemcpyAsync(c->s_prof.ptr, b->q_profile, (size_t)b->q_arena_bytes * 22, hipMemcpyHostToDevice, c->s_in));
        db.q_profile = (const int16_t *)c->s_prof.ptr;
    }
    db.q_off = (const uint64_t *)c->s_qoff.ptr; db.t_off = (const uint64_t *)c->s_toff.ptr;
    db.blk_off = (const uint64_t *)c->s_boff.ptr;
    db.q_len = (const uint32_t *)c->s_qlen.ptr; db.t_len = (const uint32_t *)c->s_tlen.ptr;
    db.n_blk = (const uint32_t *)c->s_nblk.ptr;
    gba_results dr;
    memset(&dr, 0, sizeof(dr));
    dr.pairs = (gba_pair_result *)c->s_res.ptr;
    dr.blocks_capacity = out->blocks_capacity;
    dr.cigar_capacity = out->cigar_capacity; dr.cigar_flags = out->cigar_flags;
    HostIO io;
    io.hb = b; io.hout = out; io.d_res = dr.pairs;
    gba_batch hb_noguides;
    if (!b->blocks) {
        // No guides in the batch: they are made on the device from the sequences (nothing of a guide ever
        // crosses the link).  The per-pair arrays go first; the arenas follow inside run_sdp, under the
        // producer's kernels; the pipeline below then finds everything in place.
        hb_noguides = *b;
        hb_noguides.blk_off = nullptr; hb_noguides.n_blk = nullptr;         // whatever the caller left there is not a guide
        io.hb = &hb_noguides;
        hipStream_t s0 = c->slot[0].s;
        struct { Buf *buf; const void *src; size_t bytes; } in[] = {
            {&c->s_qoff, b->q_off, (size_t)n * 8}, {&c->s_toff, b->t_off, (size_t)n * 8},
            {&c->s_qlen, b->q_len, (size_t)n * 4}, {&c->s_tlen, b->t_len, (size_t)n * 4}, {&c->s_trc, b->t_rc, b->t_rc ? (size_t)n : 0}};
        for (auto &x : in) if (x.bytes) HIPCHK(c, hipMemcpyAsync(x.buf->ptr, x.src, x.bytes, hipMemcpyHostToDevice, s0));
        uint64_t qsum = 0;
        for (uint32_t p = 0; p < n; p++) qsum += b->q_len[p];
        if ((rc = make_guides(c, &db, qsum / 16 + n, &io))) { c->cur_guide_status = nullptr; return rc; }
    }
    if (cig) io.d_cigoff = (uint64_t *)c->s_cigoff.ptr;
    // Page-locked arenas (gba_host_alloc, or pinned / registered by the caller) are copied by the copy
    // engines without the calling thread; when the runtime also has a hardware queue to spare for the
    // copy stream (GPU_MAX_HW_QUEUES of 8 or more; with the default 4 it shares one with a kernel stream,
    // whose kernels would then wait for every copy queued before them), all parts are queued at once.
    auto pinned = [](const void *p) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof(a));
        if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const char *hq = getenv("GPU_MAX_HW_QUEUES");
    io.all_ahead = hq && atoi(hq) >= 8 && pinned(b->q_arena) && (!b->blocks || pinned(b->blocks)) && (!b->t_arena || pinned(b->t_arena));
    rc = run_device(c, &db, &dr, &io);
    c->cur_guide_status = nullptr;
    out->blocks_needed = dr.blocks_needed; out->cigar_needed = dr.cigar_needed;
    return rc;
}

static void worker_main(gba_ctx *c)
{
    std::unique_lock<std::mutex> lk(c->mu);
    for (;;) {
        c->cv.wait(lk, [c] { return c->job_ready || c->quit; });
        if (c->quit) return;
        c->job_ready = false;
        lk.unlock();
        const int rc = run_host_batch(c, &c->job_batch, c->job_out);
        if (rc != GBA_OK && rc != GBA_E_NOSPACE) {
            // leave nothing of the failed batch running behind the caller's back
            (void)hipSetDevice(c->device);
            for (auto &sl : c->slot) { if (sl.s) (void)hipStreamSynchronize(sl.s); for (auto a : sl.aux) if (a) (void)hipStreamSynchronize(a); }
            if (c->s_in) (void)hipStreamSynchronize(c->s_in);
            if (c->s_out) (void)hipStreamSynchronize(c->s_out);
        }
        lk.lock();
        c->job_rc = rc; c->job_done = true;
        c->cv.notify_all();
    }
}

int gba_submit(gba_ctx *c, const gba_batch *b, gba_results *out)
{
    if (!c || !b || !out) return GBA_E_INVALID;
    if (!c->have_scoring || c->in_flight) return GBA_E_STATE;
    if (!batch_complete(c, b)) return GBA_E_INVALID;
    if (b->n_pairs && (!out->pairs || (!out->blocks && out->blocks_capacity))) return GBA_E_INVALID;
    if (out->cigar_ops && !out->cigar_off) return GBA_E_INVALID;
    if (b->n_pairs && !b->t_arena && !c->b_ref.ptr) return GBA_E_STATE;
    std::unique_lock<std::mutex> lk(c->mu);
    if (!c->worker.joinable()) {
        try { c->worker = std::thread(worker_main, c); } catch (...) { return GBA_E_NOMEM; }
    }
    c->job_batch = *b; c->job_out = out;
    c->job_ready = true; c->job_done = false;
    c->in_flight = true;
    c->cv.notify_all();
    return GBA_OK;
}

int gba_wait(gba_ctx *c)
{
    if (!c) return GBA_E_INVALID;
    if (!c->in_flight) return GBA_E_STATE;
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv.wait(lk, [c] { return c->job_done; });
    c->in_flight = false;
    const int rc = c->job_rc;
    if (rc == GBA_E_NOSPACE && c->job_out->blocks) {
        // pairs whose blocks lie beyond the caller's arena
        gba_results *out = c->job_out;
        for (uint32_t p = 0; p < c->job_batch.n_pairs; p++) {
            gba_pair_result &R = out->pairs[p];
            if (R.status == GBA_PAIR_OK && R.blk_off + R.n_blk > out->blocks_capacity) R.status = GBA_PAIR_NOSPACE;
        }
    }
    return rc;
}

void *gba_host_alloc(gba_ctx *c, uint64_t bytes)
{
    if (!c || c->in_flight) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { c->last_hip = (int)e; return nullptr; }
    try { c->host_allocs.push_back(p); } catch (...) { (void)hipHostFree(p); return nullptr; }
    return p;
}

void gba_host_free(gba_ctx *c, void *ptr)
{
    if (!c || !ptr) return;
    for (size_t i = 0; i < c->host_allocs.size(); i++)
        if (c->host_allocs[i] == ptr) {
            (void)hipSetDevice(c->device);
            (void)hipHostFree(ptr);
            c->host_allocs.erase(c->host_allocs.begin() + (long)i);
            return;
        }
}

int gba_set_guide_producer(gba_ctx *c, const gba_sdp_params *sp)
{
    if (!c) return GBA_E_INVALID;
    if (c->in_flight) return GBA_E_STATE;
    if (!sp) { c->have_guide_sp = false; return GBA_OK; }
    if (!sdp_params_ok(sp)) return GBA_E_UNSUPPORTED;
    c->guide_sp = *sp; c->have_guide_sp = true;
    return GBA_OK;
}

int gba_sdp_device(gba_ctx *c, const gba_batch *b, const gba_sdp_params *sp, gba_guides *out)
{
    if (!c || !b || !sp || !out || (b->q_packed && ((uintptr_t)b->q_arena & 3u))) return GBA_E_INVALID;
    if (c->in_flight) return GBA_E_STATE;
    if (b->n_pairs && (!b->q_arena || !b->q_off || !b->t_off || !b->q_len || !b->t_len)) return GBA_E_INVALID;
    if (b->n_pairs && (!out->status || !out->chain_score || !out->n_blk || !out->blk_off || (!out->blocks && out->blocks_capacity))) return GBA_E_INVALID;
    if (b->n_pairs && !b->t_arena && !c->b_ref.ptr) return GBA_E_STATE;
    return run_sdp(c, b, sp, out);
}

// The same from and to host buffers: the sequences travel in, the guides travel out (synchronous).
int gba_sdp(gba_ctx *c, const gba_batch *b, const gba_sdp_params *sp, gba_guides *out)
{
    if (!c || !b || !sp || !out) return GBA_E_INVALID;
    if (c->in_flight) return GBA_E_STATE;
    const uint32_t n = b->n_pairs;
    out->blocks_needed = 0;
    if (n == 0) return GBA_OK;
    if (!b->q_arena || !b->q_off || !b->t_off || !b->q_len || !b->t_len) return GBA_E_INVALID;
    if (!out->status || !out->chain_score || !out->n_blk || !out->blk_off || (!out->blocks && out->blocks_capacity)) return GBA_E_INVALID;
    if (!b->t_arena && !c->b_ref.ptr) return GBA_E_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    struct { Buf *buf; const void *src; size_t bytes; } in[] = {
        {&c->s_q, b->q_arena, b->q_packed ? ((size_t)b->q_arena_bytes + 15) / 16 * 4 : (size_t)b->q_arena_bytes},
        {&c->s_t, b->t_arena, b->t_arena ? (size_t)b->t_arena_bytes : 0},
        {&c->s_qoff, b->q_off, (size_t)n * 8}, {&c->s_toff, b->t_off, (size_t)n * 8},
        {&c->s_qlen, b->q_len, (size_t)n * 4}, {&c->s_tlen, b->t_len, (size_t)n * 4}, {&c->s_trc, b->t_rc, b->t_rc ? (size_t)n : 0}};
    hipStream_t s = c->slot[0].s;
    for (auto &x : in) {
        if ((rc = ensure(c, *x.buf, x.bytes + 16))) return rc;
        if (x.bytes) HIPCHK(c, hipMemcpyAsync(x.buf->ptr, x.src, x.bytes, hipMemcpyHostToDevice, s));
    }
    // device copies of the outputs: status | score | n_blk (4 bytes each) in s_res, offsets in s_boff, blocks in s_blk
    if ((rc = ensure(c, c->s_res, (size_t)n * 12 + 16))) return rc;
    if ((rc = ensure(c, c->s_boff, (size_t)n * 8 + 16))) return rc;
    if ((rc = ensure(c, c->s_blk, (size_t)out->blocks_capacity * sizeof(gba_block) + 16))) return rc;
    gba_batch db = *b;
    db.q_arena = (const uint8_t *)c->s_q.ptr; db.t_arena = b->t_arena ? (const uint8_t *)c->s_t.ptr : nullptr;
    db.t_rc = b->t_rc ? (const uint8_t *)c->s_trc.ptr : nullptr;
    db.q_off = (const uint64_t *)c->s_qoff.ptr; db.t_off = (const uint64_t *)c->s_toff.ptr;
    db.q_len = (const uint32_t *)c->s_qlen.ptr; db.t_len = (const uint32_t *)c->s_tlen.ptr;
    db.blocks = nullptr; db.blk_off = nullptr; db.n_blk = nullptr;
    gba_guides dg;
    dg.status = (int32_t *)c->s_res.ptr; dg.chain_score = dg.status + n; dg.n_blk = (uint32_t *)(dg.status + 2 * (size_t)n);
    dg.blk_off = (uint64_t *)c->s_boff.ptr; dg.blocks = (gba_block *)c->s_blk.ptr;
    dg.blocks_capacity = out->blocks_capacity; dg.blocks_needed = 0;
    rc = run_sdp(c, &db, sp, &dg);
    if (rc != GBA_OK && rc != GBA_E_NOSPACE) return rc;
    out->blocks_needed = dg.blocks_needed;
    HIPCHK(c, hipMemcpyAsync(out->status, dg.status, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(out->chain_score, dg.chain_score, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(out->n_blk, dg.n_blk, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(out->blk_off, dg.blk_off, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    const uint64_t nb = dg.blocks_needed < out->blocks_capacity ? dg.blocks_needed : out->blocks_capacity;
    if (nb) HIPCHK(c, hipMemcpyAsync(out->blocks, dg.blocks, (size_t)nb * sizeof(gba_block), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return rc;
}

int gba_set_reference(gba_ctx *c, const uint8_t *ref, uint64_t len)
{
    if (!c || (len && !ref)) return GBA_E_INVALID;
    if (c->in_flight) return GBA_E_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->ref_len = 0;
    if (len == 0) { release(c->b_ref); release(c->b_refm); return GBA_OK; }
    // kept packed: 2 bits per base (len / 4 bytes) and a mask bit per base (len / 8 bytes); the letters pass through a
    // staging buffer a chunk at a time
    const uint64_t groups = (len + 31) / 32;
    int rc = ensure(c, c->b_ref, (size_t)(groups * 8 + 16));
    if (rc == GBA_OK) rc = ensure(c, c->b_refm, (size_t)(groups * 4 + 16));
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync((char *)c->b_ref.ptr + groups * 8, 0, 16, c->stream));
    HIPCHK(c, hipMemsetAsync((char *)c->b_refm.ptr + groups * 4, 0, 16, c->stream));
    const uint64_t chunk = 64ull << 20;               // letters per chunk, a multiple of 32
    void *stage = nullptr;
    HIPCHK(c, hipMalloc(&stage, (size_t)(len < chunk ? len : chunk)));
    for (uint64_t at = 0; at < len; at += chunk) {
        const uint64_t n = len - at < chunk ? len - at : chunk;
        hipError_t e = hipMemcpyAsync(stage, ref + at, (size_t)n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_ref_pack, dim3((unsigned)((n + 32 * 256 - 1) / (32 * 256))), dim3(256), 0, c->stream, (const uint8_t *)stage, n, at,
                               (uint32_t *)c->b_ref.ptr, (uint32_t *)c->b_refm.ptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);       // the staging buffer is reused; the caller's letters may go away
        if (e != hipSuccess) { (void)hipFree(stage); c->last_hip = (int)e; return GBA_E_HIP; }
    }
    HIPCHK(c, hipFree(stage));
    c->ref_len = len;
    return GBA_OK;
}

int gba_get_timing(gba_ctx *c, gba_timing *t)
{
    if (!c || !t) return GBA_E_INVALID;
    *t = c->timing;
    t->scratch_retries = c->scratch_retries;
    uint64_t bytes = 0;
    for (auto &b : c->b_pair32) bytes += b.cap;
    for (auto &b : c->b_pair64) bytes += b.cap;
    Buf *bs[] = {&c->b_ref, &c->b_refm, &c->s_trc, &c->b_needed,
                 &c->s_q, &c->s_t, &c->s_blk, &c->s_qoff, &c->s_toff, &c->s_boff, &c->s_qlen, &c->s_tlen,
                 &c->s_nblk, &c->s_res, &c->s_cigoff, &c->s_gout, &c->s_prof};
    for (Buf *b : bs) bytes += b->cap;
    for (auto &b : c->kb) bytes += b.cap;
    t->slot_bytes = 0;
    for (auto &sl : c->slot) {
        Buf *sb[] = {&sl.b_plan, &sl.b_guide, &sl.b_tc, &sl.b_rec, &sl.b_blk, &sl.b_out, &sl.b_cig};
        for (Buf *b : sb) { bytes += b->cap; t->slot_bytes += b->cap; }
    }
    t->device_bytes = bytes;
    return GBA_OK;
}

}  // extern "C"

#define GBA_KBAND_HOST
#include "gba_kband.inc"
#undef GBA_KBAND_HOST
