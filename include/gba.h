/*
 * gba.h -- C ABI of libgba: guided banded alignment (BLASR refine stage) on MI355X.
 *
 * The reference has no FFI for this path: RefineAlignment calls header
 * templates of libcpp directly (iblasr/BlasrUtilsImpl.hpp:705-732).  This header
 * is the boundary a maintainer would bind instead; every entry point names the
 * reference interface it replaces.  Plain pointers and sizes only.
 *
 * Threading: one gba_ctx per (host worker, GPU); a ctx is not thread-safe,
 * different ctxs are independent (the reference's per-thread MappingBuffers,
 * Blasr.cpp:771).  Errors are return codes (< 0); a bad pair never fails a batch,
 * it gets a per-pair status (> 0) and is passed through untouched, like the
 * empty-guide case at iblasr/BlasrUtilsImpl.hpp:681.
 */
#ifndef GBA_H
#define GBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GBA_VERSION 382   /* major*10000 + minor*100 + patch */

/* return codes of the entry points */
enum {
    GBA_OK = 0,
    GBA_E_INVALID = -1,      /* bad argument */
    GBA_E_NOMEM = -2,        /* host or device allocation failed */
    GBA_E_HIP = -3,          /* a HIP call failed; gba_last_hip_error() has the code */
    GBA_E_NOSPACE = -4,      /* caller's block arena too small; gba_results.blocks_needed says how big */
    GBA_E_UNSUPPORTED = -5,  /* scoring outside what the kernels implement */
    GBA_E_STATE = -6,        /* wait without submit, submit while a batch is in flight, ... */
    GBA_E_TOOBIG = -7        /* gba_submit_device_noalloc: the batch needs more scratch than GBA_NOALLOC_SCRATCH_BYTES */
};

/* per-pair status (gba_pair_result.status) */
enum {
    GBA_PAIR_OK = 0,
    GBA_PAIR_EMPTY_GUIDE = 1,      /* no blocks: candidate left as it was (BlasrUtilsImpl.hpp:681) */
    GBA_PAIR_BAD_GUIDE = 2,        /* blocks unordered, overlapping, empty or outside the sequences; window outside the reference */
    GBA_PAIR_UNREACHABLE = 3,      /* the band is disconnected: no path reaches the end cell */
    GBA_PAIR_NOSPACE = 4,          /* blocks of this pair did not fit the caller's arena */
    GBA_PAIR_END_OUT_OF_BAND = 5,  /* k-band: |t_len - q_len| > k */
    GBA_PAIR_TOO_WIDE = 6,         /* a band row wider than the kernels support (linear, k-band: 1023 cells or a shift of 1023; affine: the same) */
    GBA_PAIR_TOO_LONG = 7,         /* more DP rows or columns than the kernels support (65535 rows, affine 32767; 2^20 columns),
                                      or rows x (128 + largest |substitution, insertion cost|) + columns x |deletion cost|
                                      reaches 2^24 (affine 2^23): the scores would not fit the kernels' cell words (a batch
                                      with a q_profile is bounded with 127, affine 63, in place of the largest matrix / gap cost,
                                      and with linear gaps also by rows x 127 + columns x (128 + 127) below 2^24);
                                      gba_sdp_device: a read or window of more than 65535 bases */
    GBA_PAIR_TOO_MANY = 8,         /* gba_sdp_device only: more maximal matches than max_fragments, or more than
                                      64 x max_fragments k-mer hits (a repeat): no guide is produced */
    GBA_PAIR_NO_ALIGNMENT = 9      /* GBA_ALIGN_LOCAL only: no cell of the band scores below zero: nothing aligns */
};

enum { GBA_MODE_LINEAR = 0,   /* replaces GuidedAlign,       BlasrUtilsImpl.hpp:713-716, :727-730 */
       GBA_MODE_AFFINE = 1,   /* replaces AffineGuidedAlign, BlasrUtilsImpl.hpp:707-710, :721-724 */
       GBA_MODE_KBAND  = 2 }; /* replaces KBandAlign(Global), BlasrUtilsImpl.hpp:657-661          */

/* End rule of the guided DP (gba_scoring.align_type).
 * GBA_ALIGN_GLOBAL: from the first cell of the DP region (first guide block's start) to its last (last block's end):
 *   RefineAlignment's calls, iblasr/BlasrUtilsImpl.hpp:707-730 (..., Global, false).
 * GBA_ALIGN_LOCAL: the alignment may begin in any cell of the band at no cost and ends in the cheapest cell of the band:
 *   GuidedAlign(subread, alignedRefSequence, idsScoreFn, 12, ..., exploded, Local, computeProbIsFalse, 6) of
 *   AlignSubreadToAlignmentTarget, iblasr/BlasrAlignImpl.hpp:1418-1421 (CCS and -concordant).  Linear gaps with guide
 *   blocks only (GBA_E_UNSUPPORTED otherwise); such batches run on the wave-per-pair kernels.  What libcpp does with Local
 *   inside the guided DP is not in the reference tree (unpinned); here: a cell takes the free start only when that is
 *   strictly cheaper than every move into it, the end is the first cheapest cell in row-major order, and a pair whose
 *   band holds no negative score gets GBA_PAIR_NO_ALIGNMENT.  q_pos / t_pos, rows / cols and ncells still describe the
 *   whole DP region and blocks stay relative to q_pos / t_pos; n_ins / n_del count the gaps inside the aligned stretch
 *   (q_start .. q_end, t_start .. t_end) only. */
enum { GBA_ALIGN_GLOBAL = 0, GBA_ALIGN_LOCAL = 1 };

/* tie-break among equal candidates of the match state */
enum { GBA_TIE_DIAG_LEFT_UP = 0, GBA_TIE_DIAG_UP_LEFT = 1 };

/* gba_config.flags */
#define GBA_FLAG_TIMING 1u    /* record HIP events around the DP-fill and traceback kernels */

typedef struct gba_ctx gba_ctx;

typedef struct {
    int32_t device;        /* HIP device ordinal */
    uint32_t flags;
    void *stream;          /* hipStream_t to run on; NULL: the ctx creates its own */
} gba_config;

/*
 * Scoring and band parameters: the hot-path subset of MappingParameters
 * (iblasr/MappingParameters.h:207-381) plus the 5x5 SMRTDistanceMatrix as main()
 * leaves it (Blasr.cpp:910-937).  Lower scores are better.
 */
/* An additional scorer the finished path is rescored with (ComputeAlignmentScore of the
 * reference): the probability score of the mapping-quality step, which rescores every refined
 * alignment with SMRTLogProbMatrix and the affine costs (iblasr/BlasrUtilsImpl.hpp:120-151), and
 * the edit distance of the SAM/BAM writers (EditDistanceMatrix, 1, 1; :1056-1063). */
typedef struct {
    int32_t sub[5][5];
    int32_t ins, del;                     /* per gap base */
    int32_t affine_open, affine_extend;   /* used instead when `affine` is nonzero: open + length*extend per gap run */
    int32_t affine;
} gba_rescorer;

#define GBA_MAX_EXTRA 2

typedef struct {
    int32_t sub[5][5];          /* [query code][target code], codes A,C,G,T,N = 0..4 */
    int32_t ins, del;           /* DP cost of a query base against a gap / a target base against a gap */
    int32_t affine_open, affine_extend;
    int32_t stats_ins, stats_del;                     /* rescoring scorer (indel, indel), BlasrUtilsImpl.hpp:633-634 */
    int32_t stats_affine_open, stats_affine_extend;
    int32_t band, max_band;     /* guidedAlignBandSize / bandSize; cap on the widened half-width */
    int32_t mode;               /* GBA_MODE_* */
    int32_t tie_break;          /* GBA_TIE_* */
    int32_t kband_k;            /* GBA_MODE_KBAND: half-width of the diagonal band */
    int32_t stats_affine;       /* nonzero: rescoring charges each gap run open + length*extend */
    int32_t n_extra;            /* 0..GBA_MAX_EXTRA further scorers, results in gba_pair_result.extra_score */
    gba_rescorer extra[GBA_MAX_EXTRA];
    int32_t align_type;         /* GBA_ALIGN_* */
} gba_scoring;

/* one gap-free run of the alignment: the reference's blasr::Block (BlasrUtilsImpl.hpp:745-746) */
typedef struct { uint32_t q_pos, t_pos, length; } gba_block;

/*
 * A batch of independent (query, target, guide) pairs, struct-of-arrays.  Pair p
 * is q_arena[q_off[p] .. +q_len[p]), t_arena[t_off[p] .. +t_len[p]) (ASCII,
 * already sliced as BlasrUtilsImpl.hpp:693-703 does) and the guide
 * blocks[blk_off[p] .. +n_blk[p]) with coordinates relative to those slices.
 * Pairs may share arena bytes (all candidates of a read view the same query).
 * All pointers are host pointers for gba_submit and device pointers for
 * gba_submit_device.
 *
 * Targets as reference windows: with t_arena == NULL the targets are windows of
 * the reference given to gba_set_reference (resident in device memory): t_off[p]
 * is the window's start on the forward strand, t_len[p] its length.  t_rc (may
 * be NULL = all forward) marks the pairs whose target is the REVERSE COMPLEMENT
 * of its arena / reference slice; the slice is complemented while it is packed
 * on the device, which replaces the per-candidate window copy
 * genome.Copy / genome.CopyAsRC of iblasr/BlasrAlignImpl.hpp:727-743.  Guide
 * blocks and results are in the coordinates of the target as the aligner sees it
 * (after the reverse complement), exactly as for an explicit arena.
 */
typedef struct {
    uint32_t n_pairs;
    const uint8_t *q_arena; const uint64_t *q_off; const uint32_t *q_len;
    const uint8_t *t_arena; const uint64_t *t_off; const uint32_t *t_len;
    const gba_block *blocks; const uint64_t *blk_off; const uint32_t *n_blk;
    uint64_t q_arena_bytes, t_arena_bytes, n_blocks_total;   /* extents of the three arenas */
    const uint8_t *t_rc;        /* per pair, nonzero: target = reverse complement of its slice; NULL: none */
    uint32_t q_packed;          /* Nonzero: q_arena holds the reads 2 bits per base instead of letters: base x of the arena is
                                   bits 2 (x & 3) .. 2 (x & 3) + 1 of byte x >> 2, A, C, G, T = 0 .. 3 (a read with other
                                   letters cannot be packed: send that batch as letters).  q_off, q_len and q_arena_bytes
                                   still count BASES; the packed arena holds ((q_arena_bytes + 15) / 16) * 4 bytes and starts
                                   on a 4-byte boundary.  A quarter of the bytes cross the link and stay that way in device
                                   memory: the kernels read the packed form.  Every entry point that takes a gba_batch
                                   accepts it, host or device buffers, with or without blocks. */
    const int16_t *q_profile;   /* NULL, or the costs of a quality-value scorer in place of matrix, ins and del for the DP of this
                                   batch: GuidedAlign(qSeq, tSeq, alignmentCandidate, idsScoreFn, band, ...) of
                                   iblasr/BlasrUtilsImpl.hpp:705-717, and with GBA_MODE_AFFINE the AffineGuidedAlign call of :707-710.
                                   11 values per base of q_arena (per BASE when the reads are packed), for the base at arena
                                   position x: q_profile[11 x + c] = Match against target code c (A, C, G, T, N),
                                   q_profile[11 x + 5] = Insertion of the base, q_profile[11 x + 6 + c] = Deletion of a
                                   target base of code c behind it; each taken within +-127 (+-63 with GBA_MODE_AFFINE, whose gap
                                   open and extend stay gba_scoring's constants).  Deletions in front of the first
                                   query base cost gba_scoring.del.  The rescoring (stats_*, extra[]) is unchanged.  Host
                                   pointer for gba_submit (copied whole before the first part), device pointer for
                                   gba_submit_device / _noalloc.  Such batches run on the wave-per-pair kernels. */
} gba_batch;

/* what RefineAlignment copies back into the candidate (BlasrUtilsImpl.hpp:745-772) */
typedef struct {
    int32_t status;             /* GBA_PAIR_* */
    int32_t dp_score;           /* optimum of the banded DP (the value GuidedAlign returns) */
    int32_t score;              /* ComputeAlignmentStats rescoring of the path, :733-736 -> candidate.score */
    uint32_t q_pos, t_pos;      /* refinedAlignment.qPos / tPos: start of the DP region in q / t */
    uint32_t n_blk;
    uint64_t blk_off;           /* first block of this pair in gba_results.blocks */
    uint32_t n_match, n_mismatch, n_ins, n_del;
    uint32_t rows, cols;        /* query / target bases inside the DP region */
    uint64_t ncells;            /* refinedAlignment.nCells: sum over guide rows of tPre+tPost+1 */
    int32_t extra_score[GBA_MAX_EXTRA];   /* the path under gba_scoring.extra[] */
    /* The aligned interval, first aligned base to one past the last, in q and in t as the aligner sees
     * them: q_pos + first block's q_pos .. q_pos + last block's q_pos + length (what GetQInterval,
     * iblasr/BlasrUtilsImpl.hpp:335, and the qStart qEnd tStart tEnd columns of -m 4 are made of,
     * ctest/bug25328.t:12, ctest/hitpolicy.t:95).  With score, n_match .. n_del and the mapping
     * quality this is the whole interval record, so a caller that prints intervals needs no blocks. */
    uint32_t q_start, q_end, t_start, t_end;
} gba_pair_result;

/* CIGAR operations in the BAM encoding (length << 4 | op), the text-ready form of a refined
 * alignment that the reference's SAM / BAM writers derive from blocks and sequences
 * (SAMOutput / BAMOutput::PrintAlignment with params.cigarUseSeqMatch and
 * params.allowAdjacentIndels, iblasr/BlasrUtilsImpl.hpp:969-974).  Soft / hard clips are the
 * writer's business (they depend on --clipping and on the full read) and are not emitted. */
enum { GBA_CIGAR_M = 0, GBA_CIGAR_I = 1, GBA_CIGAR_D = 2, GBA_CIGAR_EQ = 7, GBA_CIGAR_X = 8 };
#define GBA_CIGAR_SEQ_MATCH 1u              /* gap-free columns as runs of = and X (cigarUseSeqMatch); otherwise one M per run */
#define GBA_CIGAR_ADJACENT_INDELS 2u        /* keep an insertion next to a deletion as they are (-allowAdjacentIndels); otherwise the
                                               part the two have in common becomes aligned columns, RegisterBlasrOptions.h:254-259 */

typedef struct {
    gba_pair_result *pairs;     /* n_pairs entries */
    gba_block *blocks;          /* arena for all pairs' blocks, relative to (q_pos, t_pos); NULL: blocks are not wanted (interval
                                   and summary output, or CIGARs only): nothing is packed or copied out for them, no pair
                                   gets GBA_PAIR_NOSPACE, blocks_capacity must be 0 and blocks_needed is still reported */
    uint64_t blocks_capacity;   /* in blocks */
    uint64_t blocks_needed;     /* out: total blocks of the batch */
    /* optional (cigar_ops == NULL: not wanted): the CIGAR of every aligned pair, packed in input order */
    uint32_t *cigar_ops;        /* arena for all pairs' operations */
    uint64_t *cigar_off;        /* n_pairs + 1 entries: the operations of pair p are cigar_ops[cigar_off[p] .. cigar_off[p + 1]);
                                   pairs that were not aligned have none */
    uint64_t cigar_capacity;    /* in operations */
    uint64_t cigar_needed;      /* out: total operations of the batch; beyond cigar_capacity nothing is written and the
                                   entry point returns GBA_E_NOSPACE */
    uint32_t cigar_flags;       /* GBA_CIGAR_* */
} gba_results;

/* live measurements of the last waited batch (GBA_FLAG_TIMING) */
typedef struct {
    float fill_ms;              /* bulk DP-fill launches (HIP events around them on their stream), summed over the parts a
                                   large batch is run in; up to two more classes run beside each on other streams */
    float traceback_ms;         /* from the end of the last bulk fill to the end of the last traceback */
    float total_ms;             /* first kernel to last kernel of the batch */
    uint32_t fill_launches;
    uint64_t cells;             /* sum of ncells over pairs with status OK */
    uint64_t rows;              /* DP rows of the batch */
    uint64_t record_bytes;      /* traceback bytes the fill kernels wrote */
    uint32_t wide_pairs;        /* pairs that ran in the wave-per-pair kernels (band rows too wide for a lane) */
    uint32_t scratch_retries;   /* since gba_create, with or without GBA_FLAG_TIMING: batches that were run a second time
                                   because refined alignments had more blocks than the block scratch, sized for about three
                                   rows per block plus a shared reserve, had room for (the second run has room for a block
                                   per row; results are those of a single run) */
    uint64_t device_bytes;      /* with or without GBA_FLAG_TIMING: device memory the ctx holds now (grow-only scratch of all
                                   kinds, the resident reference, device copies of host batches) ... */
    uint64_t slot_bytes;        /* ... of which the scratch of the part pipeline (guide stream, target codes, traceback
                                   records, block scratch; three parts in flight) */
    float fill_busy_ms;         /* time with at least one of those bulk fill launches in flight: the union of the intervals
                                   that fill_ms sums (launches of consecutive parts overlap; equal to fill_ms for one part) */
} gba_timing;

/* Replaces the per-thread `MappingBuffers mappingBuffers;` (Blasr.cpp:771): device
 * scratch is owned by the ctx, grow-only, freed by gba_destroy. */
int gba_create(gba_ctx **out, const gba_config *cfg);
void gba_destroy(gba_ctx *ctx);

/* Replaces building distScoreFn / distScoreFn2 and choosing kernel and band
 * (BlasrUtilsImpl.hpp:630-646, :705-731).  GBA_E_INVALID for values outside the
 * enums / ranges, GBA_E_UNSUPPORTED for costs the kernels' word format cannot
 * hold; either way the scoring set before stays in force. */
int gba_set_scoring(gba_ctx *ctx, const gba_scoring *s);

/* Replaces the RefineAlignments loop over candidates (BlasrUtilsImpl.hpp:911-914)
 * for a whole batch held in host memory.  gba_submit checks its arguments and
 * returns at once: a worker thread of the ctx copies the batch to the device part
 * by part, runs it and copies the results of each part into host_out->pairs /
 * host_out->blocks while the next parts are still running, so that the copies in
 * both directions overlap the kernels (and the caller's own work).  gba_wait
 * blocks until host_out is complete (blocks_needed included) and returns GBA_OK,
 * GBA_E_NOSPACE (pairs whose blocks did not fit have GBA_PAIR_NOSPACE) or the
 * error the batch ran into.  The caller keeps the batch arrays, *host_out and its
 * buffers alive and unread until gba_wait has returned.  Buffers from
 * gba_host_alloc move at the full rate of the link; ordinary (pageable) memory
 * works and is slower.
 * A batch with pairs must bring every array except t_arena (NULL: windows of
 * the reference) and t_rc: GBA_E_INVALID otherwise; GBA_E_STATE without scoring,
 * with a batch in flight, or with a NULL t_arena and no reference. */
int gba_submit(gba_ctx *ctx, const gba_batch *host_batch, gba_results *host_out);
int gba_wait(gba_ctx *ctx);

/* Page-locked host memory owned by the ctx, for batch arenas and result buffers:
 * the pinned half of what the per-thread MappingBuffers (iblasr/MappingBuffers.hpp:20-49)
 * becomes on this path, the device scratch of the ctx being the other half.
 * NULL when the allocation fails or a batch is in flight.  Freed by gba_host_free
 * or, whatever is left, by gba_destroy. */
void *gba_host_alloc(gba_ctx *ctx, uint64_t bytes);
void gba_host_free(gba_ctx *ctx, void *ptr);

/* Same, for callers whose batch and result buffers already live in device
 * memory (all pointers in `dev_batch` and `dev_out` are device pointers).
 * Returns when the batch has completed; out->blocks_needed is filled in. */
int gba_submit_device(gba_ctx *ctx, const gba_batch *dev_batch, gba_results *dev_out);

/* gba_submit_device for callers that cannot allocate device memory while they run (code under HIP
 * graph capture, tight serving loops, harnesses that forbid it).  It has no counterpart in the
 * reference.  Same arguments, results and return codes as gba_submit_device, but it calls no
 * hipMalloc / hipFree: all scratch (per-pair arrays, guide stream, target codes, traceback records,
 * block scratch) is carved from a fixed array of GBA_NOALLOC_SCRATCH_BYTES that every GPU the
 * library runs on holds from load until the process ends.  A batch that needs more returns
 * GBA_E_TOOBIG and leaves the outputs untouched (roughly 330 bytes of scratch per query base at the
 * default band: about 190 pairs of 10 kb, or 13 000 of 150 b).  The batch runs in one piece.  All
 * calls on one GPU share that GPU's array, whichever ctx they come through: calls that overlap in
 * time on one GPU race; calls one after the other are independent (nothing is carried over). */
#define GBA_NOALLOC_SCRATCH_BYTES (64u << 20)
int gba_submit_device_noalloc(gba_ctx *ctx, const gba_batch *dev_batch, gba_results *dev_out);

/* ---- guide production: the step in front of the refine stage -------------------------------
 * Replaces SDPAlign(alignment->qAlignedSeq, alignment->tAlignedSeq, distScoreFn, sdpTupleSize,
 * params.sdpIns, params.sdpDel, params.indelRate*3, *alignment, mappingBuffers, Local, ...)
 * (iblasr/BlasrAlignImpl.hpp:980-987), whose result is alignment->blocks, the guide of the refine
 * stage: exact k-mer matches between each read and its window are chained into an increasing list
 * of gap-free blocks.  Fragments are the maximal exact matches of at least tuple_size ACGT bases;
 * a chain scores match per anchored base, minus ins / del per base by which consecutive fragments
 * change diagonal, minus match per base of overlap between consecutive fragments (the earlier one
 * is shortened); the best chain ending anywhere is returned (Local).  SDPAlign itself is libcpp
 * code that the reference tree does not hold: this is the published scheme as an exact optimum,
 * not a transcription (no detailed second pass with a smaller tuple, no fastSDP pruning). */
typedef struct {
    int32_t tuple_size;         /* params.sdpTupleSize (11, iblasr/MappingParameters.h:222); 4..15 */
    int32_t ins, del;           /* params.sdpIns (5), params.sdpDel (10), :220-221; 0..127 */
    int32_t match;              /* reward per anchored base: minus the match score of the scorer (5); 1..127 */
    uint32_t max_fragments;     /* fragments (and blocks) per pair, at most; 1..4096 */
} gba_sdp_params;

/* All pointers of gba_guides are device pointers.  n_blk, blk_off and blocks have the layout of
 * gba_batch: they can be handed to gba_submit_device as the guides of the same batch. */
typedef struct {
    int32_t *status;            /* n_pairs: GBA_PAIR_OK; GBA_PAIR_EMPTY_GUIDE (no match of tuple_size bases); GBA_PAIR_BAD_GUIDE (a
                                   slice outside its arena); GBA_PAIR_TOO_LONG; GBA_PAIR_TOO_MANY; GBA_PAIR_NOSPACE */
    int32_t *chain_score;       /* n_pairs */
    uint32_t *n_blk;            /* n_pairs: blocks of the chain, 0 unless status is OK (or NOSPACE) */
    uint64_t *blk_off;          /* n_pairs: first block of the pair in `blocks`, packed in input order */
    gba_block *blocks;          /* positions relative to the starts of the read and of the window */
    uint64_t blocks_capacity;
    uint64_t blocks_needed;     /* out */
} gba_guides;

/* batch: q_arena, q_off, q_len, t_arena (or the ctx reference), t_off, t_len, t_rc are used; blocks,
 * blk_off, n_blk are ignored.  Returns when the guides are complete.  GBA_E_NOSPACE: blocks_needed >
 * blocks_capacity (pairs that did not fit have GBA_PAIR_NOSPACE). */
int gba_sdp_device(gba_ctx *ctx, const gba_batch *device_batch, const gba_sdp_params *params, gba_guides *device_out);

/* The same for host buffers (every pointer of the batch and of the guides is a host pointer): copies the
 * sequences in, returns when the guides have been copied out. */
int gba_sdp(gba_ctx *ctx, const gba_batch *host_batch, const gba_sdp_params *params, gba_guides *host_out);

/* From sequences to refined alignments in one call.  With a guide producer set (params != NULL; NULL
 * switches it off again), gba_submit and gba_submit_device accept a batch whose `blocks` is NULL
 * (blk_off and n_blk are then ignored): the guide of every pair is made on the device as gba_sdp_device
 * makes it and handed to the refine stage where it lies, which is what AlignIntervals followed by
 * RefineAlignments do per candidate (iblasr/BlasrAlignImpl.hpp:980-987, then
 * iblasr/BlasrUtilsImpl.hpp:905-923); no guide crosses the link in either direction.  A pair without a
 * guide gets the producer's reason as its status (GBA_PAIR_EMPTY_GUIDE, GBA_PAIR_TOO_MANY,
 * GBA_PAIR_TOO_LONG, GBA_PAIR_BAD_GUIDE).  Batches that bring their own blocks run as before.
 * GBA_E_UNSUPPORTED for values outside gba_sdp_params' ranges; gba_submit_device_noalloc does not
 * take such batches (GBA_E_INVALID). */
int gba_set_guide_producer(gba_ctx *ctx, const gba_sdp_params *params);

/* Keeps the reference (genome, ASCII; `ref` is a host pointer) in device memory for batches whose
 * t_arena is NULL; len == 0 drops it.  It is kept 2 bits per base plus one mask bit per base for the
 * letters outside ACGT (N runs): 3 / 8 of a byte per base, 24 MB for a 64 Mb chromosome; windows are cut
 * out of the packed form (reverse strand: read backwards and complemented) where the kernels need them.
 * Replaces the per-candidate window copies out of the genome, iblasr/BlasrAlignImpl.hpp:727-743. */
int gba_set_reference(gba_ctx *ctx, const uint8_t *ref, uint64_t len);

/* ---- fixed-band aligners: the other pairwise aligners of the refine stage --------------------
 * One entry point for the reference's fixed-band family, which the guided path falls back to or
 * is followed by:
 *   KBandAlign(qSeq, tSeq, SMRTDistanceMatrix, ins, del, k, scoreMat, pathMat, alignment, scoreFn, alignType)
 *     iblasr/BlasrUtilsImpl.hpp:657-661 (-global: Global, k = drift) and
 *     iblasr/BlasrAlignImpl.hpp:1278-1294 (quality-value scorers, Fit: the legacy refine of
 *     iblasr/BlasrUtilsImpl.hpp:775-902)                       -> GBA_KB_GLOBAL / GBA_KB_FIT, GBA_KB_LINEAR
 *   AffineKBandAlign(qSeq, tSeq, SMRTDistanceMatrix, hpInsOpen, hpInsExtend, insOpen, insExtend, del, k, ...)
 *     iblasr/BlasrAlignImpl.hpp:1259-1267                         -> GBA_KB_GLOBAL, GBA_KB_AFFINE_HP
 *   ExtendAlignmentForward(readSuffix, 0, genomeSuffix, 0, extendBandSize, ..., distScoreFn, 1, maxExtendDropoff)
 *     iblasr/BlasrAlignImpl.hpp:1086-1099, ExtendAlignmentReverse :1144-1155 (-extend)
 *                                                                 -> GBA_KB_EXTEND (reverse[p] for the prefix side)
 * Cells (i, j) with |j - i| <= k[p]; lower scores are better.  FIT: the whole query against a part of
 * the target (row 0 is free, the end is the cheapest cell of the last row, the leftmost on ties).
 * EXTEND: from (0, 0); rows are taken in order, the fill ends with the (max_drops + 1)-th row in a row
 * that brought no new minimum, the alignment ends in the best cell ((0, 0): nothing is extended, no
 * blocks, score 0; the caller extends when the score is negative, BlasrAlignImpl.hpp:1102).
 * AFFINE_HP: the match state keeps linear deletions; an insertion run costs ins_open for its first base
 * and ins_extend for each further one; a homopolymer insertion (hp_ins_open / hp_ins_extend) is allowed
 * where the inserted base equals the target base to its left (hp_rule 0) or the query base before it
 * (hp_rule 1).  These routines are libcpp code absent from the reference tree: every such choice is a
 * field here.
 * Quality-value scorers (IDSScoreFunction, QualityValueScoreFunction: iblasr/BlasrUtilsImpl.hpp:636-646,
 * iblasr/BlasrAlignImpl.hpp:1241-1253) come in as a cost profile: for the query base at arena byte x,
 * profile[11 x .. 11 x + 10] = Match cost against target code A, C, G, T, N; Insertion cost of the
 * base; Deletion cost of a target base of code A, C, G, T, N skipped after this base.  Row 0 uses del.
 * Limits: k <= 255 (GBA_PAIR_TOO_WIDE), q_len + t_len < 2^19 and, for EXTEND, q_len <= 4095
 * (GBA_PAIR_TOO_LONG), |cost| <= 1023 (GBA_E_UNSUPPORTED).  GLOBAL needs |t_len - q_len| <= k, FIT
 * q_len - t_len <= k (GBA_PAIR_END_OUT_OF_BAND).  reverse[p] != 0: both sequences are read back to front; blocks and
 * positions are reported in forward coordinates.  Results as for gba_submit: q_pos / t_pos = start of the
 * first block, blocks relative to it; ncells = band cells of the rows filled.  Host pointers; returns when
 * the results are in `host_out` (GBA_E_NOSPACE as for gba_wait); CIGAR output is not offered here. */
enum { GBA_KB_GLOBAL = 0, GBA_KB_FIT = 1, GBA_KB_EXTEND = 2 };
enum { GBA_KB_LINEAR = 0, GBA_KB_AFFINE_HP = 1 };

typedef struct {
    int32_t sub[5][5];
    int32_t ins, del;
    int32_t align_type, gap_model;
    int32_t hp_ins_open, hp_ins_extend, ins_open, ins_extend;
    int32_t hp_rule;
    int32_t tie_break;          /* GBA_TIE_*; the two insertion states rank where Up does */
    int32_t max_drops;          /* EXTEND: params.maxExtendDropoff (10, iblasr/MappingParameters.h:288) */
    int32_t stats_ins, stats_del;                                     /* rescoring of the path, as gba_scoring's */
    int32_t stats_affine, stats_affine_open, stats_affine_extend;
} gba_kband_params;

typedef struct {
    uint32_t n_pairs;
    const uint8_t *q_arena; const uint64_t *q_off; const uint32_t *q_len;
    const uint8_t *t_arena; const uint64_t *t_off; const uint32_t *t_len;
    uint64_t q_arena_bytes, t_arena_bytes;
    const int32_t *k;           /* per pair: half-width of the band */
    const uint8_t *reverse;     /* per pair, may be NULL */
    const int16_t *profile;     /* NULL: matrix, ins, del; else 11 costs per byte of q_arena */
} gba_kband_batch;

int gba_kband(gba_ctx *ctx, const gba_kband_batch *host_batch, const gba_kband_params *params, gba_results *host_out);

int gba_get_timing(gba_ctx *ctx, gba_timing *t);
int gba_last_hip_error(gba_ctx *ctx);
const char *gba_strerror(int code);
int gba_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GBA_H */
