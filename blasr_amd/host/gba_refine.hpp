// gba_refine.hpp -- C++ host side of the refine path over the C ABI (include/gba.h).
//
// Header-only mirror of the reference's RefineAlignments / RefineAlignment for the guided
// branch (/root/reference/iblasr/BlasrUtilsImpl.hpp:620-923).  The reference calls
// GuidedAlign once per candidate; here the candidates of any number of reads are collected
// into one struct-of-arrays batch, aligned by one gba_submit / gba_wait, and copied back with
// the same field updates as :745-772.  Types are the minimal stand-ins for the libcpp classes
// the reference uses at this seam (blasr::Block, T_AlignmentCandidate, MappingParameters).
#ifndef GBA_REFINE_HPP
#define GBA_REFINE_HPP

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/gba.h"

namespace gba {

struct Block { uint32_t qPos, tPos, length; };      // layout-compatible with gba_block

// hot-path subset of iblasr/MappingParameters.h (defaults :207-381, bandSize after MakeSane :501-503)
struct MappingParameters {
    int insertion = 4, deletion = 5, indel = 5;
    int match = -5, mismatch = 6;
    bool affineAlign = false;
    int affineOpen = 10, affineExtend = 0;
    int guidedAlignBandSize = 10, bandSize = 16, maxBandSize = 250;
    bool useGuidedAlign = true, refineAlignments = true, sortRefinedAlignments = true;
    bool doGlobalAlignment = false;
    bool refineBetweenAnchorsOnly = false;              // -rbao
    int maxScore = -200;
    int nCandidates = 10, nBest = 10;                   // :242-243
    float minPctSimilarity = 0;                         // :299
    float minFractionToBeConsideredOverlapping = 0.75f; // :212
    int tieBreak = GBA_TIE_DIAG_LEFT_UP;
    int sdpTupleSize = 11, sdpIns = 5, sdpDel = 10;     // :220-222
    uint32_t sdpMaxFragments = 1024;                    // not a BLASR option: the device's budget of maximal matches per candidate
};

// the T_AlignmentCandidate fields RefineAlignment reads and writes (:678-772)
struct AlignmentCandidate {
    std::string tAlignedSeq, qAlignedSeq;
    uint32_t tAlignedSeqPos = 0, qAlignedSeqPos = 0, tAlignedSeqLength = 0, qAlignedSeqLength = 0;
    uint32_t qPos = 0, tPos = 0;
    std::vector<Block> blocks;
    int tIndex = 0;                // contig of the window (compared at BlasrUtilsImpl.hpp:552)
    int score = 0;
    uint64_t nCells = 0;
    uint32_t nMatch = 0, nMismatch = 0, nIns = 0, nDel = 0;
    float pctSimilarity = 0;
    int status = 0;
    float probScore = 0;           // the path under the probability scorer (StoreMapQVs, :135-151)
    int mapQV = 0, numSignificantClusters = 1;
    // from the pair result: the aligned stretch of the read / of the window (GetQInterval, the -m 4 columns)
    // and the number of blocks: what interval output and the mapping qualities need when a batch ran
    // without a block arena (blocks is then empty)
    uint32_t qStart = 0, qEnd = 0, tStart = 0, tEnd = 0, nBlocks = 0;
    bool haveInterval = false;
};

// RefineAlignment never assigns affineOpen / affineExtend of its two DistanceMatrixScoreFunction
// objects (only idsScoreFn receives params.affineOpen / affineExtend, :643-644; StoreMapQVs does
// assign them, :120-122, "bug 24363"): both run with what libcpp's constructor leaves there.
// libcpp is not in the reference tree, so the value is UNPINNED; believed to be 0.
constexpr int kDistScorerAffineOpen = 0, kDistScorerAffineExtend = 0;      // distScoreFn  (the DP of -affineAlign without QVs)
constexpr int kStatsScorerAffineOpen = 0, kStatsScorerAffineExtend = 0;    // distScoreFn2 (ComputeAlignmentStats, :733-736)

inline void MatrixFromParams(const MappingParameters &p, gba_scoring &s)
{
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) s.sub[a][b] = (a == 4 || b == 4) ? 0 : (a == b ? p.match : p.mismatch);
}

// ids: the DP runs with idsScoreFn (a batch that brings cost profiles), the one scorer that receives
// params.affineOpen / affineExtend (:643-644)
inline gba_scoring ScoringFromParams(const MappingParameters &p, bool ids = false, int alignType = GBA_ALIGN_GLOBAL)
{
    gba_scoring s{};
    MatrixFromParams(p, s);
    // distScoreFn(SMRTDistanceMatrix, params.deletion, params.insertion), read as ctor(ins, del): :630-631
    s.ins = p.deletion; s.del = p.insertion;
    s.stats_ins = s.stats_del = p.indel;                       // distScoreFn2, :633-634
    s.affine_open = ids ? p.affineOpen : kDistScorerAffineOpen; s.affine_extend = ids ? p.affineExtend : kDistScorerAffineExtend;
    s.stats_affine_open = kStatsScorerAffineOpen; s.stats_affine_extend = kStatsScorerAffineExtend;
    s.band = p.affineAlign ? p.bandSize : p.guidedAlignBandSize;   // :706-731
    s.max_band = p.maxBandSize;
    s.mode = p.affineAlign ? GBA_MODE_AFFINE : GBA_MODE_LINEAR;
    s.tie_break = p.tieBreak;
    s.stats_affine = p.affineAlign ? 1 : 0;
    s.align_type = alignType;
    return s;
}

// ComputeDrift(alignmentCandidate) (:654, :793; libcpp AlignmentUtils, not in the tree): the largest
// distance of a block from the diagonal of the first block.  UNPINNED; restated from the way both
// call sites use it (the half-width of a diagonal band that must hold the guide).
inline int ComputeDrift(const std::vector<Block> &blocks)
{
    if (blocks.empty()) return 0;
    const int64_t d0 = (int64_t)blocks[0].tPos - blocks[0].qPos;
    int64_t m = 0;
    for (const Block &b : blocks) {
        const int64_t d = (int64_t)b.tPos - b.qPos - d0;
        m = std::max(m, d < 0 ? -d : d);
    }
    return (int)m;
}

// KBandAlign(subread, tAlignedSeq, SMRTDistanceMatrix, params.insertion, params.deletion, drift, ...,
// idsScoreFn, Global) of the -global branch (:657-661), rescored by distScoreFn2 without the affine
// flag (:663-666).  Of the IDS scorer a read without quality values keeps the matrix and the two
// explicit gap costs; its per-base quality terms belong to the QV-aware scorers.
inline gba_scoring ScoringGlobal(const MappingParameters &p, int k)
{
    gba_scoring s{};
    MatrixFromParams(p, s);
    s.ins = p.insertion; s.del = p.deletion;
    s.stats_ins = s.stats_del = p.indel;
    s.mode = GBA_MODE_KBAND; s.kband_k = k;
    s.tie_break = p.tieBreak;
    return s;
}

// CIGAR of a refined candidate, the text the SAM / BAM writers print (the reference calls libcpp's
// SAMOutput / BAMOutput::PrintAlignment with params.cigarUseSeqMatch and params.allowAdjacentIndels,
// BlasrUtilsImpl.hpp:970-974).  Gap-free runs are `=` / `X` runs (or one `M`); the gap between two
// blocks is an insertion, a deletion, or both: adjacent ones are merged unless allowed
// (RegisterBlasrOptions.h:254-259), i.e. the part they have in common becomes aligned columns.
// Shape as in the literals of ctest/bamOut.t:21-22; the writers themselves are not in the tree.
inline std::string CigarString(const AlignmentCandidate &c, bool cigarUseSeqMatch = true, bool allowAdjacentIndels = false,
                               uint32_t clipLeft = 0, uint32_t clipRight = 0)
{
    std::string out;
    uint32_t runN = 0;
    char runOp = 0;
    auto push = [&](uint32_t n, char op) {
        if (n == 0) return;
        if (op == runOp) { runN += n; return; }
        if (runN) out += std::to_string(runN) + runOp;
        runN = n; runOp = op;
    };
    auto columns = [&](uint32_t q, uint32_t t, uint32_t n) {
        if (!cigarUseSeqMatch) { push(n, 'M'); return; }
        for (uint32_t k = 0; k < n; k++) push(1, c.qAlignedSeq[q + k] == c.tAlignedSeq[t + k] ? '=' : 'X');
    };
    push(clipLeft, 'S');
    for (size_t b = 0; b < c.blocks.size(); b++) {
        const uint32_t bq = c.qPos + c.blocks[b].qPos, bt = c.tPos + c.blocks[b].tPos, bl = c.blocks[b].length;
        columns(bq, bt, bl);
        if (b + 1 == c.blocks.size()) break;
        const uint32_t qGap = c.qPos + c.blocks[b + 1].qPos - (bq + bl), tGap = c.tPos + c.blocks[b + 1].tPos - (bt + bl);
        uint32_t common = 0;
        if (!allowAdjacentIndels && qGap && tGap) {
            common = qGap < tGap ? qGap : tGap;
            columns(bq + bl, bt + bl, common);
        }
        push(qGap - common, 'I');
        push(tGap - common, 'D');
    }
    push(clipRight, 'S');
    if (runN) out += std::to_string(runN) + runOp;
    return out;
}

// The per-base fields of a FASTQSequence the QV-aware scorers read (empty string = absent vector).
struct QualityValues { std::string qual, insertionQV, deletionQV, deletionTag, substitutionQV, substitutionTag; };

// IDSScoreFunction as RefineAlignment sets it up (:636-646), evaluated per query base into the 11 costs gba takes
// (see gba.h): Match = the matrix diagonal for equal bases, substitutionQV when the target base is the
// substitutionTag, substitutionPrior otherwise; Insertion = insertionQV (params.insertion without);
// Deletion = deletionQV when the target base is the deletionTag, globalDeletionPrior otherwise (params.deletion
// without deletion QVs / tags).  The class is libcpp code absent from the reference tree: UNPINNED.
inline std::vector<int16_t> IdsProfile(const std::string &read, const QualityValues &qv, const MappingParameters &p,
                                       int substitutionPrior = 20, int globalDeletionPrior = 13)
{
    auto code = [](char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } };
    gba_scoring m{};
    MatrixFromParams(p, m);
    std::vector<int16_t> prof(11 * read.size());
    const bool sub = qv.substitutionQV.size() == read.size() && qv.substitutionTag.size() == read.size();
    const bool del = qv.deletionQV.size() == read.size() && qv.deletionTag.size() == read.size();
    for (size_t i = 0; i < read.size(); i++) {
        const int qc = code(read[i]);
        for (int c = 0; c < 5; c++) {
            int v = substitutionPrior;
            if (sub && c < 4 && code(qv.substitutionTag[i]) == c) v = (uint8_t)qv.substitutionQV[i];
            if (qc == c) v = m.sub[qc][qc];
            prof[11 * i + c] = (int16_t)std::max(-1023, std::min(1023, v));
            int d = p.deletion;
            if (del) d = (c < 4 && qv.deletionTag[i] != 'N' && code(qv.deletionTag[i]) == c) ? (uint8_t)qv.deletionQV[i] : globalDeletionPrior;
            prof[11 * i + 6 + c] = (int16_t)d;
        }
        prof[11 * i + 5] = (int16_t)(qv.insertionQV.size() == read.size() ? (uint8_t)qv.insertionQV[i] : p.insertion);
    }
    return prof;
}

// Collects (read, candidate) pairs of many reads, runs them as one batch.
class RefineBatch {
public:
    explicit RefineBatch(gba_ctx *ctx) : ctx_(ctx) {}

    // slicing rules of :693-703; a candidate without blocks is kept with n_blk = 0 (:681).
    // profile (optional, all reads of a batch or none): the costs of the IDS scorer for this read, 11 per base
    // (IdsProfile below): the batch then runs GuidedAlign with idsScoreFn (:713-716 with -useQuality).
    void Add(const std::string &read, AlignmentCandidate *c, const std::vector<int16_t> *profile = nullptr)
    {
        const size_t before = q_arena_.size();
        const uint64_t qbase = Intern(read);
        if (profile) {
            if (profile->size() != 11 * read.size()) throw std::invalid_argument("profile must hold 11 costs per base of the read");
            if (q_arena_.size() != before) prof_.insert(prof_.end(), profile->begin(), profile->end());     // a newly interned read
        }
        cands_.push_back(c);
        const size_t nb = c->blocks.size();
        if (nb == 0) {
            q_off_.push_back(0); q_len_.push_back(0); t_off_.push_back(0); t_len_.push_back(0);
            b_off_.push_back(0); n_blk_.push_back(0);
            return;
        }
        const Block &first = c->blocks.front(), &last = c->blocks.back();
        const uint32_t tl = last.tPos + last.length - first.tPos, ql = last.qPos + last.length;
        if ((uint64_t)c->qAlignedSeqPos + c->qPos + ql > read.size() || (uint64_t)c->tPos + tl > c->tAlignedSeq.size())
            throw std::out_of_range("candidate guide exceeds its sequences");
        q_off_.push_back(qbase + c->qAlignedSeqPos + c->qPos); q_len_.push_back(ql);
        t_off_.push_back(t_arena_.size()); t_len_.push_back(tl);
        t_arena_.append(c->tAlignedSeq, c->tPos, tl);
        b_off_.push_back(blocks_.size()); n_blk_.push_back((uint32_t)nb);
        for (const Block &b : c->blocks) blocks_.push_back(gba_block{b.qPos, b.tPos, b.length});
    }

    // one submit/wait for everything added; copy-back of :733-772.  Returns the gba return code.
    // wantBlocks false: interval / summary output (printFormat Interval, SummaryPrint, :992-1000) prints only
    // what the pair results hold, so no block is packed or copied out.
    // alignType GBA_ALIGN_LOCAL: the end rule of GuidedAlign(..., Local, ...) (iblasr/BlasrAlignImpl.hpp:1418-1421)
    int Run(const MappingParameters &params, bool wantBlocks = true, int alignType = GBA_ALIGN_GLOBAL)
    {
        const uint32_t n = (uint32_t)cands_.size();
        if (n == 0) return GBA_OK;
        gba_scoring sc = ScoringFromParams(params, !prof_.empty(), alignType);
        int rc = gba_set_scoring(ctx_, &sc);
        if (rc != GBA_OK) return rc;
        if (t_arena_.empty()) t_arena_.push_back('N');
        if (blocks_.empty()) blocks_.push_back(gba_block{0, 0, 0});
        gba_batch b{};
        b.n_pairs = n;
        b.q_arena = (const uint8_t *)q_arena_.data(); b.q_off = q_off_.data(); b.q_len = q_len_.data();
        b.t_arena = (const uint8_t *)t_arena_.data(); b.t_off = t_off_.data(); b.t_len = t_len_.data();
        b.blocks = blocks_.data(); b.blk_off = b_off_.data(); b.n_blk = n_blk_.data();
        b.q_arena_bytes = q_arena_.size(); b.t_arena_bytes = t_arena_.size(); b.n_blocks_total = blocks_.size();
        if (!prof_.empty()) {
            if (prof_.size() != 11 * q_arena_.size()) return GBA_E_INVALID;      // some reads came with a profile, some without
            b.q_profile = prof_.data();
        }
        uint64_t cap = n;
        for (uint32_t l : q_len_) cap += l;
        if (!wantBlocks) cap = 0;
        std::vector<gba_pair_result> pairs(n);
        std::vector<gba_block> out(cap);
        gba_results r{pairs.data(), wantBlocks ? out.data() : nullptr, cap, 0};
        if ((rc = gba_submit(ctx_, &b, &r)) != GBA_OK) return rc;     // returns at once: a caller with more reads
        if ((rc = gba_wait(ctx_)) != GBA_OK) return rc;               // to prepare does that here, before waiting
        for (uint32_t i = 0; i < n; i++) {
            AlignmentCandidate &c = *cands_[i];
            const gba_pair_result &p = pairs[i];
            c.status = p.status;
            if (p.status != GBA_PAIR_OK) continue;
            c.blocks.resize(wantBlocks ? p.n_blk : 0);
            for (uint32_t k = 0; k < p.n_blk && wantBlocks; k++) {
                const gba_block &g = out[p.blk_off + k];
                c.blocks[k] = Block{g.q_pos, g.t_pos, g.length};
            }
            c.nBlocks = p.n_blk;
            c.nMatch = p.n_match; c.nMismatch = p.n_mismatch; c.nIns = p.n_ins; c.nDel = p.n_del;
            const uint64_t cols = (uint64_t)p.n_match + p.n_mismatch + p.n_ins + p.n_del;
            c.pctSimilarity = cols ? (float)(p.n_match * 2.0 / (2.0 * cols) * 100.0) : 0.f;
            c.score = p.score; c.nCells = p.ncells;
            c.tAlignedSeq = t_arena_.substr(t_off_[i], t_len_[i]);
            c.qAlignedSeq = q_arena_.substr(q_off_[i], q_len_[i]);
            c.tAlignedSeqPos += c.tPos; c.qAlignedSeqPos += c.qPos;
            c.tPos = p.t_pos; c.qPos = p.q_pos;
            c.tAlignedSeqLength = t_len_[i]; c.qAlignedSeqLength = q_len_[i];
            c.qStart = c.qAlignedSeqPos + p.q_start; c.qEnd = c.qAlignedSeqPos + p.q_end;
            c.tStart = c.tAlignedSeqPos + p.t_start; c.tEnd = c.tAlignedSeqPos + p.t_end;
            c.haveInterval = true;
        }
        return GBA_OK;
    }

private:
    // consecutive candidates of one read share its bytes in the arena; "the same read" is decided
    // by content (a caller may reuse one string buffer for different reads of equal length)
    uint64_t Intern(const std::string &read)
    {
        if (have_last_ && last_len_ == read.size() && q_arena_.compare(last_base_, last_len_, read) == 0) return last_base_;
        have_last_ = true; last_len_ = read.size(); last_base_ = q_arena_.size();
        q_arena_ += read;
        return last_base_;
    }
    gba_ctx *ctx_;
    std::vector<AlignmentCandidate *> cands_;
    std::string q_arena_, t_arena_;
    std::vector<gba_block> blocks_;
    std::vector<int16_t> prof_;
    std::vector<uint64_t> q_off_, t_off_, b_off_;
    std::vector<uint32_t> q_len_, t_len_, n_blk_;
    bool have_last_ = false; size_t last_len_ = 0; uint64_t last_base_ = 0;
};

// The guide of every candidate of a read in one call: what AlignIntervals does per candidate with
// SDPAlign(alignment->qAlignedSeq, alignment->tAlignedSeq, distScoreFn, sdpTupleSize, params.sdpIns, params.sdpDel,
// params.indelRate*3, *alignment, mappingBuffers, Local, ...) (iblasr/BlasrAlignImpl.hpp:980-987).  The query is the read
// from qAlignedSeqPos on (qAlignedSeqLength bases, the rest of the read when that is 0), the target the candidate's
// window.  On return blocks are relative to the first block and qPos / tPos give its start (:951-964); a candidate
// without a single match of sdpTupleSize bases keeps no blocks; score is minus the chain's score (lower is better).
inline int SDPAlignCandidates(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs,
                              const MappingParameters &params, gba_ctx *ctx)
{
    const uint32_t n = (uint32_t)alignmentPtrs.size();
    if (n == 0) return GBA_OK;
    std::string t_arena;
    std::vector<uint64_t> q_off(n), t_off(n), blk_off(n);
    std::vector<uint32_t> q_len(n), t_len(n), n_blk(n);
    std::vector<int32_t> status(n), score(n);
    for (uint32_t i = 0; i < n; i++) {
        const AlignmentCandidate &c = *alignmentPtrs[i];
        if (c.qAlignedSeqPos > read.size()) throw std::out_of_range("candidate starts behind its read");
        q_off[i] = c.qAlignedSeqPos;
        q_len[i] = c.qAlignedSeqLength ? c.qAlignedSeqLength : (uint32_t)(read.size() - c.qAlignedSeqPos);
        t_off[i] = t_arena.size(); t_len[i] = (uint32_t)c.tAlignedSeq.size();
        t_arena += c.tAlignedSeq;
    }
    if (t_arena.empty()) t_arena.push_back('N');
    gba_batch b{};
    b.n_pairs = n;
    b.q_arena = (const uint8_t *)read.data(); b.q_off = q_off.data(); b.q_len = q_len.data(); b.q_arena_bytes = read.size();
    b.t_arena = (const uint8_t *)t_arena.data(); b.t_off = t_off.data(); b.t_len = t_len.data(); b.t_arena_bytes = t_arena.size();
    gba_sdp_params sp{params.sdpTupleSize, params.sdpIns, params.sdpDel, -params.match, params.sdpMaxFragments};
    std::vector<gba_block> blocks((size_t)n * params.sdpMaxFragments);
    gba_guides g{status.data(), score.data(), n_blk.data(), blk_off.data(), blocks.data(), blocks.size(), 0};
    const int rc = gba_sdp(ctx, &b, &sp, &g);
    if (rc != GBA_OK) return rc;
    for (uint32_t i = 0; i < n; i++) {
        AlignmentCandidate &c = *alignmentPtrs[i];
        c.status = status[i];
        c.blocks.clear();
        c.qPos = c.tPos = 0;
        c.score = -score[i];
        if (status[i] != GBA_PAIR_OK || n_blk[i] == 0) continue;
        const gba_block *src = blocks.data() + blk_off[i];
        c.qPos = src[0].q_pos; c.tPos = src[0].t_pos;
        for (uint32_t k = 0; k < n_blk[i]; k++) c.blocks.push_back(Block{src[k].q_pos - c.qPos, src[k].t_pos - c.tPos, src[k].length});
    }
    return GBA_OK;
}

// An exact match between read and genome that the seeding stage found: an anchor of a candidate (ChainedMatchPos in the
// reference: q, t in read / genome coordinates on the forward strand, l bases).
struct Anchor { uint32_t q, t, l; };

// The -rbao branch of AlignIntervals (iblasr/BlasrAlignImpl.hpp:785-978): the guide of a candidate is its anchors with, between
// two consecutive anchors that leave a gap in BOTH sequences, the chain SDPAlign finds inside that gap (:902-908).  All gaps of
// all candidates of the read go through one gba_sdp call.  As in the reference an anchor is taken into the guide together with
// the gap behind it: an anchor followed by no gap in the query or none in the target is left out (:877); the last anchor
// always goes in (:938-949).  Blocks end up relative to the first one, qPos / tPos at its start (:951-964).  anchors[i]: those
// of alignmentPtrs[i], in order (the strand flip of :828-851 is the caller's).  SDPAlign's Global flag for these calls is
// libcpp's and unpinned: the chain is the best chain gba_sdp returns.
inline int GuidesBetweenAnchors(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs,
                                const std::vector<std::vector<Anchor>> &anchors, const MappingParameters &params, gba_ctx *ctx)
{
    struct Job { size_t c, k; uint32_t q0, t0, qGap, tGap; };
    std::vector<Job> jobs;
    std::string t_arena;
    std::vector<uint64_t> t_base;
    for (size_t ci = 0; ci < alignmentPtrs.size(); ci++) {
        const AlignmentCandidate &c = *alignmentPtrs[ci];
        t_base.push_back(t_arena.size());
        t_arena += c.tAlignedSeq;
        const std::vector<Anchor> &a = anchors[ci];
        for (size_t k = 0; k + 1 < a.size(); k++) {
            const int64_t tGap = (int64_t)a[k + 1].t - ((int64_t)a[k].t + a[k].l), qGap = (int64_t)a[k + 1].q - ((int64_t)a[k].q + a[k].l);
            if (tGap > 0 && qGap > 0)
                jobs.push_back(Job{ci, k, a[k].q + a[k].l - c.qAlignedSeqPos, a[k].t + a[k].l - c.tAlignedSeqPos, (uint32_t)qGap, (uint32_t)tGap});
        }
    }
    std::vector<int32_t> status(jobs.size()), score(jobs.size());
    std::vector<uint32_t> n_blk(jobs.size());
    std::vector<uint64_t> blk_off(jobs.size());
    std::vector<gba_block> blocks;
    if (!jobs.empty()) {
        const uint32_t n = (uint32_t)jobs.size();
        std::vector<uint64_t> q_off(n), t_off(n);
        std::vector<uint32_t> q_len(n), t_len(n);
        uint64_t cap = n;
        for (uint32_t x = 0; x < n; x++) {
            const Job &j = jobs[x];
            q_off[x] = (uint64_t)alignmentPtrs[j.c]->qAlignedSeqPos + j.q0; q_len[x] = j.qGap;
            t_off[x] = t_base[j.c] + j.t0; t_len[x] = j.tGap;
            cap += std::min(j.qGap, j.tGap);
        }
        if (t_arena.empty()) t_arena.push_back('N');
        gba_batch b{};
        b.n_pairs = n;
        b.q_arena = (const uint8_t *)read.data(); b.q_off = q_off.data(); b.q_len = q_len.data(); b.q_arena_bytes = read.size();
        b.t_arena = (const uint8_t *)t_arena.data(); b.t_off = t_off.data(); b.t_len = t_len.data(); b.t_arena_bytes = t_arena.size();
        gba_sdp_params sp{params.sdpTupleSize, params.sdpIns, params.sdpDel, -params.match, params.sdpMaxFragments};
        blocks.resize(cap);
        gba_guides g{status.data(), score.data(), n_blk.data(), blk_off.data(), blocks.data(), blocks.size(), 0};
        const int rc = gba_sdp(ctx, &b, &sp, &g);
        if (rc != GBA_OK) return rc;
    }
    size_t x = 0;
    for (size_t ci = 0; ci < alignmentPtrs.size(); ci++) {
        AlignmentCandidate &c = *alignmentPtrs[ci];
        const std::vector<Anchor> &a = anchors[ci];
        std::vector<Block> out;
        for (size_t k = 0; k + 1 < a.size(); k++) {
            if (x < jobs.size() && jobs[x].c == ci && jobs[x].k == k) {
                out.push_back(Block{a[k].q - c.qAlignedSeqPos, a[k].t - c.tAlignedSeqPos, a[k].l});
                if (status[x] == GBA_PAIR_OK)
                    for (uint32_t i = 0; i < n_blk[x]; i++) {
                        const gba_block &gb = blocks[blk_off[x] + i];
                        out.push_back(Block{gb.q_pos + jobs[x].q0, gb.t_pos + jobs[x].t0, gb.length});      // :921-923
                    }
                x++;
            }
        }
        if (!a.empty()) out.push_back(Block{a.back().q - c.qAlignedSeqPos, a.back().t - c.tAlignedSeqPos, a.back().l});
        c.qPos = c.tPos = 0;
        if (!out.empty()) {
            c.qPos = out[0].qPos; c.tPos = out[0].tPos;
            for (Block &bk : out) { bk.qPos -= c.qPos; bk.tPos -= c.tPos; }
        }
        c.blocks.swap(out);
        c.status = c.blocks.empty() ? GBA_PAIR_EMPTY_GUIDE : GBA_PAIR_OK;
    }
    return GBA_OK;
}

// SDPAlign and RefineAlignment for every candidate of a read in ONE submit (gba_set_guide_producer): the batch
// carries the read and the candidates' windows, the guides are made on the device and refined where they lie; no
// guide travels.  The candidates end up as SDPAlignCandidates followed by RefineAlignments leaves them (blocks,
// score, counts, tAlignedSeq / qAlignedSeq cut to the refined stretch, positions); a candidate without a guide
// keeps none and gets the producer's reason as its status.  wantBlocks false: interval output.
inline int AlignAndRefineCandidates(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs,
                                    const MappingParameters &params, gba_ctx *ctx, bool wantBlocks = true)
{
    const uint32_t n = (uint32_t)alignmentPtrs.size();
    if (n == 0) return GBA_OK;
    std::string t_arena;
    std::vector<uint64_t> q_off(n), t_off(n);
    std::vector<uint32_t> q_len(n), t_len(n);
    uint64_t cap = n;
    for (uint32_t i = 0; i < n; i++) {
        const AlignmentCandidate &c = *alignmentPtrs[i];
        if (c.qAlignedSeqPos > read.size()) throw std::out_of_range("candidate starts behind its read");
        q_off[i] = c.qAlignedSeqPos;
        q_len[i] = c.qAlignedSeqLength ? c.qAlignedSeqLength : (uint32_t)(read.size() - c.qAlignedSeqPos);
        t_off[i] = t_arena.size(); t_len[i] = (uint32_t)c.tAlignedSeq.size();
        t_arena += c.tAlignedSeq;
        cap += q_len[i];
    }
    if (t_arena.empty()) t_arena.push_back('N');
    gba_scoring sc = ScoringFromParams(params);
    int rc = gba_set_scoring(ctx, &sc);
    if (rc != GBA_OK) return rc;
    gba_sdp_params sp{params.sdpTupleSize, params.sdpIns, params.sdpDel, -params.match, params.sdpMaxFragments};
    if ((rc = gba_set_guide_producer(ctx, &sp)) != GBA_OK) return rc;
    gba_batch b{};
    b.n_pairs = n;
    b.q_arena = (const uint8_t *)read.data(); b.q_off = q_off.data(); b.q_len = q_len.data(); b.q_arena_bytes = read.size();
    b.t_arena = (const uint8_t *)t_arena.data(); b.t_off = t_off.data(); b.t_len = t_len.data(); b.t_arena_bytes = t_arena.size();
    if (!wantBlocks) cap = 0;
    std::vector<gba_pair_result> pairs(n);
    std::vector<gba_block> out(cap);
    gba_results r{pairs.data(), wantBlocks ? out.data() : nullptr, cap, 0};
    if ((rc = gba_submit(ctx, &b, &r)) == GBA_OK) rc = gba_wait(ctx);
    (void)gba_set_guide_producer(ctx, nullptr);
    if (rc != GBA_OK) return rc;
    for (uint32_t i = 0; i < n; i++) {
        AlignmentCandidate &c = *alignmentPtrs[i];
        const gba_pair_result &p = pairs[i];
        c.status = p.status;
        c.blocks.clear();
        c.nBlocks = 0;
        c.qPos = c.tPos = 0;
        if (p.status != GBA_PAIR_OK) continue;
        for (uint32_t k = 0; k < p.n_blk && wantBlocks; k++) {
            const gba_block &g = out[p.blk_off + k];
            c.blocks.push_back(Block{g.q_pos, g.t_pos, g.length});
        }
        c.nBlocks = p.n_blk;
        c.nMatch = p.n_match; c.nMismatch = p.n_mismatch; c.nIns = p.n_ins; c.nDel = p.n_del;
        const uint64_t cols = (uint64_t)p.n_match + p.n_mismatch + p.n_ins + p.n_del;
        c.pctSimilarity = cols ? (float)(p.n_match * 2.0 / (2.0 * cols) * 100.0) : 0.f;
        c.score = p.score; c.nCells = p.ncells;
        // the refined stretch starts at the chain's first anchor: that becomes the start of the aligned sequences
        c.qStart = c.qAlignedSeqPos + p.q_start; c.qEnd = c.qAlignedSeqPos + p.q_end;
        c.tStart = c.tAlignedSeqPos + p.t_start; c.tEnd = c.tAlignedSeqPos + p.t_end;
        c.haveInterval = true;
        c.qAlignedSeq = read.substr(q_off[i] + p.q_pos, p.rows);
        c.tAlignedSeq = c.tAlignedSeq.substr(p.t_pos, p.cols);
        c.qAlignedSeqPos += p.q_pos; c.tAlignedSeqPos += p.t_pos;
        c.qAlignedSeqLength = p.rows; c.tAlignedSeqLength = p.cols;
    }
    return GBA_OK;
}

// The GuidedAlign overload that takes only the two sequences: it chains matches of sdpTupleSize bases itself and runs the
// banded DP around that chain.  AlignSubreadToAlignmentTarget aligns every subread of a ZMW to the window its CCS (or
// longest subread) mapped to with it: GuidedAlign(subread, alignedRefSequence, idsScoreFn, 12, params.sdpIns, params.sdpDel,
// params.indelRate, mappingBuffers, exploded, Local, computeProbIsFalse, 6) (iblasr/BlasrAlignImpl.hpp:1418-1421), then
// ComputeAlignmentStats with distScoreFn2 (:1448-1450).  Here: SDPAlignCandidates with the short tuple, then the guided DP
// with band `bandSize` and the Local end rule over the chain's span (profile: the IDS scorer's costs for the read, or null).
inline int GuidedAlignFromSequences(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs, const MappingParameters &params,
                                    gba_ctx *ctx, int bandSize = 12, int sdpTupleSize = 6, const std::vector<int16_t> *profile = nullptr)
{
    MappingParameters p = params;
    p.sdpTupleSize = sdpTupleSize; p.guidedAlignBandSize = bandSize; p.affineAlign = false;
    int rc = SDPAlignCandidates(read, alignmentPtrs, p, ctx);
    if (rc != GBA_OK) return rc;
    RefineBatch batch(ctx);
    for (AlignmentCandidate *c : alignmentPtrs) batch.Add(read, c, profile);
    return batch.Run(p, true, GBA_ALIGN_LOCAL);
}

// The -global branch of RefineAlignment (:648-675) for all candidates of one read: the whole subread
// against the whole tAlignedSeq in a diagonal band of half-width ComputeDrift(candidate), RemoveEndGaps,
// rescoring with distScoreFn2; blocks, tPos, qPos (+ SubreadStart) and score are copied back and
// nothing else of the candidate changes.  One submit per distinct band half-width.
inline int RefineGlobal(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs,
                        const MappingParameters &params, gba_ctx *ctx, uint32_t subreadStart = 0, uint32_t subreadLength = UINT32_MAX)
{
    if (subreadStart > read.size()) return GBA_E_INVALID;
    if (subreadLength > read.size() - subreadStart) subreadLength = (uint32_t)(read.size() - subreadStart);
    std::vector<int> ks;
    for (AlignmentCandidate *c : alignmentPtrs) ks.push_back(ComputeDrift(c->blocks));
    std::vector<int> distinct(ks);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    for (int k : distinct) {
        std::vector<AlignmentCandidate *> cs;
        for (size_t i = 0; i < ks.size(); i++) if (ks[i] == k) cs.push_back(alignmentPtrs[i]);
        const uint32_t n = (uint32_t)cs.size();
        gba_scoring sc = ScoringGlobal(params, k);
        int rc = gba_set_scoring(ctx, &sc);
        if (rc != GBA_OK) return rc;
        std::string t_arena;
        std::vector<uint64_t> q_off(n, subreadStart), t_off(n), b_off(n, 0);
        std::vector<uint32_t> q_len(n, subreadLength), t_len(n), n_blk(n, 0);
        uint64_t cap = n;
        for (uint32_t i = 0; i < n; i++) {
            t_off[i] = t_arena.size(); t_len[i] = (uint32_t)cs[i]->tAlignedSeq.size();
            t_arena += cs[i]->tAlignedSeq;
            cap += std::min<uint64_t>(subreadLength, t_len[i]);
        }
        if (t_arena.empty()) t_arena.push_back('N');
        const gba_block none{0, 0, 0};
        gba_batch b{};
        b.n_pairs = n;
        b.q_arena = (const uint8_t *)read.data(); b.q_off = q_off.data(); b.q_len = q_len.data();
        b.t_arena = (const uint8_t *)t_arena.data(); b.t_off = t_off.data(); b.t_len = t_len.data();
        b.blocks = &none; b.blk_off = b_off.data(); b.n_blk = n_blk.data();
        b.q_arena_bytes = read.size(); b.t_arena_bytes = t_arena.size(); b.n_blocks_total = 1;
        std::vector<gba_pair_result> pairs(n);
        std::vector<gba_block> out(cap);
        gba_results r{pairs.data(), out.data(), cap, 0};
        if ((rc = gba_submit(ctx, &b, &r)) != GBA_OK) return rc;
        if ((rc = gba_wait(ctx)) != GBA_OK) return rc;
        for (uint32_t i = 0; i < n; i++) {
            AlignmentCandidate &c = *cs[i];
            const gba_pair_result &p = pairs[i];
            c.status = p.status;
            if (p.status != GBA_PAIR_OK) continue;
            // RemoveEndGaps (:662): the alignment starts at its first block
            const uint32_t dq = p.n_blk ? out[p.blk_off].q_pos : 0, dt = p.n_blk ? out[p.blk_off].t_pos : 0;
            c.blocks.resize(p.n_blk);
            for (uint32_t x = 0; x < p.n_blk; x++) {
                const gba_block &g = out[p.blk_off + x];
                c.blocks[x] = Block{g.q_pos - dq, g.t_pos - dt, g.length};
            }
            c.tPos = p.t_pos + dt;
            c.qPos = p.q_pos + dq + subreadStart;              // :672
            c.score = p.score;                                 // :673
        }
    }
    return GBA_OK;
}

// ---- the fixed-band aligners beside the guided path (gba_kband) ----------------------------------
// gba_kband_params from MappingParameters: SMRTDistanceMatrix, the two gap costs given, rescoring with
// distScoreFn2 (indel, indel); the affine costs are the ones PairwiseLocalAlign passes to AffineKBandAlign
// (/root/reference/iblasr/BlasrAlignImpl.hpp:1259-1262).
inline gba_kband_params KBandParams(const MappingParameters &p, int alignType, int gapModel, int ins, int del, int maxDrops = 10)
{
    gba_kband_params k{};
    gba_scoring m{};
    MatrixFromParams(p, m);
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) k.sub[a][b] = m.sub[a][b];
    k.ins = ins; k.del = del; k.align_type = alignType; k.gap_model = gapModel;
    k.hp_ins_open = p.indel + 2; k.hp_ins_extend = p.indel - 3; k.ins_open = p.indel + 2; k.ins_extend = p.indel - 1;
    k.hp_rule = 0; k.tie_break = p.tieBreak; k.max_drops = maxDrops;
    k.stats_ins = p.indel; k.stats_del = p.indel;
    return k;
}

// Any number of (query, target, k) triples through one gba_kband call.
class KBandBatch {
public:
    explicit KBandBatch(gba_ctx *ctx) : ctx_(ctx) {}
    // profile: 11 costs per query base (see gba.h) or nullptr; all pairs of a batch must agree on having one
    void Add(const char *q, size_t ql, const char *t, size_t tl, int k, bool reverse = false, const int16_t *profile = nullptr)
    {
        q_off_.push_back(q_arena_.size()); q_len_.push_back((uint32_t)ql); q_arena_.append(q, ql);
        t_off_.push_back(t_arena_.size()); t_len_.push_back((uint32_t)tl); t_arena_.append(t, tl);
        k_.push_back(k); rev_.push_back(reverse ? 1 : 0);
        if (profile) prof_.insert(prof_.end(), profile, profile + 11 * ql);
    }
    size_t size() const { return k_.size(); }
    int Run(const gba_kband_params &P)
    {
        const uint32_t n = (uint32_t)k_.size();
        pairs.assign(n, gba_pair_result{});
        uint64_t cap = n;
        for (uint32_t i = 0; i < n; i++) cap += std::min(q_len_[i], t_len_[i]);
        blocks.assign(cap, gba_block{});
        if (n == 0) return GBA_OK;
        if (!prof_.empty() && prof_.size() != q_arena_.size() * 11) return GBA_E_INVALID;
        if (q_arena_.empty()) q_arena_.push_back('N');
        if (t_arena_.empty()) t_arena_.push_back('N');
        gba_kband_batch b{};
        b.n_pairs = n;
        b.q_arena = (const uint8_t *)q_arena_.data(); b.q_off = q_off_.data(); b.q_len = q_len_.data();
        b.t_arena = (const uint8_t *)t_arena_.data(); b.t_off = t_off_.data(); b.t_len = t_len_.data();
        b.q_arena_bytes = q_arena_.size(); b.t_arena_bytes = t_arena_.size();
        b.k = k_.data(); b.reverse = rev_.data(); b.profile = prof_.empty() ? nullptr : prof_.data();
        gba_results r{pairs.data(), blocks.data(), cap, 0};
        return gba_kband(ctx_, &b, &P, &r);
    }
    std::vector<gba_pair_result> pairs;
    std::vector<gba_block> blocks;

private:
    gba_ctx *ctx_;
    std::string q_arena_, t_arena_;
    std::vector<uint64_t> q_off_, t_off_;
    std::vector<uint32_t> q_len_, t_len_;
    std::vector<int32_t> k_;
    std::vector<uint8_t> rev_;
    std::vector<int16_t> prof_;
};

inline void StatsInto(AlignmentCandidate &c, const gba_pair_result &p)
{
    c.nMatch = p.n_match; c.nMismatch = p.n_mismatch; c.nIns = p.n_ins; c.nDel = p.n_del;
    const uint32_t cols = c.nMatch + c.nMismatch + c.nIns + c.nDel;
    c.pctSimilarity = cols ? 100.0f * c.nMatch / cols : 0.0f;
    c.score = p.score; c.nCells = p.ncells;
}

// The branch of RefineAlignment without guided alignment (-noUseGuidedAlign, :775-902) for reads without
// usable quality values: the sequences are cut to the span of the SDP blocks, k = params.bandSize or
// |drift| * 1.5, at least 10 (:848-861), positions move into the aligned-sequence offsets (:863-887), and
// PairwiseLocalAlign runs AffineKBandAlign(..., indel+2, indel-3, indel+2, indel-1, indel, k * 1.2, ..., Global)
// (/root/reference/iblasr/BlasrAlignImpl.hpp:1259-1267), then ComputeAlignmentStats with distScoreFn2 (:1304).
inline int RefineLegacy(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs,
                        const MappingParameters &params, gba_ctx *ctx)
{
    KBandBatch batch(ctx);
    std::vector<AlignmentCandidate *> todo;
    for (AlignmentCandidate *c : alignmentPtrs) {
        if (c->blocks.empty()) { c->score = 0; continue; }                       // :800-803
        const Block &first = c->blocks.front(), &last = c->blocks.back();
        const uint32_t ql = last.qPos + last.length, tl = last.tPos + last.length - first.tPos;
        const uint32_t qs = c->qPos + c->qAlignedSeqPos;
        int k = params.bandSize ? params.bandSize : (int)(std::abs(ComputeDrift(c->blocks)) * 1.5);
        if (k < 10) k = 10;
        if (qs + (uint64_t)ql > read.size() || c->tPos + (uint64_t)tl > c->tAlignedSeq.size()) { c->status = GBA_PAIR_BAD_GUIDE; continue; }
        c->tAlignedSeq = c->tAlignedSeq.substr(c->tPos, tl);
        c->qAlignedSeq = read.substr(qs, ql);
        c->tAlignedSeqPos += c->tPos; c->tAlignedSeqLength = tl;
        c->qAlignedSeqPos += c->qPos; c->qAlignedSeqLength = last.qPos + last.length - first.qPos;
        c->tPos = c->qPos = 0;
        c->blocks.clear();
        batch.Add(c->qAlignedSeq.data(), ql, c->tAlignedSeq.data(), tl, (int)(k * 1.2));
        todo.push_back(c);
    }
    const int rc = batch.Run(KBandParams(params, GBA_KB_GLOBAL, GBA_KB_AFFINE_HP, params.indel, params.indel));
    if (rc != GBA_OK) return rc;
    for (size_t i = 0; i < todo.size(); i++) {
        AlignmentCandidate &c = *todo[i];
        const gba_pair_result &p = batch.pairs[i];
        c.status = p.status;
        if (p.status != GBA_PAIR_OK) continue;
        c.blocks.resize(p.n_blk);
        for (uint32_t x = 0; x < p.n_blk; x++) { const gba_block &g = batch.blocks[p.blk_off + x]; c.blocks[x] = Block{g.q_pos, g.t_pos, g.length}; }
        c.nBlocks = p.n_blk;
        c.qPos = p.q_pos; c.tPos = p.t_pos;
        StatsInto(c, p);
        // read- / window-absolute, as RefineBatch::Run reports them
        c.qStart = c.qAlignedSeqPos + p.q_start; c.qEnd = c.qAlignedSeqPos + p.q_end;
        c.tStart = c.tAlignedSeqPos + p.t_start; c.tEnd = c.tAlignedSeqPos + p.t_end; c.haveInterval = true;
    }
    return GBA_OK;
}

// The -extend step of AlignIntervals (/root/reference/iblasr/BlasrAlignImpl.hpp:1011-1201) for the candidates
// of one read on the FORWARD strand of `genome` (a reverse-strand candidate wants the reverse complement
// of the genome here, whose positions its tAlignedSeqPos counts along): behind the last block up to 500
// bases of read and genome go through ExtendAlignmentForward (:1086-1099), in front of the first block up
// to 500 through ExtendAlignmentReverse (:1144-1155), both with AlignIntervals' distScoreFn (:576),
// extendBandSize and maxExtendDropoff; an extension with a negative score is spliced on (:1102-1120,
// :1158-1196); counts and score of the pieces add up (:1214 with linear gap costs).  *extended: ends extended.
inline int ExtendAlignments(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs, const std::string &genome,
                            const MappingParameters &params, gba_ctx *ctx, int *extended = nullptr,
                            int extendBandSize = 10, int maxExtendDropoff = 10, uint64_t contigStart = 0, uint64_t contigEnd = UINT64_MAX)
{
    const uint64_t maximumExtendLength = 500;                                    // :1039
    if (contigEnd > genome.size()) contigEnd = genome.size();
    struct Job { AlignmentCandidate *c; int side; uint64_t q0, t0, rl, gl; };
    std::vector<Job> jobs;
    KBandBatch batch(ctx);
    // the prelude of :1020-1037, for every candidate: the aligned sequences are cut to the alignment's own boundaries
    // (from the first block's start over QEnd() / TEnd() bases) and qPos / tPos become 0, so that what is spliced on
    // below lands directly behind the last aligned base and in front of the first
    for (AlignmentCandidate *c : alignmentPtrs) {
        const uint32_t qend = c->blocks.empty() ? 0 : c->blocks.back().qPos + c->blocks.back().length;
        const uint32_t tend = c->blocks.empty() ? 0 : c->blocks.back().tPos + c->blocks.back().length;
        if ((uint64_t)c->qAlignedSeqPos + c->qPos + qend > read.size() || (uint64_t)c->tPos + tend > c->tAlignedSeq.size())
            throw std::out_of_range("candidate alignment exceeds its sequences");
        c->qAlignedSeqPos += c->qPos; c->qAlignedSeqLength = qend; c->qPos = 0;
        c->qAlignedSeq = read.substr(c->qAlignedSeqPos, qend);
        c->tAlignedSeqPos += c->tPos; c->tAlignedSeq = c->tAlignedSeq.substr(c->tPos, tend); c->tAlignedSeqLength = tend; c->tPos = 0;
    }
    for (AlignmentCandidate *c : alignmentPtrs) {
        if (c->blocks.empty()) continue;
        const Block &last = c->blocks.back();
        const uint64_t lastQ = (uint64_t)last.qPos + last.length + c->qAlignedSeqPos;
        const uint64_t lastT = (uint64_t)last.tPos + last.length + c->tAlignedSeqPos;
        if (lastQ < read.size() && lastT < contigEnd) {
            const uint64_t rl = std::min<uint64_t>(read.size() - lastQ, maximumExtendLength), gl = std::min<uint64_t>(contigEnd - lastT, maximumExtendLength);
            jobs.push_back(Job{c, 0, lastQ, lastT, rl, gl});
            batch.Add(read.data() + lastQ, rl, genome.data() + lastT, gl, extendBandSize, false);
        }
        const uint64_t firstQ = c->qAlignedSeqPos, firstT = c->tAlignedSeqPos;
        if (firstQ > 0 && firstT > contigStart) {
            const uint64_t rl = std::min<uint64_t>(firstQ, maximumExtendLength), gl = std::min<uint64_t>(firstT - contigStart, maximumExtendLength);
            jobs.push_back(Job{c, 1, firstQ - rl, firstT - gl, rl, gl});
            batch.Add(read.data() + firstQ - rl, rl, genome.data() + firstT - gl, gl, extendBandSize, true);
        }
    }
    int done = 0;
    if (extended) *extended = 0;
    if (jobs.empty()) return GBA_OK;
    const int rc = batch.Run(KBandParams(params, GBA_KB_EXTEND, GBA_KB_LINEAR, params.insertion, params.deletion, maxExtendDropoff));
    if (rc != GBA_OK) return rc;
    for (int order = 0; order < 2; order++) {
        for (size_t x = 0; x < jobs.size(); x++) {
            const Job &j = jobs[x];
            const gba_pair_result &p = batch.pairs[x];
            if (j.side != order || p.status != GBA_PAIR_OK || p.dp_score >= 0 || p.n_blk == 0) continue;
            AlignmentCandidate &c = *j.c;
            const gba_block *eb = &batch.blocks[p.blk_off];
            const uint64_t oq = (uint64_t)c.qPos + c.qAlignedSeqPos, ot = (uint64_t)c.tPos + c.tAlignedSeqPos;
            const uint64_t qend = p.q_pos + eb[p.n_blk - 1].q_pos + eb[p.n_blk - 1].length, tend = p.t_pos + eb[p.n_blk - 1].t_pos + eb[p.n_blk - 1].length;
            const uint32_t qlen = (uint32_t)c.qAlignedSeq.size();
            if (j.side == 0) {                      // :1111-1119
                for (uint32_t i = 0; i < p.n_blk; i++)
                    c.blocks.push_back(Block{(uint32_t)(j.q0 + p.q_pos + eb[i].q_pos - oq), (uint32_t)(j.t0 + p.t_pos + eb[i].t_pos - ot), eb[i].length});
                c.tAlignedSeq.append(genome, j.t0, tend);
                c.qAlignedSeqLength = qlen + (uint32_t)qend;
                c.qAlignedSeq = read.substr(c.qAlignedSeqPos, c.qAlignedSeqLength);
            } else {
                std::vector<Block> nb;
                for (uint32_t i = 0; i < p.n_blk; i++) nb.push_back(Block{eb[i].q_pos, eb[i].t_pos, eb[i].length});
                const uint64_t nq = j.q0 + p.q_pos, nt = j.t0 + p.t_pos;                 // new origin of the blocks
                for (const Block &b : c.blocks) nb.push_back(Block{(uint32_t)(b.qPos + oq - nq), (uint32_t)(b.tPos + ot - nt), b.length});
                c.blocks.swap(nb);
                // :1170: the WHOLE prefix goes in front (Append(tAlignedSeq, length - tPos) with tPos = 0 after the prelude)
                c.tAlignedSeq = genome.substr(j.t0, j.gl) + c.tAlignedSeq;
                c.tAlignedSeqPos = (uint32_t)j.t0; c.qAlignedSeqPos = (uint32_t)j.q0;
                c.tPos = p.t_pos; c.qPos = p.q_pos;
                c.qAlignedSeqLength = (uint32_t)j.rl + qlen;
                c.qAlignedSeq = read.substr(c.qAlignedSeqPos, c.qAlignedSeqLength);
            }
            c.tAlignedSeqLength = (uint32_t)c.tAlignedSeq.size();
            c.nBlocks = (uint32_t)c.blocks.size();
            c.nMatch += p.n_match; c.nMismatch += p.n_mismatch; c.nIns += p.n_ins; c.nDel += p.n_del;
            c.score += p.score;
            const uint32_t cols = c.nMatch + c.nMismatch + c.nIns + c.nDel;
            c.pctSimilarity = cols ? 100.0f * c.nMatch / cols : 0.0f;
            // read- / genome-absolute, as RefineBatch::Run reports them
            c.qStart = c.qAlignedSeqPos + c.qPos + c.blocks.front().qPos; c.qEnd = c.qAlignedSeqPos + c.qPos + c.blocks.back().qPos + c.blocks.back().length;
            c.tStart = c.tAlignedSeqPos + c.tPos + c.blocks.front().tPos; c.tEnd = c.tAlignedSeqPos + c.tPos + c.blocks.back().tPos + c.blocks.back().length;
            c.haveInterval = true;
            done++;
        }
    }
    if (extended) *extended = done;
    return GBA_OK;
}

// RefineAlignments for ONE read (:905-923): all its candidates in one batch, then the re-sort.
inline int RefineAlignments(const std::string &read, std::vector<AlignmentCandidate *> &alignmentPtrs,
                            const MappingParameters &params, gba_ctx *ctx, bool wantBlocks = true)
{
    if (!params.refineAlignments) return GBA_OK;
    int rc;
    if (params.doGlobalAlignment) rc = RefineGlobal(read, alignmentPtrs, params, ctx);
    else if (!params.useGuidedAlign) rc = RefineLegacy(read, alignmentPtrs, params, ctx);      // legacy k-band branch (:775-902)
    else {
        RefineBatch batch(ctx);
        for (AlignmentCandidate *c : alignmentPtrs) batch.Add(read, c);
        rc = batch.Run(params, wantBlocks);
    }
    if (rc == GBA_OK && params.sortRefinedAlignments)
        std::stable_sort(alignmentPtrs.begin(), alignmentPtrs.end(),
                         [](const AlignmentCandidate *a, const AlignmentCandidate *b) { return a->score < b->score; });
    return rc;
}

// Among the first nCandidates candidates, drop from the first one that has no blocks or scores above maxScore to the
// end (:485-509), as MapRead does right after refining (iblasr/BlasrAlignImpl.hpp:381).  The vector holds borrowed
// pointers here: nothing is deleted.
inline size_t RemoveLowQualityAlignments(std::vector<AlignmentCandidate *> &alignmentPtrs, const MappingParameters &params)
{
    const size_t lim = std::min<size_t>((size_t)params.nCandidates, alignmentPtrs.size());
    for (size_t i = 0; i < lim; i++) {
        const AlignmentCandidate &c = *alignmentPtrs[i];
        const bool empty = c.blocks.empty() && c.nBlocks == 0;
        if (empty || c.score > params.maxScore) { alignmentPtrs.resize(i); break; }
    }
    return alignmentPtrs.size();
}

// GenomicTBegin() / GenomicTEnd() (libcpp's Alignment, not in the reference tree: UNPINNED): the stretch of the target
// strand between the first aligned base and the end of the last block, tAlignedSeqPos + tPos (+ TEnd()).
inline void GenomicTInterval(const AlignmentCandidate &c, int64_t &begin, int64_t &end)
{
    begin = end = (int64_t)c.tAlignedSeqPos + c.tPos;
    if (!c.blocks.empty()) end = begin + c.blocks.back().tPos + c.blocks.back().length;
    else if (c.haveInterval && c.nBlocks) { begin = c.tStart; end = c.tEnd; }
}

// RemoveOverlappingAlignments (:523-605; called at iblasr/BlasrAlignImpl.hpp:382): of two candidates on the same contig
// whose target stretches contain one another, the contained one goes when its score is no better; a candidate below
// minPctSimilarity is not used as the outer one; the survivors keep their order.  Borrowed pointers: nothing is deleted.
inline size_t RemoveOverlappingAlignments(std::vector<AlignmentCandidate *> &alignmentPtrs, const MappingParameters &params)
{
    const size_t n = alignmentPtrs.size();
    std::vector<unsigned char> contained(n, 0);
    for (size_t i = 0; i + 1 < n; i++) {
        const AlignmentCandidate &a = *alignmentPtrs[i];
        if (a.pctSimilarity < params.minPctSimilarity) continue;
        int64_t ab, ae;
        GenomicTInterval(a, ab, ae);
        for (size_t j = i + 1; j < n; j++) {
            const AlignmentCandidate &b = *alignmentPtrs[j];
            if (contained[j] || a.tIndex != b.tIndex) continue;
            int64_t bb, be;
            GenomicTInterval(b, bb, be);
            if (ab <= bb && ae >= be) { if (a.score <= b.score) contained[j] = 1; }
            else if (bb <= ab && be >= ae) { if (b.score <= a.score) contained[i] = 1; }
        }
    }
    size_t keep = 0;
    for (size_t i = 0; i < n; i++) if (!contained[i]) alignmentPtrs[keep++] = alignmentPtrs[i];
    alignmentPtrs.resize(keep);
    return keep;
}

// ---- mapping qualities of one read's candidates: StoreMapQVs (:108-309) and its helpers.
// What the device contributes is rawProbScore[i] = ComputeAlignmentScore of candidate i's path with
// the probability scorer (SMRTLogProbMatrix, insertion / deletion / affine costs, :117-123): one of the
// extra rescoring scorers of gba_scoring, returned in gba_pair_result.extra_score.  GetQInterval,
// LogSumOfTwo and Phred are libcpp functions that are not in the reference tree (UNPINNED restatements).
constexpr int kMaxPhredScore = 254, kMapQVEndAlignWiggle = 5;       // iblasr/BlasrHeaders.h:19-20

inline void GetQInterval(const AlignmentCandidate &c, int &start, int &end)
{
    if (c.blocks.empty() && c.haveInterval) { start = (int)c.qStart; end = (int)c.qEnd; return; }
    start = end = (int)(c.qAlignedSeqPos + c.qPos);
    if (c.blocks.empty()) return;
    end = start + (int)(c.blocks.back().qPos + c.blocks.back().length);
    start += (int)c.blocks.front().qPos;
}
inline double LogSumOfTwo(double a, double b) { const double hi = a > b ? a : b, lo = a > b ? b : a; return hi + std::log10(1.0 + std::pow(10.0, lo - hi)); }
inline int Phred(double p) { return (int)(-10.0 * std::log10(p)); }

inline bool AlignmentsOverlap(const AlignmentCandidate &a, const AlignmentCandidate &b, float minPercentOverlap)      // :369-409
{
    int a0, a1, b0, b1, ovp = 0;
    GetQInterval(a, a0, a1); GetQInterval(b, b0, b1);
    if (a0 <= b0 && a1 >= b1) return true;
    if (b0 <= a0 && b1 >= a1) return true;
    if (a1 >= b0 && a1 <= b1) ovp = a1 - b0;
    else if (a0 >= b0 && a0 <= b1) ovp = b1 - a0;
    float pct = 0;
    if (a1 - a0 > 0 && b1 - b0 > 0) pct = std::max(float(ovp) / float(a1 - a0), float(ovp) / float(b1 - b0));
    return pct > minPercentOverlap;
}

inline void StoreMapQVs(std::vector<AlignmentCandidate *> &alignmentPtrs, const std::vector<int> &rawProbScore,
                        float minFractionToBeConsideredOverlapping = 0.75f, int nCandidates = 10,
                        bool scaleMapQVByNumSignificantClusters = false)
{
    const size_t n = alignmentPtrs.size();
    for (size_t i = 0; i < n; i++) alignmentPtrs[i]->probScore = (float)(-rawProbScore[i] / 10.0);
    std::vector<std::vector<int>> partitions;                        // PartitionOverlappingAlignments, :411-444
    for (size_t i = 0; i < n; i++) {
        bool found = false;
        for (auto &part : partitions) {
            for (int j : part) {
                int i0, i1, j0, j1;
                GetQInterval(*alignmentPtrs[i], i0, i1); GetQInterval(*alignmentPtrs[j], j0, j1);
                if (AlignmentsOverlap(*alignmentPtrs[i], *alignmentPtrs[j], minFractionToBeConsideredOverlapping) || (i0 <= j0 && i1 > j1)) {
                    part.push_back((int)i); found = true; break;
                }
            }
            if (found) break;
        }
        if (!found) partitions.push_back(std::vector<int>{(int)i});
    }
    for (auto &part : partitions) {
        int begin, end;
        GetQInterval(*alignmentPtrs[part[0]], begin, end);
        for (size_t k = 1; k < part.size(); k++) {
            int s, e; GetQInterval(*alignmentPtrs[part[k]], s, e);
            if (e - s > end - begin) { begin = s; end = e; }
        }
        for (int i : part) {                                         // :214-233, SumMismatches without quality values :362-365
            int s, e; GetQInterval(*alignmentPtrs[i], s, e);
            int mm = 0;
            if (s - begin > kMapQVEndAlignWiggle || end - e > kMapQVEndAlignWiggle) mm = 15 * ((s - begin) + (end - e));
            alignmentPtrs[i]->probScore = (float)(alignmentPtrs[i]->probScore + -(mm) * 0.5);
        }
        float denom = alignmentPtrs[part[0]]->probScore;             // `float mapQVDenominator`, :238
        for (size_t k = 1; k < part.size(); k++) denom = (float)LogSumOfTwo(denom, alignmentPtrs[part[k]]->probScore);
        for (int i : part) {
            AlignmentCandidate &c = *alignmentPtrs[i];
            if (part.size() == 1) c.mapQV = kMaxPhredScore;
            else if (c.probScore - denom < -20) c.mapQV = 0;
            else {
                const double sub = c.probScore - denom;              // float - float, then `double sub`, :278
                const double expo = std::exp(std::log(10.0) * sub), diff = 1.0 - expo;
                int q = expo == 0 ? 0 : diff == 0 ? kMaxPhredScore : Phred(diff);
                c.mapQV = q > kMaxPhredScore ? kMaxPhredScore : q;
            }
            if (scaleMapQVByNumSignificantClusters) {                // ScaleMapQVByClusterSize, :97-106
                if (c.numSignificantClusters > nCandidates)
                    c.mapQV = Phred((1 - std::pow(10.0, c.mapQV / -10.0)) * ((float)nCandidates / c.numSignificantClusters));
                else if (c.numSignificantClusters == 0) c.mapQV = 0;
            }
        }
    }
}

// ---- the worker pool of the reference on GPUs ------------------------------------------------------
// The reference runs -nproc worker threads, each with its own MappingBuffers (Blasr.cpp:1457-1488, :771).  On this path a
// worker is a context (gba_ctx); HostPipeline is what the workers that share ONE GPU become, MultiGpuRefiner the whole
// pool: one pipeline per GPU, every batch cut into one contiguous range of reads per GPU, no collective, results in
// input order (the -start / -stride partition of iblasr/RegisterBlasrOptions.h:93-94 gives every process whole reads
// too; contiguous ranges instead of a stride keep each GPU's share of the arenas one stretch, so a GPU stages and holds
// 1 / nGPU of the bytes).

// Several contexts on ONE GPU used in turn, so that host-buffer batches overlap: batch k + 1 is submitted before batch k
// is waited for, its inputs travel while batch k computes, and batch k's last results travel while batch k + 1 starts.
// Batches that start together stay together (both wait for their first inputs and both copy their last results out at
// the same time, with no kernel running in between), so a batch submitted while the one before it has run for less than
// `stagger` of a batch's usual time is held back inside Submit() for the difference; once apart they stay apart.
// Set GPU_MAX_HW_QUEUES=16 in the environment before the HIP runtime starts (8 streams per context).
class HostPipeline {
public:
    explicit HostPipeline(int device, int depth = 2, double stagger = 0.4) : stagger_(stagger)
    {
        for (int i = 0; i < depth; i++) {
            gba_ctx *c = nullptr;
            gba_config cfg{device, 0, nullptr};
            const int rc = gba_create(&c, &cfg);
            if (rc != GBA_OK) { for (gba_ctx *x : ctxs_) gba_destroy(x); throw std::runtime_error(std::string("gba_create: ") + gba_strerror(rc)); }
            ctxs_.push_back(c);
        }
    }
    ~HostPipeline() { for (gba_ctx *c : ctxs_) gba_destroy(c); }
    HostPipeline(const HostPipeline &) = delete;
    HostPipeline &operator=(const HostPipeline &) = delete;

    int depth() const { return (int)ctxs_.size(); }
    gba_ctx *ctx(int i) const { return ctxs_[(size_t)i]; }
    int SetScoring(const gba_scoring &s) { for (gba_ctx *c : ctxs_) { const int rc = gba_set_scoring(c, &s); if (rc) return rc; } return GBA_OK; }
    int SetReference(const uint8_t *ref, uint64_t len) { for (gba_ctx *c : ctxs_) { const int rc = gba_set_reference(c, ref, len); if (rc) return rc; } return GBA_OK; }
    int SetGuideProducer(const gba_sdp_params *p) { for (gba_ctx *c : ctxs_) { const int rc = gba_set_guide_producer(c, p); if (rc) return rc; } return GBA_OK; }
    // page-locked memory of context i (freed with it)
    void *HostAlloc(uint64_t bytes, int i = 0) { return gba_host_alloc(ctxs_[(size_t)i], bytes); }

    // gba_submit on the next context in turn; the batch arrays, *out and its buffers stay the caller's until Wait() returns
    int Submit(const gba_batch &b, gba_results *out)
    {
        if (sub_ - wait_ >= ctxs_.size()) return GBA_E_STATE;
        if (!t_submit_.empty() && service_ > 0 && stagger_ > 0) {
            const double hold = t_submit_.back() + stagger_ * service_ - Now();
            if (hold > 0) std::this_thread::sleep_for(std::chrono::duration<double>(hold));
        }
        const int rc = gba_submit(ctxs_[sub_ % ctxs_.size()], &b, out);
        if (rc != GBA_OK) return rc;
        t_submit_.push_back(Now());
        sub_++;
        return GBA_OK;
    }
    // gba_wait for the oldest submitted batch
    int Wait()
    {
        if (wait_ >= sub_) return GBA_E_STATE;
        const int rc = gba_wait(ctxs_[wait_ % ctxs_.size()]);
        wait_++;
        const double now = Now(), started = t_submit_.front();
        t_submit_.pop_front();
        const double took = now - (have_done_ && t_done_ > started ? t_done_ : started);
        service_ = service_ > 0 ? 0.5 * service_ + 0.5 * took : took;
        t_done_ = now; have_done_ = true;
        return rc;
    }
    // Runs a batch to completion on the context whose batch Wait() returned last (it is idle; the batches still in flight
    // are on the others): for a caller that repeats that batch, e.g. with a larger block arena after GBA_E_NOSPACE.
    int RerunLast(const gba_batch &b, gba_results *out)
    {
        if (wait_ == 0) return GBA_E_STATE;
        gba_ctx *c = ctxs_[(wait_ - 1) % ctxs_.size()];
        const int rc = gba_submit(c, &b, out);
        return rc != GBA_OK ? rc : gba_wait(c);
    }

private:
    static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    std::vector<gba_ctx *> ctxs_;
    size_t sub_ = 0, wait_ = 0;
    double stagger_, service_ = 0, t_done_ = 0;
    bool have_done_ = false;
    std::deque<double> t_submit_;
};

// The results of the pairs [lo, hi) of a batch as one GPU left them; pairs[k].blk_off indexes `blocks`.
struct Segment { uint32_t lo = 0, hi = 0; gba_pair_result *pairs = nullptr; gba_block *blocks = nullptr; uint64_t blocks_needed = 0; int rc = GBA_OK; };

class MultiGpuRefiner {
public:
    // devices: one worker per entry (a GPU may be named more than once); depth: contexts per worker, used in turn
    explicit MultiGpuRefiner(const std::vector<int> &devices, int depth = 1, double stagger = 0.4) : depth_(depth)
    {
        for (int d : devices) pipes_.emplace_back(new HostPipeline(d, depth, stagger));
        work_.resize(devices.size() * (size_t)depth);
        last_needed_.assign(devices.size(), 0);
        bytes_staged.assign(devices.size(), 0);
    }
    size_t workers() const { return pipes_.size(); }
    HostPipeline &pipe(size_t g) { return *pipes_[g]; }
    int SetScoring(const gba_scoring &s) { for (auto &p : pipes_) { const int rc = p->SetScoring(s); if (rc) return rc; } return GBA_OK; }
    int SetReference(const uint8_t *ref, uint64_t len) { for (auto &p : pipes_) { const int rc = p->SetReference(ref, len); if (rc) return rc; } return GBA_OK; }
    int SetGuideProducer(const gba_sdp_params *sp) { for (auto &p : pipes_) { const int rc = p->SetGuideProducer(sp); if (rc) return rc; } return GBA_OK; }

    // What the caller knows about the blocks a worker's share of a batch will hold (otherwise: what its last batch needed;
    // a first batch: one block per four query bases).  Too small a figure costs that worker a second run, nothing else.
    void ExpectBlocks(uint64_t perWorker) { last_needed_.assign(last_needed_.size(), perWorker); }

    // Pair ranges, one per worker, that tile the batch in order: cut only between reads (read_id[p], null: every pair is
    // its own read) and balanced by query bases, the predictor of a pair's DP cells.
    static std::vector<std::pair<uint32_t, uint32_t>> ContiguousRanges(const gba_batch &b, const uint32_t *read_id, size_t world)
    {
        const uint32_t n = b.n_pairs;
        std::vector<std::pair<uint32_t, uint32_t>> out(world, {0u, 0u});
        if (n == 0) return out;
        std::vector<uint32_t> starts;               // first pair of every read
        std::vector<uint64_t> before;               // query bases in front of it
        uint64_t acc = 0;
        for (uint32_t p = 0; p < n; p++) {
            if (p == 0 || !read_id || read_id[p] != read_id[p - 1]) { starts.push_back(p); before.push_back(acc); }
            acc += b.q_len[p];
        }
        const size_t R = starts.size();
        std::vector<uint32_t> cuts{0};
        size_t k_prev = 0;
        for (size_t r = 1; r < world; r++) {
            const double want = (double)acc * (double)r / (double)world;
            size_t k = (size_t)(std::lower_bound(before.begin(), before.end(), want, [](uint64_t a, double w) { return (double)a < w; }) - before.begin());
            if (k > 0 && k < R && want - (double)before[k - 1] < (double)before[k] - want) k--;      // the nearest read boundary
            if (R >= world) k = std::min(std::max(k, k_prev + 1), R - (world - r));              // nobody idles when there are reads enough
            k = std::max(k, k_prev);
            cuts.push_back(k < R ? starts[k] : n);
            k_prev = k;
        }
        cuts.push_back(n);
        for (size_t r = 0; r < world; r++) out[r] = {cuts[r], cuts[r + 1]};
        return out;
    }

    // Starts the batch on every worker (gba_submit does not block) and returns; Wait() collects it.  The batch arrays stay
    // the caller's until then.  Batches with packed reads or a cost profile go to one context (or HostPipeline) whole.
    int Submit(const gba_batch &b, const uint32_t *read_id = nullptr, bool wantBlocks = true)
    {
        if (flight_.size() >= (size_t)depth_) return GBA_E_STATE;
        if (b.q_packed || b.q_profile) return GBA_E_UNSUPPORTED;
        const size_t slot = sub_ % (size_t)depth_;
        Ticket t;
        t.wantBlocks = wantBlocks; t.slot = slot;
        const auto ranges = ContiguousRanges(b, read_id, pipes_.size());
        for (size_t g = 0; g < pipes_.size(); g++) {
            Work &w = work_[g * (size_t)depth_ + slot];
            w.lo = ranges[g].first; w.hi = ranges[g].second;
            const uint32_t n = w.hi - w.lo;
            w.submitted = false;
            if (n == 0) continue;
            // this range as a batch of its own: per-pair arrays shifted, the three arenas cut to the stretch its pairs point
            // into and the offsets rebased, so that the GPU stages and holds its own share only
            gba_batch &sb = w.batch;
            sb = b;
            sb.n_pairs = n;
            sb.q_len = b.q_len + w.lo; sb.t_len = b.t_len + w.lo;
            if (b.n_blk) sb.n_blk = b.n_blk + w.lo;
            if (b.t_rc) sb.t_rc = b.t_rc + w.lo;
            uint64_t staged = 0, qsum = 0;
            Rebase(b.q_off, b.q_len, w.lo, n, b.q_arena_bytes, w.q_off, &sb.q_arena, 1, &sb.q_arena_bytes, &staged);
            sb.q_off = w.q_off.data();
            if (b.t_arena) Rebase(b.t_off, b.t_len, w.lo, n, b.t_arena_bytes, w.t_off, &sb.t_arena, 1, &sb.t_arena_bytes, &staged);
            else w.t_off.assign(b.t_off + w.lo, b.t_off + w.hi);                       // windows of the resident reference: absolute
            sb.t_off = w.t_off.data();
            if (b.blocks) {
                const uint8_t *base = (const uint8_t *)b.blocks;
                Rebase(b.blk_off, b.n_blk, w.lo, n, b.n_blocks_total, w.b_off, &base, sizeof(gba_block), &sb.n_blocks_total, &staged);
                sb.blocks = (const gba_block *)base; sb.blk_off = w.b_off.data();
            }
            bytes_staged[g] = staged;
            for (uint32_t p = 0; p < n; p++) qsum += sb.q_len[p];
            const uint64_t cap = !wantBlocks ? 0 : last_needed_[g] ? last_needed_[g] + last_needed_[g] / 8 + 64 : qsum / 4 + n + 64;
            int rc = Buffers(g, w, n, cap);
            if (rc != GBA_OK) return rc;
            if ((rc = pipes_[g]->Submit(sb, &w.res)) != GBA_OK) return rc;
            w.submitted = true;
        }
        flight_.push_back(t);
        sub_++;
        return GBA_OK;
    }

    // Blocks until the oldest submitted batch is complete on every worker; segs (one per worker, in order) then tile the
    // batch in input order.  A worker whose block arena, sized from its last batch, was too small runs its range again at
    // the size it reported.  Returns the first error of any worker.
    int Wait(std::vector<Segment> *segs)
    {
        if (flight_.empty()) return GBA_E_STATE;
        const Ticket t = flight_.front();
        flight_.pop_front();
        segs->clear();
        int rc_all = GBA_OK;
        for (size_t g = 0; g < pipes_.size(); g++) {
            Work &w = work_[g * (size_t)depth_ + t.slot];
            Segment s;
            s.lo = w.lo; s.hi = w.hi;
            if (w.submitted) {
                int rc = pipes_[g]->Wait();
                if (rc == GBA_E_NOSPACE && t.wantBlocks) {
                    rc = Buffers(g, w, w.hi - w.lo, w.res.blocks_needed + 64);
                    if (rc == GBA_OK) rc = pipes_[g]->RerunLast(w.batch, &w.res);
                }
                last_needed_[g] = w.res.blocks_needed;
                s.pairs = w.res.pairs; s.blocks = w.res.blocks; s.blocks_needed = w.res.blocks_needed; s.rc = rc;
                if (rc != GBA_OK && rc_all == GBA_OK) rc_all = rc;
            }
            segs->push_back(s);
        }
        return rc_all;
    }

    // The segments as the two arrays one context would have returned: pairs in input order, blocks packed in input order,
    // every blk_off rewritten (two sequential copies and one prefix per worker).
    static void Merge(const std::vector<Segment> &segs, std::vector<gba_pair_result> *pairs, std::vector<gba_block> *blocks)
    {
        pairs->clear(); blocks->clear();
        for (const Segment &s : segs) {
            const uint64_t base = blocks->size();
            const size_t at = pairs->size();
            if (s.pairs) pairs->insert(pairs->end(), s.pairs, s.pairs + (s.hi - s.lo));
            for (size_t k = at; k < pairs->size(); k++) (*pairs)[k].blk_off += base;
            if (s.blocks) blocks->insert(blocks->end(), s.blocks, s.blocks + s.blocks_needed);
        }
    }

    std::vector<uint64_t> bytes_staged;      // arena bytes of the last batch, per worker

private:
    struct Work {
        uint32_t lo = 0, hi = 0;
        bool submitted = false;
        gba_batch batch{};
        gba_results res{};
        std::vector<uint64_t> q_off, t_off, b_off;
        gba_pair_result *pairs = nullptr; uint64_t pairs_cap = 0;
        gba_block *blocks = nullptr; uint64_t blocks_cap = 0;
    };
    struct Ticket { bool wantBlocks = true; size_t slot = 0; };

    // offsets of pairs [lo, lo + n) into an arena of `extent` elements: the stretch they cover becomes the arena
    static void Rebase(const uint64_t *off, const uint32_t *len, uint32_t lo, uint32_t n, uint64_t extent, std::vector<uint64_t> &out,
                       const uint8_t **arena, size_t esz, uint64_t *new_extent, uint64_t *staged)
    {
        uint64_t first = UINT64_MAX, last = 0;
        for (uint32_t p = lo; p < lo + n; p++) {
            const uint64_t o = off[p], l = len[p];
            if (l == 0 || o > extent || l > extent - o) continue;          // the kernels refuse such a pair without touching it
            first = std::min(first, o); last = std::max(last, o + l);
        }
        if (first > last) { first = 0; last = 0; }
        out.resize(n);
        for (uint32_t p = 0; p < n; p++) out[p] = off[lo + p] - first;     // a refused pair's offset wraps: it stays outside the arena
        *arena += first * esz;
        *new_extent = last - first;
        *staged += (last - first) * esz;
    }

    // page-locked result buffers of a worker's slot, grow-only, from the context that fills them
    int Buffers(size_t g, Work &w, uint32_t n, uint64_t cap)
    {
        const int ci = (int)(&w - &work_[g * (size_t)depth_]);
        if (w.pairs_cap < n) {
            w.pairs = (gba_pair_result *)pipes_[g]->HostAlloc(((uint64_t)n + n / 8 + 1) * sizeof(gba_pair_result), ci);
            if (!w.pairs) return GBA_E_NOMEM;
            w.pairs_cap = (uint64_t)n + n / 8 + 1;
        }
        if (w.blocks_cap < cap) {
            w.blocks = (gba_block *)pipes_[g]->HostAlloc((cap + cap / 8 + 1) * sizeof(gba_block), ci);
            if (!w.blocks) return GBA_E_NOMEM;
            w.blocks_cap = cap + cap / 8 + 1;
        }
        w.res = gba_results{};
        w.res.pairs = w.pairs;
        w.res.blocks = cap ? w.blocks : nullptr;
        w.res.blocks_capacity = cap;
        return GBA_OK;
    }

    int depth_;
    std::vector<std::unique_ptr<HostPipeline>> pipes_;
    std::vector<Work> work_;                 // [worker][slot]
    std::vector<uint64_t> last_needed_;
    std::deque<Ticket> flight_;
    size_t sub_ = 0;
};

}  // namespace gba
#endif
