// Small driver of gba_refine.hpp used by the tests: reads pairs from stdin
//   n
//   read  target  nblk  (qPos tPos len)*nblk      -- one candidate per line, one read each
// refines them in ONE batch and prints per candidate: status score nCells qPos tPos nblk blocks...
// Arguments:
//   cigar    also print the two CIGAR forms (adjacent indels merged / allowed) behind each candidate
//   reuse    hand every read to RefineBatch::Add through ONE std::string buffer that is overwritten
//            for each candidate (what a parsing loop does)
//   global   run the -global branch (RefineGlobal) on each (read, candidate) instead
//   sdp      the guides come from the device first (SDPAlignCandidates: the blocks of the input are ignored), then the batch
//   ccs      GuidedAlignFromSequences on each (read, candidate): guides from 6-mers, band 12, the Local end rule (the blocks of
//            the input are ignored)
//   rbao     every candidate line carries, behind its blocks, "nanchors (q t l)*": the guide comes from GuidesBetweenAnchors (the
//            blocks of the input are ignored), then the batch
//   fused    SDPAlign and RefineAlignment in one submit (AlignAndRefineCandidates; the blocks of the input are ignored)
//   intervals  run without a block arena and print per candidate: status score nBlocks qStart qEnd tStart tEnd
//   qv       every candidate line carries five more strings behind its blocks (insertionQV deletionQV deletionTag substitutionQV
//            substitutionTag, quality values as letters: value + 33): the batch runs with the IDS scorer's costs (IdsProfile)
//   legacy   run the branch without guided alignment (RefineLegacy: AffineKBandAlign in a band around the guide's drift)
//   extend   stdin is n, the genome, then per candidate "read tAlignedSeqPos windowLength qAlignedSeqPos qPos tPos nblk blocks..."
//            (its window is the genome from tAlignedSeqPos over windowLength bases, the blocks are relative to qPos / tPos):
//            the -extend step (ExtendAlignments); prints per candidate
//            status score qAlignedSeqPos tAlignedSeqPos qPos tPos qStart qEnd tStart tEnd tAlignedSeq nblk blocks...
//   pool     the worker pool: every input line is one candidate of read (line index / 3) (three candidates per read); the batch
//            runs (a) on one context, (b) twice in a row through a HostPipeline of two contexts on GPU 0, (c) twice through a
//            MultiGpuRefiner of three workers with two contexts each, all on GPU 0, the first time with a block arena guessed
//            far too small; prints "ok <pairs> <blocks> <staged bytes of each worker>" when (b) and (c) give (a)'s bytes
//   mapqv    no GPU: stdin is n, then "qStart qEnd rawProbScore" per candidate of ONE read; prints their mapQVs
//   filters  no GPU: stdin is n, then "tIndex tBegin tEnd score pctSimilarity nBlocks" per candidate of ONE read, in order;
//            runs RemoveLowQualityAlignments, then RemoveOverlappingAlignments (iblasr/BlasrAlignImpl.hpp:381-382) and
//            prints the input indices of the survivors
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "gba_refine.hpp"

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    const bool cigar = mode == "cigar", reuse = mode == "reuse", global = mode == "global", sdp = mode == "sdp";
    const bool intervals = mode == "intervals", fused = mode == "fused";
    size_t n;
    if (!(std::cin >> n)) return 2;
    if (mode == "mapqv") {
        std::vector<gba::AlignmentCandidate> cs(n);
        std::vector<gba::AlignmentCandidate *> ptrs;
        std::vector<int> raw(n);
        for (size_t i = 0; i < n; i++) {
            uint32_t q0, q1;
            std::cin >> q0 >> q1 >> raw[i];
            cs[i].qAlignedSeqPos = q0;
            cs[i].blocks.push_back(gba::Block{0, 0, q1 - q0});
            ptrs.push_back(&cs[i]);
        }
        gba::StoreMapQVs(ptrs, raw);
        for (auto &c : cs) std::printf("%d ", c.mapQV);
        std::printf("\n");
        return 0;
    }
    if (mode == "filters") {
        std::vector<gba::AlignmentCandidate> cs(n);
        std::vector<gba::AlignmentCandidate *> ptrs;
        for (size_t i = 0; i < n; i++) {
            uint32_t t0, t1, nb;
            std::cin >> cs[i].tIndex >> t0 >> t1 >> cs[i].score >> cs[i].pctSimilarity >> nb;
            cs[i].tAlignedSeqPos = t0;
            if (nb) cs[i].blocks.push_back(gba::Block{0, 0, t1 - t0});
            ptrs.push_back(&cs[i]);
        }
        gba::MappingParameters params;
        if (argc > 2) params.minPctSimilarity = (float)atof(argv[2]);
        if (argc > 3) params.nCandidates = atoi(argv[3]);
        gba::RemoveLowQualityAlignments(ptrs, params);
        gba::RemoveOverlappingAlignments(ptrs, params);
        for (auto *c : ptrs) std::printf("%d ", (int)(c - cs.data()));
        std::printf("\n");
        return 0;
    }
    if (mode == "extend") {
        std::string genome;
        std::cin >> genome;
        std::vector<std::string> rd(n);
        std::vector<gba::AlignmentCandidate> cs(n);
        for (size_t i = 0; i < n; i++) {
            size_t nb, wl;
            std::cin >> rd[i] >> cs[i].tAlignedSeqPos >> wl >> cs[i].qAlignedSeqPos >> cs[i].qPos >> cs[i].tPos >> nb;
            cs[i].blocks.resize(nb);
            for (auto &b : cs[i].blocks) std::cin >> b.qPos >> b.tPos >> b.length;
            cs[i].tAlignedSeq = genome.substr(cs[i].tAlignedSeqPos, wl);
            cs[i].tAlignedSeqLength = (uint32_t)wl;
            cs[i].qAlignedSeqLength = (uint32_t)(rd[i].size() - cs[i].qAlignedSeqPos);
            cs[i].qAlignedSeq = rd[i].substr(cs[i].qAlignedSeqPos);
        }
        gba_ctx *ctx = nullptr;
        gba_config cfg{0, 0, nullptr};
        int rc = gba_create(&ctx, &cfg);
        if (rc != GBA_OK) { std::fprintf(stderr, "gba_create: %s\n", gba_strerror(rc)); return 1; }
        gba::MappingParameters params;
        for (size_t i = 0; i < n && rc == GBA_OK; i++) {
            std::vector<gba::AlignmentCandidate *> one{&cs[i]};
            rc = gba::ExtendAlignments(rd[i], one, genome, params, ctx);
        }
        if (rc != GBA_OK) { std::fprintf(stderr, "ExtendAlignments: %s\n", gba_strerror(rc)); return 1; }
        for (auto &c : cs) {
            std::printf("%d %d %u %u %u %u %u %u %u %u %s %zu", c.status, c.score, c.qAlignedSeqPos, c.tAlignedSeqPos, c.qPos, c.tPos,
                        c.qStart, c.qEnd, c.tStart, c.tEnd, c.tAlignedSeq.c_str(), c.blocks.size());
            for (auto &b : c.blocks) std::printf(" %u %u %u", b.qPos, b.tPos, b.length);
            std::printf("\n");
        }
        gba_destroy(ctx);
        return 0;
    }
    std::vector<std::string> reads(n);
    std::vector<gba::AlignmentCandidate> cands(n);
    std::vector<gba::QualityValues> qvs;
    std::vector<std::vector<gba::Anchor>> anchor_sets;
    for (size_t i = 0; i < n; i++) {
        size_t nb;
        std::cin >> reads[i] >> cands[i].tAlignedSeq >> nb;
        cands[i].blocks.resize(nb);
        for (auto &b : cands[i].blocks) std::cin >> b.qPos >> b.tPos >> b.length;
        if (mode == "rbao") {
            size_t na;
            std::cin >> na;
            std::vector<gba::Anchor> a(na);
            for (auto &m : a) std::cin >> m.q >> m.t >> m.l;
            anchor_sets.push_back(a);
        }
        if (mode == "qv") {
            gba::QualityValues q;
            std::cin >> q.insertionQV >> q.deletionQV >> q.deletionTag >> q.substitutionQV >> q.substitutionTag;
            for (std::string *s : {&q.insertionQV, &q.deletionQV, &q.substitutionQV}) for (char &ch : *s) ch = (char)(ch - 33);
            qvs.push_back(q);
        }
    }
    if (mode == "pool") {
        std::string qa, ta;
        std::vector<uint64_t> q_off, t_off, b_off;
        std::vector<uint32_t> q_len, t_len, n_blk, read_id;
        std::vector<gba_block> blk;
        for (size_t i = 0; i < n; i++) {
            q_off.push_back(qa.size()); q_len.push_back((uint32_t)reads[i].size()); qa += reads[i];
            t_off.push_back(ta.size()); t_len.push_back((uint32_t)cands[i].tAlignedSeq.size()); ta += cands[i].tAlignedSeq;
            b_off.push_back(blk.size()); n_blk.push_back((uint32_t)cands[i].blocks.size());
            for (auto &b : cands[i].blocks) blk.push_back(gba_block{b.qPos, b.tPos, b.length});
            read_id.push_back((uint32_t)(i / 3));
        }
        gba_batch b{};
        b.n_pairs = (uint32_t)n;
        b.q_arena = (const uint8_t *)qa.data(); b.q_off = q_off.data(); b.q_len = q_len.data(); b.q_arena_bytes = qa.size();
        b.t_arena = (const uint8_t *)ta.data(); b.t_off = t_off.data(); b.t_len = t_len.data(); b.t_arena_bytes = ta.size();
        b.blocks = blk.data(); b.blk_off = b_off.data(); b.n_blk = n_blk.data(); b.n_blocks_total = blk.size();
        const gba_scoring sc = gba::ScoringFromParams(gba::MappingParameters());
        const uint64_t cap = qa.size() + n;
        // (a) one context
        std::vector<gba_pair_result> wantP(n);
        std::vector<gba_block> wantB(cap);
        uint64_t needed = 0;
        {
            gba_ctx *one = nullptr;
            gba_config cfg{0, 0, nullptr};
            int rc = gba_create(&one, &cfg);
            if (rc == GBA_OK) rc = gba_set_scoring(one, &sc);
            gba_results r{};
            r.pairs = wantP.data(); r.blocks = wantB.data(); r.blocks_capacity = cap;
            if (rc == GBA_OK) rc = gba_submit(one, &b, &r);
            if (rc == GBA_OK) rc = gba_wait(one);
            if (rc != GBA_OK) { std::fprintf(stderr, "one context: %s\n", gba_strerror(rc)); return 1; }
            needed = r.blocks_needed;
            gba_destroy(one);
        }
        auto same = [&](const gba_pair_result *p, const gba_block *bl, uint64_t nb) {
            return nb == needed && std::memcmp(p, wantP.data(), n * sizeof(gba_pair_result)) == 0 && std::memcmp(bl, wantB.data(), needed * sizeof(gba_block)) == 0;
        };
        // (b) two contexts on the GPU, two batches in flight
        {
            gba::HostPipeline pipe(0, 2, 0.4);
            if (pipe.SetScoring(sc) != GBA_OK) return 1;
            std::vector<gba_pair_result> p0(n), p1(n);
            std::vector<gba_block> b0(cap), b1(cap);
            gba_results r0{}, r1{};
            r0.pairs = p0.data(); r0.blocks = b0.data(); r0.blocks_capacity = cap;
            r1.pairs = p1.data(); r1.blocks = b1.data(); r1.blocks_capacity = cap;
            if (pipe.Submit(b, &r0) != GBA_OK || pipe.Submit(b, &r1) != GBA_OK) { std::fprintf(stderr, "pipeline submit\n"); return 1; }
            gba_results r2{};
            if (pipe.Submit(b, &r2) != GBA_E_STATE) { std::fprintf(stderr, "a third batch in flight was accepted\n"); return 1; }
            if (pipe.Wait() != GBA_OK || pipe.Wait() != GBA_OK || pipe.Wait() != GBA_E_STATE) { std::fprintf(stderr, "pipeline wait\n"); return 1; }
            if (!same(p0.data(), b0.data(), r0.blocks_needed) || !same(p1.data(), b1.data(), r1.blocks_needed)) { std::fprintf(stderr, "pipeline results differ\n"); return 1; }
        }
        // (c) the pool: three workers of two contexts each
        gba::MultiGpuRefiner pool({0, 0, 0}, 2, 0.0);
        if (pool.SetScoring(sc) != GBA_OK) return 1;
        std::vector<gba::Segment> segs;
        std::vector<gba_pair_result> gotP;
        std::vector<gba_block> gotB;
        pool.ExpectBlocks(1);            // far too few: every worker's first run ends in GBA_E_NOSPACE and is repeated at size
        for (int round = 0; round < 2; round++) {
            if (pool.Submit(b, read_id.data()) != GBA_OK || pool.Submit(b, read_id.data()) != GBA_OK) { std::fprintf(stderr, "pool submit\n"); return 1; }
            for (int k = 0; k < 2; k++) {
                const int rc = pool.Wait(&segs);
                if (rc != GBA_OK) { std::fprintf(stderr, "pool wait: %s\n", gba_strerror(rc)); return 1; }
                for (size_t g = 0; g < segs.size(); g++) {
                    if (g && segs[g].lo != segs[g - 1].hi) { std::fprintf(stderr, "segments do not tile the batch\n"); return 1; }
                    if (segs[g].lo < n && segs[g].lo > 0 && read_id[segs[g].lo] == read_id[segs[g].lo - 1]) { std::fprintf(stderr, "a read was cut\n"); return 1; }
                }
                gba::MultiGpuRefiner::Merge(segs, &gotP, &gotB);
                if (gotP.size() != n || !same(gotP.data(), gotB.data(), gotB.size())) { std::fprintf(stderr, "pool results differ (round %d)\n", round); return 1; }
            }
        }
        std::printf("ok %zu %llu", n, (unsigned long long)needed);
        for (uint64_t v : pool.bytes_staged) std::printf(" %llu", (unsigned long long)v);
        std::printf(" %zu\n", qa.size() + ta.size() + blk.size() * sizeof(gba_block));
        return 0;
    }
    gba_ctx *ctx = nullptr;
    gba_config cfg{0, 0, nullptr};
    int rc = gba_create(&ctx, &cfg);
    if (rc != GBA_OK) { std::fprintf(stderr, "gba_create: %s\n", gba_strerror(rc)); return 1; }
    gba::MappingParameters params;
    if (mode == "legacy") {
        params.useGuidedAlign = false;
        params.bandSize = 0;
        params.sortRefinedAlignments = false;
        for (size_t i = 0; i < n && rc == GBA_OK; i++) {
            std::vector<gba::AlignmentCandidate *> one{&cands[i]};
            rc = gba::RefineAlignments(reads[i], one, params, ctx);
        }
    } else if (global) {
        params.doGlobalAlignment = true;
        params.sortRefinedAlignments = false;
        for (size_t i = 0; i < n && rc == GBA_OK; i++) {
            std::vector<gba::AlignmentCandidate *> one{&cands[i]};
            rc = gba::RefineAlignments(reads[i], one, params, ctx);
        }
    } else if (mode == "ccs") {
        for (size_t i = 0; i < n && rc == GBA_OK; i++) {
            std::vector<gba::AlignmentCandidate *> one{&cands[i]};
            rc = gba::GuidedAlignFromSequences(reads[i], one, params, ctx);
        }
    } else if (fused) {
        params.sortRefinedAlignments = false;
        for (size_t i = 0; i < n && rc == GBA_OK; i++) {
            std::vector<gba::AlignmentCandidate *> one{&cands[i]};
            rc = gba::AlignAndRefineCandidates(reads[i], one, params, ctx);
        }
    } else {
        if (mode == "rbao") {
            for (size_t i = 0; i < n && rc == GBA_OK; i++) {
                std::vector<gba::AlignmentCandidate *> one{&cands[i]};
                std::vector<std::vector<gba::Anchor>> a{anchor_sets[i]};
                rc = gba::GuidesBetweenAnchors(reads[i], one, a, params, ctx);
            }
            if (rc != GBA_OK) { std::fprintf(stderr, "GuidesBetweenAnchors: %s\n", gba_strerror(rc)); return 1; }
        }
        if (sdp) {
            for (size_t i = 0; i < n && rc == GBA_OK; i++) {
                std::vector<gba::AlignmentCandidate *> one{&cands[i]};
                rc = gba::SDPAlignCandidates(reads[i], one, params, ctx);
            }
            if (rc != GBA_OK) { std::fprintf(stderr, "SDPAlignCandidates: %s\n", gba_strerror(rc)); return 1; }
        }
        gba::RefineBatch batch(ctx);
        std::string buffer;
        for (size_t i = 0; i < n; i++) {
            if (reuse) { buffer.assign(reads[i]); batch.Add(buffer, &cands[i]); }
            else if (mode == "qv") { const std::vector<int16_t> prof = gba::IdsProfile(reads[i], qvs[i], params); batch.Add(reads[i], &cands[i], &prof); }
            else batch.Add(reads[i], &cands[i]);
        }
        rc = batch.Run(params, !intervals);
    }
    if (rc != GBA_OK) { std::fprintf(stderr, "Run: %s\n", gba_strerror(rc)); return 1; }
    for (auto &c : cands) {
        if (intervals) {
            int q0, q1;
            gba::GetQInterval(c, q0, q1);
            if ((uint32_t)q0 != c.qStart || (uint32_t)q1 != c.qEnd || !c.blocks.empty()) { std::fprintf(stderr, "interval mismatch\n"); return 1; }
            std::printf("%d %d %u %u %u %u %u\n", c.status, c.score, c.nBlocks, c.qStart, c.qEnd, c.tStart, c.tEnd);
            continue;
        }
        std::printf("%d %d %llu %u %u %zu", c.status, c.score, (unsigned long long)c.nCells, c.qPos, c.tPos, c.blocks.size());
        for (auto &b : c.blocks) std::printf(" %u %u %u", b.qPos, b.tPos, b.length);
        if (cigar && c.status == 0)
            std::printf(" %s %s", gba::CigarString(c).c_str(), gba::CigarString(c, true, true).c_str());
        std::printf("\n");
    }
    gba_destroy(ctx);
    return 0;
}
