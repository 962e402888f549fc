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
//   fused    SDPAlign and RefineAlignment in one submit (AlignAndRefineCandidates; the blocks of the input are ignored)
//   intervals  run without a block arena and print per candidate: status score nBlocks qStart qEnd tStart tEnd
//   qv       every candidate line carries five more strings behind its blocks (insertionQV deletionQV deletionTag substitutionQV
//            substitutionTag, quality values as letters: value + 33): the batch runs with the IDS scorer's costs (IdsProfile)
//   legacy   run the branch without guided alignment (RefineLegacy: AffineKBandAlign in a band around the guide's drift)
//   extend   stdin is n, the genome, then per candidate "read tAlignedSeqPos windowLength qAlignedSeqPos qPos tPos nblk blocks..."
//            (its window is the genome from tAlignedSeqPos over windowLength bases, the blocks are relative to qPos / tPos):
//            the -extend step (ExtendAlignments); prints per candidate
//            status score qAlignedSeqPos tAlignedSeqPos qPos tPos qStart qEnd tStart tEnd tAlignedSeq nblk blocks...
//   mapqv    no GPU: stdin is n, then "qStart qEnd rawProbScore" per candidate of ONE read; prints their mapQVs
//   filters  no GPU: stdin is n, then "tIndex tBegin tEnd score pctSimilarity nBlocks" per candidate of ONE read, in order;
//            runs RemoveLowQualityAlignments, then RemoveOverlappingAlignments (iblasr/BlasrAlignImpl.hpp:381-382) and
//            prints the input indices of the survivors
#include <cstdio>
#include <cstdlib>
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
    for (size_t i = 0; i < n; i++) {
        size_t nb;
        std::cin >> reads[i] >> cands[i].tAlignedSeq >> nb;
        cands[i].blocks.resize(nb);
        for (auto &b : cands[i].blocks) std::cin >> b.qPos >> b.tPos >> b.length;
        if (mode == "qv") {
            gba::QualityValues q;
            std::cin >> q.insertionQV >> q.deletionQV >> q.deletionTag >> q.substitutionQV >> q.substitutionTag;
            for (std::string *s : {&q.insertionQV, &q.deletionQV, &q.substitutionQV}) for (char &ch : *s) ch = (char)(ch - 33);
            qvs.push_back(q);
        }
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
    } else if (fused) {
        params.sortRefinedAlignments = false;
        for (size_t i = 0; i < n && rc == GBA_OK; i++) {
            std::vector<gba::AlignmentCandidate *> one{&cands[i]};
            rc = gba::AlignAndRefineCandidates(reads[i], one, params, ctx);
        }
    } else {
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
