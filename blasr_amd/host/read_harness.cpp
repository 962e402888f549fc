// read_harness.cpp -- the read-level loop around the refine stage, end to end, in C++ over the C ABI.
//
// BLASR maps one read at a time: MapRead finds up to nCandidates windows, AlignIntervals chains each (SDPAlign),
// RefineAlignments refines them, then the per-read steps run: sort by score, RemoveLowQualityAlignments,
// RemoveOverlappingAlignments (/root/reference/iblasr/BlasrAlignImpl.hpp:358-383), StoreMapQVs and the writer
// (/root/reference/Blasr.cpp:419-430, :832-840).  Batching across reads splits that loop in three phases:
//   A  candidates of a few thousand reads are collected (here: they come from the synthetic generator, the reference's
//      seeding and clustering being outside this library);
//   B  ONE gba_submit from sequences for all candidates of all reads of the batch (guides made on the device, refined
//      where they lie), pair results only: interval output needs no blocks;
//   C  per read, in input order: the candidates are sorted and filtered as above, mapping qualities are stored, and the
//      -m 4 lines of the survivors are written.
// Two contexts on the GPU are used in turn (gba::HostPipeline), so phase C of batch k runs on host threads while the GPU
// works on batch k + 1.  Prints one JSON line: reads per second end to end and where the time went.
//
//   read_harness [--reads N] [--cands K] [--read-len L] [--jitter J] [--ref-len G] [--n-frac F] [--batches B] [--seed S]
//                [--threads T] [--device D] [--dump FILE]
// --dump: the m4 lines of the LAST batch go to FILE (tests compare them with the Python mirrors' lines).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gba_refine.hpp"

extern "C" {
typedef struct { uint32_t q_pos, t_pos, length; } gbs_block;
typedef struct {
    uint32_t read_len;
    double p_ins, p_del, p_sub;
    uint32_t min_block;
    double drop_frac;
    uint64_t seed;
    uint32_t q_stride, t_stride, blk_stride;
    uint32_t decoy_every;
    uint32_t len_jitter;
    uint32_t shard_rank, shard_world;
    uint32_t share_reads;
} gbs_cfg;
void gbs_reference(uint8_t *ref, uint64_t len, uint64_t seed, double n_frac, uint32_t n_run);
uint32_t gbs_generate(const uint8_t *ref, uint64_t ref_len, const gbs_cfg *c, uint32_t first, uint32_t count,
                      uint8_t *q_arena, uint64_t *q_off, uint32_t *q_len, uint8_t *t_arena, uint64_t *t_off, uint32_t *t_len,
                      gbs_block *blocks, uint64_t *blk_off, uint32_t *n_blk, uint64_t *win_start, uint8_t *win_rc);
}

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The probability scorer of the mapping-quality step: SMRTLogProbMatrix with params.insertion / params.deletion
// (/root/reference/iblasr/BlasrUtilsImpl.hpp:117-123).  The matrix is libcpp data that the reference tree does not hold:
// UNPINNED; stated here as 0 for equal bases, 15 otherwise (the flat mismatch price SumMismatches uses, :224), 0 against N.
void ProbScorer(const gba::MappingParameters &p, gba_rescorer &r)
{
    std::memset(&r, 0, sizeof(r));
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) r.sub[a][b] = (a == 4 || b == 4 || a == b) ? 0 : 15;
    r.ins = p.insertion; r.del = p.deletion;
    r.affine = p.affineAlign ? 1 : 0; r.affine_open = p.affineOpen; r.affine_extend = p.affineExtend;
}

struct Batch {                     // one batch of reads x candidates, in page-locked memory of a context
    uint32_t n = 0;
    uint8_t *q_arena = nullptr; uint64_t q_bytes = 0;
    uint64_t *q_off = nullptr, *t_off = nullptr;
    uint32_t *q_len = nullptr, *t_len = nullptr;
    uint8_t *t_rc = nullptr;
    gba_pair_result *pairs = nullptr;
    gba_results res{};
    gba_batch b{};
};

// Phase C for the reads [r0, r1) of a batch; the m4 lines are appended to `out`.  Returns the alignments printed.
size_t PhaseC(const Batch &B, uint32_t K, uint32_t r0, uint32_t r1, uint64_t read_base, uint64_t ref_len, const gba::MappingParameters &params,
              std::string &out)
{
    size_t printed = 0;
    std::vector<gba::AlignmentCandidate> cs(K);
    std::vector<gba::AlignmentCandidate *> ptrs;
    std::vector<int> raw, rawOf(K);
    char line[256];
    for (uint32_t r = r0; r < r1; r++) {
        ptrs.clear();
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t p = r * K + k;
            const gba_pair_result &R = B.pairs[p];
            gba::AlignmentCandidate &c = cs[k];
            c = gba::AlignmentCandidate();
            c.status = R.status;
            c.tIndex = 0;
            // positions of a reverse-strand window count along the reverse complement of the reference, as the reference's do
            // until AssignRefContigLocations (/root/reference/iblasr/BlasrAlignImpl.hpp:504)
            c.tAlignedSeqPos = (uint32_t)(B.t_rc[p] ? ref_len - B.t_off[p] - B.t_len[p] : B.t_off[p]);
            c.qAlignedSeqPos = 0;
            c.numSignificantClusters = (int)K;
            if (R.status != GBA_PAIR_OK) continue;            // no guide, nothing aligned: the candidate has no blocks and drops out below
            c.score = R.score; c.nCells = R.ncells; c.nBlocks = R.n_blk;
            c.nMatch = R.n_match; c.nMismatch = R.n_mismatch; c.nIns = R.n_ins; c.nDel = R.n_del;
            const uint64_t cols = (uint64_t)R.n_match + R.n_mismatch + R.n_ins + R.n_del;
            c.pctSimilarity = cols ? (float)(R.n_match * 2.0 / (2.0 * cols) * 100.0) : 0.f;
            c.qPos = R.q_pos; c.tPos = R.t_pos;
            c.qStart = R.q_start; c.qEnd = R.q_end;
            c.tStart = c.tAlignedSeqPos + R.t_start; c.tEnd = c.tAlignedSeqPos + R.t_end;
            c.haveInterval = true;
            rawOf[k] = R.extra_score[0];                     // the path under the probability scorer, for StoreMapQVs
        }
        for (uint32_t k = 0; k < K; k++) ptrs.push_back(&cs[k]);
        // AlignIntervals' candidates without an alignment sort behind the others (score 0 against negative scores)
        std::stable_sort(ptrs.begin(), ptrs.end(), [](const gba::AlignmentCandidate *a, const gba::AlignmentCandidate *b) { return a->score < b->score; });
        gba::RemoveLowQualityAlignments(ptrs, params);
        gba::RemoveOverlappingAlignments(ptrs, params);
        raw.clear();
        for (auto *c : ptrs) raw.push_back(rawOf[(size_t)(c - cs.data())]);
        gba::StoreMapQVs(ptrs, raw, params.minFractionToBeConsideredOverlapping, params.nCandidates);
        const size_t nprint = std::min<size_t>(ptrs.size(), (size_t)params.nBest);
        for (size_t x = 0; x < nprint; x++) {
            const gba::AlignmentCandidate &c = *ptrs[x];
            const uint32_t p = r * K + (uint32_t)(&c - cs.data());
            const int len = std::snprintf(line, sizeof(line), "read%llu ref %d %g 0 %u %u %u %d %u %u %llu %d\n",
                                          (unsigned long long)(read_base + r), c.score, (double)c.pctSimilarity, c.qStart, c.qEnd, B.q_len[p],
                                          (int)B.t_rc[p], c.tStart, c.tEnd, (unsigned long long)ref_len, c.mapQV);
            out.append(line, (size_t)len);
        }
        printed += nprint;
    }
    return printed;
}

}  // namespace

int main(int argc, char **argv)
{
    // before the HIP runtime starts: a context runs a batch's parts on a dozen streams, HIP's default is four hardware queues
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    uint32_t reads = 16384, K = 10, read_len = 10000, jitter = 0, batches = 4, threads = 0;
    uint64_t ref_len = 64444167, seed = 4000;
    double n_frac = 0.01;
    int device = 0;
    const char *dump = nullptr;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        const char *v = argv[i + 1];
        if (a == "--reads") reads = (uint32_t)atol(v);
        else if (a == "--cands") K = (uint32_t)atol(v);
        else if (a == "--read-len") read_len = (uint32_t)atol(v);
        else if (a == "--jitter") jitter = (uint32_t)atol(v);
        else if (a == "--ref-len") ref_len = (uint64_t)atoll(v);
        else if (a == "--n-frac") n_frac = atof(v);
        else if (a == "--batches") batches = (uint32_t)atol(v);
        else if (a == "--seed") seed = (uint64_t)atoll(v);
        else if (a == "--threads") threads = (uint32_t)atol(v);
        else if (a == "--device") device = atoi(v);
        else if (a == "--dump") dump = v;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (K == 0 || reads == 0 || batches == 0) return 2;
    if (threads == 0) {
        threads = std::thread::hardware_concurrency();
        const char *env = std::getenv("OMP_NUM_THREADS");             // the box exports its CPU share this way
        if (env && atoi(env) > 0 && (uint32_t)atoi(env) < threads) threads = (uint32_t)atoi(env);
        if (threads == 0 || threads > 16) threads = 16;
    }
    const uint32_t n = reads * K;
    gba::MappingParameters params;

    std::vector<uint8_t> ref(ref_len);
    gbs_reference(ref.data(), ref_len, 20250218, n_frac, 1000);

    gba::HostPipeline pipe(device, 2, 0.4);
    gba_scoring sc = gba::ScoringFromParams(params);
    sc.n_extra = 1;
    ProbScorer(params, sc.extra[0]);
    int rc = pipe.SetScoring(sc);
    if (rc == GBA_OK) rc = pipe.SetReference(ref.data(), ref_len);
    gba_sdp_params sp{params.sdpTupleSize, params.sdpIns, params.sdpDel, -params.match, params.sdpMaxFragments};
    if (rc == GBA_OK) rc = pipe.SetGuideProducer(&sp);
    if (rc != GBA_OK) { std::fprintf(stderr, "setup: %s\n", gba_strerror(rc)); return 1; }

    // phase A stand-in: the candidates of `reads` reads (one true window and K - 1 decoy windows each, all viewing the read's
    // bytes), generated once per buffer and resubmitted; the generator's guides are NOT used (the device makes its own)
    gbs_cfg cfg{};
    cfg.read_len = read_len; cfg.p_ins = 0.09; cfg.p_del = 0.04; cfg.p_sub = 0.02; cfg.min_block = 11; cfg.drop_frac = 0.10;
    cfg.q_stride = read_len; cfg.t_stride = (uint32_t)(read_len * 1.35) + 256; cfg.blk_stride = read_len / 11 + 2;
    cfg.decoy_every = K > 1 ? K : 0; cfg.len_jitter = jitter; cfg.shard_rank = 0; cfg.shard_world = 1; cfg.share_reads = K > 1 ? 1 : 0;
    Batch B[2];
    {
        std::vector<uint8_t> t_arena((size_t)n * cfg.t_stride);
        std::vector<gbs_block> blocks((size_t)n * cfg.blk_stride);
        std::vector<uint64_t> blk_off(n), t_off(n);
        std::vector<uint32_t> n_blk(n);
        for (int s = 0; s < 2; s++) {
            Batch &b = B[s];
            b.n = n; b.q_bytes = (uint64_t)n * cfg.q_stride;
            b.q_arena = (uint8_t *)pipe.HostAlloc(b.q_bytes, s);
            b.q_off = (uint64_t *)pipe.HostAlloc((uint64_t)n * 8, s); b.t_off = (uint64_t *)pipe.HostAlloc((uint64_t)n * 8, s);
            b.q_len = (uint32_t *)pipe.HostAlloc((uint64_t)n * 4, s); b.t_len = (uint32_t *)pipe.HostAlloc((uint64_t)n * 4, s);
            b.t_rc = (uint8_t *)pipe.HostAlloc(n, s);
            b.pairs = (gba_pair_result *)pipe.HostAlloc((uint64_t)n * sizeof(gba_pair_result), s);
            if (!b.q_arena || !b.q_off || !b.t_off || !b.q_len || !b.t_len || !b.t_rc || !b.pairs) { std::fprintf(stderr, "pinned memory\n"); return 1; }
            cfg.seed = seed + (uint64_t)s;
            // the generator is single-threaded per call: split the pairs over the host threads (whole reads per chunk)
            std::vector<std::thread> th;
            std::atomic<uint32_t> bad{0};
            const uint32_t chunk = ((reads + threads - 1) / threads) * K;
            for (uint32_t first = 0; first < n; first += chunk)
                th.emplace_back([&, first] {
                    bad += gbs_generate(ref.data(), ref_len, &cfg, first, std::min(chunk, n - first), b.q_arena, b.q_off, b.q_len, t_arena.data(),
                                        t_off.data(), b.t_len, blocks.data(), blk_off.data(), n_blk.data(), b.t_off, b.t_rc);
                });
            for (auto &t : th) t.join();
            gba_batch &g = b.b;
            g = gba_batch{};
            g.n_pairs = n;
            g.q_arena = b.q_arena; g.q_off = b.q_off; g.q_len = b.q_len; g.q_arena_bytes = b.q_bytes;
            g.t_arena = nullptr; g.t_off = b.t_off; g.t_len = b.t_len; g.t_rc = b.t_rc;          // windows of the resident reference
            b.res = gba_results{};
            b.res.pairs = b.pairs;                                                                // interval output: no block arena
        }
    }

    std::string text[2];
    size_t printed_total = 0, bytes_total = 0;
    double t_phase_c = 0, t_wait = 0;
    auto phase_c = [&](int s, uint64_t read_base) {
        const double t0 = now();
        std::vector<std::string> parts(threads);
        std::vector<size_t> cnt(threads, 0);
        std::vector<std::thread> th;
        const uint32_t per = (reads + threads - 1) / threads;
        for (uint32_t t = 0; t < threads; t++)
            th.emplace_back([&, t] {
                const uint32_t r0 = std::min(reads, t * per), r1 = std::min(reads, r0 + per);
                cnt[t] = PhaseC(B[s], K, r0, r1, read_base, ref_len, params, parts[t]);
            });
        for (auto &x : th) x.join();
        text[s].clear();
        for (uint32_t t = 0; t < threads; t++) { text[s] += parts[t]; printed_total += cnt[t]; }      // in input order
        bytes_total += text[s].size();
        t_phase_c += now() - t0;
    };

    // warm-up: both contexts once (their device buffers grow to size)
    for (int s = 0; s < 2; s++) {
        if ((rc = pipe.Submit(B[s].b, &B[s].res)) != GBA_OK || (rc = pipe.Wait()) != GBA_OK) { std::fprintf(stderr, "warm-up: %s\n", gba_strerror(rc)); return 1; }
    }
    printed_total = 0; bytes_total = 0;
    const double t0 = now();
    if ((rc = pipe.Submit(B[0].b, &B[0].res)) != GBA_OK) { std::fprintf(stderr, "submit: %s\n", gba_strerror(rc)); return 1; }
    for (uint32_t k = 0; k < batches; k++) {
        const int s = (int)(k & 1);
        if (k + 1 < batches && (rc = pipe.Submit(B[s ^ 1].b, &B[s ^ 1].res)) != GBA_OK) { std::fprintf(stderr, "submit: %s\n", gba_strerror(rc)); return 1; }
        const double tw = now();
        if ((rc = pipe.Wait()) != GBA_OK) { std::fprintf(stderr, "wait: %s\n", gba_strerror(rc)); return 1; }
        t_wait += now() - tw;
        phase_c(s, (uint64_t)k * reads);             // while the GPU runs batch k + 1
    }
    const double dt = now() - t0;
    const int last = (int)((batches - 1) & 1);
    uint64_t ok = 0, cells = 0;
    for (uint32_t p = 0; p < n; p++) if (B[last].pairs[p].status == GBA_PAIR_OK) { ok++; cells += B[last].pairs[p].ncells; }
    if (dump) {
        FILE *f = std::fopen(dump, "w");
        if (!f) { std::fprintf(stderr, "cannot write %s\n", dump); return 1; }
        std::fwrite(text[last].data(), 1, text[last].size(), f);
        std::fclose(f);
    }
    std::printf("{\"reads_per_sec\": %.1f, \"candidates_per_sec\": %.1f, \"reads_per_batch\": %u, \"candidates_per_read\": %u, \"read_len\": %u, "
                "\"batches\": %u, \"seconds\": %.4f, \"wait_for_gpu_s\": %.4f, \"phase_c_s\": %.4f, \"host_threads\": %u, "
                "\"alignments_printed\": %zu, \"m4_bytes\": %zu, \"pairs_aligned_last_batch\": %llu, \"refine_gcups\": %.2f}\n",
                (double)reads * batches / dt, (double)n * batches / dt, reads, K, read_len, batches, dt, t_wait, t_phase_c, threads,
                printed_total, bytes_total, (unsigned long long)ok, (double)cells * batches / dt / 1e9);
    return 0;
}
