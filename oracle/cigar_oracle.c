/*
 * cigar_oracle.c -- CPU restatement of the CIGAR of a refined alignment.  TEST INFRASTRUCTURE ONLY:
 * nothing under blasr_amd/ may call this; it is the checker of the device's k_cigar and of the host
 * writer blasr_amd/emit.py, written against neither.
 *
 * What it restates: the SAM / BAM writers the reference calls at
 * /root/reference/iblasr/BlasrUtilsImpl.hpp:969-974 (SAMOutput / BAMOutput::PrintAlignment with
 * params.cigarUseSeqMatch and params.allowAdjacentIndels) are libcpp code that the reference tree does
 * not hold, so parity with them is UNPINNED.  What the reference's own files show of their output:
 * the alphabet = X I D of the two literal strings at ctest/bamOut.t:21-22; one M per gap-free run
 * without cigarUseSeqMatch; "adjacent insertion and deletions are merged into one operation" unless
 * -allowAdjacentIndels (iblasr/RegisterBlasrOptions.h:254-259; ctest/cigarAdjecentIndels.t:12-16 greps
 * the output for ID / DI).
 *
 * Method (deliberately not the block-by-block form of the writers under test): the alignment is spelt
 * out column by column into a string over {a: aligned column, i: query base against a gap, d: target
 * base against a gap}; a run of i directly followed by a run of d (one gap between two blocks) is
 * rewritten so that the part both runs have in common becomes aligned columns, which come first; then
 * aligned columns are classified by comparing base codes (case folded, every letter outside ACGT is N
 * and N equals N) and the string is run-length encoded into BAM operations (length << 4 | op).
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint32_t q_pos, t_pos, length; } cgo_block;

enum { CGO_M = 0, CGO_I = 1, CGO_D = 2, CGO_EQ = 7, CGO_X = 8 };
#define CGO_SEQ_MATCH 1u
#define CGO_ADJACENT_INDELS 2u

static int code_of(uint8_t c)
{
    switch (c & 0xdf) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

/* Returns the number of operations (written up to cap), or -1 for blocks that overlap / go backwards. */
int cgo_cigar(const uint8_t *q, const uint8_t *t, const cgo_block *b, int n_blk, uint32_t q_pos, uint32_t t_pos,
              uint32_t flags, uint32_t *ops, int cap)
{
    if (n_blk <= 0) return 0;
    /* 1. columns */
    uint64_t ncol = 0;
    for (int k = 0; k < n_blk; k++) {
        ncol += b[k].length;
        if (k + 1 < n_blk) {
            if (b[k + 1].q_pos < b[k].q_pos + b[k].length || b[k + 1].t_pos < b[k].t_pos + b[k].length) return -1;
            ncol += (b[k + 1].q_pos - b[k].q_pos - b[k].length) + (b[k + 1].t_pos - b[k].t_pos - b[k].length);
        }
    }
    char *col = (char *)malloc(ncol + 1);
    uint64_t n = 0;
    for (int k = 0; k < n_blk; k++) {
        for (uint32_t x = 0; x < b[k].length; x++) col[n++] = 'a';
        if (k + 1 < n_blk) {
            const uint32_t gi = b[k + 1].q_pos - b[k].q_pos - b[k].length, gd = b[k + 1].t_pos - b[k].t_pos - b[k].length;
            for (uint32_t x = 0; x < gi; x++) col[n++] = 'i';
            for (uint32_t x = 0; x < gd; x++) col[n++] = 'd';
        }
    }
    /* 2. merge an insertion run that touches a deletion run */
    if (!(flags & CGO_ADJACENT_INDELS)) {
        uint64_t w = 0, r = 0;
        while (r < n) {
            if (col[r] != 'i') { col[w++] = col[r++]; continue; }
            uint64_t ni = 0, nd = 0;
            while (r + ni < n && col[r + ni] == 'i') ni++;
            while (r + ni + nd < n && col[r + ni + nd] == 'd') nd++;
            const uint64_t common = ni < nd ? ni : nd;
            for (uint64_t x = 0; x < common; x++) col[w++] = 'a';
            for (uint64_t x = common; x < ni; x++) col[w++] = 'i';
            for (uint64_t x = common; x < nd; x++) col[w++] = 'd';
            r += ni + nd;
        }
        n = w;
    }
    /* 3. walk the columns along both sequences, classify, run-length encode */
    uint64_t qi = (uint64_t)q_pos + b[0].q_pos, ti = (uint64_t)t_pos + b[0].t_pos;
    int nops = 0;
    uint32_t run = 0, run_op = 99;
    for (uint64_t x = 0; x < n; x++) {
        uint32_t op;
        if (col[x] == 'i') { op = CGO_I; qi++; }
        else if (col[x] == 'd') { op = CGO_D; ti++; }
        else {
            op = !(flags & CGO_SEQ_MATCH) ? CGO_M : code_of(q[qi]) == code_of(t[ti]) ? CGO_EQ : CGO_X;
            qi++; ti++;
        }
        if (op == run_op) { run++; continue; }
        if (run) { if (nops < cap) ops[nops] = run << 4 | run_op; nops++; }
        run = 1; run_op = op;
    }
    if (run) { if (nops < cap) ops[nops] = run << 4 | run_op; nops++; }
    free(col);
    return nops;
}
