// backend_oracle.cpp — TEST INFRASTRUCTURE ONLY.  Binds the host call surface
// (prographmsa_amd/host) to the CPU oracle so that the scaffolding + oracle can be pinned against
// the reference binary's FASTA output (tests/test_oracle_golden.py).  Never linked into the product.
#include "../prographmsa_amd/host/pgm_host.h"

#include <algorithm>
#include <cstdlib>

extern "C" {
int pgmo_align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                            const pgm_model *const *model, const pgm_scores *scores, pgm_align_out *out);
int pgmo_nw_pairs_batch(uint32_t dim, const int32_t *score, int32_t gap_open, int32_t gap_extend, uint32_t nseq,
                        const int8_t *syms, const uint32_t *offs, uint32_t npairs, const uint32_t *pi,
                        const uint32_t *pj, int32_t *counts, uint32_t *gaps);
int pgmo_csprofile_create(uint32_t K, uint32_t ncols, const double *lprofiles, const double *centre,
                          const double *priors, const int8_t *seq, uint32_t L, double tau, const double *pi,
                          const double *p_uniform, double *out);
int pgmo_kmer_cosine(uint32_t nseq, uint32_t ncols, const int32_t *counts, double *out);
int pgmo_merge_profiles(const pgm_merge_job *job);
int pgmo_prealigned_counts(uint32_t dim, uint32_t nrows, uint32_t ncols, const int8_t *rows, uint32_t npairs, const uint32_t *pi,
                           const uint32_t *pj, int32_t *counts, uint32_t *gaps);
int pgmo_mldist(const pgm_mldist_model *m, uint32_t npairs, const int32_t *counts, const uint32_t *gaps, const double *seqlen,
                double *dist, double *var);
}

namespace pgm {
namespace {
struct OracleBackend : Backend {
    const char *name() const override { return "oracle"; }
    void align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                            const pgm_model *const *model, const pgm_scores *scores, pgm_align_out *out, int, const pgm_site_ref *, const pgm_site_ref *) override {
        pgmo_align_graphs_batch(njobs, g1, g2, model, scores, out);
    }
    // PGM_FARM_WORKERS=k: the farms of the host scaffolding (all-pairs tiles, jobs of a guide-tree level, leaves) run with k
    // host threads over this (stateless, re-entrant) CPU oracle: the 1-vs-k-workers identity tests of tests/test_cpu_host.py
    int workers() const override { const char *e = getenv("PGM_FARM_WORKERS"); return e ? std::max(1, atoi(e)) : 1; }
    int nw_pairs_submit(uint32_t dim, const int32_t *score, int32_t go, int32_t ge, uint32_t nseq, const int8_t *syms,
                        const uint32_t *offs, uint32_t npairs, const uint32_t *pi, const uint32_t *pj, uint32_t flags, int32_t *counts,
                        uint32_t *gaps, int) override {
        if (flags & 1u) {   // PGM_NW_REDUCED: (ident, total) of the count matrix
            std::vector<int32_t> full((size_t)npairs * dim * dim);
            if (pgmo_nw_pairs_batch(dim, score, go, ge, nseq, syms, offs, npairs, pi, pj, full.data(), gaps) != PGM_OK) error("error while backtracking");
            for (uint32_t p = 0; p < npairs; ++p) {
                int32_t ident = 0, total = 0;
                for (uint32_t a = 0; a < dim; ++a)
                    for (uint32_t c = 0; c < dim; ++c) { total += full[((size_t)p * dim + c) * dim + a]; if (a == c) ident += full[((size_t)p * dim + c) * dim + a]; }
                counts[2 * (size_t)p] = ident; counts[2 * (size_t)p + 1] = total;
            }
        } else if (pgmo_nw_pairs_batch(dim, score, go, ge, nseq, syms, offs, npairs, pi, pj, counts, gaps) != PGM_OK)
            error("error while backtracking");
        return 0;
    }
    void nw_pairs_wait(int, int) override {}
    // f1 / f3 through the oracle's own restatements, so that the golden fixtures (FASTA, --profile_out, newick) pin them too
    void kmer_cosine(uint32_t nseq, uint32_t ncols, const int32_t *counts, double *cosine, int) override {
        if (pgmo_kmer_cosine(nseq, ncols, counts, cosine) != PGM_OK) error("pgmo_kmer_cosine failed");
    }
    bool merge_profiles_batch(uint32_t njobs, const pgm_merge_job *jobs, int) override {
        for (uint32_t i = 0; i < njobs; ++i)
            if (pgmo_merge_profiles(&jobs[i]) != PGM_OK) error("pgmo_merge_profiles failed");
        return true;
    }
    bool prealigned_counts_batch(uint32_t dim, uint32_t nrows, uint32_t ncols, const int8_t *rows, uint32_t npairs, const uint32_t *pi,
                                 const uint32_t *pj, int32_t *counts, uint32_t *gaps, int) override {
        if (pgmo_prealigned_counts(dim, nrows, ncols, rows, npairs, pi, pj, counts, gaps) != PGM_OK) error("pgmo_prealigned_counts failed");
        return true;
    }
    bool mldist_batch(const pgm_mldist_model &m, uint32_t npairs, const int32_t *counts, const uint32_t *gaps, const double *seqlen,
                      double *dist, double *var, int) override {   // (used with PGM_DEVICE_MLDIST=1, as in the product)
        if (pgmo_mldist(&m, npairs, counts, gaps, seqlen, dist, var) != PGM_OK) error("pgmo_mldist failed");
        return true;
    }
    void csprofile_create_batch(const CSProfile &lib, uint32_t nseq, const int8_t *syms, const uint32_t *offs,
                                const double *tau, const double *pi, const double *p_uniform, double *out,
                                const uint64_t *out_offs, int) override {
        for (uint32_t s = 0; s < nseq; ++s)
            pgmo_csprofile_create((uint32_t)lib.nprof(), (uint32_t)lib.ncols(), lib.lprofiles().data(), lib.centre().data(),
                                  lib.priors().data(), syms + offs[s], offs[s + 1] - offs[s], tau[s], pi,
                                  p_uniform + (size_t)s * 20, out + out_offs[s]);
    }
};
}  // namespace
Backend &default_backend() {
    static OracleBackend be;
    return be;
}
}  // namespace pgm
