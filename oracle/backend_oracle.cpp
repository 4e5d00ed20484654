// backend_oracle.cpp — TEST INFRASTRUCTURE ONLY.  Binds the host call surface
// (prographmsa_amd/host) to the CPU oracle so that the scaffolding + oracle can be pinned against
// the reference binary's FASTA output (tests/test_oracle_golden.py).  Never linked into the product.
#include "../prographmsa_amd/host/pgm_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

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
    // PGM_ORACLE_RESIDENT=1: the "device-resident" profiles of the host scaffolding on this CPU backend — one arena of host memory per
    // worker, addresses handed out like device addresses, and EVERY use checked against the arena of the worker that runs the call: the
    // tests of the subtree-sharded pass (tests/test_cpu_host.py) fail on a profile matrix that was not copied to the worker using it.
    struct Arena { std::vector<std::pair<std::unique_ptr<double[]>, size_t>> blocks; };
    mutable std::mutex mu;
    std::vector<Arena> arenas;
    double *arena_alloc(int w, size_t count) {
        std::lock_guard<std::mutex> g(mu);
        if (arenas.size() <= (size_t)w) arenas.resize((size_t)w + 1);
        arenas[(size_t)w].blocks.emplace_back(std::unique_ptr<double[]>(new double[std::max<size_t>(count, 1)]()), count);
        return arenas[(size_t)w].blocks.back().first.get();
    }
    int arena_of(const double *p) const {   // worker whose arena holds p, -1: not a resident address
        std::lock_guard<std::mutex> g(mu);
        for (size_t w = 0; w < arenas.size(); ++w)
            for (const auto &b : arenas[w].blocks) if (p >= b.first.get() && p < b.first.get() + std::max<size_t>(b.second, 1)) return (int)w;
        return -1;
    }
    void check_here(const double *p, int w, const char *what) const {
        const int o = arena_of(p);
        if (o >= 0 && o != w) error("oracle backend: %s of worker %d used on worker %d", what, o, w);
    }
    bool resident() const override { return getenv("PGM_ORACLE_RESIDENT") != nullptr; }
    void resident_reset() override { std::lock_guard<std::mutex> g(mu); arenas.clear(); }
    bool resident_onehot(uint32_t dim, uint32_t nseq, const int8_t *syms, const uint32_t *offs, const double **dev, int w) override {
        for (uint32_t s = 0; s < nseq; ++s) {   // SequenceGraph.h:101-109: START, one column per residue (uniform without a value), END
            const uint32_t L = offs[s + 1] - offs[s];
            double *m = arena_alloc(w, (size_t)dim * (L + 2));
            for (uint32_t i = 0; i < L; ++i) {
                const int v = syms[offs[s] + i];
                for (uint32_t k = 0; k < dim; ++k) m[(size_t)dim * (i + 1) + k] = v < 0 ? 1.0 / dim : ((int)k == v ? 1.0 : 0.0);
            }
            dev[s] = m;
        }
        return true;
    }
    bool merge_profiles_batch_res(uint32_t njobs, const pgm_merge_job *jobs, const double **dev, int w) override {
        for (uint32_t i = 0; i < njobs; ++i) {
            pgm_merge_job j = jobs[i];
            check_here(j.sites1, w, "a child's profiles"); check_here(j.sites2, w, "a child's profiles");
            double *m = arena_alloc(w, (size_t)j.dim * j.nnodes);
            j.profiles = m;
            if (pgmo_merge_profiles(&j) != PGM_OK) error("pgmo_merge_profiles failed");
            dev[i] = m;
        }
        return true;
    }
    const double *resident_import(int dst, int src, const double *p, size_t count) override {
        if (arena_of(p) != src) error("oracle backend: resident_import of an address that is not worker %d's", src);
        double *m = arena_alloc(dst, count);
        memcpy(m, p, 8 * count);
        return m;
    }
    void align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                            const pgm_model *const *model, const pgm_scores *scores, pgm_align_out *out, int w, const pgm_site_ref *res1, const pgm_site_ref *res2) override {
        if (!res1 && !res2) { pgmo_align_graphs_batch(njobs, g1, g2, model, scores, out); return; }
        // resident graphs: the profiles gathered through the cleaned graphs' node maps, as the device does (include/pgm_hip.h: pgm_site_ref)
        std::vector<pgm_graph> f1(njobs), f2(njobs);
        std::vector<const pgm_graph *> p1(njobs), p2(njobs);
        std::vector<std::vector<double>> keep;
        keep.reserve(2 * (size_t)njobs);
        auto gather = [&](const pgm_graph &g, const pgm_site_ref *r) {
            pgm_graph f = g;
            if (r && r->dev_sites) {
                check_here(r->dev_sites, w, "resident profiles");
                keep.emplace_back((size_t)g.dim * g.n);
                for (uint32_t v = 0; v < g.n; ++v) {
                    const uint32_t c = r->node_map ? r->node_map[v] : v;
                    if (c >= r->ncols) error("oracle backend: node map beyond the resident matrix");
                    memcpy(&keep.back()[(size_t)g.dim * v], r->dev_sites + (size_t)g.dim * c, 8 * (size_t)g.dim);
                }
                f.sites = keep.back().data();
            }
            return f;
        };
        for (uint32_t i = 0; i < njobs; ++i) {
            f1[i] = gather(*g1[i], res1 ? &res1[i] : nullptr); f2[i] = gather(*g2[i], res2 ? &res2[i] : nullptr);
            p1[i] = &f1[i]; p2[i] = &f2[i];
        }
        pgmo_align_graphs_batch(njobs, p1.data(), p2.data(), model, scores, out);
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
    bool csprofile_create_batch_res(const CSProfile &lib, uint32_t nseq, const int8_t *syms, const uint32_t *offs, const double *tau, const double *pi,
                                    const double *p_uniform, const double **dev, int w) override {
        for (uint32_t s = 0; s < nseq; ++s) {
            double *m = arena_alloc(w, (size_t)20 * (offs[s + 1] - offs[s] + 2));
            pgmo_csprofile_create((uint32_t)lib.nprof(), (uint32_t)lib.ncols(), lib.lprofiles().data(), lib.centre().data(),
                                  lib.priors().data(), syms + offs[s], offs[s + 1] - offs[s], tau[s], pi, p_uniform + (size_t)s * 20, m);
            dev[s] = m;
        }
        return true;
    }
};
}  // namespace
Backend &default_backend() {
    static OracleBackend be;
    return be;
}
}  // namespace pgm
