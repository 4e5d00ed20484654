// backend_hip.cpp — binds the host call surface to libpgm_hip.so (the HIP kernels behind the C ABI
// of include/pgm_hip.h).  There is deliberately no CPU fallback: without a usable MI355X context
// every hot-path call fails loudly.
#include "pgm_host.h"

#include <cstdlib>

namespace pgm {
namespace {
struct HipBackend : Backend {
    pgm_ctx *ctx = nullptr;
    const CSProfile *loaded = nullptr;
    HipBackend() {
        int dev = 0;
        if (const char *e = getenv("PGM_DEVICE")) dev = atoi(e);
        if (pgm_ctx_create(dev, &ctx) != PGM_OK || !ctx)
            error("libpgm_hip: cannot create a context on device %d: %s", dev, pgm_last_error());
    }
    ~HipBackend() override { if (ctx) pgm_ctx_destroy(ctx); }
    const char *name() const override { return "hip"; }
    void align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                            const pgm_model *const *model, const pgm_scores *scores, pgm_align_out *out) override {
        int rc = pgm_align_graphs_batch(ctx, njobs, g1, g2, model, scores, out);
        if (rc != PGM_OK && rc != PGM_ERR_BACKTRACK) error("pgm_align_graphs_batch failed (%d): %s", rc, pgm_last_error());
    }
    void nw_pairs_batch(uint32_t dim, const int32_t *score, int32_t go, int32_t ge, uint32_t nseq, const int8_t *syms,
                        const uint32_t *offs, uint32_t npairs, const uint32_t *pi, const uint32_t *pj, int32_t *counts,
                        uint32_t *gaps) override {
        int rc = pgm_nw_pairs_batch(ctx, dim, score, go, ge, nseq, syms, offs, npairs, pi, pj, counts, gaps);
        if (rc != PGM_OK) error("pgm_nw_pairs_batch failed (%d): %s", rc, pgm_last_error());
    }
    void csprofile_create_batch(const CSProfile &lib, uint32_t nseq, const int8_t *syms, const uint32_t *offs,
                                const double *tau, const double *pi, const double *p_uniform, double *out,
                                const uint64_t *out_offs) override {
        if (loaded != &lib) {
            int rc = pgm_csprofile_load(ctx, (uint32_t)lib.nprof(), (uint32_t)lib.ncols(), lib.lprofiles().data(),
                                        lib.centre().data(), lib.priors().data());
            if (rc != PGM_OK) error("pgm_csprofile_load failed (%d): %s", rc, pgm_last_error());
            loaded = &lib;
        }
        int rc = pgm_csprofile_create_batch(ctx, nseq, syms, offs, tau, pi, p_uniform, out, out_offs);
        if (rc != PGM_OK) error("pgm_csprofile_create_batch failed (%d): %s", rc, pgm_last_error());
    }
};
}  // namespace

Backend &default_backend() {
    static HipBackend be;
    return be;
}
}  // namespace pgm
