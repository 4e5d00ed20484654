// backend_hip.cpp — binds the host call surface to libpgm_hip.so (the HIP kernels behind the C ABI
// of include/pgm_hip.h).  There is deliberately no CPU fallback: without a usable MI355X context
// every hot-path call fails loudly.
#include "pgm_host.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace pgm {
namespace {
struct HipBackend : Backend {
    // One context per device.  PGM_DEVICES="0,2,5" lists them explicitly, PGM_DEVICE=k selects one (what a one-process-
    // per-GPU launcher sets), otherwise every visible device is used.  Every batch call runs on the context the caller names
    // (`worker`): the host code deals a batch's independent units — jobs of a guide-tree level, leaves, merges, sequence
    // pairs — to the contexts, one host thread each (farm_shards / computePwDistances).
    std::vector<pgm_ctx *> ctxs;
    std::vector<const CSProfile *> loaded;   // the profile library resident on context w
    // Two contexts on ONE device (PGM_DEVICES=0,0: a test set-up) must not run their alignGraphs batches at the same time: every
    // batch sizes its persistent grids for the whole device (no grid of a stage waits for a CU).  One lock per device.
    std::map<int, std::unique_ptr<std::mutex>> device_lock;
    std::vector<int> device_of;
    HipBackend() {
        std::vector<int> devs;
        if (const char *e = getenv("PGM_DEVICES")) {
            for (const char *p = e; *p;) { devs.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        } else if (const char *e = getenv("PGM_DEVICE")) {
            devs.push_back(atoi(e));
        } else {
            const int n = pgm_device_count();
            for (int d = 0; d < std::max(n, 1); ++d) devs.push_back(d);
        }
        for (int d : devs) {
            pgm_ctx *c = nullptr;
            if (pgm_ctx_create(d, &c) != PGM_OK || !c)
                error("libpgm_hip: cannot create a context on device %d: %s", d, pgm_last_error());
            ctxs.push_back(c);
            device_of.push_back(d);
            if (!device_lock.count(d)) device_lock[d].reset(new std::mutex);
        }
        loaded.assign(ctxs.size(), nullptr);
    }
    pgm_ctx *ctx_of(int worker) const { return ctxs[(size_t)worker % ctxs.size()]; }
    ~HipBackend() override { for (pgm_ctx *c : ctxs) pgm_ctx_destroy(c); }
    const char *name() const override { return "hip"; }
    int workers() const override { return (int)ctxs.size(); }
    void align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                            const pgm_model *const *model, const pgm_scores *scores, pgm_align_out *out, int worker,
                            const pgm_site_ref *res1, const pgm_site_ref *res2) override {
        std::lock_guard<std::mutex> one_batch_per_device(*device_lock.at(device_of[(size_t)worker % ctxs.size()]));
        int rc = pgm_align_graphs_batch_res(ctx_of(worker), njobs, g1, g2, model, scores, res1, res2, out);
        if (rc != PGM_OK && rc != PGM_ERR_BACKTRACK) error("pgm_align_graphs_batch failed (%d): %s", rc, pgm_last_error());
    }
    // resident profiles (PGM_NO_RESIDENT switches them off): with several contexts the pass is sharded by subtree and the few matrices
    // a parent needs from another device are copied over (progressive.cpp assign_owners, resident_import)
    bool resident() const override { return !host_switches().no_resident; }
    const double *resident_import(int dst, int src, const double *p, size_t count) override {
        const double *out = nullptr;
        int rc = pgm_resident_import(ctx_of(dst), ctx_of(src), p, (uint64_t)count, &out);
        if (rc != PGM_OK) error("pgm_resident_import failed (%d): %s", rc, pgm_last_error());
        return out;
    }
    bool resident_onehot(uint32_t dim, uint32_t nseq, const int8_t *syms, const uint32_t *offs, const double **dev, int worker) override {
        int rc = pgm_resident_onehot(ctx_of(worker), dim, nseq, syms, offs, dev);
        if (rc != PGM_OK) error("pgm_resident_onehot failed (%d): %s", rc, pgm_last_error());
        return true;
    }
    void resident_reset() override { for (pgm_ctx *c : ctxs) (void)pgm_resident_reset(c); }
    bool merge_profiles_batch_res(uint32_t njobs, const pgm_merge_job *jobs, const double **dev, int worker) override {
        int rc = pgm_merge_profiles_batch_ex(ctx_of(worker), njobs, jobs, PGM_MERGE_RESIDENT, dev);
        if (rc != PGM_OK) error("pgm_merge_profiles_batch_ex failed (%d): %s", rc, pgm_last_error());
        return true;
    }
    int nw_pairs_submit(uint32_t dim, const int32_t *score, int32_t go, int32_t ge, uint32_t nseq, const int8_t *syms,
                        const uint32_t *offs, uint32_t npairs, const uint32_t *pi, const uint32_t *pj, uint32_t flags, int32_t *counts,
                        uint32_t *gaps, int worker) override {
        int ticket = -1;
        int rc = pgm_nw_pairs_submit(ctxs[(size_t)worker % ctxs.size()], dim, score, go, ge, nseq, syms, offs, npairs, pi, pj, flags, counts, gaps, &ticket);
        if (rc != PGM_OK) error("pgm_nw_pairs_submit failed (%d): %s", rc, pgm_last_error());
        return ticket;
    }
    void nw_pairs_wait(int ticket, int worker) override {
        int rc = pgm_nw_pairs_wait(ctxs[(size_t)worker % ctxs.size()], ticket);
        if (rc != PGM_OK) error("pgm_nw_pairs_wait failed (%d): %s", rc, pgm_last_error());
    }
    void *host_alloc(size_t bytes) override {
        void *p = pgm_host_alloc(bytes);
        if (!p) error("pgm_host_alloc(%zu) failed", bytes);
        return p;
    }
    void host_free(void *p) override { pgm_host_free(p); }
    bool mldist_batch(const pgm_mldist_model &m, uint32_t npairs, const int32_t *counts, const uint32_t *gaps, const double *seqlen,
                      double *dist, double *var, int worker) override {
        int rc = pgm_mldist_batch(ctxs[(size_t)worker % ctxs.size()], &m, npairs, counts, gaps, seqlen, dist, var);
        if (rc != PGM_OK) error("pgm_mldist_batch failed (%d): %s", rc, pgm_last_error());
        return true;
    }
    bool prealigned_counts_batch(uint32_t dim, uint32_t nrows, uint32_t ncols, const int8_t *rows, uint32_t npairs, const uint32_t *pi,
                                 const uint32_t *pj, int32_t *counts, uint32_t *gaps, int worker) override {
        int rc = pgm_prealigned_counts_batch(ctx_of(worker), dim, nrows, ncols, rows, npairs, pi, pj, counts, gaps);
        if (rc != PGM_OK) error("pgm_prealigned_counts_batch failed (%d): %s", rc, pgm_last_error());
        return true;
    }
    void kmer_cosine(uint32_t nseq, uint32_t ncols, const int32_t *counts, double *cosine, int worker) override {
        int rc = pgm_kmer_cosine(ctx_of(worker), nseq, ncols, counts, cosine);
        if (rc != PGM_OK) error("pgm_kmer_cosine failed (%d): %s", rc, pgm_last_error());
    }
    bool merge_profiles_batch(uint32_t njobs, const pgm_merge_job *jobs, int worker) override {
        int rc = pgm_merge_profiles_batch(ctx_of(worker), njobs, jobs);
        if (rc != PGM_OK) error("pgm_merge_profiles_batch failed (%d): %s", rc, pgm_last_error());
        return true;
    }
    void csprofile_create_batch(const CSProfile &lib, uint32_t nseq, const int8_t *syms, const uint32_t *offs,
                                const double *tau, const double *pi, const double *p_uniform, double *out,
                                const uint64_t *out_offs, int worker) override {
        pgm_ctx *ctx = ctx_of(worker);
        const size_t slot = (size_t)worker % ctxs.size();
        if (loaded[slot] != &lib) {
            int rc = pgm_csprofile_load(ctx, (uint32_t)lib.nprof(), (uint32_t)lib.ncols(), lib.lprofiles().data(),
                                        lib.centre().data(), lib.priors().data());
            if (rc != PGM_OK) error("pgm_csprofile_load failed (%d): %s", rc, pgm_last_error());
            loaded[slot] = &lib;
        }
        int rc = pgm_csprofile_create_batch(ctx, nseq, syms, offs, tau, pi, p_uniform, out, out_offs);
        if (rc != PGM_OK) error("pgm_csprofile_create_batch failed (%d): %s", rc, pgm_last_error());
    }
    void csprofile_preload(const CSProfile &lib) override {
        for (size_t slot = 0; slot < ctxs.size(); ++slot) {
            if (loaded[slot] == &lib) continue;
            int rc = pgm_csprofile_load(ctxs[slot], (uint32_t)lib.nprof(), (uint32_t)lib.ncols(), lib.lprofiles().data(), lib.centre().data(), lib.priors().data());
            if (rc != PGM_OK) error("pgm_csprofile_load failed (%d): %s", rc, pgm_last_error());
            loaded[slot] = &lib;
        }
    }
    bool csprofile_create_batch_res(const CSProfile &lib, uint32_t nseq, const int8_t *syms, const uint32_t *offs, const double *tau, const double *pi,
                                    const double *p_uniform, const double **dev, int worker) override {
        pgm_ctx *ctx = ctx_of(worker);
        const size_t slot = (size_t)worker % ctxs.size();
        if (loaded[slot] != &lib) {
            int rc = pgm_csprofile_load(ctx, (uint32_t)lib.nprof(), (uint32_t)lib.ncols(), lib.lprofiles().data(), lib.centre().data(), lib.priors().data());
            if (rc != PGM_OK) error("pgm_csprofile_load failed (%d): %s", rc, pgm_last_error());
            loaded[slot] = &lib;
        }
        int rc = pgm_csprofile_create_batch_res(ctx, nseq, syms, offs, tau, pi, p_uniform, dev);
        if (rc != PGM_OK) error("pgm_csprofile_create_batch_res failed (%d): %s", rc, pgm_last_error());
        return true;
    }
};
}  // namespace

Backend &default_backend() {
    static HipBackend be;
    return be;
}
}  // namespace pgm
