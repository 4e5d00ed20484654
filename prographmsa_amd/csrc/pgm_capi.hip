// pgm_capi.hip — host side of the C ABI declared in include/pgm_hip.h (libpgm_hip.so).
// Flattens the caller's graphs into HBM-resident job descriptors and launches the HIP kernels of
// pgm_align_kernels.h / pgm_nw_kernels.h / pgm_csprofile_kernels.h.  No CPU fallback exists: every
// entry point needs a working gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <thread>
#include <atomic>
#include <chrono>
#include <queue>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>
#include <sys/mman.h>

#include "pgm_align_kernels.h"
#include "pgm_crit_kernels.h"
#include "pgm_nw_kernels.h"
#include "pgm_csprofile_kernels.h"
#include "pgm_dist_kernels.h"
#include "pgm_merge_kernels.h"
#include "pgm_pool.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &m) { g_err = m; return code; }
// Experiment switches of tools/ (timeline, strip-down variants, work-list parameters) exist only in the tools build of the
// library (make tools: -DPGM_TOOLS, lib/libpgm_hip_tools.so); the release library reads none of them.
#ifdef PGM_TOOLS
static const char *tools_env(const char *k) { return getenv(k); }
#else
static const char *tools_env(const char *) { return nullptr; }
#endif
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
            return fail(PGM_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_));            \
    } while (0)

struct PgmNwState;
struct pgm_ctx {
    int device = 0;
    PgmNwState *nw = nullptr;        // the all-pairs stage's two tiles in flight (pgm_nw_capi.inc)
    int nw_per_cu = 0;
    // resident merge results (pgm_merge_profiles_batch_ex): bump allocation over chunks that live until pgm_resident_reset
    // (a guide tree is not balanced: a level-1 graph may wait many levels for its sibling)
    struct ResChunk { uint8_t *p; size_t cap; };
    std::vector<ResChunk> res_chunks;
    size_t res_chunk = 0, res_off = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // the lean kernel runs beside the fill kernel (pgm_lean_kernel)
    hipStream_t stream3 = nullptr;   // ... and so does the band kernel (pgm_band_kernel)
    hipStream_t stream4 = nullptr;   // ... and the fill kernel's launch for the longest chains
    hipStream_t stream5 = nullptr;   // ... and the traceback kernel that runs beside them
    // The big buffers of a destroyed batch are kept for the next one (a progressive alignment issues one batch per tree
    // level: hipMalloc / hipFree of several GB per call would dominate the call).  Slot k holds at most one buffer.
    enum { C_IN, C_WORK, C_CELLS, C_OUT, C_S, C_HOST, C_HIN, C_SMALL, C_SLOTS };   // C_HOST, C_HIN: pinned host memory; C_SMALL: the batch's counters, job descriptors, work list
    void *cache_ptr[C_SLOTS] = {};
    size_t cache_bytes[C_SLOTS] = {};
    hipDeviceProp_t prop;
    float nw_ms = 0, cs_ms = 0, ml_ms = 0, merge_ms = 0;
    // grow-only scratch buffers of the all-pairs / context-profile calls (slot = position in the call's buffer list): a
    // guide-tree stage issues many calls (one per pair tile), hipMalloc / hipFree of up to 2 GB per call would dominate them
    enum { SC_DEV = 24, SC_HOST = 4 };
    void *sc_dev[SC_DEV] = {};
    size_t sc_dev_bytes[SC_DEV] = {};
    void *sc_host[SC_HOST] = {};      // pinned
    size_t sc_host_bytes[SC_HOST] = {};
    hipEvent_t sc_ev[2] = {nullptr, nullptr};
    // context-profile library resident in HBM
    uint32_t csK = 0, csC = 0;
    double *cs_lprofiles = nullptr, *cs_centre = nullptr, *cs_priors = nullptr;
};

// the library's host threads (flattening and chunked upload of the jobs, staging copies, first touch of pinned blocks)
static pgm_pool::Pool &lib_pool() {
    static pgm_pool::Pool pool(std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    return pool;
}

// ---- pinned host blocks ---------------------------------------------------------------------------
// hipHostMalloc of the staging block of a 128-job level (76 MB) takes 10-14 ms in a cold process and stalls every other
// thread that touches the address space meanwhile (hipHostFree: another 8 ms).  A 2 MB aligned block with MADV_HUGEPAGE,
// first touched by a few threads and registered afterwards, is pinned in ~1 ms (tools/micro/pin_bench.hip: fill 0.9 ms,
// hipHostRegister 0.2 ms, same copy rate).  Blocks that could not be registered fall back to hipHostMalloc.
static std::mutex g_pinned_mu;
static std::unordered_set<void *> g_pinned_registered;
static hipError_t pinned_alloc(size_t bytes, void **out) {
    const size_t H = (size_t)2 << 20, len = (std::max<size_t>(bytes, 1) + H - 1) / H * H;
    void *p = getenv("PGM_PINNED_MALLOC") ? nullptr : aligned_alloc(H, len);
    if (p) {
        (void)madvise(p, len, MADV_HUGEPAGE);
        (void)lib_pool().run(len / H, len >= 8 * H ? 16u : 1u, [&](size_t c) { for (size_t o = 0; o < H; o += 4096) ((volatile char *)p)[c * H + o] = 0; });
        if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
            std::lock_guard<std::mutex> g(g_pinned_mu);
            g_pinned_registered.insert(p);
            *out = p;
            return hipSuccess;
        }
        (void)hipGetLastError();
        free(p);
    }
    return hipHostMalloc(out, bytes, hipHostMallocDefault);
}
static void pinned_free(void *p) {
    if (!p) return;
    bool reg;
    { std::lock_guard<std::mutex> g(g_pinned_mu); reg = g_pinned_registered.erase(p) != 0; }
    if (reg) { (void)hipHostUnregister(p); free(p); }
    else (void)hipHostFree(p);
}
static void slot_free(int slot, void *p) {
    if (slot == pgm_ctx::C_HIN) pinned_free(p);
    else if (slot == pgm_ctx::C_HOST) (void)hipHostFree(p);
    else (void)hipFree(p);
}

// ---- arena: one host staging buffer mirrored by one device allocation ------------------------
struct Arena {   // bump allocator over a slice [off, end) of an external host buffer; offsets are relative to `base`
    uint8_t *base = nullptr;
    size_t off = 0, end = 0;
    bool overflow = false;
    size_t put(const void *src, size_t bytes, size_t align = 16) {
        const size_t o = (off + align - 1) / align * align;
        if (o + bytes > end) { overflow = true; return 0; }
        off = o + bytes;
        if (src && bytes) memcpy(base + o, src, bytes);
        return o;
    }
};
struct DevLayout {  // sizes of device-only regions
    size_t bytes = 0;
    size_t take(size_t b, size_t align = 256) {
        size_t off = (bytes + align - 1) / align * align;
        bytes = off + b;
        return off;
    }
};

#define PGM_LEAN_RSHIFT_DEFAULT 1   /* rows per lane of the lean sweep: R = 1 << rshift (the release library: 2) */
struct pgm_align_batch {
    uint32_t njobs = 0;
    uint64_t cells = 0;
    uint32_t maxdim = 0, maxnb = 0, maxn = 0;
    std::vector<PgmJob> jobs;         // host copy of the descriptors (device pointers inside)
    std::vector<uint32_t> order;      // launch order: largest job first
    uint8_t *d_in = nullptr;          // uploaded inputs (arena image)
    uint8_t *d_work = nullptr;        // prep outputs, brow, maps, results, scratch
    uint8_t *d_cells = nullptr;       // DP storage
    uint8_t *d_out = nullptr;         // device working copy of results + mappings (the walk pushes them in reverse order)
    size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // real sizes of the buffers taken from the context's cache
    uint8_t *h_out = nullptr;          // pinned result block, same layout: written by the kernel itself, read by fetch
    uint8_t *h_in = nullptr;           // pinned staging buffer of the flattened inputs (one H2D copy per create)
    int *h_flag = nullptr;            // (inside h_out, after the results)
    uint8_t *d_small = nullptr;       // d_sync, d_jobs, d_order, d_items live in this one cached allocation (no hipMalloc / hipFree per batch)
    uint8_t *d_S = nullptr;           // emission scores in fill order
    int *d_sync = nullptr;            // [0] abort flag, [1] band-list ticket, then the per-band progress counters of every job
    size_t sync_ints = 0, s_bytes = 0;
    PgmItem *d_items = nullptr;       // band list of the batch (fill work queue)
    uint32_t *d_lean = nullptr;       // the lean jobs (pgm_lean_kernel's work queue), largest first
    uint32_t nlean = 0, nlean_workers = 0;
    uint32_t nbands = 0, nband_workers = 0;   // pgm_band_kernel: bands of the MODE 0 / 1 jobs, one per wavefront; its workers (CUs)
    uint32_t nbands_narrow = 0, nwide_workers = 0;   // ... the first nbands_narrow of the list: sweeps that fit an eighth of a CU's LDS; the rest: a quarter, swept by the last nwide_workers workers (four wavefronts each)
    PgmItem *d_bands = nullptr;
    unsigned long long *d_times = nullptr;   // per job {last band complete, traceback published}, then the launch's start (ticks of 10 ns)
    hipEvent_t ev_join_b = nullptr;
    uint32_t ntb_beside_workers = 0, tbq_off = 0;   // pgm_tb_kernel beside the sweeps: its workers (0: the tracebacks follow their launches), its ready queue inside d_sync
    hipEvent_t ev_join_t = nullptr;
    bool crit_c3 = false, rest_c3 = false;   // every item of the launch for the longest chains / of the main launch belongs to a crit3 job: pgm_crit_kernel sweeps that list
    uint32_t ncrit = 0, ncrit_workers = 0, ntb_c = 0;   // the first ncrit items of the work list: the jobs with the longest chains, swept by a launch of their own on their own CUs; their tracebacks
    hipEvent_t ev_join_c = nullptr;
    uint32_t ntb_b = 0, ntb_b_workers = 0;          // ... of the jobs pgm_band_kernel sweeps: their instance of pgm_tb_kernel follows it on its stream
    uint32_t lq_off = 0, ntb = 0, ntb_workers = 0;   // pre-link announcements inside d_sync; jobs of the general path (one traceback each); workers of pgm_tb_kernel
    int2 *d_tblist = nullptr;         // those jobs, largest first: (job, its last item of the work list)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // stream -> stream2 after the emission kernel, stream2 -> stream after the lean kernel
    unsigned long long *d_trace = nullptr;   // PGM_FILL_TRACE=file: per-item timeline, written by fetch (tools only)
    unsigned long long *d_c3dbg = nullptr;   // PGM_C3_DBG (tools build): wait statistics of pgm_crit_kernel, 64 words per item, printed by fetch
    uint32_t test_spin_limit = 0, test_stall_job = 0xFFFFFFFFu, test_stall_band = 0;   // pgm_align_batch_test_stall
    double acc_ms[3] = {0, 0, 0};     // device time of prep / emission / fill (+ lean kernel + tracebacks) summed over the launches fetched since the last reset
    uint32_t acc_n = 0;
    bool ev_pending = false;          // the last launch recorded its stage events and they have not been read yet
    uint32_t nitems = 0, lean_rshift = PGM_LEAN_RSHIFT_DEFAULT;
    uint32_t nworkers = 0, maxnblk = 0;
    PgmJob *d_jobs = nullptr;
    uint32_t *d_order = nullptr;
    int *d_tabhdr = nullptr;   // class headers of the lean jobs (PgmJob::tabhdr), PGM_TAB_HDR ints per job of the batch; NULL: no lean job
    size_t in_bytes = 0, work_bytes = 0, cell_bytes = 0, out_bytes = 0;
    std::vector<size_t> res_off, map1_off, map2_off;  // offsets inside d_out
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

extern "C" {

const char *pgm_last_error(void) { return g_err.c_str(); }

int pgm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// (launched once per context: the device code of the library is loaded when the context is created, not in the middle of
// the first batch)
__global__ void pgm_warm_kernel() {}

int pgm_ctx_create(int device, pgm_ctx **out) {
    if (!out) return fail(PGM_ERR_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(PGM_ERR_DEVICE, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    pgm_ctx *c = new pgm_ctx;
    c->device = device;
    HIPCHK(hipGetDeviceProperties(&c->prop, device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream5, hipStreamNonBlocking));
    {   // (the launch of the longest chains on the highest stream priority the device offers)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if (hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, hi) != hipSuccess) {   // (no priorities: an ordinary stream)
            (void)hipGetLastError();
            HIPCHK(hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking));
        }
    }
    hipLaunchKernelGGL(pgm_warm_kernel, dim3(1), dim3(64), 0, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    {   // ... and so is the copy path: the first host-to-device copy of a process takes ~8 ms longer than any later one
        void *h = nullptr, *d = nullptr;
        const size_t wb = (size_t)1 << 20;
        if (hipHostMalloc(&h, wb, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, wb) == hipSuccess) {
            (void)hipMemcpyAsync(d, h, wb, hipMemcpyHostToDevice, c->stream);
            (void)hipMemcpyAsync(h, d, wb, hipMemcpyDeviceToHost, c->stream);
            (void)hipStreamSynchronize(c->stream);
        }
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
    }
    // ... and the library's host threads (flattening, staging copies, first touch of pinned blocks): sixteen thread starts are 0.4 ms
    // of whatever call needs them first
    (void)lib_pool().run(64, 16u, [](size_t) {});
    {   // ... and a first staging block for the inputs of a batch in the context's cache (128 MB: a level of 128 jobs of 1000 x 1000
        // takes 37, one of 512 jobs of 600 x 600 95; a larger batch replaces it — 4 ms for the release of a pinned block): allocating, touching and registering it is 0.8-1 ms of the first batch otherwise
        void *p = nullptr;
        const size_t sb = (size_t)128 << 20;
        if (!getenv("PGM_NO_STAGING_RESERVE") && pinned_alloc(sb, &p) == hipSuccess) { c->cache_ptr[pgm_ctx::C_HIN] = p; c->cache_bytes[pgm_ctx::C_HIN] = sb; }
        else (void)hipGetLastError();
        // (and for the results the kernels write over PCIe: 8 MB hold the mappings of 128 jobs of 1000 x 1000 four times over)
        const size_t rb = (size_t)8 << 20;
        if (!getenv("PGM_NO_STAGING_RESERVE") && hipHostMalloc(&p, rb, hipHostMallocDefault) == hipSuccess) { c->cache_ptr[pgm_ctx::C_HOST] = p; c->cache_bytes[pgm_ctx::C_HOST] = rb; }
        else (void)hipGetLastError();
        // The device side of the same cache: 3.4 GB of the 288 (inputs 128 MB, prep outputs / codes 256 MB, DP cells 2 GB, results 8 MB,
        // emission scores 1 GB, descriptors 4 MB) — what the levels of a 256 x 1000 or a 1024 x 600 pass take; a larger batch replaces a block.  On
        // most hosts of the pool these six hipMalloc calls take 0.3 ms together; on some (or in some states of a host) the driver
        // hands out device memory at ~30 ms per GB, and the first two levels of a pass then waited 50-70 ms for their buffers
        // (DESIGN section 4): a runtime pays that when it starts, not in the middle of its first call.  PGM_NO_DEVICE_RESERVE=1: off.
        if (!getenv("PGM_NO_STAGING_RESERVE") && !getenv("PGM_NO_DEVICE_RESERVE")) {
            static const struct { int slot; size_t mb; } pool[] = {{pgm_ctx::C_IN, 128}, {pgm_ctx::C_WORK, 256}, {pgm_ctx::C_CELLS, 2048}, {pgm_ctx::C_OUT, 8}, {pgm_ctx::C_S, 1024}, {pgm_ctx::C_SMALL, 4}};
            for (const auto &e : pool) {
                void *d = nullptr;
                if (hipMalloc(&d, e.mb << 20) == hipSuccess) { c->cache_ptr[e.slot] = d; c->cache_bytes[e.slot] = e.mb << 20; }
                else { (void)hipGetLastError(); break; }
            }
        }
    }
    *out = c;
    return PGM_OK;
}

static void nw_state_free(PgmNwState *st);
void pgm_ctx_destroy(pgm_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    nw_state_free(ctx->nw);
    for (auto &c : ctx->res_chunks) if (c.p) (void)hipFree(c.p);
    if (ctx->cs_lprofiles) (void)hipFree(ctx->cs_lprofiles);
    if (ctx->cs_centre) (void)hipFree(ctx->cs_centre);
    if (ctx->cs_priors) (void)hipFree(ctx->cs_priors);
    for (int k = 0; k < pgm_ctx::SC_DEV; ++k) if (ctx->sc_dev[k]) (void)hipFree(ctx->sc_dev[k]);
    for (int k = 0; k < pgm_ctx::SC_HOST; ++k) if (ctx->sc_host[k]) pinned_free(ctx->sc_host[k]);
    for (int k = 0; k < 2; ++k) if (ctx->sc_ev[k]) (void)hipEventDestroy(ctx->sc_ev[k]);
    for (int k = 0; k < pgm_ctx::C_SLOTS; ++k)
        if (ctx->cache_ptr[k]) slot_free(k, ctx->cache_ptr[k]);
    if (ctx->stream5) (void)hipStreamDestroy(ctx->stream5);
    if (ctx->stream4) (void)hipStreamDestroy(ctx->stream4);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pgm_ctx_device_info(pgm_ctx *ctx, char *name, size_t name_len, int *cu_count) {
    if (!ctx) return fail(PGM_ERR_INVALID, "null ctx");
    if (name && name_len) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    return PGM_OK;
}

}  // extern "C"

// ---- flattening of one graph side -------------------------------------------------------------
namespace {
struct SideOff {
    size_t sites, ni, xp, xc, xv, pp, pc, pv, pu, fp, fe, ov;
    size_t smap = 0;              // resident profiles: the node -> column map of the side
    bool has_smap = false;
    uint32_t nodes_with_extras;   // nodes with a predecessor other than the chain neighbour
    uint32_t has_long;            // some edge outside the near slots is longer than PGM_DCAP
    uint32_t maxd_cap;            // largest distance <= PGM_DCAP of an edge outside the chain slot (>= 1)
    uint32_t maxd_kf8;            // ... among the nodes with at most PGM_KF8 far candidates, none of them long
    // set by finalize_side, once the job's sweep mode is known:
    uint32_t far_nodes;           // nodes with entries served from the LDS history
    uint32_t maxd;                // largest on-chip predecessor distance of the graph (>= 1)
    uint32_t far_dmin;            // smallest distance of a far entry (PGM_DCAP + 1 if there is none)
    uint32_t remote;              // MODE 2: entries served from the cell storage by the far helpers
    uint32_t nov;                 // MODE 2, columns: records of the overflow table in use
    uint32_t ngeneric;            // nodes served by the generic path
    uint32_t nkill;               // interior nodes without predecessors
    // host only: far candidates (every finite edge outside the near slots) of node v: [cp[v], cp[v+1])
    std::vector<uint32_t> cp, cd;
    std::vector<float> cv;
};

static int flatten_side(const pgm_graph *g, const pgm_scores &sc, Arena &A, SideOff &o, const pgm_site_ref *res = nullptr) {
    const uint32_t n = g->n;
    const bool resident = res && res->dev_sites;
    if (n < 2 || (!g->sites && !resident) || !g->e_rowptr) return PGM_ERR_INVALID;
    // (scratch of the calling pool thread, kept between jobs: sixteen threads allocating and freeing ~150 KB per side
    // contend for the address space with the allocation thread's hipMalloc)
    static thread_local std::vector<float> xv, pv;
    static thread_local std::vector<int32_t> xp, pp;
    static thread_local std::vector<uint32_t> xc, pc, pu;
    static thread_local std::vector<PgmNode2> ni;
    xv.clear(); pv.clear(); xc.clear(); pc.clear(); pu.clear();
    xp.assign(n + 1, 0); pp.assign(n + 1, 0);
    ni.resize(n);
    o.nodes_with_extras = 0; o.has_long = 0; o.maxd_cap = 1; o.maxd_kf8 = 1; o.nkill = 0;
    o.cp.assign(n + 1, 0); o.cd.clear(); o.cv.clear();
    for (uint32_t v = 0; v < n; ++v) {
        PgmNode2 &I = ni[v];
        memset(&I, 0, sizeof I);
        I.cc = I.c2 = I.c3 = INFINITY;
        for (int k = 0; k < PGM_KF8; ++k) I.fc[k] = INFINITY;
        // near slots: the first finite-cost edge from node-1 / node-2 / node-3; everything else is a far candidate (an edge
        // of infinite cost contributes -inf to every maximum: it only stays in the CSR lists)
        auto place = [&](uint32_t from, float val) {
            const uint32_t d = v - from;
            if (d == 1 && I.cc == INFINITY && val != INFINITY) { I.cc = val; return; }
            xc.push_back(from); xv.push_back(val);
            if (val == INFINITY) return;
            if (d <= (uint32_t)PGM_DCAP) o.maxd_cap = std::max(o.maxd_cap, d); else o.has_long = 1;
            if (d == 2 && I.c2 == INFINITY) { I.c2 = val; return; }
            if (d == 3 && I.c3 == INFINITY) { I.c3 = val; return; }
            o.cd.push_back(d); o.cv.push_back(val);
        };
        const int32_t eb = g->e_rowptr[v], ee = g->e_rowptr[v + 1];
        if (eb > ee || eb < 0) return PGM_ERR_INVALID;
        for (int32_t e = eb; e < ee; ++e) {
            const uint32_t from = g->e_col[e];
            if (from >= v) return PGM_ERR_INVALID;  // edges must point to earlier nodes (Graph.h:43, GraphAlign.h:631-656)
            const float c = g->e_val[e];
            const float val = (c == 0) ? INFINITY : c + 10000.0f;  // PredIterator::value, Graph.h:223-231
            pc.push_back(from); pv.push_back(val); pu.push_back(0u);
            place(from, val);
        }
        if (g->r_rowptr) {
            if (g->r_rowptr[v] > g->r_rowptr[v + 1] || g->r_rowptr[v] < 0) return PGM_ERR_INVALID;
            for (int32_t e = g->r_rowptr[v]; e < g->r_rowptr[v + 1]; ++e) {
                const uint32_t from = g->r_col[e];
                if (from >= v) return PGM_ERR_INVALID;
                const uint32_t units = g->r_units[e];
                const float val = (units == 0) ? INFINITY : sc.repeat_init + sc.repeat_ext * (float)(units - 1);  // Graph.h:232-238
                pc.push_back(from); pv.push_back(val); pu.push_back(0x80000000u | units);
                place(from, val);
            }
        }
        xp[v + 1] = (int32_t)xc.size();
        pp[v + 1] = (int32_t)pc.size();
        o.cp[v + 1] = (uint32_t)o.cd.size();
        {
            uint32_t dm = I.c3 != INFINITY ? 3u : (I.c2 != INFINITY ? 2u : 1u);
            bool small = o.cp[v + 1] - o.cp[v] <= (uint32_t)PGM_KF8;
            for (uint32_t k = o.cp[v]; k < o.cp[v + 1] && small; ++k) { if (o.cd[k] > (uint32_t)PGM_DCAP) small = false; else dm = std::max(dm, o.cd[k]); }
            if (small) o.maxd_kf8 = std::max(o.maxd_kf8, dm);
        }
        if (v > 0 && v + 1 < n && pp[v + 1] == pp[v]) { I.flags |= PGM_NF_KILL; ++o.nkill; }  // interior node without predecessors
        o.nodes_with_extras += (xp[v + 1] > xp[v]);
    }
    // at least one element each so that pointers are valid
    if (xc.empty()) { xc.push_back(0); xv.push_back(0); }
    if (pc.empty()) { pc.push_back(0); pv.push_back(0); pu.push_back(0); }
    o.smap = 0; o.has_smap = false;
    if (resident) {   // the profiles are in HBM already (pgm_merge_profiles_batch_ex): only the node -> column map travels
        // (the prep kernel gathers column node_map[v] of the device matrix unchecked: the range is checked here)
        if (res->ncols == 0 || (!res->node_map && n > res->ncols)) return PGM_ERR_INVALID;
        if (res->node_map) for (uint32_t v = 0; v < n; ++v) if (res->node_map[v] >= res->ncols) return PGM_ERR_INVALID;
        o.sites = 0;
        if (res->node_map) { o.smap = A.put(res->node_map, 4 * (size_t)n); o.has_smap = true; }
    } else o.sites = A.put(g->sites, sizeof(double) * (size_t)g->dim * n);
    o.ni = A.put(ni.data(), sizeof(PgmNode2) * ni.size());
    o.xp = A.put(xp.data(), 4 * xp.size());
    o.xc = A.put(xc.data(), 4 * xc.size());
    o.xv = A.put(xv.data(), 4 * xv.size());
    o.pp = A.put(pp.data(), 4 * pp.size());
    o.pc = A.put(pc.data(), 4 * pc.size());
    o.pv = A.put(pv.data(), 4 * pv.size());
    o.pu = A.put(pu.data(), 4 * pu.size());
    o.fp = A.put(nullptr, 4 * ((size_t)n + 1));                             // filled by finalize_side
    o.fe = A.put(nullptr, 8 * std::max<size_t>(1, o.cd.size()));
    o.ov = A.put(nullptr, 8 * (size_t)PGM_OV_REC * PGM_OV_ENT);
    return PGM_OK;
}

// Second half of the flattening, once the sweep mode of the job is known: where the far candidates of every node go.
//   self-contained sweep (MODE 1): up to PGM_KF entries of distance <= PGM_DCAP in the node summary, else the node is generic
//   MODE 2, rows (side 0): every candidate into the row CSR fp / fe, remote if farther than PGM_DCAP or above the virtual
//           lanes of the row's band; at most PGM_REMOTE_MAX remote and 512 entries per band (rows beyond that: generic)
//   MODE 2, columns (side 1): up to PGM_KF8 entries in the node summary, at most one of them LONG (slot 7)
static void finalize_side(uint8_t *base, uint32_t n, SideOff &o, int side, bool mode2, bool allow_long, bool allow_ov) {
    PgmNode2 *ni = (PgmNode2 *)(base + o.ni);
    int32_t *fp = (int32_t *)(base + o.fp);
    uint2 *fe = (uint2 *)(base + o.fe), *ov = (uint2 *)(base + o.ov);
    o.nov = 0;
    o.far_nodes = 0; o.maxd = 1; o.far_dmin = PGM_DCAP + 1; o.remote = 0; o.ngeneric = 0;
    uint32_t band_entries = 0, band_remote = 0, nfe = 0;
    fp[0] = 0;
    for (uint32_t v = 0; v < n; ++v) {
        PgmNode2 &I = ni[v];
        const uint32_t kill = I.flags & PGM_NF_KILL;
        const uint32_t c0 = o.cp[v], c1 = o.cp[v + 1], nc = c1 - c0;
        if (side == 0 && (v & 63u) == 0) { band_entries = 0; band_remote = 0; }
        uint32_t dmax = 1, nloc = 0, nrem = 0, novf = 0, ovi = 0;
        if (I.c2 != INFINITY) dmax = 2;
        if (I.c3 != INFINITY) dmax = 3;
        bool generic = false;
        if (!mode2) {
            if (nc > (uint32_t)PGM_KF) generic = true;
            for (uint32_t k = c0; k < c1 && !generic; ++k) {
                if (o.cd[k] > (uint32_t)PGM_DCAP) { generic = true; break; }
                I.fd[nloc] = o.cd[k]; I.fc[nloc] = o.cv[k]; ++nloc;
                dmax = std::max(dmax, o.cd[k]);
                o.far_dmin = std::min(o.far_dmin, o.cd[k]);
            }
        } else if (side == 0) {
            const uint32_t lane = v & 63u;
            for (uint32_t k = c0; k < c1; ++k) nrem += (o.cd[k] > (uint32_t)PGM_DCAP || o.cd[k] > lane + (uint32_t)PGM_VL);
            if (band_entries + nc > 512u || band_remote + nrem > (uint32_t)PGM_REMOTE_MAX || nc > 255u || (nrem && !allow_long)) generic = true;
            else {
                for (uint32_t k = c0; k < c1; ++k) {
                    const uint32_t d = o.cd[k];
                    const bool rem = d > (uint32_t)PGM_DCAP || d > lane + (uint32_t)PGM_VL;
                    fe[nfe++] = make_uint2(d | (rem ? 0x80000000u : 0u), __builtin_bit_cast(uint32_t, o.cv[k]));
                    if (!rem) { dmax = std::max(dmax, d); o.far_dmin = std::min(o.far_dmin, d); ++nloc; }
                }
                band_entries += nc; band_remote += nrem;
            }
        } else {
            for (uint32_t k = c0; k < c1; ++k) nrem += o.cd[k] > (uint32_t)PGM_DCAP;
            const uint32_t nl_all = nc - nrem, ring_cap = (uint32_t)PGM_KF8 - std::min(nrem, (uint32_t)PGM_KF8);
            novf = nl_all > ring_cap ? nl_all - ring_cap : 0u;
            if (nrem > (uint32_t)PGM_NLONG || (nrem && !allow_long) || novf > (uint32_t)PGM_OV_ENT || (novf && (o.nov >= (uint32_t)PGM_OV_REC || !allow_ov))) generic = true;
            else {
                uint2 *rec = ov + (size_t)o.nov * PGM_OV_ENT;
                uint32_t nl = 0, no = 0;
                for (uint32_t k = c0; k < c1; ++k) {
                    const uint32_t d = o.cd[k];
                    if (d > (uint32_t)PGM_DCAP) { I.fd[PGM_KF8 - 1 - nl] = d; I.fc[PGM_KF8 - 1 - nl] = o.cv[k]; ++nl; continue; }
                    if (nloc < ring_cap) { I.fd[nloc] = d; I.fc[nloc] = o.cv[k]; ++nloc; }
                    else rec[no++] = make_uint2(d, __builtin_bit_cast(uint32_t, o.cv[k]));
                    dmax = std::max(dmax, d); o.far_dmin = std::min(o.far_dmin, d);
                }
                if (novf) { ovi = o.nov++; for (; no < (uint32_t)PGM_OV_ENT; ++no) rec[no] = make_uint2(1u, __builtin_bit_cast(uint32_t, (float)INFINITY)); }
            }
        }
        fp[v + 1] = (int32_t)nfe;
        if (generic) {   // every non-chain predecessor of this node goes through the CSR lists and the cell storage
            I.c2 = I.c3 = INFINITY;
            for (int k = 0; k < PGM_KF8; ++k) { I.fd[k] = 0; I.fc[k] = INFINITY; }
            I.flags = PGM_NF_GENERIC | (1u << 8) | kill;
            ++o.ngeneric;
        } else {
            const bool rows2 = mode2 && side == 0;
            I.flags = (rows2 ? 0u : nloc) | (dmax << 8) | kill | ((mode2 && side == 1) ? (nrem << 16) | (novf << 20) | (ovi << 25) : 0u);
            o.maxd = std::max(o.maxd, dmax);
            o.far_nodes += (nloc + nrem) != 0;
            o.remote += nrem;
        }
    }
}

// Device-only regions of one job (offsets inside the batch's work / cell / result / score buffers) and its slice of the
// progress counters (pass 1 of pgm_align_batch_create).
struct JobOff { SideOff s1, s2; size_t M, pi, g1f, a1, t2, aux2, map1, map2, ms, mp, res, cells, tb1, tb2, S, prog, codes, endcell, ltab, lready, cls; uint32_t lrows, lcols; };
struct BatchLayout { DevLayout W, C, O, SL; size_t sync_ints = 64; };   // sync: [0] abort flag, [1] ticket counter of the band list, [2] of the lean list; on a cache line of their own, [32] pre-link tasks announced, [33] tracebacks finished (polled by every idle worker)
static void layout_job(BatchLayout &L, uint32_t n1, uint32_t n2, uint32_t dim, uint32_t rshift, bool lean, bool keep, JobOff &o) {
    const uint32_t R = 1u << rshift, rows = PGM_ROWS * R;
    const uint32_t dp = dim <= 20 ? 20 : 64, nb = (n1 - 1 + rows - 1) / rows, tsteps = (n2 - 1) + 63;
    const uint32_t nblk = (tsteps + PGM_BLOCK - 1) / PGM_BLOCK, maxn = std::max(n1, n2);
    o.g1f = L.W.take(sizeof(float) * (size_t)dp * n1);
    o.a1 = L.W.take(sizeof(float) * n1);
    o.t2 = L.W.take(sizeof(float) * (size_t)dp * n2);
    o.aux2 = L.W.take(sizeof(float) * (size_t)n2);
    o.map1 = L.O.take(4 * (size_t)(n1 + n2), 16);
    o.map2 = L.O.take(4 * (size_t)(n1 + n2), 16);
    o.tb1 = L.W.take(sizeof(PgmTbNode) * (size_t)n1);
    o.tb2 = L.W.take(sizeof(PgmTbNode) * (size_t)n2);
    o.ms = L.W.take(4 * (size_t)maxn);
    o.mp = L.W.take(4 * (size_t)maxn);
    o.res = L.O.take(sizeof(PgmJob::Result), 16);
    o.cells = L.C.take(keep ? sizeof(float4) * (size_t)nb * tsteps * 64u * R : 16, 1024);   // (a lean job without the test hook: codes only)
    o.codes = L.W.take(lean ? 4 * (size_t)nb * nblk * 64u * R : 16);   // one word per lane, row and block of eight steps
    o.endcell = L.W.take(16, 16);
    o.cls = L.W.take((lean && !keep) ? (size_t)n1 + n2 : 16, 16);   // classes of the nodes of a lean job (PgmJob::cls1 / cls2)
    o.S = L.SL.take(sizeof(float) * (size_t)nb * nblk * 64u * PGM_BLOCK * R, 1024);
    o.prog = L.sync_ints;
    L.sync_ints += (nb + 3) / 4 * 4;
    // pre-linked traceback tiles (PgmJob::ltab): the long jobs of the general path (from PGM_LK_MIN_ROWS rows: the ones whose
    // tracebacks end a batch; pre-linking every job of the headline batch — 26 000 tiles — cost the sweeps still running 30 %)
    o.lrows = (!lean && n1 - 1 >= PGM_LK_MIN_ROWS && n2 - 1 >= 4 * PGM_LK_T) ? (n1 - 1 + PGM_LK_T - 1) / PGM_LK_T : 0u;
#ifdef PGM_NO_PRELINK   /* A/B builds only: same kernels, no pre-linking */
    o.lrows = 0u;
#endif
    o.lcols = (n2 - 1 + PGM_LK_T - 1) / PGM_LK_T;
    o.ltab = L.W.take(std::max<size_t>((size_t)o.lrows * PGM_LK_W * PGM_LK_TAB * 2, 16), 256);
    o.lready = L.sync_ints;                              // tiles complete per grid row, then the claim counter and the walker's row (zeroed with the progress counters)
    L.sync_ints += ((size_t)o.lrows + 2 + 3) / 4 * 4;
}
// A plain chain 0 -> 1 -> ... -> n-1 with finite edge costs and no repeat edges (a sequence graph).  Decided before the layout
// pass because jobs of two such graphs get the lean sweep's storage (R rows per lane, code bytes, matrices only on request);
// anything else — also a chain with a missing or infinite edge — takes the general path.
static bool graph_is_chain(const pgm_graph *g) {
    if (g->r_rowptr && g->r_rowptr[g->n] != 0) return false;
    if (g->e_rowptr[0] != 0 || g->e_rowptr[1] != 0) return false;
    for (uint32_t v = 1; v < g->n; ++v) {   // exactly the edge v-1 -> v, at finite cost (stored value 0 means +inf, Graph.h:223-231)
        const int32_t eb = g->e_rowptr[v], ee = g->e_rowptr[v + 1];
        if (eb < 0 || ee - eb != 1 || g->e_col[eb] != v - 1 || g->e_val[eb] == 0.0f) return false;
    }
    return true;
}


// upper bound of the flattened input of one graph side with n nodes and E edges (regular + repeat)
static size_t side_bound_bytes(size_t n, size_t dim, size_t E) {
    E = std::max<size_t>(E, 1);
    return n * dim * 8 + n * sizeof(PgmNode2) + 3 * (n + 1) * 4 + E * 28 + 8 * (size_t)PGM_OV_REC * PGM_OV_ENT + 16 * 16;
}
static size_t model_bound_bytes(size_t dim) { return (dim * dim + dim) * 8 + 64; }

// take a buffer of at least `bytes` from the context's cache slot, or allocate one (device memory; slot C_HOST: pinned host)
static hipError_t cache_take(pgm_ctx *ctx, int slot, size_t bytes, void **out, size_t *got) {
    if (ctx->cache_ptr[slot] && ctx->cache_bytes[slot] >= bytes) {
        *out = ctx->cache_ptr[slot]; *got = ctx->cache_bytes[slot];
        ctx->cache_ptr[slot] = nullptr; ctx->cache_bytes[slot] = 0;
        return hipSuccess;
    }
    if (ctx->cache_ptr[slot]) {   // too small: replace
        slot_free(slot, ctx->cache_ptr[slot]);
        ctx->cache_ptr[slot] = nullptr; ctx->cache_bytes[slot] = 0;
    }
    *got = bytes;
    if (slot == pgm_ctx::C_HIN) return pinned_alloc(bytes, out);
    return slot == pgm_ctx::C_HOST ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
}
static void cache_give(pgm_ctx *ctx, int slot, void *p, size_t bytes) {
    if (!p) return;
    if (ctx && (!ctx->cache_ptr[slot] || ctx->cache_bytes[slot] < bytes)) {
        if (ctx->cache_ptr[slot]) slot_free(slot, ctx->cache_ptr[slot]);
        ctx->cache_ptr[slot] = p; ctx->cache_bytes[slot] = bytes;
    } else {
        slot_free(slot, p);
    }
}

// scratch buffer `slot` of the context with room for `bytes` (device memory, or pinned host memory)
static hipError_t scratch_dev(pgm_ctx *ctx, int slot, size_t bytes, void **out) {
    bytes = std::max<size_t>(bytes, 16);
    if (ctx->sc_dev_bytes[slot] < bytes) {
        if (ctx->sc_dev[slot]) (void)hipFree(ctx->sc_dev[slot]);
        ctx->sc_dev[slot] = nullptr; ctx->sc_dev_bytes[slot] = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&ctx->sc_dev[slot], want);
        if (e != hipSuccess) return e;
        ctx->sc_dev_bytes[slot] = want;
    }
    *out = ctx->sc_dev[slot];
    return hipSuccess;
}
static hipError_t scratch_host(pgm_ctx *ctx, int slot, size_t bytes, void **out) {
    bytes = std::max<size_t>(bytes, 16);
    if (ctx->sc_host_bytes[slot] < bytes) {
        if (ctx->sc_host[slot]) pinned_free(ctx->sc_host[slot]);
        ctx->sc_host[slot] = nullptr; ctx->sc_host_bytes[slot] = 0;
        const size_t want = bytes + bytes / 4;
        hipError_t e = pinned_alloc(want, &ctx->sc_host[slot]);
        if (e != hipSuccess) return e;
        ctx->sc_host_bytes[slot] = want;
    }
    *out = ctx->sc_host[slot];
    return hipSuccess;
}
static hipError_t scratch_events(pgm_ctx *ctx) {
    for (int k = 0; k < 2; ++k)
        if (!ctx->sc_ev[k]) { hipError_t e = hipEventCreate(&ctx->sc_ev[k]); if (e != hipSuccess) return e; }
    return hipSuccess;
}

// How the CUs of the device are dealt to the launches of a batch's fill stage (pure arithmetic; pgm_test_cu_shares exports it for
// the CPU tests).  Every launch is a grid of persistent workers, one per CU, and all of them are resident together (no grid ever
// waits for a CU: pgm_tb_kernel's header says why), so the shares add up to at most `cus` and every queue with work gets at least one.
//   crit   the launch of the longest chains: one CU per band, at most half of the device, and only if it leaves every other
//          queue with work at least one CU (else 0: the caller leaves those jobs in the main launch)
//   then the time to beat is t_goal = max(longest chain of sweeps, all other work / the other CUs).  A batch bound by that chain
//   (chain >= 1.5 x the parallel time) wants the other launches' traffic out of the chain's way early: the lean queue gets the
//   fewest CUs with which it ends within 0.6 t_goal, the band queue within 0.75 t_goal (measured on the headline batch, round 3);
//   a batch bound by throughput wants every queue to end together: the factors go to 1 as the chain's lead shrinks to nothing.
//   tb     the traceback kernel that runs beside the sweeps (ntb = 0: the tracebacks follow their launches instead): a third of its
//          work over t_goal — the jobs of a level end together, late in the stage, when the CUs of the sweep kernels join in (an
//          instance of the kernel follows each of them); the workers here take the early finishers — at most a third of the CUs
//   rest   the main launch: what is left, never less than its own work needs to end within t_goal — if the shares do not fit,
//          they are cut back in proportion.
struct CuShares { uint32_t lean, band, crit, rest, tb, rest_need; double t_goal, fl, fb; };
static CuShares cu_shares(uint32_t cus, double lean_cost, uint32_t nlean, double band_cost, uint32_t nbands, double rest_cost, uint32_t nrest, uint32_t ncrit, double rsweep,
                          double tb_cost = 0.0, uint32_t ntb = 0, double tb_frac = 0.35) {
    CuShares r = {0u, 0u, 0u, 0u, 0u, 0u, 0.0, 1.0, 1.0};
    cus = std::max(1u, cus);
    const uint32_t queues = (nlean != 0) + (nbands != 0) + (nrest != 0) + (ntb != 0);
    if (ncrit != 0 && cus > queues) r.crit = std::min(std::min(ncrit, cus / 2u), cus - queues);
    const uint32_t cap = std::max(1u, cus - r.crit);
    const double sum = (nlean ? lean_cost : 0.0) + (nbands ? band_cost : 0.0) + (nrest ? rest_cost : 0.0) + (ntb ? tb_cost : 0.0);
    const double t_par = std::max(1e-3, sum / cap);
    r.t_goal = std::max(std::max(rsweep, t_par), 1e-3);
    const double w = std::min(1.0, std::max(0.0, (rsweep / t_par - 1.0) / 0.5));
    r.fl = 1.0 - 0.4 * w; r.fb = 1.0 - 0.25 * w;
    auto need = [](double cost, double t, uint32_t most) { return (uint32_t)std::min<double>(most, std::max(1.0, std::ceil(cost / t))); };
    const uint32_t band_most = (nbands + PGM_WAVES - 1) / PGM_WAVES;
    uint32_t lean = nlean ? need(lean_cost, r.fl * r.t_goal, nlean) : 0u, band = nbands ? need(band_cost, r.fb * r.t_goal, band_most) : 0u;
    uint32_t rest = nrest ? need(rest_cost, r.t_goal, nrest) : 0u;
    uint32_t tb = ntb ? need(tb_frac * tb_cost, r.t_goal, std::max(1u, std::min(ntb, cap / 3u))) : 0u;
    if (lean + band + rest + tb > cap) {   // cut back in proportion to the work, at least one CU each (cap >= queues unless the device has fewer CUs than queues)
        const double scale = (double)cap / (double)(lean + band + rest + tb);
        auto cut = [&](uint32_t v) { return v ? std::max(1u, (uint32_t)std::floor(v * scale)) : 0u; };
        lean = cut(lean); band = cut(band); rest = cut(rest); tb = cut(tb);
        while (lean + band + rest + tb > cap) {   // (rounding up to one CU each)
            uint32_t *big = &rest; if (band > *big) big = &band; if (lean > *big) big = &lean; if (tb > *big) big = &tb;
            if (*big <= 1u) break;
            --*big;
        }
    }
    r.rest_need = rest;
    const uint32_t left = cap > lean + band + rest + tb ? cap - lean - band - rest - tb : 0u;
    if (nrest) rest = std::min(nrest, rest + left);          // the main launch takes what is left ...
    else if (nbands) band = std::min(band_most, band + left);   // ... or the band queue, or the lean queue
    else if (nlean) lean = std::min(nlean, lean + left);
    r.lean = lean; r.band = band; r.rest = rest; r.tb = tb;
    return r;
}

// classes of the nodes of the lean jobs (PgmJob::cls1): once per batch, behind the upload of the inputs and the job descriptors
static hipError_t classify_lean_jobs(pgm_ctx *ctx, pgm_align_batch *b) {
    if (!b->d_tabhdr || b->njobs == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(b->d_tabhdr, 0, 4 * (size_t)PGM_TAB_HDR * b->njobs, ctx->stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pgm_classify_kernel, dim3(b->njobs, 2, (b->maxn + 1023) / 1024), dim3(256), 0, ctx->stream, b->d_jobs);
    return hipGetLastError();
}

#define PGM_STATUS_PENDING 0x7ffffff0   /* status word of a job's result record in the pinned block until its traceback worker has written it */
static hipError_t launch_all(pgm_ctx *ctx, pgm_align_batch *b, bool timed) {
    hipStream_t s = ctx->stream;
    hipError_t e;
    for (uint32_t i = 0; i < b->njobs; ++i) ((PgmJob::Result *)(b->h_out + b->res_off[i]))->status = PGM_STATUS_PENDING;
    if (timed && (e = hipEventRecord(b->ev[0], s)) != hipSuccess) return e;
    if (b->maxdim <= 20) {
        const size_t prep_lds = ((size_t)b->maxdim * b->maxdim + b->maxdim + 256 * ((size_t)b->maxdim + 1)) * sizeof(float);
        hipLaunchKernelGGL((pgm_prep_kernel<20, 256>), dim3(b->njobs, 2, (b->maxn + 255) / 256), dim3(256), prep_lds, s, b->d_jobs, b->d_sync, (uint32_t)b->sync_ints);
    } else {
        const size_t prep_lds = ((size_t)b->maxdim * b->maxdim + b->maxdim + 64 * ((size_t)b->maxdim + 1)) * sizeof(float);
        hipLaunchKernelGGL((pgm_prep_kernel<64, 64>), dim3(b->njobs, 2, (b->maxn + 63) / 64), dim3(64), prep_lds, s, b->d_jobs, b->d_sync, (uint32_t)b->sync_ints);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (timed && (e = hipEventRecord(b->ev[1], s)) != hipSuccess) return e;
    // (RB = 2 bands per thread — one LDS read of a column pair for four cells — was measured at half the speed: 0.55 -> 1.13 ms)
    const dim3 eg((b->maxnblk + PGM_EM_TB - 1) / PGM_EM_TB, (b->maxnb + 3) / 4, b->njobs);
    if (b->maxdim <= 20) hipLaunchKernelGGL((pgm_emission_skew_kernel<20, 1>), eg, dim3(4 * PGM_ROWS), 0, s, b->d_jobs);
    else hipLaunchKernelGGL((pgm_emission_skew_kernel<64, 1>), eg, dim3(4 * PGM_ROWS), 0, s, b->d_jobs);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (timed && (e = hipEventRecord(b->ev[2], s)) != hipSuccess) return e;
    // One kernel does the DP fill of every band and, right after a job's last band, that job's traceback.
    const char *dbg = tools_env("PGM_FILL_DBG");   // 8: the fill alone, no traceback (tools build)
    const int dbgv = dbg ? atoi(dbg) : 0;
    // test knobs for the hand-off time-out path (tests/test_gpu_align.py): a shorter spin limit, and one band of one job
    // that never publishes its progress ("job:band"), so that the band below it times out and the batch aborts
    const uint32_t spin_limit = b->test_spin_limit ? b->test_spin_limit : PGM_SPIN_LIMIT, stall_job = b->test_stall_job, stall_band = b->test_stall_band;   // (pgm_align_batch_test_stall)
    // timing experiments only (results are garbage): 1 = the cell stores are dropped, 2 = the sweeping wavefront of a MODE 2 band
    // does not merge the helpers' terms, 4 = no helpers, 8 = no history records
    const uint32_t dbg_flags = tools_env("PGM_TEST_NOSTORE") ? (uint32_t)atoi(tools_env("PGM_TEST_NOSTORE")) : 0u;   // (tools build)
    const bool fork = b->nlean != 0;
    const bool bandk = b->nbands != 0;
    const bool tbk = (b->nitems != 0 || bandk) && (b->ntb + b->ntb_c + b->ntb_b) != 0 && dbgv != 8 && !tools_env("PGM_NO_TBK");   // the traceback kernel behind the fill kernel (PGM_NO_TBK, tools build: the sweeps alone, every result stays pending)
    const bool critk = b->ncrit != 0;
    const uint32_t ntb_all = b->ntb + b->ntb_c + b->ntb_b;
    const bool beside = tbk && b->ntb_beside_workers != 0 && !b->d_trace;   // the tracebacks run beside the sweeps (else: behind their launches)
    const uint32_t tbq = beside ? b->tbq_off : 0u;
    if ((fork || bandk || critk || beside) && (e = hipEventRecord(b->ev_fork, s)) != hipSuccess) return e;
    if (beside) {   // first of all, so that its workers are resident when the sweep grids fill the rest of the device
        if ((e = hipStreamWaitEvent(ctx->stream5, b->ev_fork, 0)) != hipSuccess) return e;
        hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->ntb_beside_workers), dim3(64 * PGM_WAVES), 0, ctx->stream5, b->d_jobs, b->d_tblist, ntb_all, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off, 0u, tbq, 1u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(b->ev_join_t, ctx->stream5)) != hipSuccess) return e;
    }
    if (critk && (e = hipStreamWaitEvent(ctx->stream4, b->ev_fork, 0)) != hipSuccess) return e;
    if (fork && (e = hipStreamWaitEvent(ctx->stream2, b->ev_fork, 0)) != hipSuccess) return e;
    if (bandk && (e = hipStreamWaitEvent(ctx->stream3, b->ev_fork, 0)) != hipSuccess) return e;
    const uint32_t nrest = b->nitems - b->ncrit;
    // The launch of the longest chains goes out on the PRIMARY stream, right behind the emission kernel; the main launch on stream4, behind
    // the fork event like the lean and band kernels (a kernel behind an event of another stream starts 50-60 us later: measured on the
    // root of the headline batch, whose traceback is the last thing the stage waits for, when it was the other way round).
    const bool swap_streams = critk && !tools_env("PGM_C3_DBG");
    hipStream_t sc = swap_streams ? s : ctx->stream4, sr = swap_streams ? ctx->stream4 : s;
#ifdef PGM_TOOLS
    if (tools_env("PGM_C3_DBG") && !b->d_c3dbg && b->nitems && hipMalloc((void **)&b->d_c3dbg, 512 * (size_t)b->nitems) != hipSuccess) b->d_c3dbg = nullptr;
    if (b->d_c3dbg) (void)hipMemsetAsync(b->d_c3dbg, 0, 512 * (size_t)b->nitems, s);
    if (critk && b->crit_c3 && b->d_c3dbg) hipLaunchKernelGGL((pgm_crit_kernel<true>), dim3(b->ncrit_workers), dim3(64 * PGM_C3_WAVES), 0, sc, b->d_jobs, b->d_items, b->ncrit, b->d_sync, spin_limit, stall_job, stall_band, (uint32_t)PGM_SY_CRIT_TICKET, b->d_c3dbg, tbq);
    else
#endif
    if (critk && b->crit_c3) hipLaunchKernelGGL((pgm_crit_kernel<false>), dim3(b->ncrit_workers), dim3(64 * PGM_C3_WAVES), 0, sc, b->d_jobs, b->d_items, b->ncrit, b->d_sync, spin_limit, stall_job, stall_band, (uint32_t)PGM_SY_CRIT_TICKET, (unsigned long long *)nullptr, tbq);
    else if (critk) hipLaunchKernelGGL((pgm_fill_kernel<false, false>), dim3(b->ncrit_workers), dim3(64 * PGM_WAVES), 0, sc, b->d_jobs, b->d_items, b->ncrit, b->d_sync, b->d_trace, spin_limit, stall_job, stall_band, dbg_flags, (uint32_t)PGM_SY_CRIT_TICKET, tbq);
    if (nrest == 0) {}   // (no job for this launch)
#ifdef PGM_TOOLS
    else if (b->rest_c3 && b->d_c3dbg) hipLaunchKernelGGL((pgm_crit_kernel<true>), dim3(b->nworkers), dim3(64 * PGM_C3_WAVES), 0, sr, b->d_jobs, b->d_items + b->ncrit, nrest, b->d_sync, spin_limit, stall_job, stall_band, 1u, b->d_c3dbg + 64 * (size_t)b->ncrit, tbq);
#endif
    else if (b->rest_c3) hipLaunchKernelGGL((pgm_crit_kernel<false>), dim3(b->nworkers), dim3(64 * PGM_C3_WAVES), 0, sr, b->d_jobs, b->d_items + b->ncrit, nrest, b->d_sync, spin_limit, stall_job, stall_band, 1u, (unsigned long long *)nullptr, tbq);
    else if (dbgv == 8) hipLaunchKernelGGL((pgm_fill_kernel<true, true>), dim3(b->nworkers), dim3(64 * PGM_WAVES), 0, sr, b->d_jobs, b->d_items + b->ncrit, nrest, b->d_sync, b->d_trace, spin_limit, stall_job, stall_band, dbg_flags, 1u, tbq);
    else if (b->d_trace || dbg_flags) hipLaunchKernelGGL((pgm_fill_kernel<false, true>), dim3(b->nworkers), dim3(64 * PGM_WAVES), 0, sr, b->d_jobs, b->d_items + b->ncrit, nrest, b->d_sync, b->d_trace, spin_limit, stall_job, stall_band, dbg_flags, 1u, tbq);
    else hipLaunchKernelGGL((pgm_fill_kernel<false, false>), dim3(b->nworkers), dim3(64 * PGM_WAVES), 0, sr, b->d_jobs, b->d_items + b->ncrit, nrest, b->d_sync, b->d_trace, spin_limit, stall_job, stall_band, dbg_flags, 1u, tbq);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (fork) {
        // the lean jobs' kernel, launched after the fill kernel (whose grid leaves nlean_workers CUs free)
        unsigned long long *tr2 = b->d_trace ? b->d_trace + 22 * (size_t)b->nitems : nullptr;
#ifdef PGM_TOOLS
        if (b->lean_rshift == 2u) hipLaunchKernelGGL((pgm_lean_kernel<4>), dim3(b->nlean_workers), dim3(64 * PGM_WAVES), 0, ctx->stream2, b->d_jobs, b->d_lean, b->nlean, b->d_sync, tr2, spin_limit);
        else if (b->lean_rshift == 0u) hipLaunchKernelGGL((pgm_lean_kernel<1>), dim3(b->nlean_workers), dim3(64 * PGM_WAVES), 0, ctx->stream2, b->d_jobs, b->d_lean, b->nlean, b->d_sync, tr2, spin_limit);
        else
#endif
        hipLaunchKernelGGL((pgm_lean_kernel<2>), dim3(b->nlean_workers), dim3(64 * PGM_WAVES), 0, ctx->stream2, b->d_jobs, b->d_lean, b->nlean, b->d_sync, tr2, spin_limit);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    const uint32_t njp = (b->njobs + 3u) / 4u * 4u;
    if (bandk) {
        // the bands of the MODE 0 / 1 jobs, one per wavefront, on their share of the CUs; their tracebacks follow on the same
        // stream and the same CUs (the band queue is done well before the chains of the fill kernel are)
        hipLaunchKernelGGL(pgm_band_kernel, dim3(b->nband_workers), dim3(64 * PGM_WAVES), 0, ctx->stream3, b->d_jobs, b->d_bands, b->nbands_narrow, b->nbands, b->nband_workers - b->nwide_workers, b->d_sync, spin_limit, stall_job, stall_band, tbq);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // (tracebacks beside the sweeps: what follows a sweep kernel on its CUs joins in — the jobs that are ready and not yet claimed, then pre-linking)
        if (beside) hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->nband_workers), dim3(64 * PGM_WAVES), 0, ctx->stream3, b->d_jobs, b->d_tblist, ntb_all, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off, 0u, tbq, 1u);
        else if (tbk && b->ntb_b) hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->ntb_b_workers), dim3(64 * PGM_WAVES), 0, ctx->stream3, b->d_jobs, b->d_tblist + b->ntb + b->ntb_c, b->ntb_b, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off + njp, 8u, 0u, 1u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(b->ev_join_b, ctx->stream3)) != hipSuccess) return e;
    }
    if (critk) {
        if (beside) hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->ncrit_workers), dim3(64 * PGM_WAVES), 0, sc, b->d_jobs, b->d_tblist, ntb_all, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off, 0u, tbq, 1u);
        else if (tbk && b->ntb_c) hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->ncrit_workers), dim3(64 * PGM_WAVES), 0, sc, b->d_jobs, b->d_tblist + b->ntb, b->ntb_c, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off + 2 * njp, 16u, 0u, 1u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (beside && nrest != 0) {
        hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->nworkers), dim3(64 * PGM_WAVES), 0, sr, b->d_jobs, b->d_tblist, ntb_all, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off, 0u, tbq, 1u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    } else if (!beside && tbk && b->ntb) {
        // the tracebacks of the fill kernel's jobs, behind it on its stream
        if (b->d_trace) hipLaunchKernelGGL((pgm_tb_kernel<true>), dim3(b->ntb_workers), dim3(64 * PGM_WAVES), 0, sr, b->d_jobs, b->d_tblist, b->ntb, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off, 0u, 0u, 1u);
        else hipLaunchKernelGGL((pgm_tb_kernel<false>), dim3(b->ntb_workers), dim3(64 * PGM_WAVES), 0, sr, b->d_jobs, b->d_tblist, b->ntb, b->d_sync, b->d_trace, b->test_spin_limit, b->lq_off, 0u, 0u, 1u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (critk && (e = hipEventRecord(b->ev_join_c, ctx->stream4)) != hipSuccess) return e;   // (behind whatever went to stream4)
    if (beside && (e = hipStreamWaitEvent(s, b->ev_join_t, 0)) != hipSuccess) return e;
    if (bandk && (e = hipStreamWaitEvent(s, b->ev_join_b, 0)) != hipSuccess) return e;
    if (critk && (e = hipStreamWaitEvent(s, b->ev_join_c, 0)) != hipSuccess) return e;
    if (fork && ((e = hipEventRecord(b->ev_join, ctx->stream2)) != hipSuccess || (e = hipStreamWaitEvent(s, b->ev_join, 0)) != hipSuccess)) return e;
    if (timed && (e = hipEventRecord(b->ev[3], s)) != hipSuccess) return e;
    if (timed && (e = hipEventRecord(b->ev[4], s)) != hipSuccess) return e;
    return hipSuccess;
}
}  // namespace

extern "C" {

int pgm_test_cu_shares(uint32_t cus, double lean_cost, uint32_t nlean, double band_cost, uint32_t nbands, double rest_cost, uint32_t nrest,
                       uint32_t ncrit, double longest_chain, double tb_cost, uint32_t ntb, uint32_t *out5) {
    if (!out5) return fail(PGM_ERR_INVALID, "null argument");
    const CuShares r = cu_shares(cus, lean_cost, nlean, band_cost, nbands, rest_cost, nrest, ncrit, longest_chain, tb_cost, ntb);
    out5[0] = r.lean; out5[1] = r.band; out5[2] = r.crit; out5[3] = r.rest; out5[4] = r.tb;
    return PGM_OK;
}

int pgm_align_batch_create(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                           const pgm_model *const *model, const pgm_scores *scores, pgm_align_batch **out) {
    return pgm_align_batch_create_ex(ctx, njobs, g1, g2, model, scores, 0u, out);
}

int pgm_align_batch_create_ex(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                              const pgm_model *const *model, const pgm_scores *scores, uint32_t flags, pgm_align_batch **out) {
    return pgm_align_batch_create_res(ctx, njobs, g1, g2, model, scores, flags, nullptr, nullptr, out);
}

int pgm_align_batch_create_res(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                               const pgm_model *const *model, const pgm_scores *scores, uint32_t flags,
                               const pgm_site_ref *res1, const pgm_site_ref *res2, pgm_align_batch **out) {
    if (!ctx || !out || (njobs && (!g1 || !g2 || !model || !scores))) return fail(PGM_ERR_INVALID, "null argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    const double tcs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    pgm_align_batch *b = new pgm_align_batch;
    b->njobs = njobs;
    b->jobs.resize(njobs);
    BatchLayout L;
    typedef JobOff Off;
    std::vector<Off> off(njobs);
    b->res_off.resize(njobs); b->map1_off.resize(njobs); b->map2_off.resize(njobs);
    // pass 1 (serial, O(jobs)): sizes, device layouts, and an upper bound of each job's flattened input
    auto side_bound = [](const pgm_graph *g, const pgm_site_ref *res) -> size_t {
        const size_t n = g->n;
        size_t E = (size_t)std::max(0, g->e_rowptr ? g->e_rowptr[n] : 0);
        if (g->r_rowptr) E += (size_t)std::max(0, g->r_rowptr[n]);
        const size_t full = side_bound_bytes(n, g->dim, E);
        return (res && res->dev_sites) ? full - n * g->dim * 8 + n * 4 + 16 : full;   // (resident profiles: a map of n words instead)
    };
    auto ref_of = [&](const pgm_site_ref *r, uint32_t i) -> const pgm_site_ref * { return (r && r[i].dev_sites) ? &r[i] : nullptr; };
    std::vector<size_t> in_base(njobs + 1, 0);
    // (which jobs are two chains: a walk over both graphs' edges, on the host threads — 0.25 ms of a 128-job leaf level otherwise)
    std::vector<char> two_chains(njobs, 0);
    (void)lib_pool().run(njobs, njobs >= 16 ? 16u : 1u, [&](size_t i) {
        const pgm_graph *a = g1[i], *c = g2[i];
        two_chains[i] = (a && c && a->n >= 2 && c->n >= 2 && a->e_rowptr && c->e_rowptr && a->e_col && c->e_col && a->e_val && c->e_val && graph_is_chain(a) && graph_is_chain(c)) ? 1 : 0;
    });
    for (uint32_t i = 0; i < njobs; ++i) {
        const pgm_graph *a = g1[i], *c = g2[i];
        if (!a || !c || !model[i] || a->dim != c->dim || a->dim == 0 || a->dim > 64 || a->n < 2 || c->n < 2 || !model[i]->M || !model[i]->pi ||
            (!a->sites && !ref_of(res1, i)) || !a->e_rowptr || (!c->sites && !ref_of(res2, i)) || !c->e_rowptr) {
            delete b;
            return fail(PGM_ERR_INVALID, "invalid job " + std::to_string(i));
        }
        PgmJob &J = b->jobs[i];
        memset(&J, 0, sizeof J);
        J.n1 = a->n; J.n2 = c->n; J.dim = a->dim;
        J.dp = a->dim <= 20 ? 20 : 64;
        J.ncol = c->n - 1;
        J.tsteps = J.ncol + 63;
        // chain-only jobs: the lean sweep, R rows per lane (the band's buffer descriptor must stay below 1 GiB: see pgm_sweep_chain)
        uint32_t lean_rshift = PGM_LEAN_RSHIFT_DEFAULT;
        if (const char *v = tools_env("PGM_LEAN_RSHIFT")) lean_rshift = (uint32_t)std::min(2, std::max(0, atoi(v)));   // (tools build: R = 1, 2, 4)
        b->lean_rshift = lean_rshift;
        J.lean = (!tools_env("PGM_NO_LEAN") && two_chains[i] && ((uint64_t)J.tsteps * 1024u << lean_rshift) < (1ull << 30)) ? 1u : 0u;
        J.rshift = J.lean ? lean_rshift : 0u;
        J.nb = (a->n - 1 + (PGM_ROWS << J.rshift) - 1) / (PGM_ROWS << J.rshift);
        J.nblk = (J.tsteps + PGM_BLOCK - 1) / PGM_BLOCK;
        b->maxnblk = std::max(b->maxnblk, J.nblk);
        J.maxn = std::max(a->n, c->n);
        b->maxn = std::max(b->maxn, J.maxn);
        J.sc = scores[i];
        b->maxdim = std::max(b->maxdim, a->dim);
        b->maxnb = std::max(b->maxnb, J.nb << J.rshift);   // (bands of the emission kernel: R virtual bands per band)
        b->cells += (uint64_t)(a->n - 2) * (c->n - 2);
        Off &o = off[i];
        in_base[i + 1] = in_base[i] + side_bound(a, ref_of(res1, i)) + side_bound(c, ref_of(res2, i)) + model_bound_bytes(a->dim);
        J.keep_cells = (!J.lean || (flags & PGM_BATCH_KEEP_MATRICES)) ? 1u : 0u;
        layout_job(L, J.n1, J.n2, J.dim, J.rshift, J.lean != 0, J.keep_cells != 0, o);
        b->res_off[i] = o.res; b->map1_off[i] = o.map1; b->map2_off[i] = o.map2;
    }
    // The device buffers and the pinned result block are allocated (or taken from the context's cache) on a thread of their own
    // while the jobs are being flattened: all sizes are known after pass 1.
    b->in_bytes = std::max<size_t>(in_base[njobs], 16);
    b->work_bytes = std::max<size_t>(L.W.bytes, 16);
    b->cell_bytes = std::max<size_t>(L.C.bytes, 16);
    b->out_bytes = std::max<size_t>(L.O.bytes, 16);
    b->s_bytes = std::max<size_t>(L.SL.bytes, 16);
    b->lq_off = (uint32_t)L.sync_ints;                   // ids of the jobs whose tracebacks have started (pre-link announcements)
    L.sync_ints += 3 * (((size_t)njobs + 3) / 4 * 4);  // (one array per instance of pgm_tb_kernel)
    b->tbq_off = (uint32_t)L.sync_ints;                  // ready queue of the traceback kernel that runs beside the sweeps
    L.sync_ints += ((size_t)njobs + 3) / 4 * 4 + 4;
    const size_t sync_ints = L.sync_ints;
    b->sync_ints = sync_ints;
    hipError_t alloc_err = hipSuccess, alloc_host_err = hipSuccess;
    uint8_t *h_out_dev = nullptr;
    const bool cprof = getenv("PGM_HOST_PROFILE") != nullptr;   // tools: where the time of create goes
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tc0 = now_ms();
    double tc_alloc = 0, tc_hostalloc = 0, tc_slot[6] = {0, 0, 0, 0, 0, 0};
    // one small allocation: progress counters, job descriptors, size order, work list (at most one item per band)
    size_t total_bands = 0;
    for (uint32_t i = 0; i < njobs; ++i) total_bands += b->jobs[i].nb;
    DevLayout SM;
    const size_t small_sync = SM.take(sync_ints * sizeof(int)), small_jobs = SM.take(sizeof(PgmJob) * std::max(1u, njobs)),
                 small_order = SM.take(4 * (size_t)std::max(1u, njobs)), small_items = SM.take(sizeof(PgmItem) * std::max<size_t>(1, total_bands)),
                 small_lean = SM.take(4 * (size_t)std::max(1u, njobs)), small_tblist = SM.take(8 * (size_t)std::max(1u, njobs)),
                 small_bands = SM.take(sizeof(PgmItem) * std::max<size_t>(1, total_bands)), small_times = SM.take(16 * (size_t)std::max(1u, njobs) + 16);
    bool any_lean = false;
    for (uint32_t i = 0; i < njobs; ++i) any_lean = any_lean || (b->jobs[i].lean && !b->jobs[i].keep_cells);
    if (tools_env("PGM_NO_LEAN_TABLE") || getenv("PGM_X_NO_LEAN_TABLE")) any_lean = false;   // (the second: a release-build switch for tools/ab scripts)
    const size_t small_tabhdr = SM.take(any_lean ? 4 * (size_t)PGM_TAB_HDR * njobs : 16);
    const size_t small_bytes = SM.bytes;
    std::atomic<int> alloc_state(0);   // 1: the device buffers exist (the flattening threads then upload their jobs' slices), -1: failed
    std::atomic<int> upload_err((int)hipSuccess);
    std::thread alloc_thread([&]() {
        const double ta0 = now_ms();
        hipError_t e2 = hipSetDevice(ctx->device);
        double tprev = ta0;
        auto lap = [&](int k) { const double t = now_ms(); tc_slot[k] = t - tprev; tprev = t; };
        if (e2 == hipSuccess) { e2 = cache_take(ctx, pgm_ctx::C_IN, b->in_bytes, (void **)&b->d_in, &b->cap[pgm_ctx::C_IN]); lap(0); }
        if (e2 == hipSuccess) { e2 = cache_take(ctx, pgm_ctx::C_WORK, b->work_bytes, (void **)&b->d_work, &b->cap[pgm_ctx::C_WORK]); lap(1); }
        if (e2 == hipSuccess) { e2 = cache_take(ctx, pgm_ctx::C_CELLS, b->cell_bytes, (void **)&b->d_cells, &b->cap[pgm_ctx::C_CELLS]); lap(2); }
        if (e2 == hipSuccess) { e2 = cache_take(ctx, pgm_ctx::C_OUT, b->out_bytes, (void **)&b->d_out, &b->cap[pgm_ctx::C_OUT]); lap(3); }
        if (e2 == hipSuccess) { e2 = cache_take(ctx, pgm_ctx::C_S, b->s_bytes, (void **)&b->d_S, &b->cap[pgm_ctx::C_S]); lap(4); }
        if (e2 == hipSuccess) { e2 = cache_take(ctx, pgm_ctx::C_SMALL, small_bytes, (void **)&b->d_small, &b->cap[pgm_ctx::C_SMALL]); lap(5); }
        if (e2 == hipSuccess) {
            b->d_sync = (int *)(b->d_small + small_sync);
            b->d_jobs = (PgmJob *)(b->d_small + small_jobs);
            b->d_order = (uint32_t *)(b->d_small + small_order);
            b->d_items = (PgmItem *)(b->d_small + small_items);
            b->d_lean = (uint32_t *)(b->d_small + small_lean);
            b->d_tblist = (int2 *)(b->d_small + small_tblist);
            b->d_bands = (PgmItem *)(b->d_small + small_bands);
            b->d_times = (unsigned long long *)(b->d_small + small_times);
            b->d_tabhdr = any_lean ? (int *)(b->d_small + small_tabhdr) : nullptr;
        }
        alloc_err = e2;
        alloc_state.store(e2 == hipSuccess ? 1 : -1, std::memory_order_release);
        tc_alloc = now_ms() - ta0;
        // pinned result block, same layout as d_out: the traceback workers write the finished mappings and result records
        // into it over PCIe while the kernel is still running
        if (e2 == hipSuccess) {
            hipError_t e3 = cache_take(ctx, pgm_ctx::C_HOST, b->out_bytes + 64, (void **)&b->h_out, &b->cap[pgm_ctx::C_HOST]);
            if (e3 == hipSuccess) b->h_flag = (int *)(b->h_out + (b->out_bytes + 15) / 16 * 16);
            if (e3 == hipSuccess) e3 = hipHostGetDevicePointer((void **)&h_out_dev, b->h_out, 0);
            alloc_host_err = e3;
        }
        tc_hostalloc = now_ms() - ta0 - tc_alloc;
    });
    // pass 2 (the library's host threads): flatten every job straight into a pinned staging buffer (kept by the context)
    hipError_t e;
    if ((e = cache_take(ctx, pgm_ctx::C_HIN, b->in_bytes, (void **)&b->h_in, &b->cap[pgm_ctx::C_HIN])) != hipSuccess) {
        alloc_thread.join();
        pgm_align_batch_destroy(ctx, b);
        return fail(PGM_ERR_DEVICE, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    const double tc1 = now_ms();
    {
        std::atomic<int> bad(-1);
        const bool job_stats = tools_env("PGM_JOB_STATS") != nullptr;                               // tools build only
        const bool no_helper = tools_env("PGM_NO_HELPER") != nullptr;
        const int mode2_min_bands = tools_env("PGM_MODE2_BANDS") ? atoi(tools_env("PGM_MODE2_BANDS")) : 20;   // tools build only
        const uint32_t mode2_min_hd = tools_env("PGM_MODE2_HD") ? (uint32_t)atoi(tools_env("PGM_MODE2_HD")) : 32u;
        // (the chain of sweeps the batch's largest job would have in pgm_crit_kernel, from the sizes alone: what the other jobs' chains are held against)
        double longest_crit_chain = 0.0;
        for (uint32_t i = 0; i < njobs; ++i)
            longest_crit_chain = std::max(longest_crit_chain, ((double)((g1[i]->n - 1 + PGM_ROWS - 1) / PGM_ROWS - 1) * 80.0 + (double)(g2[i]->n - 1 + 63)) * 0.42);
        // A batch that would not fill the device as MODE 2 sweeps (one band per CU: 0.42 us per step of every band) — a guide-tree level of
        // 16 or 32 jobs on its own, not the 255 jobs of a whole pass — is bound by its longest chain of sweeps, and that chain is 2-3 x
        // shorter in pgm_crit_kernel than on a wavefront of pgm_band_kernel: every job of 8 bands or more goes there then (levels 3 and 4
        // of the headline family alone: 2.05 -> 1.56 and 2.29 -> 1.40 ms per call; level 2, 64 jobs, would fill the device 1.6 times over
        // and stays: 2.2 against 2.4 ms).
        uint32_t promote_bands = 0xffffffffu;
        if (!tools_env("PGM_MODE2_BANDS") && !tools_env("PGM_NO_PROMOTE")) {
            double crit_load = 0.0;
            for (uint32_t i = 0; i < njobs; ++i)
                if (!b->jobs[i].lean) crit_load += (double)((g1[i]->n - 1 + PGM_ROWS - 1) / PGM_ROWS) * (double)(g2[i]->n - 1 + 63) * 0.42;
            if (crit_load <= 1.25 * (double)ctx->prop.multiProcessorCount * longest_crit_chain) promote_bands = 8u;
        }
        const uint32_t chunk_jobs = (uint32_t)std::max<size_t>(1, ((size_t)8 << 20) / std::max<size_t>(1, in_base[njobs] / std::max(1u, njobs)));
        std::vector<std::atomic<uint32_t>> chunk_done((njobs + chunk_jobs - 1) / chunk_jobs + 1);
        for (auto &cd : chunk_done) cd.store(0);
        auto work = [&](size_t job_index) {
            {
                const uint32_t i = (uint32_t)job_index;
                Arena A;
                A.base = b->h_in; A.off = in_base[i]; A.end = in_base[i + 1];
                PgmJob &J = b->jobs[i];
                Off &o = off[i];
                if (flatten_side(g1[i], J.sc, A, o.s1, ref_of(res1, i)) != PGM_OK || flatten_side(g2[i], J.sc, A, o.s2, ref_of(res2, i)) != PGM_OK) { bad.store((int)i); return; }
                J.has_extras = (o.s1.nodes_with_extras + o.s2.nodes_with_extras) > 0 ? 1u : 0u;
                if (J.lean && J.has_extras) { bad.store((int)i); return; }   // (graph_is_chain and flatten_side disagree: cannot happen)
                // LDS of one sweeping wavefront: W / Y history of hD steps x (64 lanes + 16 virtual lanes), X history of hDX
                // steps x 64 lanes, 128 column summaries.  A pair (y - dy, x - dx) is read dy + dx steps back and the virtual
                // lanes are written a block ahead: hD >= maxd1 + maxd2 + 8, hDX >= maxd2 + 1 (powers of two).
                uint32_t hD = 16, hDX = 4;
                while (hD < o.s1.maxd_kf8 + o.s2.maxd_kf8 + (uint32_t)PGM_BLOCK) hD *= 2;
                while (hDX < o.s2.maxd_kf8 + 1) hDX *= 2;
                // Jobs on the batch's critical path (many bands, or a deep history that leaves room for one or two sweeps per
                // worker anyway) and jobs with edges longer than the on-chip history are swept one band per worker: the other
                // three wavefronts take every term but the chain terms off the sweeping wavefront (pgm_terms_helper), which
                // shortens its step by a factor of 2-3, and serve the long edges from the cell storage with a prefetch.
                const uint32_t nb_job = (g1[i]->n - 1 + PGM_ROWS - 1) / PGM_ROWS;
                // (the helpers address the job's cell storage with 32-bit byte offsets)
                const bool allow_long = !tools_env("PGM_NO_LONG") && (uint64_t)J.nb * J.tsteps * 1024u < (1ull << 32);
                const bool has_long = (o.s1.has_long | o.s2.has_long) != 0 && allow_long;
                // (a job whose self-contained sweep fits a quarter of the LDS and that is not on the critical path — fewer than 20 bands —
                // goes to pgm_band_kernel's WIDE workers, four bands per CU, instead of one band per CU with helper wavefronts)
                // ... provided its chain of self-contained sweeps (slower per step the more of its nodes have far edges: 0.75 us at none,
                // 1 us at 2.5 %, measured on levels 4 and 5 of the headline family) still ends well before the batch's longest chain
                const uint32_t slot1 = 2u * hD * (64u + PGM_VL) * 4u + hDX * 64u * 4u + PGM_NRING * 48u;
                const double far_density = 0.5 * ((double)o.s1.cp[g1[i]->n] / g1[i]->n + (double)o.s2.cp[g2[i]->n] / g2[i]->n);
                const double chain_wide = ((double)(nb_job - 1) * 78.0 + (double)(g2[i]->n - 1 + 63)) * (0.7 + 10.0 * far_density);
                const bool fits_wide = slot1 <= (uint32_t)(PGM_POOL / PGM_WIDE_WAVES / 16 * 16) && chain_wide <= 0.9 * longest_crit_chain && !tools_env("PGM_NO_WIDE");
                J.mode2 = (J.has_extras && ((hD >= mode2_min_hd && !fits_wide) || nb_job >= (uint32_t)mode2_min_bands || nb_job >= promote_bands || has_long) && !no_helper) ? 1u : 0u;
                if (J.mode2) {   // (a MODE 2 sweep keeps every on-chip distance of the graphs, whatever the number of entries of a node)
                    while (hD < o.s1.maxd_cap + o.s2.maxd_cap + (uint32_t)PGM_BLOCK) hD *= 2;
                    while (hDX < o.s2.maxd_cap + 1) hDX *= 2;
                }
                J.hD = hD; J.hDX = hDX;
                J.slot_bytes = 2u * hD * (64u + PGM_VL) * 4u + hDX * 64u * 4u;
                J.slot_bytes += PGM_NRING * (J.mode2 ? 80u : 48u);   // column ring: 5 or 3 float4 per column
                J.aux_off = J.slot_bytes;
                if (J.mode2) J.slot_bytes += PGM_AUX_BYTES;
                const uint32_t ov_bytes = 8u * PGM_OV_REC * PGM_OV_ENT;
                finalize_side(b->h_in, g1[i]->n, o.s1, 0, J.mode2 != 0, allow_long, false);
                finalize_side(b->h_in, g2[i]->n, o.s2, 1, J.mode2 != 0, allow_long, J.slot_bytes + ov_bytes <= (uint32_t)PGM_POOL);
                J.nov2 = J.mode2 ? o.s2.nov : 0u;
                J.ov_off = J.slot_bytes;
                if (J.nov2) J.slot_bytes += ov_bytes;
                J.has_far = (o.s1.far_nodes + o.s2.far_nodes) > 0 ? 1u : 0u;
                J.long1 = (J.mode2 && o.s1.remote) ? 1u : 0u;
                J.long2 = (J.mode2 && o.s2.remote) ? 1u : 0u;
                J.rh_off = J.slot_bytes;
                if (J.long1 | J.long2) J.slot_bytes += 3u * 32u * 64u * 4u;   // W of the last 32 columns of every remote row's walk (one ring per row helper)
                // MODE 2 jobs whose every predecessor is near or in the LDS history (no long / remote entries, no overflow columns, no
                // generic nodes) are swept by pgm_crit_kernel: the chain terms on one wavefront, everything else on fifteen others
                J.c3_off = J.slot_bytes;
                // (and no interior node without predecessors: the chain wavefront carries no code for them)
                J.crit3 = (J.mode2 && !J.long1 && !J.long2 && J.nov2 == 0 && o.s1.ngeneric + o.s2.ngeneric == 0 && o.s1.nkill + o.s2.nkill == 0 &&
                           J.slot_bytes + (uint32_t)PGM_C3_BYTES <= (uint32_t)PGM_POOL && !tools_env("PGM_NO_CRIT3") && !tools_env("PGM_FILL_TRACE") &&
                           !tools_env("PGM_TEST_NOSTORE") && !tools_env("PGM_FILL_DBG")) ? 1u : 0u;   // (the timeline and strip-down switches of the tools build belong to pgm_fill_kernel)
                if (J.crit3 && J.hDX < 8u) {   // (the chain wavefront addresses a block of eight steps from one base: no ring wraps inside a block)
                    J.slot_bytes += (8u - J.hDX) * 256u; J.aux_off += (8u - J.hDX) * 256u; J.ov_off += (8u - J.hDX) * 256u; J.rh_off += (8u - J.hDX) * 256u; J.c3_off += (8u - J.hDX) * 256u;
                    J.hDX = 8u;
                    if (J.slot_bytes + (uint32_t)PGM_C3_BYTES > (uint32_t)PGM_POOL) J.crit3 = 0u;   // (cannot happen: a deep W / Y history comes with a deep X history)
                }
                if (J.crit3) J.slot_bytes += (uint32_t)PGM_C3_BYTES;
                J.far_slack = std::max(1u, std::min(4u, std::min(o.s1.far_dmin, o.s2.far_dmin)));
                J.nslots = J.mode2 ? 1u : std::max(1u, std::min((uint32_t)PGM_WAVES, (uint32_t)PGM_POOL / J.slot_bytes));
                if (J.lean) J.nslots = PGM_WAVES;
                if (job_stats) {   // tools: how the nodes of this job are served
                    uint32_t gen[2] = {0, 0}, lng = 0;
                    for (int side = 0; side < 2; ++side) {
                        const PgmNode2 *ni = (const PgmNode2 *)(b->h_in + (side ? o.s2.ni : o.s1.ni));
                        const uint32_t nn = side ? g2[i]->n : g1[i]->n;
                        for (uint32_t v = 0; v < nn; ++v) { gen[side] += (ni[v].flags & PGM_NF_GENERIC) != 0; if (side) lng += PGM_NF_NLONG(ni[v].flags); }
                    }
                    fprintf(stderr, "pgm job %u: %u x %u mode2 %u crit3 %u slot %u B hD %u hDX %u far_slack %u generic rows %u cols %u remote row entries %u long col entries %u far nodes %u + %u overflow cols %u\n",
                            i, g1[i]->n, g2[i]->n, J.mode2, J.crit3, J.slot_bytes, J.hD, J.hDX, J.far_slack, gen[0], gen[1], o.s1.remote, lng, o.s1.far_nodes, o.s2.far_nodes, J.nov2);
                    for (int side = 0; side < 2; ++side) {
                        const SideOff &so = side ? o.s2 : o.s1;
                        const PgmNode2 *ni = (const PgmNode2 *)(b->h_in + so.ni);
                        const uint32_t nn = side ? g2[i]->n : g1[i]->n;
                        for (uint32_t v = 0; v < nn; ++v)
                            if (ni[v].flags & PGM_NF_GENERIC) {
                                uint32_t nl = 0;
                                for (uint32_t k = so.cp[v]; k < so.cp[v + 1]; ++k) nl += so.cd[k] > (uint32_t)PGM_DCAP;
                                fprintf(stderr, "   generic %s %u: %u far candidates, %u of them long\n", side ? "col" : "row", v, so.cp[v + 1] - so.cp[v], nl);
                            }
                    }
                }
                o.M = A.put(model[i]->M, sizeof(double) * J.dim * J.dim);
                o.pi = A.put(model[i]->pi, sizeof(double) * J.dim);
                if (A.overflow) { bad.store((int)i); return; }
                // the input image goes to the device in chunks of consecutive jobs (~8 MB: a copy has ~10 us of fixed cost) while
                // the other jobs are still being flattened: whoever completes a chunk's last job sends it
                const uint32_t c = i / chunk_jobs, c0 = c * chunk_jobs, c1 = std::min(njobs, c0 + chunk_jobs);
                if (chunk_done[c].fetch_add(1, std::memory_order_acq_rel) + 1 == c1 - c0) {
                    int st;
                    while ((st = alloc_state.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
                    if (st == 1) {
                        (void)hipSetDevice(ctx->device);
                        const hipError_t eu = hipMemcpyAsync(b->d_in + in_base[c0], b->h_in + in_base[c0], in_base[c1] - in_base[c0], hipMemcpyHostToDevice, ctx->stream);
                        if (eu != hipSuccess) upload_err.store((int)eu);
                    }
                }
            }
        };
        (void)lib_pool().run(njobs, 16, work);
        if (bad.load() >= 0) {
            const int i = bad.load();
            alloc_thread.join();
            pgm_align_batch_destroy(ctx, b);
            return fail(PGM_ERR_INVALID, "invalid graph in job " + std::to_string(i));
        }
    }
    const double tc2 = now_ms();
    alloc_thread.join();
    const double tc3 = now_ms();
    if (alloc_err != hipSuccess) {
        pgm_align_batch_destroy(ctx, b);
        return fail(alloc_err == hipErrorOutOfMemory ? PGM_ERR_NOMEM : PGM_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(alloc_err));
    }
    if (alloc_host_err != hipSuccess) {
        pgm_align_batch_destroy(ctx, b);
        return fail(PGM_ERR_DEVICE, std::string("hipHostMalloc: ") + hipGetErrorString(alloc_host_err));
    }
    for (uint32_t i = 0; i < njobs; ++i) {
        PgmJob &J = b->jobs[i];
        const Off &o = off[i];
        uint8_t *in = b->d_in, *w = b->d_work, *ob = b->d_out;
        J.sites1 = ref_of(res1, i) ? res1[i].dev_sites : (const double *)(in + o.s1.sites);
        J.sites2 = ref_of(res2, i) ? res2[i].dev_sites : (const double *)(in + o.s2.sites);
        J.smap1 = o.s1.has_smap ? (const uint32_t *)(in + o.s1.smap) : nullptr;
        J.smap2 = o.s2.has_smap ? (const uint32_t *)(in + o.s2.smap) : nullptr;
        J.M = (const double *)(in + o.M); J.pi = (const double *)(in + o.pi);
        J.ni1 = (const PgmNode2 *)(in + o.s1.ni); J.ni2 = (const PgmNode2 *)(in + o.s2.ni);
        J.xp1 = (const int32_t *)(in + o.s1.xp); J.xp2 = (const int32_t *)(in + o.s2.xp);
        J.xc1 = (const uint32_t *)(in + o.s1.xc); J.xc2 = (const uint32_t *)(in + o.s2.xc);
        J.xv1 = (const float *)(in + o.s1.xv); J.xv2 = (const float *)(in + o.s2.xv);
        J.fp1 = (const int32_t *)(in + o.s1.fp); J.fe1 = (const uint2 *)(in + o.s1.fe);
        J.ov2 = (const uint2 *)(in + o.s2.ov);
        J.pp1 = (const int32_t *)(in + o.s1.pp); J.pp2 = (const int32_t *)(in + o.s2.pp);
        J.pc1 = (const uint32_t *)(in + o.s1.pc); J.pc2 = (const uint32_t *)(in + o.s2.pc);
        J.pv1 = (const float *)(in + o.s1.pv); J.pv2 = (const float *)(in + o.s2.pv);
        J.pu1 = (const uint32_t *)(in + o.s1.pu); J.pu2 = (const uint32_t *)(in + o.s2.pu);
        J.g1f = (float *)(w + o.g1f); J.a1 = (float *)(w + o.a1);
        J.t2 = (float *)(w + o.t2); J.b2 = (float *)(w + o.aux2);
        J.map1 = (uint32_t *)(ob + o.map1); J.map2 = (uint32_t *)(ob + o.map2);
        J.mark_score = (float *)(w + o.ms); J.mark_prev = (uint32_t *)(w + o.mp);
        J.tb1 = (PgmTbNode *)(w + o.tb1); J.tb2 = (PgmTbNode *)(w + o.tb2);
        J.result = (PgmJob::Result *)(ob + o.res);
        J.hmap1 = (uint32_t *)(h_out_dev + o.map1); J.hmap2 = (uint32_t *)(h_out_dev + o.map2);
        J.hresult = (PgmJob::Result *)(h_out_dev + o.res);
        J.cells = (float4 *)(b->d_cells + o.cells);
        J.codes = (uint32_t *)(w + o.codes);
        J.endcell = (float4 *)(w + o.endcell);
        if (J.lean && !J.keep_cells && b->d_tabhdr) { J.cls1 = w + o.cls; J.cls2 = w + o.cls + J.n1; J.tabhdr = b->d_tabhdr + (size_t)PGM_TAB_HDR * i; }
        else { J.cls1 = nullptr; J.cls2 = nullptr; J.tabhdr = nullptr; }
        J.S = (float *)(b->d_S + o.S);
        J.prog = b->d_sync + o.prog;
        J.ltab = (uint16_t *)(w + o.ltab); J.lready = b->d_sync + o.lready; J.lrows = o.lrows; J.lcols = o.lcols;
        J.times = b->d_times + 2 * (size_t)i;
    }
    b->order.resize(njobs);
    std::iota(b->order.begin(), b->order.end(), 0u);
    std::stable_sort(b->order.begin(), b->order.end(), [&](uint32_t x, uint32_t y) {
        return (uint64_t)b->jobs[x].n1 * b->jobs[x].n2 > (uint64_t)b->jobs[y].n1 * b->jobs[y].n2;
    });
    // ---- fill work list.  An item is a band (MODE 2 jobs) or a group of up to eight bands (all others); item k of a job
    // can start once item k-1 has been running for the band-to-band lag, and the workers take items in list order.  The
    // order is the result of simulating the persistent workers on the host with estimated times: whenever a worker is
    // free it takes, among the items that are READY by then, the one with the longest remaining path (the time until its
    // job is complete: the lags still ahead, one full sweep, the traceback).  Within a job the items keep ascending
    // order, as the kernel requires; taking only ready items keeps workers from idling in front of a predecessor band.
    // Step times (us, measured with the whole batch resident): ~0.45 for a chain-only band (the leaf level is bound by the
    // HBM write bandwidth), ~0.65 with the near window and the far history in the sweeping wavefront, ~0.6 with helpers.
    std::vector<PgmItem> items;
    std::vector<uint32_t> lean_list;
    double lean_cost = 0.0, other_cost = 0.0;   // worker-microseconds of the two kernels' queues
    uint32_t capacity = (uint32_t)ctx->prop.multiProcessorCount;
    // persistent workers: one workgroup of 8 wavefronts per CU (it owns the CU's LDS for its sweeps' histories)
    if (const char *env_c = tools_env("PGM_FILL_WORKERS"))   // tools build only
        capacity = std::min<uint32_t>((uint32_t)ctx->prop.multiProcessorCount, (uint32_t)std::max(1, atoi(env_c)));
    std::vector<PgmItem> bands;   // pgm_band_kernel's list: one band per entry
    if (njobs) {
        struct Item { double rem, dur, gap; uint32_t job, band, count; };
        std::vector<std::vector<Item>> per_job(njobs), per_job_b(njobs), per_job_c(njobs), per_job_w(njobs);   // main launch, narrow bands, longest chains, wide bands
        std::vector<double> chain_of(njobs, 0.0);   // chain of sweeps of the jobs of the fill kernel
        const double lag = PGM_ROWS + 3.0 * PGM_BLOCK;
        auto envd = [](const char *k, double d) { const char *v = tools_env(k); return v ? atof(v) : d; };   // tools build only
        const double tau_x = envd("PGM_SIM_TAU_X", 0.65), tau_c = envd("PGM_SIM_TAU_C", 0.45), tau_2 = envd("PGM_SIM_TAU_2", 0.6), eager = envd("PGM_SIM_EAGER", 0.7);
        const double tau_3 = envd("PGM_SIM_TAU_3", 0.42);   // a band of a crit3 job (pgm_crit_kernel)
        const double tau_l = envd("PGM_SIM_TAU_L", 0.34);   // lean sweep: us per step of R rows per lane
        // The jobs without helper wavefronts go to pgm_band_kernel, band by band (not with the timeline of the tools build, whose
        // slots are the fill kernel's items, and not a job whose sweep would not fit an eighth of the LDS)
        const bool use_bands = !tools_env("PGM_FILL_TRACE") && !tools_env("PGM_NO_BANDK");
        size_t total = 0, total_b = 0, total_w = 0;
        double rmax = 1.0, rsweep = 1.0;   // longest remaining path with / without the traceback behind it
        for (uint32_t q = 0; q < njobs; ++q) {
            const uint32_t i = b->order[q];   // (largest first: the order of the lean queue)
            const PgmJob &J = b->jobs[i];
            const double tau = J.crit3 ? tau_3 : (J.mode2 ? tau_2 : (J.has_extras ? tau_x : tau_c));     // us per step
            const double tb = (J.has_extras ? 0.3 : 0.2) * (double)(J.n1 + J.n2);   // the traceback follows the last band (us)
            if (J.lean) {   // pgm_lean_kernel's queue: a worker's wavefronts cycle over the job's bands (72 steps behind each other), then the walk
                const double rounds = std::ceil((double)J.nb / PGM_WAVES), first = std::min<double>(J.nb, PGM_WAVES);
                lean_cost += tau_l * (rounds * J.tsteps + (first - 1.0) * 72.0) + 0.04 * (double)(J.n1 + J.n2);
                lean_list.push_back(i);
                continue;
            }
            const bool narrow = use_bands && !J.mode2 && J.slot_bytes <= (uint32_t)(PGM_POOL / PGM_WAVES / 16 * 16);
            const bool wide = use_bands && !J.mode2 && !narrow && J.slot_bytes <= (uint32_t)(PGM_POOL / PGM_WIDE_WAVES / 16 * 16);
            const bool per_band = narrow || wide;
            if (!per_band) chain_of[i] = tau * ((double)(J.nb - 1) * lag + J.tsteps);
            const uint32_t group = per_band ? 1u : J.nslots;       // bands per item, one per wavefront of the worker
            for (uint32_t band = 0; band < J.nb; band += group) {
                const uint32_t cnt = std::min(group, J.nb - band);
                Item it;
                it.rem = tau * ((double)(J.nb - 1 - band) * lag + J.tsteps) + tb;
                it.dur = tau * ((double)(cnt - 1) * lag + J.tsteps);   // (the traceback is another kernel's: pgm_tb_kernel)
                it.gap = tau * (double)cnt * lag;                 // the next item may start this long after this one
                it.job = i; it.band = band; it.count = cnt;
                (wide ? per_job_w : (narrow ? per_job_b : per_job))[i].push_back(it);
                rmax = std::max(rmax, it.rem);
                rsweep = std::max(rsweep, it.rem - tb);
            }
            total += per_job[i].size();
            total_b += per_job_b[i].size();
            total_w += per_job_w[i].size();
        }
        // The CUs are split between the kernels (one worker per CU in each), see cu_shares(): the jobs with the longest chains of
        // sweeps (within 15 % of the longest: the root of a guide tree, as a rule) get a launch of the fill kernel of their own, one CU
        // per band — their tracebacks are the last thing a batch waits for, and this way the other jobs' tracebacks are out of the
        // way before they start (each launch is followed by its own instance of pgm_tb_kernel); only if other jobs stay behind for
        // the main launch.  The rest of the CUs is dealt to the lean queue, the band queue and the main launch by their costs.
        double band_cost = 0.0, crit_cost = 0.0;
        double wide_cost = 0.0;
        for (uint32_t i = 0; i < njobs; ++i) {
            for (const Item &it : per_job[i]) other_cost += it.dur;
            for (const Item &it : per_job_b[i]) band_cost += it.dur / PGM_WAVES;
            for (const Item &it : per_job_w[i]) wide_cost += it.dur / PGM_WIDE_WAVES;
        }
        band_cost += wide_cost;   // one queue for the shares: pgm_band_kernel's workers, split below
        size_t total_c = 0;
        if (use_bands && total != 0) {
            uint32_t ncj = 0, nrestj = 0;
            for (uint32_t i = 0; i < njobs; ++i) if (!per_job[i].empty()) { if (chain_of[i] >= 0.85 * rsweep) ++ncj; else ++nrestj; }
            if (ncj != 0 && nrestj != 0)
                for (uint32_t i = 0; i < njobs; ++i)
                    if (!per_job[i].empty() && chain_of[i] >= 0.85 * rsweep) { total_c += per_job[i].size(); for (const Item &it : per_job[i]) crit_cost += it.dur; }
        }
        // the tracebacks of the jobs of the fill, crit and band kernels run beside the sweeps on CUs of their own (not with the timeline
        // of the tools build, whose slots belong to the kernels that follow each other)
        double tb_cost = 0.0;
        uint32_t ntb_all = 0;
        // (measured on the headline batch, round 4: the jobs of a guide-tree level end together, late in the stage — beside the sweeps the
        // workers of this kernel idle until then, and the root's walk loses the pre-linkers that the launch of the longest chains hands it
        // when its tracebacks follow it: 3.3 ms against 2.9.  Kept for batches whose jobs end at different times: PGM_TB_BESIDE=1.)
        const bool tb_beside = getenv("PGM_TB_BESIDE") != nullptr && !tools_env("PGM_FILL_TRACE") && !tools_env("PGM_NO_TBK") && !tools_env("PGM_FILL_DBG");
        for (uint32_t i = 0; i < njobs; ++i) if (!b->jobs[i].lean) { ++ntb_all; tb_cost += (b->jobs[i].has_extras ? 0.3 : 0.2) * (double)(b->jobs[i].n1 + b->jobs[i].n2); }
        if (!tb_beside) { ntb_all = 0; tb_cost = 0.0; }
        CuShares sh = cu_shares(capacity, lean_cost, (uint32_t)lean_list.size(), band_cost, (uint32_t)(total_b + 2 * total_w), other_cost - crit_cost, (uint32_t)(total - total_c), (uint32_t)total_c, rsweep, tb_cost, ntb_all);
        if (total_c != 0 && sh.crit == 0) {   // no CU to spare for a launch of their own: the longest chains stay in the main launch
            total_c = 0; crit_cost = 0.0;
            sh = cu_shares(capacity, lean_cost, (uint32_t)lean_list.size(), band_cost, (uint32_t)(total_b + 2 * total_w), other_cost, (uint32_t)total, 0u, rsweep, tb_cost, ntb_all);
        }
        b->ntb_beside_workers = sh.tb;
        if (total_c != 0)
            for (uint32_t i = 0; i < njobs; ++i)
                if (!per_job[i].empty() && chain_of[i] >= 0.85 * rsweep) per_job_c[i].swap(per_job[i]);
        total -= total_c;
        const double t_goal = sh.t_goal;
        uint32_t lean_cus = sh.lean, band_cus = sh.band, crit_cus = sh.crit;
        // (a lean job is one worker's: the queue ends after ceil(jobs / workers) rounds — the fewest workers with that many rounds do)
        if (lean_cus) { const uint32_t rounds = ((uint32_t)lean_list.size() + lean_cus - 1u) / lean_cus; lean_cus = ((uint32_t)lean_list.size() + rounds - 1u) / rounds; }
        if (const char *v = tools_env("PGM_LEAN_CUS")) if (lean_cus) lean_cus = std::max(1u, std::min(std::min(capacity - 1u, (uint32_t)lean_list.size()), (uint32_t)atoi(v)));
        b->nlean = (uint32_t)lean_list.size();
        b->nlean_workers = lean_cus;
        // event simulation: free workers (min-heap of times), ready items (max-heap of remaining paths), pending successors
        auto simulate = [&](std::vector<std::vector<Item>> &pj, size_t count, uint32_t workers, std::vector<PgmItem> &out) -> double {
            out.clear();
            double end = 0.0;
            typedef std::pair<double, uint32_t> TE;   // (time, job)
            std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
            for (uint32_t w = 0; w < std::max(1u, workers); ++w) free_at.push(0.0);
            std::priority_queue<TE> ready;                                               // (rem, job): next item of that job
            std::priority_queue<TE, std::vector<TE>, std::greater<TE>> pending;         // (ready time, job)
            std::vector<uint32_t> next(njobs, 0);
            for (uint32_t i = 0; i < njobs; ++i) if (!pj[i].empty()) ready.push({pj[i][0].rem, i});
            out.reserve(count);
            double now = 0.0;
            while (out.size() < count) {
                now = std::max(now, free_at.top());
                while (!pending.empty() && pending.top().first <= now) {
                    const uint32_t j = pending.top().second; pending.pop();
                    ready.push({pj[j][next[j]].rem, j});
                }
                if (ready.empty()) { now = pending.top().first; continue; }              // every free worker would have to wait
                const uint32_t j = ready.top().second; ready.pop();
                const Item &it = pj[j][next[j]];
                // wave priority (s_setprio): the longest paths of the batch win the issue arbitration on their SIMDs
                out.push_back(PgmItem{it.job, it.band, it.rem > 0.6 * rmax ? 3u : (it.rem > 0.35 * rmax ? 2u : (it.rem > 0.2 * rmax ? 1u : 0u)), it.count});
                free_at.pop();
                free_at.push(now + it.dur);
                end = std::max(end, now + it.dur);
                // the longest paths of the batch are not held back: their next band gets a worker at once (it spins until the
                // predecessor is far enough, but then follows it without any queueing delay)
                if (++next[j] < pj[j].size()) pending.push({pj[j][next[j]].rem > eager * rmax ? now : now + it.gap, j});
            }
            return end;
        };
        double band_end = 0.0;
        uint32_t wide_cus = 0;
        std::vector<PgmItem> bands_w;
        if (total_b + total_w != 0) {
            // (a simulation of a 1000-band list is 0.1 ms: the share grows by how far the simulated schedule overshoots, three times at
            // most, and only into CUs the main launch does not need for its own share).  The share is split between the workers of
            // the narrow bands (eight at a time per CU) and of the wide ones (four at a time) by their work.
            const uint32_t most = std::max(band_cus, band_cus + (sh.rest > sh.rest_need ? sh.rest - sh.rest_need : 0u));
            auto split = [&](uint32_t cus) {
                if (total_w == 0) return 0u;
                if (total_b == 0) return cus;
                const uint32_t w = (uint32_t)std::lround(cus * wide_cost / band_cost);
                return std::max(1u, std::min(cus > 1u ? cus - 1u : 1u, w));
            };
            auto run = [&](uint32_t cus) {
                wide_cus = split(cus);
                const uint32_t ncus = cus > wide_cus ? cus - wide_cus : (total_b ? 1u : 0u);
                double e = 0.0;
                if (total_b) e = simulate(per_job_b, total_b, ncus * PGM_WAVES, bands);
                if (total_w) e = std::max(e, simulate(per_job_w, total_w, wide_cus * PGM_WIDE_WAVES, bands_w));
                return e;
            };
            if (const char *v = tools_env("PGM_BAND_CUS")) band_cus = std::max(1u, std::min(most, (uint32_t)atoi(v)));
            else if (total != 0) {
                for (int it = 0; it < 3 && band_cus < most; ++it) {
                    band_end = run(band_cus);
                    if (band_end <= sh.fb * t_goal) break;
                    band_cus = std::min(most, std::max(band_cus + 1u, (uint32_t)std::ceil(band_cus * std::min(2.0, band_end / (sh.fb * t_goal)))));
                }
            }
            band_cus = std::max(1u, std::min<uint32_t>(std::min(band_cus, most), (uint32_t)((total_b + PGM_WAVES - 1) / PGM_WAVES + (total_w + PGM_WIDE_WAVES - 1) / PGM_WIDE_WAVES)));
            if (total_b && total_w) band_cus = std::max(band_cus, 2u);
            band_end = run(band_cus);
            if (total_b == 0) bands.clear();
            b->nbands_narrow = (uint32_t)(total_b ? bands.size() : 0);
            bands.insert(bands.end(), bands_w.begin(), bands_w.end());
        }
        if (total_b + total_w != 0 && !tools_env("PGM_BAND_CUS")) {
            // the tracebacks of the band kernel's jobs follow it on its CUs, one worker per job: with fewer workers than jobs the last ones
            // wait a whole walk longer — a round less if the main launch can spare the CUs for it
            uint32_t nbj = 0;
            for (uint32_t i = 0; i < njobs; ++i) nbj += (!per_job_b[i].empty() || !per_job_w[i].empty());
            const uint32_t most = std::max(band_cus, band_cus + (sh.rest > sh.rest_need ? sh.rest - sh.rest_need : 0u));
            if (nbj > band_cus) {
                const uint32_t rounds = (nbj + band_cus - 1u) / band_cus, want = rounds > 1u ? (nbj + rounds - 2u) / (rounds - 1u) : band_cus;
                if (want > band_cus && want <= most && want <= band_cus + band_cus / 8u + 1u) {
                    band_cus = want;
                    wide_cus = total_w == 0 ? 0u : (total_b == 0 ? band_cus : std::max(1u, std::min(band_cus - 1u, (uint32_t)std::lround(band_cus * wide_cost / band_cost))));
                }
            }
        }
        b->nwide_workers = wide_cus;
        b->nband_workers = band_cus;
        if (total_c != 0) {
            // the launch of the longest chains has a worker per band — but a job never has more than tsteps / lag + 2 of its bands under way
            // at the same time (the first are through before the last may start): the workers beyond that go to the main launch
            uint32_t need = 0;
            for (uint32_t i = 0; i < njobs; ++i)
                if (!per_job_c[i].empty()) need += std::min<uint32_t>(b->jobs[i].nb, b->jobs[i].tsteps / (uint32_t)std::max(1.0, lag) + 2u);
            crit_cus = std::max(1u, std::min(crit_cus, need));
        }
        capacity = std::max(1u, capacity > lean_cus + band_cus + crit_cus + sh.tb ? capacity - lean_cus - band_cus - crit_cus - sh.tb : 1u);   // the main launch's CUs
        b->ncrit_workers = crit_cus;
        std::vector<PgmItem> items_rest;
        const double crit_end = total_c ? simulate(per_job_c, total_c, crit_cus, items) : 0.0;
        b->ncrit = (uint32_t)items.size();
        const double fill_end = simulate(per_job, total, capacity, items_rest);
        items.insert(items.end(), items_rest.begin(), items_rest.end());
        (void)crit_end;
        if (cprof) fprintf(stderr, "    work lists: longest chain of sweeps %.0f us, goal %.0f us; lean %zu jobs %.0f us-worker on %u CUs; bands %zu, %.0f us-worker on %u CUs (simulated end %.0f us); items %zu, %.0f us-worker on %u CUs (simulated end %.0f us), of the longest chains %zu on %u CUs (%.0f us)\n",
                           rsweep, t_goal, lean_list.size(), lean_cost, lean_cus, total_b, band_cost, band_cus, band_end, total, other_cost, capacity, fill_end, total_c, crit_cus, crit_end);
        (void)fill_end;
    }
    b->nbands = (uint32_t)bands.size();
    if (bands.empty()) bands.push_back(PgmItem{0u, 0u, 0u, 0u});
    const double tc4 = now_ms();
    b->nitems = (uint32_t)items.size();
    b->nworkers = std::max(1u, std::min(capacity, b->nitems - b->ncrit));
    b->crit_c3 = b->ncrit != 0; b->rest_c3 = b->nitems > b->ncrit;
    for (size_t k = 0; k < items.size(); ++k)
        if (!b->jobs[items[k].job].crit3) { if (k < b->ncrit) b->crit_c3 = false; else b->rest_c3 = false; }
    if (lean_list.empty()) lean_list.push_back(0u);
    if (tools_env("PGM_FILL_TRACE") && njobs &&   // (tools build) 6 words per item + 16 per item for the helper wavefronts, then 6 words per lean job
        hipMalloc((void **)&b->d_trace, 176 * items.size() + 48 * (size_t)njobs) != hipSuccess) {
        b->d_trace = nullptr;
        pgm_align_batch_destroy(ctx, b);
        return fail(PGM_ERR_NOMEM, "no device memory for the timeline of PGM_FILL_TRACE");
    }
    if (items.size() > std::max<size_t>(1, total_bands)) {   // (cannot happen: an item holds at least one band)
        pgm_align_batch_destroy(ctx, b);
        return fail(PGM_ERR_DEVICE, "work list longer than the number of bands");
    }
    std::vector<int2> tblist;   // the jobs of the fill kernel's main launch (largest first), of its launch for the longest chains, of the band kernel: (job, its last item of the work list)
    {
        std::vector<int> last_item(njobs, 0), group(njobs, 0);
        for (size_t k = 0; k < items.size(); ++k) {
            if (items[k].band + items[k].count == b->jobs[items[k].job].nb) last_item[items[k].job] = (int)k;
            if (k < b->ncrit) group[items[k].job] = 1;
        }
        if (b->nbands) for (const PgmItem &it : bands) group[it.job] = 2;
        uint32_t cnt[3] = {0, 0, 0};
        for (int pass = 0; pass < 3; ++pass)
            for (uint32_t q = 0; q < njobs; ++q) { const uint32_t i = b->order[q]; if (!b->jobs[i].lean && group[i] == pass) { tblist.push_back(make_int2((int)i, last_item[i])); ++cnt[pass]; } }
        b->ntb = cnt[0]; b->ntb_c = cnt[1]; b->ntb_b = cnt[2];
        // workers: the CUs of the kernel each instance follows (they are free by then; nothing of either grid is left waiting
        // for a CU while other kernels of the batch still run)
        const uint32_t all_cus = (uint32_t)std::max(1, ctx->prop.multiProcessorCount) - b->nlean_workers;
        b->ntb_workers = std::max(1u, (b->ntb_b || b->ntb_c) ? b->nworkers : all_cus);
        b->ntb_b_workers = std::max(1u, (b->ntb || b->ntb_c) ? b->nband_workers : all_cus);
        if (tblist.empty()) tblist.push_back(make_int2(0, 0));
    }
    if ((e = (hipError_t)upload_err.load()) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_tblist, tblist.data(), 8 * tblist.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_jobs, b->jobs.data(), sizeof(PgmJob) * njobs, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_order, b->order.data(), 4 * (size_t)njobs, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_items, items.data(), sizeof(PgmItem) * items.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_lean, lean_list.data(), 4 * lean_list.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(b->d_bands, bands.data(), sizeof(PgmItem) * bands.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = classify_lean_jobs(ctx, b)) != hipSuccess ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) {
        pgm_align_batch_destroy(ctx, b);
        return fail(PGM_ERR_DEVICE, std::string("upload: ") + hipGetErrorString(e));
    }
    for (int k = 0; k < 5; ++k) (void)hipEventCreate(&b->ev[k]);
    (void)hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&b->ev_join_b, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&b->ev_join_c, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&b->ev_join_t, hipEventDisableTiming);
    if (cprof)
        fprintf(stderr, "    create: sizes %.2f ms, pinned input block %.2f, flatten %.2f, wait for the allocations %.2f (device %.2f, pinned results %.2f), work list %.2f, upload of %.1f MB %.2f\n",
                tc0 - tcs, tc1 - tc0, tc2 - tc1, tc3 - tc2, tc_alloc, tc_hostalloc, tc4 - tc3, b->in_bytes / 1e6, now_ms() - tc4);
    if (cprof)
        fprintf(stderr, "    device buffers: inputs %.1f MB %.2f ms, work %.1f MB %.2f, cells %.1f MB %.2f, results %.1f MB %.2f, S %.1f MB %.2f, small %.2f\n",
                b->in_bytes / 1e6, tc_slot[0], b->work_bytes / 1e6, tc_slot[1], b->cell_bytes / 1e6, tc_slot[2], b->out_bytes / 1e6, tc_slot[3], b->s_bytes / 1e6, tc_slot[4], tc_slot[5]);
    *out = b;
    return PGM_OK;
}

int pgm_align_batch_run(pgm_ctx *ctx, pgm_align_batch *b) {
    if (!ctx || !b) return fail(PGM_ERR_INVALID, "null argument");
    if (b->njobs == 0) return PGM_OK;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(launch_all(ctx, b, true));   // (stage events: four event records per launch, read by fetch once the stream has completed)
    b->ev_pending = true;
    return PGM_OK;
}

int pgm_align_batch_time(pgm_ctx *ctx, pgm_align_batch *b, int reps, float *ms_prep, float *ms_emission, float *ms_fill,
                         float *ms_traceback) {
    if (!ctx || !b || reps <= 0) return fail(PGM_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps && b->njobs; ++r) {
        HIPCHK(launch_all(ctx, b, true));
        b->ev_pending = false;
        HIPCHK(hipEventSynchronize(b->ev[4]));
        for (int k = 0; k < 4; ++k) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, b->ev[k], b->ev[k + 1]));
            acc[k] += ms;
        }
    }
    if (ms_prep) *ms_prep = (float)(acc[0] / reps);
    if (ms_emission) *ms_emission = (float)(acc[1] / reps);
    if (ms_fill) *ms_fill = (float)(acc[2] / reps);
    if (ms_traceback) *ms_traceback = (float)(acc[3] / reps);
    return PGM_OK;
}

int pgm_align_batch_job_times(pgm_ctx *ctx, pgm_align_batch *b, uint64_t *ticks) {
    if (!ctx || !b || !ticks) return fail(PGM_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (b->njobs) HIPCHK(hipMemcpy(ticks, b->d_times, 16 * (size_t)b->njobs, hipMemcpyDeviceToHost));
    return PGM_OK;
}

int pgm_align_batch_fetch(pgm_ctx *ctx, pgm_align_batch *b, pgm_align_out *out) {
    if (!ctx || !b || (b->njobs && !out)) return fail(PGM_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (b->njobs == 0) { HIPCHK(hipStreamSynchronize(ctx->stream)); return PGM_OK; }
    // results + mappings were written into the pinned block by the kernel itself (PgmJob::hmap1/hmap2/hresult): wait for
    // the stream, then scatter
    int rc = PGM_OK;
    if (tools_env("PGM_FILL_DBG"))   // instrumented kernel variants leave their counters in the device block
        HIPCHK(hipMemcpyAsync(b->h_out, b->d_out, b->out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(b->h_flag, b->d_sync, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    // Finished jobs are copied out while the kernel is still running: a job's traceback worker writes the reversed mappings
    // and then the result record (status word last) into the pinned block; all but the largest jobs are done long before the
    // batch is, so only a few records are left when the stream completes.
    std::vector<uint8_t> copied(b->njobs, 0);
    auto collect = [&]() -> int {
        uint32_t n = 0;
        for (uint32_t i = 0; i < b->njobs; ++i) {
            if (copied[i]) { ++n; continue; }
            const PgmJob::Result *hr = (const PgmJob::Result *)(b->h_out + b->res_off[i]);
            const int32_t st = __atomic_load_n(&hr->status, __ATOMIC_ACQUIRE);
            if (st == PGM_STATUS_PENDING) continue;
            if (!out[i].map1 || !out[i].map2) return -1;
            const uint32_t len = hr->len;
            if (len > b->jobs[i].n1 + b->jobs[i].n2) return -2;
            out[i].score = hr->score; out[i].n_tr_indels = hr->n_tr_indels; out[i].len = len; out[i].status = st;
            memcpy(out[i].map1, b->h_out + b->map1_off[i], 4 * (size_t)len);
            memcpy(out[i].map2, b->h_out + b->map2_off[i], 4 * (size_t)len);
            copied[i] = 1; ++n;
        }
        return (int)n;
    };
    if (!tools_env("PGM_FILL_DBG") && !b->d_trace) {
        for (;;) {
            const int n = collect();
            if (n < 0) { (void)hipStreamSynchronize(ctx->stream); return fail(n == -1 ? PGM_ERR_INVALID : PGM_ERR_DEVICE, n == -1 ? "null mapping buffer" : "corrupt result length"); }
            if ((uint32_t)n == b->njobs) break;
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) break;                       // (an aborted launch leaves records pending)
            if (q != hipErrorNotReady) return fail(PGM_ERR_DEVICE, std::string("fill kernel: ") + hipGetErrorString(q));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (b->ev_pending) {   // stage times of the launch just completed (pgm_align_batch_stage_times)
        b->ev_pending = false;
        float ms[3] = {0, 0, 0};
        bool ok = true;
        for (int k = 0; k < 3; ++k) ok = ok && hipEventElapsedTime(&ms[k], b->ev[k], b->ev[k + 1]) == hipSuccess;
        if (ok) { for (int k = 0; k < 3; ++k) b->acc_ms[k] += ms[k]; ++b->acc_n; }
        else (void)hipGetLastError();
    }
    const int aborted = *b->h_flag;
#ifdef PGM_TOOLS
    if (aborted || tools_env("PGM_DUMP_SYNC")) {   // the counters of the launch, for a post-mortem
        std::vector<int> sy(b->sync_ints);
        (void)hipMemcpy(sy.data(), b->d_sync, 4 * b->sync_ints, hipMemcpyDeviceToHost);
        fprintf(stderr, "sync: abort %d ticket %d lean %d | lq_n %d tb_done %d tb ticket %d | nitems %u nworkers %u nlean %u lean_workers %u ntb %u\n", sy[0], sy[1], sy[2],
                sy[PGM_SY_LQ_N], sy[PGM_SY_TB_DONE], sy[PGM_SY_TBQ_N], b->nitems, b->nworkers, b->nlean, b->nlean_workers, b->ntb);
        for (uint32_t i = 0; i < b->njobs; ++i) {
            const PgmJob &J = b->jobs[i];
            if (J.lean) continue;
            const int *pr = sy.data() + (J.prog - b->d_sync);
            bool done = true;
            for (uint32_t q = 0; q < J.nb; ++q) done = done && pr[q] >= (int)J.tsteps;
            if (!done) { fprintf(stderr, "  job %u (%u x %u, mode2 %u, %u bands): prog", i, J.n1, J.n2, J.mode2, J.nb); for (uint32_t q = 0; q < J.nb; ++q) fprintf(stderr, " %d", pr[q]); fprintf(stderr, " (of %u)\n", J.tsteps); }
        }
    }
#endif
#ifdef PGM_TOOLS
    if (b->d_c3dbg && tools_env("PGM_C3_DBG")) {   // who waits for whom in pgm_crit_kernel: per job, means over its bands (ticks of 10 ns -> us)
        std::vector<unsigned long long> dg(64 * (size_t)b->nitems);
        std::vector<PgmItem> its(b->nitems);
        (void)hipMemcpy(dg.data(), b->d_c3dbg, 512 * (size_t)b->nitems, hipMemcpyDeviceToHost);
        (void)hipMemcpy(its.data(), b->d_items, sizeof(PgmItem) * b->nitems, hipMemcpyDeviceToHost);
        const uint32_t want = (uint32_t)atoi(tools_env("PGM_C3_DBG"));   // job index + 1 (0: every job)
        for (uint32_t j = 0; j < b->njobs; ++j) {
            if (!b->jobs[j].crit3 || (want && want != j + 1)) continue;
            double acc[64] = {0}; uint32_t n = 0;
            for (uint32_t k = 0; k < b->nitems; ++k) if (its[k].job == j) { for (int w = 0; w < 64; ++w) acc[w] += (double)dg[64 * (size_t)k + w]; ++n; }
            if (!n) continue;
            for (double &v : acc) v /= n * 100.0;
            fprintf(stderr, "c3dbg job %u (%u x %u, %u bands, %u steps): chain %.0f us, of it waiting for fold %.0f, for the band above %.0f (%.0f waits per band) | fold 0: %.0f us, waiting for the record %.0f, for helpers %.0f (%.0f late) | fold 1: %.0f / %.0f / %.0f (%.0f) | near: %.0f us waiting %.0f; %.0f / %.0f\n",
                    j, b->jobs[j].n1, b->jobs[j].n2, b->jobs[j].nb, b->jobs[j].tsteps, acc[0], acc[1], acc[2], acc[3] * 100.0, acc[4], acc[5], acc[6], acc[7] * 100.0, acc[8], acc[9], acc[10], acc[11] * 100.0, acc[12], acc[13], acc[16], acc[17]);
            if (want) for (uint32_t k = 0; k < b->nitems; ++k) if (its[k].job == j && (its[k].band % 4u == 0u || its[k].band + 1u == b->jobs[j].nb)) {   // every fourth band of the job asked for
                const unsigned long long *g = dg.data() + 64 * (size_t)k;
                fprintf(stderr, "   band %2u: chain %.0f us (waiting for fold %.0f, for the band above %.0f), %.2f GHz | fold 0 waits: record %.0f, helpers %.0f | near waits %.0f | helpers poll/all: cols", its[k].band, g[0] / 100.0, g[1] / 100.0, g[2] / 100.0,
                        (double)g[60] / std::max<double>(1.0, (double)g[0]) / 10.0, g[5] / 100.0, g[6] / 100.0, g[13] / 100.0);
                for (int q = 0; q < 10; ++q) { const int w = q < 8 ? 20 + q : 28 + q; fprintf(stderr, "%s %.0f/%.0f", q == 4 ? " rows" : "", g[w] / 100.0, g[w + 8] / 100.0); }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "   helpers (poll us / all us): columns");
            for (int k = 0; k < 10; ++k) { const int w = k < 8 ? 20 + k : 28 + k; fprintf(stderr, "%s %.0f/%.0f", k == 4 ? " | rows" : "", acc[w], acc[w + 8]); }
            fprintf(stderr, "\n");
        }
    }
#endif
    if (aborted) return fail(PGM_ERR_DEVICE, "fill kernel: a band hand-off timed out");
    if (b->d_trace) {
        // timeline dump for tools/probe_trace.py: nitems x {worker, start, band end, traceback end} + the item list
        std::vector<unsigned long long> tr(22 * (size_t)b->nitems + 6 * (size_t)b->nlean + 1);
        std::vector<PgmItem> its(std::max(1u, b->nitems));
        std::vector<uint32_t> lj(std::max(1u, b->nlean));
        HIPCHK(hipMemcpy(tr.data(), b->d_trace, 176 * (size_t)b->nitems + 48 * (size_t)b->nlean, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(its.data(), b->d_items, sizeof(PgmItem) * b->nitems, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lj.data(), b->d_lean, 4 * (size_t)b->nlean, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(tools_env("PGM_FILL_TRACE") ? tools_env("PGM_FILL_TRACE") : "/dev/null", "wb")) {
            // nitems, items, 22 words per item; then nlean, the lean jobs in queue order, 6 words per lean job
            fwrite(&b->nitems, 4, 1, f); fwrite(its.data(), sizeof(PgmItem), b->nitems, f); fwrite(tr.data(), 8, 22 * (size_t)b->nitems, f);
            fwrite(&b->nlean, 4, 1, f); fwrite(lj.data(), 4, b->nlean, f); fwrite(tr.data() + 22 * (size_t)b->nitems, 8, 6 * (size_t)b->nlean, f);
            fclose(f);
        }
    }
    for (uint32_t i = 0; i < b->njobs; ++i) {
        PgmJob::Result res;
        memcpy(&res, b->h_out + b->res_off[i], sizeof res);
        if (res.status == PGM_STATUS_PENDING && !tools_env("PGM_FILL_DBG")) return fail(PGM_ERR_DEVICE, "fill kernel: a job's result record was never written");
        if (res.status != PGM_OK) rc = res.status;
        if (copied[i]) continue;
        out[i].score = res.score;
        out[i].n_tr_indels = res.n_tr_indels;
        out[i].len = res.len;
        out[i].status = res.status;
        if (res.status != PGM_OK) rc = res.status;
        if (!out[i].map1 || !out[i].map2) return fail(PGM_ERR_INVALID, "null mapping buffer");
        if (res.len > b->jobs[i].n1 + b->jobs[i].n2) return fail(PGM_ERR_DEVICE, "corrupt result length");
        memcpy(out[i].map1, b->h_out + b->map1_off[i], 4 * (size_t)res.len);
        memcpy(out[i].map2, b->h_out + b->map2_off[i], 4 * (size_t)res.len);
    }
    if (rc != PGM_OK) g_err = "backtracking failed";
    return rc;
}

void pgm_align_batch_destroy(pgm_ctx *ctx, pgm_align_batch *b) {
    if (!b) return;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }   // nothing of the batch is in flight when its buffers go back to the cache
    if (ctx && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx && ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    if (ctx && ctx->stream4) (void)hipStreamSynchronize(ctx->stream4);
    if (ctx && ctx->stream5) (void)hipStreamSynchronize(ctx->stream5);
    for (int k = 0; k < 5; ++k)
        if (b->ev[k]) (void)hipEventDestroy(b->ev[k]);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    if (b->ev_join_b) (void)hipEventDestroy(b->ev_join_b);
    if (b->ev_join_c) (void)hipEventDestroy(b->ev_join_c);
    if (b->ev_join_t) (void)hipEventDestroy(b->ev_join_t);
    cache_give(ctx, pgm_ctx::C_IN, b->d_in, b->cap[pgm_ctx::C_IN]);
    cache_give(ctx, pgm_ctx::C_WORK, b->d_work, b->cap[pgm_ctx::C_WORK]);
    cache_give(ctx, pgm_ctx::C_CELLS, b->d_cells, b->cap[pgm_ctx::C_CELLS]);
    cache_give(ctx, pgm_ctx::C_OUT, b->d_out, b->cap[pgm_ctx::C_OUT]);
    cache_give(ctx, pgm_ctx::C_S, b->d_S, b->cap[pgm_ctx::C_S]);
    cache_give(ctx, pgm_ctx::C_SMALL, b->d_small, b->cap[pgm_ctx::C_SMALL]);
    cache_give(ctx, pgm_ctx::C_HOST, b->h_out, b->cap[pgm_ctx::C_HOST]);
    cache_give(ctx, pgm_ctx::C_HIN, b->h_in, b->cap[pgm_ctx::C_HIN]);
    if (b->d_trace) (void)hipFree(b->d_trace);
    if (b->d_c3dbg) (void)hipFree(b->d_c3dbg);
    delete b;
}

uint64_t pgm_align_batch_cells(const pgm_align_batch *b) { return b ? b->cells : 0; }

int pgm_align_batch_stage_times(pgm_align_batch *b, int reset, float *ms_prep, float *ms_emission, float *ms_fill, uint32_t *launches) {
    if (!b) return fail(PGM_ERR_INVALID, "null batch");
    const double n = b->acc_n ? (double)b->acc_n : 1.0;
    if (ms_prep) *ms_prep = (float)(b->acc_ms[0] / n);
    if (ms_emission) *ms_emission = (float)(b->acc_ms[1] / n);
    if (ms_fill) *ms_fill = (float)(b->acc_ms[2] / n);
    if (launches) *launches = b->acc_n;
    if (reset) { b->acc_ms[0] = b->acc_ms[1] = b->acc_ms[2] = 0; b->acc_n = 0; }
    return PGM_OK;
}

int pgm_align_batch_test_stall(pgm_align_batch *b, uint32_t job, uint32_t band, uint32_t spin_limit) {
    if (!b) return fail(PGM_ERR_INVALID, "null batch");
    b->test_stall_job = job; b->test_stall_band = band; b->test_spin_limit = spin_limit;
    return PGM_OK;
}

int pgm_align_graphs_batch(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                           const pgm_model *const *model, const pgm_scores *scores, pgm_align_out *out) {
    return pgm_align_graphs_batch_res(ctx, njobs, g1, g2, model, scores, nullptr, nullptr, out);
}

int pgm_align_graphs_batch_res(pgm_ctx *ctx, uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                               const pgm_model *const *model, const pgm_scores *scores, const pgm_site_ref *res1, const pgm_site_ref *res2,
                               pgm_align_out *out) {
    pgm_align_batch *b = nullptr;
    const bool prof = getenv("PGM_HOST_PROFILE") != nullptr;   // tools: where the time of one call goes
    const auto t0 = std::chrono::steady_clock::now();
    int rc = pgm_align_batch_create_res(ctx, njobs, g1, g2, model, scores, 0u, res1, res2, &b);
    if (rc != PGM_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    {
        // One stage of persistent grids per DEVICE at a time: every batch sizes its grids for the whole device (no grid of a stage waits
        // for a CU, see pgm_tb_kernel's header), so two contexts on one device take turns here.  (Callers of the create / run / fetch
        // interface with several contexts on a device have to do the same.)
        static std::mutex device_turn[64];
        std::lock_guard<std::mutex> turn(device_turn[(unsigned)ctx->device % 64u]);
        rc = pgm_align_batch_run(ctx, b);
        if (rc == PGM_OK) rc = pgm_align_batch_fetch(ctx, b, out);
    }
    const auto t2 = std::chrono::steady_clock::now();
    pgm_align_batch_destroy(ctx, b);
    if (prof)
        fprintf(stderr, "  pgm_align_graphs_batch: %u jobs, create (flatten, allocate, upload, work list) %.2f ms, run + fetch %.2f ms, destroy %.2f ms\n", njobs,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    return rc;
}

int pgm_align_batch_read_matrices(pgm_ctx *ctx, pgm_align_batch *b, uint32_t job, float *M, float *X, float *Y, float *W, float *S) {
    if (!ctx || !b || job >= b->njobs) return fail(PGM_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const PgmJob &J = b->jobs[job];
    const size_t N = (size_t)J.n1 * J.n2;
    if ((M || X || Y || W) && !J.keep_cells)
        return fail(PGM_ERR_INVALID, "the DP matrices of a chain-only job are only kept for a batch created with PGM_BATCH_KEEP_MATRICES");
    if (S && J.tabhdr) {   // (a job of two sequence graphs looks its scores up in a class table: nothing was written to S)
        int bad = 1;
        HIPCHK(hipMemcpy(&bad, J.tabhdr, sizeof(int), hipMemcpyDeviceToHost));
        if (bad == 0) return fail(PGM_ERR_INVALID, "the score matrix of a job of two sequence graphs is only stored for a batch created with PGM_BATCH_KEEP_MATRICES");
    }
    if (M || X || Y || W) {
        const uint32_t sh = J.rshift, R = 1u << sh;
        const size_t ncell = (size_t)J.nb * J.tsteps * 64u * R;
        std::vector<float4> cells(ncell);
        HIPCHK(hipMemcpy(cells.data(), J.cells, ncell * sizeof(float4), hipMemcpyDeviceToHost));
        float *dst[4] = {M, X, Y, W};
        for (int k = 0; k < 4; ++k)
            if (dst[k]) std::fill(dst[k], dst[k] + N, -INFINITY);
        for (uint32_t y = 0; y + 1 < J.n1; ++y)
            for (uint32_t x = 0; x < J.ncol; ++x) {
                const uint32_t bb = y >> (6u + sh), w = y & ((64u << sh) - 1u), l = w >> sh, r = w & (R - 1u);
                const float4 c = cells[((((size_t)bb * J.tsteps + (x + l)) << sh) | r) * 64u + l];  // {M, X, W, Y}
                const size_t i = (size_t)y + (size_t)J.n1 * x;
                if (M) M[i] = c.x;
                if (X) X[i] = c.y;
                if (Y) Y[i] = c.w;
                if (W) W[i] = c.z;
            }
    }
    if (S) {
        // the emission scores exactly as the fill kernel consumes them (PgmJob::S, written by pgm_emission_skew_kernel), de-skewed
        const uint32_t sh = J.rshift, R = 1u << sh;
        const size_t ns = (size_t)J.nb * J.nblk * 64u * PGM_BLOCK * R;
        std::vector<float> sk(ns);
        HIPCHK(hipMemcpy(sk.data(), J.S, ns * sizeof(float), hipMemcpyDeviceToHost));
        std::fill(S, S + N, 0.0f);
        for (uint32_t y = 0; y + 1 < J.n1; ++y)
            for (uint32_t x = 0; x < J.ncol; ++x) {
                const uint32_t bb = y >> (6u + sh), w = y & ((64u << sh) - 1u), l = w >> sh, r = w & (R - 1u), t = x + l;
                S[(size_t)y + (size_t)J.n1 * x] = sk[(((((size_t)bb * J.nblk + t / PGM_BLOCK) << sh) | r) * 64u + l) * PGM_BLOCK + t % PGM_BLOCK];
            }
    }
    return PGM_OK;
}

}  // extern "C"

#include "pgm_nw_capi.inc"
#include "pgm_csprofile_capi.inc"
#include "pgm_dist_capi.inc"
#include "pgm_merge_capi.inc"
