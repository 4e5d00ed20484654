// graph_align.cpp — host side of reference src/GraphAlign.h: DynProgScores (:98-143),
// the alignGraphs call surface (:200-534, executed by the backend behind the C ABI) and
// mergeGraphs (:550-727) with the reference's quirks kept.
// NOTE: planMerge / finishMerge below are DERIVED FROM reference src/GraphAlign.h:551-727 (mergeGraphs, restated with its quirks,
// identifiers kept): host scaffolding around the device kernels; the arithmetic of the merge (node profiles) runs on the GPU
// (csrc/pgm_merge_kernels.h), the edge bookkeeping here is the reference's.
#include "pgm_host.h"

#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>

namespace pgm {

#define LOG(x) (std::log(x) / std::log(2))  // GraphAlign.h:46-48 (USE_LS_LOG build)

// averageAlignmentLength (GraphAlign.h:56-96): the mean length of the paths from START to END over edges of cost 0, defined by a
// memoised recursion over the predecessors — value(v) = mean over the cost-0 predecessors p with a path of (value(p) + 1), "no path"
// if there is none.  Edges point to earlier nodes, so one pass in ascending node order evaluates the same expression for every node
// from the same operands in the same order (the recursion's depth reaches n: 1000 stack frames for a leaf graph, 1 ms of a 128-job
// level on sixteen threads); nodes the recursion would not have visited get a value nobody reads.
double averageAlignmentLength(const Graph &g) {
    if (g.size() == 0) return 0;
    const index_t n = g.size();
    std::vector<double> value(n, -2.0);   // -2: no path from START
    value[0] = 0;
    for (index_t v = 1; v < n; ++v) {
        double sum = 0.0;
        index_t paths = 0;
        for (Graph::PredIterator it = g.getPreds(v, INFINITY, INFINITY); it; ++it)
            if (it.value() == 0.0) {
                const double res = value[*it];
                if (res >= 0.0) { sum += res + 1.0; ++paths; }
            }
        value[v] = paths > 0 ? sum / paths : -2.0;
    }
    return value[n - 1];
}

pgm_scores DynProgScores(const Graph &g1, const Graph &g2, const Model &model) {  // GraphAlign.h:100-131
    const double l1 = averageAlignmentLength(g1);
    const double l2 = averageAlignmentLength(g2);
    const double exp_length = std::max(l1, l2) * std::exp(model.distance * cmdlineopts.indel_rate * (model.epsilon / (1.0 - model.epsilon) + 1.0));
    const double nu = 2.0 / (2.0 + l1 + l2);
    double ttau = 1.0 / (1.0 + exp_length);
    if (model.epsilon + ttau >= 1.0) ttau = (1.0 - model.epsilon) / 2.0;
    const double tau = ttau;
    const double E = cmdlineopts.end_indel_prob;
    pgm_scores s;
    s.gap_init = (float)LOG(model.delta * (1.0 - model.epsilon - tau) / (1.0 - nu));
    s.gap_extend = (float)LOG(model.epsilon / (1.0 - nu));
    s.match_init = (float)LOG((1.0 - 2.0 * model.delta) * (1.0 - tau) / (1.0 - nu) / (1.0 - nu));
    s.end_skip = (float)LOG(tau);
    if (E >= 0 && E <= 1) {
        s.end_match = (float)LOG(tau * (1.0 - E) / (1.0 - 2.0 * model.delta) / (1.0 - tau));
        s.end_gap = (float)LOG(tau * E / 2.0 / (1.0 - model.epsilon - tau) / model.delta);
        s.start_gap = (float)LOG(E / 2.0 * (1.0 - model.epsilon - tau) / (1.0 - E) / (1.0 - nu));
        s.start_init = (float)LOG((1.0 - tau) * (1.0 - E));
    } else {
        s.end_match = (float)LOG(tau / (1.0 - tau));
        s.end_gap = (float)LOG(tau / (1.0 - model.epsilon - tau));
        s.start_gap = (float)LOG(model.delta * (1.0 - model.epsilon - tau) / (1.0 - nu));
        s.start_init = (float)LOG(1.0 - tau);
    }
    const double repeat_prob = 1.0 - std::exp(-model.distance * cmdlineopts.repeat_rate);
    s.repeat_init = (float)-LOG(std::min<double>(1, repeat_prob / (1 - repeat_prob) * (1 - cmdlineopts.repeatext_prob)));
    s.repeat_ext = (float)-LOG(std::min<double>(1, std::max<double>(0, cmdlineopts.repeatext_prob)));
    return s;
}

// ---------------------------------------------------------------------------------------
// Job dump: a flat binary record per alignGraphs call so that tests and bench.py can replay the
// exact jobs of a progressive pass through the C ABI (format documented in prographmsa_amd/jobs.py).
static std::string g_dump_path;
const HostSwitches &host_switches() {
    static const HostSwitches sw = []() {
        HostSwitches h;
        h.profile = getenv("PGM_HOST_PROFILE") != nullptr;
        h.host_merge = getenv("PGM_HOST_MERGE") != nullptr;
        h.no_resident = getenv("PGM_NO_RESIDENT") != nullptr;
        h.host_counts = getenv("PGM_HOST_COUNTS") != nullptr;
        h.device_mldist = getenv("PGM_DEVICE_MLDIST") != nullptr;
        return h;
    }();
    return sw;
}
std::string HostSwitches::describe() const {
    std::string s;
    auto add = [&](bool on, const char *name) { if (on) { if (!s.empty()) s += ","; s += name; } };
    add(host_merge, "PGM_HOST_MERGE"); add(no_resident, "PGM_NO_RESIDENT"); add(host_counts, "PGM_HOST_COUNTS"); add(device_mldist, "PGM_DEVICE_MLDIST");
    return s;
}

void set_job_dump(const std::string &path) {
    g_dump_path = path;
    if (!path.empty()) { FILE *f = fopen(path.c_str(), "wb"); if (f) fclose(f); }
}
bool job_dump_active() { return !g_dump_path.empty(); }
static void dump_graph(FILE *f, const pgm_graph &g) {
    uint32_t hdr[4] = {g.n, g.dim, (uint32_t)g.e_rowptr[g.n], g.r_rowptr ? (uint32_t)g.r_rowptr[g.n] : 0u};
    fwrite(hdr, 4, 4, f);
    fwrite(g.sites, 8, (size_t)g.n * g.dim, f);
    fwrite(g.e_rowptr, 4, g.n + 1, f);
    fwrite(g.e_col, 4, hdr[2], f);
    fwrite(g.e_val, 4, hdr[2], f);
    if (hdr[3]) {
        fwrite(g.r_rowptr, 4, g.n + 1, f);
        fwrite(g.r_col, 4, hdr[3], f);
        fwrite(g.r_units, 4, hdr[3], f);
    }
}
static void dump_job(const pgm_graph &g1, const pgm_graph &g2, const pgm_model &m, const pgm_scores &s) {
    FILE *f = fopen(g_dump_path.c_str(), "ab");
    if (!f) return;
    uint32_t magic = 0x4a4d4750u;  // "PGMJ"
    fwrite(&magic, 4, 1, f);
    dump_graph(f, g1);
    dump_graph(f, g2);
    fwrite(m.M, 8, (size_t)g1.dim * g1.dim, f);
    fwrite(m.pi, 8, g1.dim, f);
    fwrite(&s, sizeof s, 1, f);
    fclose(f);
}

std::vector<std::vector<uint32_t>> farm_shards(const std::vector<uint64_t> &cost, int nw) {
    const uint32_t n = (uint32_t)cost.size();
    nw = std::max(1, std::min<int>(nw, (int)std::max(1u, n)));
    std::vector<std::vector<uint32_t>> sh((size_t)nw);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<uint64_t> load((size_t)nw, 0);
    for (uint32_t i : order) {
        size_t w = 0;
        for (size_t k = 1; k < load.size(); ++k) if (load[k] < load[w]) w = k;
        sh[w].push_back(i);
        load[w] += std::max<uint64_t>(cost[i], 1);
    }
    while (sh.size() > 1 && sh.back().empty()) sh.pop_back();
    return sh;
}

void farm_run(const std::vector<std::vector<uint32_t>> &shards, const std::function<void(int)> &fn) {
    std::vector<std::thread> th;
    for (size_t w = 1; w < shards.size(); ++w) if (!shards[w].empty()) th.emplace_back(fn, (int)w);
    if (!shards.empty() && !shards[0].empty()) fn(0);
    for (auto &t : th) t.join();
}

std::vector<AlignmentResult> alignGraphsBatch(const std::vector<const Graph *> &g1, const std::vector<const Graph *> &g2,
                                              const std::vector<const Model *> &model, const std::vector<pgm_site_ref> &res1,
                                              const std::vector<pgm_site_ref> &res2, const std::vector<int> *worker_of) {
    const uint32_t n = (uint32_t)g1.size();
    std::vector<pgm_graph> f1(n), f2(n);
    std::vector<pgm_model> fm(n);
    std::vector<pgm_scores> sc(n);
    std::vector<const pgm_graph *> p1(n), p2(n);
    std::vector<const pgm_model *> pm(n);
    std::vector<pgm_align_out> out(n);
    std::vector<AlignmentResult> res(n);
    Backend &be = default_backend();
    parallel_for(n, [&](size_t i) {   // (DynProgScores walks both graphs: averageAlignmentLength)
        f1[i] = g1[i]->flat();
        f2[i] = g2[i]->flat();
        fm[i].M = model[i]->M.data();
        fm[i].pi = model[i]->pi.data();
        sc[i] = DynProgScores(*g1[i], *g2[i], *model[i]);
        p1[i] = &f1[i]; p2[i] = &f2[i]; pm[i] = &fm[i];
        res[i].mapping1.assign(g1[i]->size() + g2[i]->size(), 0);
        res[i].mapping2.assign(g1[i]->size() + g2[i]->size(), 0);
        out[i].map1 = res[i].mapping1.data();
        out[i].map2 = res[i].mapping2.data();
        out[i].len = 0;
        out[i].status = 0;
    });
    for (uint32_t i = 0; i < n; ++i) {
        be.cells_aligned += (uint64_t)(g1[i]->size() - 2) * (g2[i]->size() - 2);
        if (!g_dump_path.empty()) {
            if (!f1[i].sites || !f2[i].sites) error("--dump_jobs needs the profiles on the host (PGM_NO_RESIDENT=1)");
            dump_job(f1[i], f2[i], fm[i], sc[i]);
        }
    }
    auto t0 = std::chrono::steady_clock::now();
    // The jobs of a level are independent (sibling subtrees, ProgressiveAlignment.cpp:50-51): with several device contexts
    // they are dealt to the workers by DP cells, longest first; every worker runs its shard as one batch on its own context.
    std::vector<uint64_t> cost(n);
    for (uint32_t i = 0; i < n; ++i) cost[i] = (uint64_t)g1[i]->size() * g2[i]->size();
    std::vector<std::vector<uint32_t>> shards;
    if (worker_of && be.workers() > 1) {
        // a pass sharded by subtree: every job runs where its children's profiles are (largest first within a worker, like farm_shards)
        shards.assign((size_t)be.workers(), std::vector<uint32_t>());
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
        for (uint32_t i : order) shards[(size_t)(*worker_of)[i] % shards.size()].push_back(i);
    } else shards = farm_shards(cost, be.workers());
    size_t used = 0;
    for (const auto &sh : shards) used += !sh.empty();
    if (shards.size() <= 1) {
        be.align_graphs_batch(n, p1.data(), p2.data(), pm.data(), sc.data(), out.data(), 0, res1.empty() ? nullptr : res1.data(), res2.empty() ? nullptr : res2.data());
    } else if (worker_of) {
        farm_run(shards, [&](int w) {
            const std::vector<uint32_t> &sh = shards[(size_t)w];
            const uint32_t m = (uint32_t)sh.size();
            std::vector<const pgm_graph *> q1(m), q2(m);
            std::vector<const pgm_model *> qm(m);
            std::vector<pgm_scores> qs(m);
            std::vector<pgm_align_out> qo(m);
            std::vector<pgm_site_ref> r1(res1.empty() ? 0 : m), r2(res2.empty() ? 0 : m);
            for (uint32_t k = 0; k < m; ++k) {
                q1[k] = p1[sh[k]]; q2[k] = p2[sh[k]]; qm[k] = pm[sh[k]]; qs[k] = sc[sh[k]]; qo[k] = out[sh[k]];
                if (!res1.empty()) r1[k] = res1[sh[k]];
                if (!res2.empty()) r2[k] = res2[sh[k]];
            }
            be.align_graphs_batch(m, q1.data(), q2.data(), qm.data(), qs.data(), qo.data(), w, r1.empty() ? nullptr : r1.data(), r2.empty() ? nullptr : r2.data());
            for (uint32_t k = 0; k < m; ++k) out[sh[k]] = qo[k];
        });
    } else {
        if (!res1.empty() || !res2.empty()) error("alignGraphsBatch: resident profiles need the jobs' workers (worker_of)");
        farm_run(shards, [&](int w) {
            const std::vector<uint32_t> &sh = shards[(size_t)w];
            const uint32_t m = (uint32_t)sh.size();
            std::vector<const pgm_graph *> q1(m), q2(m);
            std::vector<const pgm_model *> qm(m);
            std::vector<pgm_scores> qs(m);
            std::vector<pgm_align_out> qo(m);
            for (uint32_t k = 0; k < m; ++k) { q1[k] = p1[sh[k]]; q2[k] = p2[sh[k]]; qm[k] = pm[sh[k]]; qs[k] = sc[sh[k]]; qo[k] = out[sh[k]]; }
            be.align_graphs_batch(m, q1.data(), q2.data(), qm.data(), qs.data(), qo.data(), w);
            for (uint32_t k = 0; k < m; ++k) out[sh[k]] = qo[k];
        });
    }
    be.farm_level_workers = std::max(be.farm_level_workers, (int)std::max<size_t>(used, 1));
    be.seconds_align += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t i = 0; i < n; ++i) {
        if (out[i].status != PGM_OK) error("backtracking failed");  // GraphAlign.h:410
        res[i].score = out[i].score;
        res[i].n_tr_indels = out[i].n_tr_indels;
        res[i].mapping1.resize(out[i].len);
        res[i].mapping2.resize(out[i].len);
    }
    return res;
}

AlignmentResult alignGraphs(const Graph &g1, const Graph &g2, const Model &model) {
    return alignGraphsBatch({&g1}, {&g2}, {&model})[0];
}

// ---------------------------------------------------------------------------------------
// y = P * g(:,i), P column-major
// Association follows Eigen 3.0-3.2's column-major gemv (four columns at a time,
// res += (c0 v0 + c1 v1) + (c2 v2 + c3 v3), leftover columns one by one): the binary's Eigen
// version is not pinned, and only the last bit of a double depends on this.
static void matvec(const std::vector<double> &P, const double *v, int n, std::vector<double> &out) {
    out.assign(n, 0.0);
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        const double *c0 = &P[(size_t)n * j], *c1 = c0 + n, *c2 = c1 + n, *c3 = c2 + n;
        const double v0 = v[j], v1 = v[j + 1], v2 = v[j + 2], v3 = v[j + 3];
        for (int i = 0; i < n; ++i) out[i] += (c0[i] * v0 + c1[i] * v1) + (c2[i] * v2 + c3[i] * v3);
    }
    for (; j < n; ++j) {
        const double vj = v[j];
        const double *pc = &P[(size_t)n * j];
        for (int i = 0; i < n; ++i) out[i] += pc[i] * vj;
    }
}
static void normalize(std::vector<double> &p) {  // p.norm()==0 ? p : p.normalized()
    double s = 0;
    for (double v : p) s += v * v;
    double nrm = std::sqrt(s);
    if (nrm == 0) return;
    double inv = 1.0 / nrm;  // Eigen 3.0-3.2 scalar quotient = multiplication by the reciprocal
    for (double &v : p) v *= inv;
}

// The edge records of a merged graph in the order of the reference's ordered maps — by (to, from), of equal keys the smallest value
// (updateEdge, GraphAlign.h:538-547) — without a comparison sort: the records arrive as a few runs that ascend in `to`, a node has a
// handful of predecessors, so a counting pass over `to` and an insertion sort inside every node's few records does it in O(n).
template <class Rec, class Value>
static void order_records(std::vector<Rec> &v, size_t nnodes, Value value_of) {
    std::vector<uint32_t> start(nnodes + 1, 0);
    for (const Rec &r : v) start[r.to + 1]++;
    for (size_t i = 0; i < nnodes; ++i) start[i + 1] += start[i];
    std::vector<Rec> out(v.size());
    {
        std::vector<uint32_t> at(start.begin(), start.end() - 1);
        for (const Rec &r : v) out[at[r.to]++] = r;
    }
    size_t w = 0;
    for (size_t node = 0; node < nnodes; ++node) {
        const size_t b = start[node], e = start[node + 1];
        for (size_t i = b + 1; i < e; ++i) {
            const Rec r = out[i];
            size_t j = i;
            while (j > b && (out[j - 1].from > r.from || (out[j - 1].from == r.from && value_of(out[j - 1]) > value_of(r)))) { out[j] = out[j - 1]; --j; }
            out[j] = r;
        }
        for (size_t i = b; i < e; ++i)
            if (i == b || out[i].from != out[i - 1].from) out[w++] = out[i];   // the cheapest of equal keys comes first
    }
    out.resize(w);
    v.swap(out);
}

MergePlan planMerge(const Graph &g1, const Graph &g2, const std::vector<index_t> &mapping1, const std::vector<index_t> &mapping2) {
    const index_t NONE = (index_t)-1;
    MergePlan plan;
    auto node = [&](index_t k1, index_t k2, bool matched, bool p1_for_g2) {
        plan.mapping1.push_back(k1); plan.mapping2.push_back(k2);
        plan.is_matched.push_back(matched); plan.g2_with_P1.push_back(p1_for_g2 ? 1 : 0);
    };
    /* unify graphs (GraphAlign.h:569-620) */
    for (index_t i1 = 0, i2 = 0, j = 0; j < mapping1.size(); ++j) {
        index_t k1 = mapping1[j], k2 = mapping2[j];
        if (k1 != NONE) {
            for (; i1 != k1; ++i1) node(i1, NONE, false, false);
            ++i1;
        }
        if (k2 != NONE) {
            for (; i2 != k2; ++i2) node(NONE, i2, false, true);   // model1 (sic), GraphAlign.h:591
            ++i2;
        }
        if (k1 == NONE && k2 == NONE) error("error in mapping");
        node(k1, k2, true, false);
    }
    (void)g1; (void)g2;
    return plan;
}

void mergeProfilesHost(const Graph &g1, const Graph &g2, const Model &model1, const Model &model2, const MergePlan &plan,
                       std::vector<double> &profiles) {
    const int D = g1.dim();
    const index_t NONE = (index_t)-1;
    const size_t nn = plan.mapping1.size();
    profiles.assign((size_t)D * nn, 0.0);
    std::vector<double> p, q;
    for (size_t v = 0; v < nn; ++v) {
        const index_t k1 = plan.mapping1[v], k2 = plan.mapping2[v];
        if (k1 != NONE && k2 != NONE) {
            matvec(model1.P, g1.col(k1), D, p);
            matvec(plan.g2_with_P1[v] ? model1.P : model2.P, g2.col(k2), D, q);
            for (int a = 0; a < D; ++a) p[a] *= q[a];
        } else if (k1 != NONE) {
            matvec(model1.P, g1.col(k1), D, p);
        } else {
            matvec(plan.g2_with_P1[v] ? model1.P : model2.P, g2.col(k2), D, p);
        }
        normalize(p);
        std::copy(p.begin(), p.end(), profiles.begin() + (size_t)D * v);
    }
}

AncestralResult finishMerge(const Graph &g1, const Graph &g2, const MergePlan &plan, const double *profiles,
                            double support1, double support2) {
    const int D = g1.dim();
    const index_t NONE = (index_t)-1;
    AncestralResult result;
    result.mapping1 = plan.mapping1;
    result.mapping2 = plan.mapping2;
    result.is_matched = plan.is_matched;
    // The reference collects the merged graph's edges in two ordered maps keyed (to, from), an existing edge keeping the smaller
    // cost (updateEdge, GraphAlign.h:539-548).  Here they are appended to flat lists, sorted by the same key and reduced with the
    // same minimum: the same set of (key, cost) in the same order, without a tree node per edge.
    const size_t nnodes = plan.mapping1.size();
    std::vector<Graph::EdgeRec> edges;
    std::vector<Graph::RepeatRec> repeats;
    edges.reserve(3 * nnodes);
    auto updateEdge = [](std::vector<Graph::EdgeRec> &l, index_t from, index_t to, dp_score_t cost) { l.push_back(Graph::EdgeRec{to, from, cost}); };
    auto updateRepeat = [](std::vector<Graph::RepeatRec> &l, index_t from, index_t to, index_t units) { l.push_back(Graph::RepeatRec{to, from, units}); };

    /* homologous path (GraphAlign.h:626-657) */
    index_t last_xy = 0, last_x = 0, last_y = 0, last_mapped = 0;
    for (index_t i = 1; i < nnodes; ++i) {
        if (!result.is_matched[i]) continue;
        updateEdge(edges, last_mapped, i, (dp_score_t)0);
        last_mapped = i;
        if (result.mapping1[i] != NONE && result.mapping2[i] != NONE) {
            if (last_xy != i - 1) updateEdge(edges, last_xy, i, (dp_score_t)0);
            last_xy = i;
        }
        if (result.mapping1[i] != NONE) {
            if (last_y != i - 1) updateEdge(edges, last_y, i, (dp_score_t)0);
            last_y = i;
        }
        if (result.mapping2[i] != NONE) {
            if (last_x != i - 1) updateEdge(edges, last_x, i, (dp_score_t)0);
            last_x = i;
        }
    }

    /* inverse mappings (GraphAlign.h:661-673) */
    std::vector<index_t> inv_mapping1(g1.size(), 0), inv_mapping2(g2.size(), 0);
    for (index_t i = 0; i < result.mapping1.size(); ++i)
        if (result.mapping1[i] != NONE) inv_mapping1[result.mapping1[i]] = i;
    for (index_t i = 0; i < result.mapping2.size(); ++i)
        if (result.mapping2[i] != NONE) inv_mapping2[result.mapping2[i]] = i;

    /* penalties for unused edges (GraphAlign.h:677-681) */
    double unused_prob1 = cmdlineopts.altsplice_prob + (1.0 - cmdlineopts.altsplice_prob) * (1.0 - support1);
    dp_score_t unused_penalty1 = (dp_score_t)-LOG(unused_prob1);
    double unused_prob2 = cmdlineopts.altsplice_prob + (1.0 - cmdlineopts.altsplice_prob) * (1.0 - support2);
    dp_score_t unused_penalty2 = (dp_score_t)-LOG(unused_prob2);

    /* add missing edges (GraphAlign.h:685-722); is_matched is indexed with SOURCE-graph indices
     * although it is in merged indexing — kept as in the reference. */
    for (int side = 0; side < 2; ++side) {
        const Graph &g = side == 0 ? g1 : g2;
        const std::vector<index_t> &inv = side == 0 ? inv_mapping1 : inv_mapping2;
        const dp_score_t pen = side == 0 ? unused_penalty1 : unused_penalty2;
        for (index_t to = 0; to < g.size(); ++to) {
            for (Graph::PredIterator from = g.getPreds(to, 0, 0); from; ++from) {
                index_t y = inv[*from];
                index_t x = inv[to];
                if (!from.isRepeat()) {
                    if (result.is_matched[*from] && result.is_matched[to]) updateEdge(edges, y, x, (dp_score_t)(from.value() + pen));
                    else if (result.is_matched[*from] || result.is_matched[to]) updateEdge(edges, y, x, (dp_score_t)(from.value() + pen / 2));
                    else updateEdge(edges, y, x, from.value());
                } else {
                    updateRepeat(repeats, y, x, from.repeatUnits());
                }
            }
        }
    }
    order_records(edges, nnodes, [](const Graph::EdgeRec &r) { return r.cost; });
    order_records(repeats, nnodes, [](const Graph::RepeatRec &r) { return r.units; });
    result.graph = Graph(D, (index_t)nnodes, profiles, edges, repeats);
    return result;
}

AncestralResult mergeGraphs(const Graph &g1, const Graph &g2, const std::vector<index_t> &mapping1,
                            const std::vector<index_t> &mapping2, const Model &model1, const Model &model2,
                            double support1, double support2) {
    const MergePlan plan = planMerge(g1, g2, mapping1, mapping2);
    std::vector<double> profiles;
    mergeProfilesHost(g1, g2, model1, model2, plan, profiles);
    return finishMerge(g1, g2, plan, profiles.data(), support1, support2);
}

// mergeGraphsIncremental (GraphAlign.h:729-882).  Same walk as planMerge, one model: a node of the ancestral graph keeps its
// column, a node of `graph` shows P g, a matched pair their element-wise product; every column normalised.  The homologous path is
// the one of finishMerge; the edges of both graphs are carried over at their own cost (no penalties for unused edges).
AncestralResult mergeGraphsIncremental(const Graph &anc_graph, const Graph &graph, const std::vector<index_t> &anc_mapping,
                                       const std::vector<index_t> &mapping, const Model &model) {
    const int D = anc_graph.dim();
    const index_t NONE = (index_t)-1;
    if (anc_mapping.size() != mapping.size()) error("mergeGraphsIncremental: mappings of different lengths");
    AncestralResult result;
    std::vector<double> profiles, p, q;
    auto node = [&](index_t k1, index_t k2, bool matched) {
        if (k1 != NONE && k2 != NONE) {
            matvec(model.P, graph.col(k2), D, q);
            p.assign(anc_graph.col(k1), anc_graph.col(k1) + D);
            for (int a = 0; a < D; ++a) p[(size_t)a] *= q[(size_t)a];
        } else if (k1 != NONE) p.assign(anc_graph.col(k1), anc_graph.col(k1) + D);
        else matvec(model.P, graph.col(k2), D, p);
        normalize(p);
        profiles.insert(profiles.end(), p.begin(), p.end());
        result.mapping1.push_back(k1); result.mapping2.push_back(k2); result.is_matched.push_back(matched);
    };
    /* unify graphs (:747-797) */
    for (index_t i1 = 0, i2 = 0, j = 0; j < anc_mapping.size(); ++j) {
        const index_t k1 = anc_mapping[j], k2 = mapping[j];
        if (k1 != NONE) {
            for (; i1 != k1; ++i1) node(i1, NONE, false);
            ++i1;
        }
        if (k2 != NONE) {
            for (; i2 != k2; ++i2) node(NONE, i2, false);
            ++i2;
        }
        if (k1 == NONE && k2 == NONE) error("error in mapping");
        node(k1, k2, true);
    }
    const size_t nnodes = result.mapping1.size();
    std::vector<Graph::EdgeRec> edges;
    std::vector<Graph::RepeatRec> repeats;
    auto updateEdge = [&](index_t from, index_t to, dp_score_t cost) { edges.push_back(Graph::EdgeRec{to, from, cost}); };
    /* homologous path and allow skipping newly inserted gaps (:803-834) */
    index_t last_xy = 0, last_x = 0, last_y = 0, last_mapped = 0;
    for (index_t i = 1; i < nnodes; ++i) {
        if (!result.is_matched[i]) continue;
        updateEdge(last_mapped, i, (dp_score_t)0);
        last_mapped = i;
        if (result.mapping1[i] != NONE && result.mapping2[i] != NONE) {
            if (last_xy != i - 1) updateEdge(last_xy, i, (dp_score_t)0);
            last_xy = i;
        }
        if (result.mapping1[i] != NONE) {
            if (last_y != i - 1) updateEdge(last_y, i, (dp_score_t)0);
            last_y = i;
        }
        if (result.mapping2[i] != NONE) {
            if (last_x != i - 1) updateEdge(last_x, i, (dp_score_t)0);
            last_x = i;
        }
    }
    /* inverse mappings (:838-851), the edges of both graphs (:855-880) */
    std::vector<index_t> inv1(anc_graph.size(), 0), inv2(graph.size(), 0);
    for (index_t i = 0; i < nnodes; ++i) {
        if (result.mapping1[i] != NONE) inv1[result.mapping1[i]] = i;
        if (result.mapping2[i] != NONE) inv2[result.mapping2[i]] = i;
    }
    for (int side = 0; side < 2; ++side) {
        const Graph &g = side == 0 ? anc_graph : graph;
        const std::vector<index_t> &inv = side == 0 ? inv1 : inv2;
        for (index_t to = 0; to < g.size(); ++to)
            for (Graph::PredIterator from = g.getPreds(to, 0, 0); from; ++from) {
                if (!from.isRepeat()) updateEdge(inv[*from], inv[to], from.value());
                else repeats.push_back(Graph::RepeatRec{inv[to], inv[*from], from.repeatUnits()});
            }
    }
    // (updateEdge, GraphAlign.h:538-547: of two edges with the same ends the cheaper one, of two repeats the one with fewer units)
    order_records(edges, nnodes, [](const Graph::EdgeRec &r) { return r.cost; });
    order_records(repeats, nnodes, [](const Graph::RepeatRec &r) { return r.units; });
    result.graph = Graph(D, (index_t)nnodes, profiles.data(), edges, repeats);
    return result;
}

}  // namespace pgm
