// distance.cpp — pairwise distances and the BioNJ guide tree for the `-a` path
// (reference src/DistanceFactoryAlign.{h,cpp}, DistanceFactoryML.{h,cpp}, TreeNJ.{h,cpp}).
// The O(L^2) Needleman-Wunsch of every pair (alignPair) runs behind the C ABI
// (pgm_nw_pairs_batch); ML distance estimation and neighbour joining stay on the host.
#include "pgm_host.h"
#include <quadmath.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <fstream>
#include <thread>

namespace pgm {

// ---- DistanceFactoryML ---------------------------------------------------------------------
static void consts(const Alphabet &a, double &DIST_MAX, double &VAR_MAX, double &VAR_MIN) {  // DistanceFactoryML.cpp
    if (a.kind == ALPHA_AA) { DIST_MAX = 2.2; VAR_MAX = 1e3; VAR_MIN = 1e-5; }
    else { DIST_MAX = 5.2; VAR_MAX = 5e3; VAR_MIN = 1e-5; }
}

static void matmul(const std::vector<double> &A, const std::vector<double> &B, int n, std::vector<double> &C) {
    C.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) {
            double b = B[k + n * j];
            for (int i = 0; i < n; ++i) C[i + n * j] += A[i + n * k] * b;
        }
}

distvar_t DistanceFactoryML::computeMLDist(const std::vector<int32_t> &counts, index_t gaps, double seqlen, double dist0,
                                           double var0) const {  // DistanceFactoryML.h:66-135
    const int n = alphabet.DIM;
    double DIST_MAX, VAR_MAX, VAR_MIN;
    consts(alphabet, DIST_MAX, VAR_MAX, VAR_MIN);
    const double EPSILON = 1e-5;
    const index_t MAXITER = 20;
    double dist_min = 0, dist_max = INFINITY;
    double dist = dist0, var = var0;
    double delta = 1;
    index_t iteration = 0;
    std::vector<double> pp, ppp;
    while (std::abs(delta) > EPSILON) {
        if (iteration > MAXITER) {
            if (dist_max == INFINITY) { dist = DIST_MAX; var = VAR_MAX; }
            else { dist = dist0; var = var0; }
            break;
        }
        Model model = model_factory->getModel(dist);
        const std::vector<double> &p = model.P;
        matmul(model.Q, p, n, pp);
        matmul(model.Q, pp, n, ppp);
        double f = 0, ff = 0;
        for (size_t i = 0; i < p.size(); ++i) {
            double c = counts[i];
            f += c * pp[i] / p[i];
            ff += (c * (ppp[i] * p[i] - pp[i] * pp[i])) / (p[i] * p[i]);
        }
        if (cmdlineopts.mldist_gap_flag) {
            double grate = cmdlineopts.indel_rate * seqlen * dist;
            f += (-grate + gaps) / dist;
            ff += -(double)gaps / (dist * dist);
        }
        var = -1.0 / ff;
        if (f > 0) dist_min = std::max(dist_min, dist);
        else dist_max = std::min(dist_max, dist);
        double new_dist = dist - f / ff;
        if (!(new_dist < dist_max && new_dist > dist_min)) {
            double upper = (dist_max == INFINITY) ? dist * 3 : dist_max;
            new_dist = (upper + dist_min) / 2.0;
        }
        delta = 1.0 - new_dist / dist;
        dist = new_dist;
        ++iteration;
    }
    return distvar_t{dist, var};
}

distvar_t DistanceFactoryML::computeDistance(const std::vector<int32_t> &counts, index_t gaps, double seqlen) const {
    const int n = alphabet.DIM;  // DistanceFactoryML.h:137-190
    double DIST_MAX, VAR_MAX, VAR_MIN;
    consts(alphabet, DIST_MAX, VAR_MAX, VAR_MIN);
    double ident = 0, total = 0;
    for (int i = 0; i < n; ++i) ident += counts[i + n * i];
    for (int32_t c : counts) total += c;
    return computeDistance(ident, total, &counts, gaps, seqlen);
}

// (ident, total) are sums of integers, exact in any order: the all-pairs stage reduces them on the device when nothing else of the
// count matrix is read (no --mldist: counts == nullptr)
distvar_t DistanceFactoryML::computeDistance(double ident, double total, const std::vector<int32_t> *counts, index_t gaps, double seqlen) const {
    double DIST_MAX, VAR_MAX, VAR_MIN;
    consts(alphabet, DIST_MAX, VAR_MAX, VAR_MIN);
    double dist0 = 1.0 - ident / total;
    double dist, var;
    if (cmdlineopts.mldist_flag || cmdlineopts.mldist_gap_flag) {
        if (total == 0 || dist0 > 0.85) { dist = dist0 = DIST_MAX; var = VAR_MAX; }
        else { dist = dist0 = -std::log(1.0 - dist0 - 0.2 * dist0 * dist0); var = dist / total; }
        if (total > 0 && ident != total) {
            if (!counts) error("computeDistance: the ML estimate needs the count matrix");
            distvar_t dv = computeMLDist(*counts, gaps, seqlen, dist, var);
            dist = dv.dist;
            var = dv.var;
        }
    } else {
        if (total == 0) { dist = dist0 = 1.0; var = VAR_MAX; }
        else { dist = dist0; var = dist0 / total; }
    }
    if (!(dist < DIST_MAX)) { dist = DIST_MAX; var = VAR_MAX; }
    if (dist > cmdlineopts.cutoff_dist) dist = cmdlineopts.cutoff_dist;
    if (var < VAR_MIN) var = VAR_MIN;
    if (!(var < VAR_MAX)) var = VAR_MAX;
    return distvar_t{dist, var};
}

static std::string g_dist_dump;
void set_dist_dump(const std::string &path) { g_dist_dump = path; }
static void dump_distances(const DistanceMatrix &d) {
    if (g_dist_dump.empty()) return;
    std::ofstream f(g_dist_dump.c_str(), std::ios::binary | std::ios::app);
    const int32_t n = d.dim;
    f.write((const char *)&n, 4);
    f.write((const char *)d.distances.data(), 8 * d.distances.size());
    f.write((const char *)d.variances.data(), 8 * d.variances.size());
}

void DistanceFactoryML::computeDistances(const int32_t *counts, const uint32_t *gaps, const std::vector<double> &seqlen,
                                         const std::vector<uint32_t> &pi, const std::vector<uint32_t> &pj, DistanceMatrix &distances) const {
    const uint32_t np = (uint32_t)pi.size(), D = (uint32_t)alphabet.DIM;
    Backend &be = default_backend();
    auto t1 = std::chrono::steady_clock::now();
    bool done = false;
    if (host_switches().device_mldist && model_factory->has_eigen() && D <= 20 && np) {
        // the whole batch in one kernel (one wavefront per pair); same arithmetic as computeDistance below except for the
        // device library's exp / log (last-bit differences: see csrc/pgm_dist_kernels.h)
        double DIST_MAX, VAR_MAX, VAR_MIN;
        consts(alphabet, DIST_MAX, VAR_MAX, VAR_MIN);
        pgm_mldist_model m;
        m.dim = D; m.Q = model_factory->Qmat().data(); m.V = model_factory->eigV().data(); m.Vi = model_factory->eigVi().data();
        m.sigma = model_factory->eigSigma().data();
        m.dist_max = DIST_MAX; m.var_max = VAR_MAX; m.var_min = VAR_MIN; m.cutoff_dist = cmdlineopts.cutoff_dist;
        m.min_dist = cmdlineopts.min_dist; m.max_dist = cmdlineopts.max_dist; m.indel_rate = cmdlineopts.indel_rate;
        m.mldist = cmdlineopts.mldist_flag ? 1 : 0; m.mldist_gap = cmdlineopts.mldist_gap_flag ? 1 : 0;
        std::vector<double> dist(np), var(np);
        done = be.mldist_batch(m, np, counts, gaps, seqlen.data(), dist.data(), var.data());
        if (done)
            for (uint32_t p = 0; p < np; ++p) {
                distances.D(pi[p], pj[p]) = distances.D(pj[p], pi[p]) = dist[p];
                distances.V(pi[p], pj[p]) = distances.V(pj[p], pi[p]) = var[p];
            }
    }
    if (!done) {
        // ML distance per pair (Newton on d, each step a 20x20 P(d)): independent per pair, so the pairs are dealt to host
        // threads; every pair's arithmetic is the single-threaded one, the matrix entries written are disjoint
        unsigned nt = std::thread::hardware_concurrency();
        if (const char *e = getenv("PGM_HOST_THREADS")) nt = (unsigned)atoi(e);
        nt = std::max(1u, std::min(nt, 16u));
        nt = (unsigned)std::min<uint32_t>(nt, std::max(1u, np));
        auto work = [&](unsigned t) {
            std::vector<int32_t> c((size_t)D * D);
            for (uint32_t p = t; p < np; p += nt) {
                std::copy(counts + (size_t)p * D * D, counts + (size_t)(p + 1) * D * D, c.begin());
                distvar_t dv = computeDistance(c, gaps[p], seqlen[p]);
                distances.D(pi[p], pj[p]) = distances.D(pj[p], pi[p]) = dv.dist;
                distances.V(pi[p], pj[p]) = distances.V(pj[p], pi[p]) = dv.var;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    be.seconds_mldist += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
}

// ---- DistanceFactoryAlign ---------------------------------------------------------------------
DistanceFactoryAlign::DistanceFactoryAlign(const Alphabet &a, const ModelFactory *mf) : DistanceFactoryML(a, mf) {
    const int sd = a.DIM + 1;  // initMatrix (DistanceFactoryAlign.cpp:5-35, 38-235)
    std::string file = data_dir() + (a.kind == ALPHA_AA ? "/nw_aa.imat" : "/nw_codon.imat");
    std::ifstream in(file.c_str());
    int r = 0, c = 0;
    in >> r >> c;
    if (!in || r != sd || c != sd) error("cannot read NW scoring matrix %s", file.c_str());
    scoring_matrix_.resize((size_t)sd * sd);
    for (int32_t &v : scoring_matrix_) in >> v;
    gap_open = -10;
    gap_extend = -2;
}

DistanceMatrix DistanceFactoryAlign::computePwDistances(const std::map<std::string, sequence_t> &sequences,
                                                        const std::vector<std::string> &order) {
    const uint32_t n = (uint32_t)order.size();
    const uint32_t D = (uint32_t)alphabet.DIM;
    DistanceMatrix distances((int)n);
    // symbols: value(), negative -> 20 for every alphabet (the reference's quirk, DistanceFactoryAlign.h:72,79)
    std::vector<int8_t> syms;
    std::vector<uint32_t> offs(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const sequence_t &s = sequences.at(order[i]);
        for (int8_t c : s) {
            int v = alphabet.value(c);
            syms.push_back((int8_t)(v < 0 ? 20 : v));
        }
        offs[i + 1] = (uint32_t)syms.size();
    }
    // The reference's i < j double loop (DistanceFactoryAlign.h:35-53) is a farm of independent alignPair jobs.  Here: the
    // pairs sorted by cost (longest first), cut into tiles, and one host thread per device context pulling tile numbers from
    // an atomic counter (no collective, no static partition: a slower device simply takes fewer tiles).  Every tile is one
    // pgm_nw_pairs_batch call on the worker's own context; the outputs of a pair land at the pair's position in the sorted
    // order, whoever computed it, so the result does not depend on the number of workers.
    std::vector<uint32_t> pi, pj;
    {
        std::vector<std::pair<uint32_t, uint32_t>> pr;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t j = i + 1; j < n; ++j) pr.push_back({i, j});
        auto cost = [&](const std::pair<uint32_t, uint32_t> &p) { return (uint64_t)(offs[p.first + 1] - offs[p.first]) * (offs[p.second + 1] - offs[p.second]); };
        std::stable_sort(pr.begin(), pr.end(), [&](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b2) { return cost(a) > cost(b2); });
        for (auto &p : pr) { pi.push_back(p.first); pj.push_back(p.second); }
    }
    const uint32_t np = (uint32_t)pi.size();
    Backend &be = default_backend();
    // Without --mldist / --mldist_gap the distance of a pair reads (ident, total) of its count matrix and nothing else
    // (DistanceFactoryML.h:143-146, 175-178): the device reduces them and 8 B per pair come back instead of 4 D^2.
    const bool reduced = !(cmdlineopts.mldist_flag || cmdlineopts.mldist_gap_flag);
    const size_t per = reduced ? 2 : (size_t)D * D;
    // result buffers in pinned memory (the D2H copies write them directly), not zero-filled: every pair's slice is written by its tile
    int32_t *counts = (int32_t *)be.host_alloc(std::max<size_t>(sizeof(int32_t) * (size_t)np * per, 16));
    uint32_t *gaps = (uint32_t *)be.host_alloc(std::max<size_t>(4 * (size_t)np, 16));
    for (uint32_t p = 0; p < np; ++p)
        be.cells_nw += (uint64_t)(offs[pi[p] + 1] - offs[pi[p]]) * (offs[pj[p] + 1] - offs[pj[p]]);
    auto t0 = std::chrono::steady_clock::now();
    {
        const int nw = std::max(1, be.workers());
        // Tile size.  A call costs ~0.3 ms beside its kernel (staging of the inputs, launch, the last D2H: bench.py all_pairs_nw
        // rank0_fixed_ms_per_call) and a worker hides that of tile k under the kernel of tile k+1 (two tiles in flight), so what
        // matters is (a) that a tile fills a device — its persistent grid holds 7168 pairs at once; fewer pairs leave CUs idle —
        // and (b) that the last tiles of the ticket queue are small against a worker's share.  Three tiles per worker, at least
        // 256 pairs; the pairs are sorted by cost, so the last tiles are also the cheapest.  PGM_NW_TILE overrides.
        uint32_t tile = std::max<uint32_t>(256u, (np + 3u * (uint32_t)nw - 1u) / (3u * (uint32_t)nw));
        if (const char *e = getenv("PGM_NW_TILE")) tile = (uint32_t)std::max(1, atoi(e));
        const uint32_t ntiles = np ? (np + tile - 1) / tile : 0;
        std::atomic<uint32_t> next_tile(0);
        auto farm = [&](int w) {
            int pending = -1;
            for (;;) {
                const uint32_t t = next_tile.fetch_add(1);
                if (t >= ntiles) break;
                const uint32_t p0 = t * tile, cnt = std::min(tile, np - p0);
                const int ticket = be.nw_pairs_submit(D, scoring_matrix_.data(), gap_open, gap_extend, n, syms.data(), offs.data(), cnt, pi.data() + p0,
                                                      pj.data() + p0, reduced ? 1u : 0u, counts + (size_t)p0 * per, gaps + p0, w);
                if (pending >= 0) be.nw_pairs_wait(pending, w);
                pending = ticket;
            }
            if (pending >= 0) be.nw_pairs_wait(pending, w);
        };
        std::vector<std::thread> devs;
        for (int w = 1; w < nw; ++w) devs.emplace_back(farm, w);
        farm(0);
        for (auto &th : devs) th.join();
        be.farm_workers = nw; be.farm_tiles = (int)ntiles;
    }
    be.seconds_nw += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> seqlen(np);
    for (uint32_t p = 0; p < np; ++p) seqlen[p] = ((double)(offs[pi[p] + 1] - offs[pi[p]]) + (double)(offs[pj[p] + 1] - offs[pj[p]])) / 2.0;
    if (reduced) {
        for (uint32_t p = 0; p < np; ++p) {
            const distvar_t dv = computeDistance((double)counts[2 * (size_t)p], (double)counts[2 * (size_t)p + 1], nullptr, gaps[p], seqlen[p]);
            distances.D(pi[p], pj[p]) = distances.D(pj[p], pi[p]) = dv.dist;
            distances.V(pi[p], pj[p]) = distances.V(pj[p], pi[p]) = dv.var;
        }
    } else {
        computeDistances(counts, gaps, seqlen, pi, pj, distances);
    }
    be.host_free(counts);
    be.host_free(gaps);
    dump_distances(distances);
    return distances;
}

// ---- BioNJ (TreeNJ.cpp:22-29, 132-281; no fixed-topology plan) --------------------------------------
static double support(double d) {
    double s = 1.0 - std::exp(-std::log(2.0) * d / cmdlineopts.edge_halflife);
    s = std::min(1.0, std::max(0.0, s));
    if (std::isnan(s)) s = 0.0;
    return s;
}

// Column sum in the association Eigen's vectorised reduction uses for `distances.colwise().sum()` (TreeNJ.cpp:157):
// SSE2 packets of two doubles starting at the first 16-byte aligned element of the column (the matrix is column-major and
// 16-byte aligned, so column j starts aligned iff j*dim is even), two packet accumulators over alternating packets, the
// accumulators added, an odd last packet added, the two lanes added, then the unaligned head and the tail element.
// It matters: when four clusters are left the criterion has the exact tie Q(0,1) = Q(2,3), and the last bit of these sums
// decides which pair is joined, i.e. where the guide tree is rooted (tests/golden: c1.nw_ml.tree, t9.nw_p.tree, t13.nw_p.tree).
// The matrix is not rebuilt after a join (the reference's reduce() copies dim^2 doubles twice per join): it stays in its
// n0 x n0 storage and `act` lists the rows / columns still in it, in the order of the reduced matrix.  The matrix need not be
// symmetric in its last bits (the k-mer angle distances are not, see angleDistances) and BioNJ reads it by rows, by columns and
// at (index1, index2) as well as (index2, index1): a transposed copy keeps the column reads of the sums and of the criterion
// on contiguous memory.
static double eigen_column_sum(const std::vector<double> &tr, size_t ld, const std::vector<int> &act, int n, int j) {
    const double *col = &tr[(size_t)act[(size_t)j] * ld];   // column act[j] of the distances: a row of their transpose
    auto at = [&](int i) { return col[(size_t)act[(size_t)i]]; };
    const int start = std::min<int>(((size_t)j * n) & 1, n);
    const int end2 = start + ((n - start) / 4) * 4, end = start + ((n - start) / 2) * 2;
    if (end == start) {
        double res = at(0);
        for (int k = 1; k < n; ++k) res += at(k);
        return res;
    }
    double a0 = at(start), a1 = at(start + 1);
    if (end - start > 2) {
        double b0 = at(start + 2), b1 = at(start + 3);
        for (int k = start + 4; k < end2; k += 4) { a0 += at(k); a1 += at(k + 1); b0 += at(k + 2); b1 += at(k + 3); }
        a0 += b0; a1 += b1;
        if (end > end2) { a0 += at(end2); a1 += at(end2 + 1); }
    }
    double res = a0 + a1;
    for (int k = 0; k < start; ++k) res += at(k);
    for (int k = end; k < n; ++k) res += at(k);
    return res;
}

// O(N^2) per join, N - 3 joins.  What the reference does per join — clamp every entry, column sums, the scan of the criterion, a
// copy of the matrix without the joined column — is here: the clamp of the entries the previous join wrote (the others were
// clamped when they were written, and nothing reads an entry between its join and the next clamp), sums and scan on the host
// threads from 512 clusters on (ranges of columns; the scan keeps the FIRST minimum in column-major order like Eigen's
// minCoeff: a range keeps its first, the ranges are combined in order with the same strict comparison), and no copy.
PhyTree *buildNJTree(std::vector<std::string> seqs_order, DistanceMatrix dist) {
    const double MIN_DIST = 1e-4, MIN_VAR = 1e-5;
    const int n0 = (int)seqs_order.size();
    std::vector<PhyTree *> subtrees;
    for (const std::string &s : seqs_order) subtrees.push_back(new PhyTree(s));
    std::vector<int> act((size_t)n0);
    for (int i = 0; i < n0; ++i) act[(size_t)i] = i;
    auto D = [&](int i, int j) -> double & { return dist.D(act[(size_t)i], act[(size_t)j]); };   // (reduced indices)
    auto V = [&](int i, int j) -> double & { return dist.V(act[(size_t)i], act[(size_t)j]); };
    int fresh = -1;   // reduced index of the row / column the previous join wrote (not clamped yet); -1: the whole matrix is new
    std::vector<double> sums;
    const size_t ld = (size_t)n0;
    std::vector<double> tr((size_t)n0 * n0);   // tr[j ld + i] = dist.D(i, j)
    auto T = [&](int i, int j) -> double & { return tr[(size_t)act[(size_t)j] * ld + (size_t)act[(size_t)i]]; };   // the same element as D(i, j)
    for (int dim = n0; dim > 3; --dim) {
        if (fresh < 0) {
            for (double &d : dist.distances) d = std::max(d, MIN_DIST);
            for (double &v : dist.variances) v = std::max(v, MIN_VAR);
            for (int i = 0; i < dim; ++i) { D(i, i) = 0; V(i, i) = 0; }
            for (int i = 0; i < n0; ++i)
                for (int j = 0; j < n0; ++j) tr[(size_t)j * ld + (size_t)i] = dist.distances[(size_t)i * ld + (size_t)j];
        } else {
            for (int i = 0; i < dim; ++i) {   // (the entries a join writes are symmetric)
                if (i == fresh) continue;
                D(fresh, i) = D(i, fresh) = T(fresh, i) = T(i, fresh) = std::max(D(i, fresh), MIN_DIST);
                V(fresh, i) = V(i, fresh) = std::max(V(i, fresh), MIN_VAR);
            }
        }
        sums.assign((size_t)dim, 0.0);  // colwise sums
        const bool threads = dim >= 512;   // (a section of half a millisecond and more; tests/test_oracle_golden.py: the 1024-taxon tree)
        const size_t nranges = threads ? 16 : 1;
        auto range = [&](size_t r, int &c0, int &c1) { c0 = (int)((size_t)dim * r / nranges); c1 = (int)((size_t)dim * (r + 1) / nranges); };
        auto sum_range = [&](size_t r) { int c0, c1; range(r, c0, c1); for (int j = c0; j < c1; ++j) sums[(size_t)j] = eigen_column_sum(tr, ld, act, dim, j); };
        if (threads) parallel_for(nranges, sum_range); else sum_range(0);
        // Q = 0.5 d - 0.5/(dim-2) (S + S^T); minCoeff scans column-major (row index fastest) and keeps the first minimum
        struct Best { double min; int row, col; };
        std::vector<Best> best(nranges, Best{INFINITY, 0, 0});
        auto scan_range = [&](size_t r) {
            int c0, c1; range(r, c0, c1);
            Best bq{INFINITY, 0, 0};
            const double f = 0.5 / (dim - 2.0);
            for (int col = c0; col < c1; ++col) {
                const double *colp = &tr[(size_t)act[(size_t)col] * ld];   // column col of the distances
                const double sc = sums[(size_t)col];
                for (int row = 0; row < dim; ++row) {
                    if (row == col) continue;
                    const double q = 0.5 * colp[(size_t)act[(size_t)row]] - f * (sc + sums[(size_t)row]);
                    if (q < bq.min) { bq.min = q; bq.row = row; bq.col = col; }
                }
            }
            best[r] = bq;
        };
        if (threads) parallel_for(nranges, scan_range); else scan_range(0);
        int index1 = 0, index2 = 0;
        double min = INFINITY;
        for (size_t r = 0; r < nranges; ++r) if (best[r].min < min) { min = best[r].min; index2 = best[r].row; index1 = best[r].col; }
        if (index2 < index1) std::swap(index1, index2);
        std::string name1 = seqs_order[index1], name2 = seqs_order[index2];
        double dist1 = (D(index1, index2) + (sums[index1] - sums[index2]) / (dim - 2.0)) / 2.0;
        dist1 = std::min(std::max(dist1, MIN_DIST), D(index1, index2));
        double dist2 = std::max(D(index2, index1) - dist1, MIN_DIST);
        double vsum = 0;
        for (int i = 0; i < dim; ++i) vsum += V(index2, i) - V(index1, i);
        double lambda = .5 + vsum / (2 * (dim - 2) * V(index1, index2));
        if (std::isnan(lambda)) lambda = .5;
        else lambda = std::min(std::max(0.0, lambda), 1.0);

        // reduce(index2) + the joined cluster in row / column index1 (TreeNJ.cpp:230-262)
        const double v12 = V(index1, index2);
        for (int i = 0; i < dim; ++i) {
            if (i == index2) continue;
            double nd = lambda * (D(index1, i) - dist1) + (1.0 - lambda) * (D(index2, i) - dist2);
            double nv = lambda * V(index1, i) + (1.0 - lambda) * V(index2, i) - lambda * (1.0 - lambda) * v12;
            if (i == index1) { nd = 0; nv = 0; }
            D(index1, i) = D(i, index1) = T(index1, i) = T(i, index1) = nd;
            V(index1, i) = V(i, index1) = nv;
        }
        act.erase(act.begin() + index2);
        fresh = index1;   // (index1 < index2: its reduced index stays)
        seqs_order.erase(seqs_order.begin() + index2);
        seqs_order[index1] = name1 + "," + name2;
        PhyTree *tree = new PhyTree(seqs_order[index1]);
        tree->addChild(subtrees[index1], dist1, support(dist1));
        tree->addChild(subtrees[index2], dist2, support(dist2));
        subtrees.erase(subtrees.begin() + index2);
        subtrees[index1] = tree;
    }
    PhyTree *tree = new PhyTree("root");
    if (seqs_order.size() == 2) {
        double d = D(0, 1) / 2;
        tree->addChild(subtrees[0], d, support(d));
        tree->addChild(subtrees[1], d, support(d));
    } else {
        double d0 = (D(0, 1) + D(0, 2) - D(1, 2)) / 2.0;
        d0 = std::min(std::max(d0, MIN_DIST), std::min(D(1, 0), D(2, 0)));
        double d1 = std::max(D(1, 0) - d0, MIN_DIST);
        double d2 = std::max(D(2, 0) - d0, MIN_DIST);
        PhyTree *tree2 = new PhyTree("root2");
        tree2->addChild(subtrees[0], d0, support(d0));
        tree2->addChild(subtrees[1], d1, support(d1));
        tree->addChild(subtrees[2], d2 / 2, support(d2));
        tree->addChild(tree2, d2 / 2, support(d2));
    }
    return tree;
}

// ---- distances induced by an alignment (DistanceFactoryPrealigned.h:34-90) -----------------------------------------
// Pair counts over the columns where both rows have a residue (only residues with value() in 0..19, for every alphabet:
// the reference's literal 20), one gap per opening of a run in which exactly one of the two rows has a residue.
DistanceMatrix DistanceFactoryPrealigned::computePwDistances(const std::map<std::string, sequence_t> &aligned,
                                                            const std::vector<std::string> &order) {
    const uint32_t n = (uint32_t)order.size(), D = (uint32_t)alphabet.DIM;
    DistanceMatrix distances((int)n);
    std::vector<const sequence_t *> rows(n);
    for (uint32_t i = 0; i < n; ++i) rows[i] = &aligned.at(order[i]);
    std::vector<uint32_t> pi, pj;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = i + 1; j < n; ++j) { pi.push_back(i); pj.push_back(j); }
    const uint32_t np = (uint32_t)pi.size();
    const size_t L = n ? rows[0]->size() : 0;
    for (uint32_t i = 0; i < n; ++i)
        if (rows[i]->size() != L) error("prealigned distances: rows of different length");
    std::vector<int32_t> counts((size_t)np * D * D, 0);
    std::vector<uint32_t> gaps(np, 0);
    Backend &be = default_backend();
    bool done = false;
    if (!host_switches().host_counts && np) {
        // the N^2 L column scan on the device (integer counts, bit-exact: on by default, unlike the ML estimates that follow):
        // value() per residue, -1 for a gap, -2 for a residue without a value
        auto t0 = std::chrono::steady_clock::now();
        std::vector<int8_t> mat((size_t)n * L);
        for (uint32_t i = 0; i < n; ++i)
            for (size_t k = 0; k < L; ++k) {
                const int8_t c = (*rows[i])[k];
                const int v = alphabet.isGap(c) ? -1 : alphabet.value(c);
                mat[(size_t)i * L + k] = (int8_t)(alphabet.isGap(c) ? -1 : (v < 0 ? -2 : v));
            }
        {
            // every pair costs the same (one scan of the columns): contiguous ranges of pairs, one per device context
            const int nw = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, be.workers()), np));
            std::vector<char> ok((size_t)nw, 0);
            auto part = [&](int w) {
                const uint32_t p0 = (uint32_t)((uint64_t)np * (uint32_t)w / (uint32_t)nw), p1 = (uint32_t)((uint64_t)np * ((uint32_t)w + 1u) / (uint32_t)nw);
                ok[(size_t)w] = (p1 == p0 || be.prealigned_counts_batch(D, n, (uint32_t)L, mat.data(), p1 - p0, pi.data() + p0, pj.data() + p0,
                                                                       counts.data() + (size_t)p0 * D * D, gaps.data() + p0, w)) ? 1 : 0;
            };
            std::vector<std::thread> th;
            for (int w = 1; w < nw; ++w) th.emplace_back(part, w);
            part(0);
            for (auto &t : th) t.join();
            done = true;
            for (char c : ok) done = done && c;
        }
        be.seconds_mldist += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!done) {
        auto t1 = std::chrono::steady_clock::now();
        unsigned nt = std::thread::hardware_concurrency();
        if (const char *e = getenv("PGM_HOST_THREADS")) nt = (unsigned)atoi(e);
        nt = std::max(1u, std::min(nt, 16u));
        nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, np));
        auto work = [&](unsigned t) {
            for (size_t p = t; p < np; p += nt) {
                const sequence_t &s1 = *rows[pi[p]], &s2 = *rows[pj[p]];
                int32_t *c = counts.data() + p * D * D;
                index_t g = 0;
                bool open1 = false, open2 = false;
                for (size_t k = 0; k < s1.size(); ++k) {
                    const bool g1 = alphabet.isGap(s1[k]), g2 = alphabet.isGap(s2[k]);
                    if (!g1 && !g2) {
                        const int c1 = alphabet.value(s1[k]), c2 = alphabet.value(s2[k]);
                        if (c1 >= 0 && c1 < 20 && c2 >= 0 && c2 < 20) ++c[(size_t)c1 + (size_t)D * c2];
                        open1 = false; open2 = false;
                    } else if (g1 && g2) {
                        // skip
                    } else if (!g1 && !open1) {
                        ++g; open1 = true; open2 = false;
                    } else if (!g2 && !open2) {
                        ++g; open1 = false; open2 = true;
                    }
                }
                gaps[p] = g;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
        be.seconds_mldist += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    }
    std::vector<double> seqlen(np, ((double)L + (double)L) / 2.0);
    const auto tq0 = std::chrono::steady_clock::now();
    computeDistances(counts.data(), gaps.data(), seqlen, pi, pj, distances);
    if (host_switches().profile)
        fprintf(stderr, "  prealigned distances: pair counts %s, estimates %.1f ms\n", done ? "on the device" : "on the host",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count());
    dump_distances(distances);
    return distances;
}

DistanceMatrix angleDistances(const Alphabet &a, const std::map<std::string, sequence_t> &sequences, const std::vector<std::string> &order) {
    const uint32_t n = (uint32_t)order.size(), D = (uint32_t)a.DIM, ncols = D * D;   // K = 2 for both alphabets
    DistanceMatrix distances((int)n);
    std::vector<int32_t> counts((size_t)n * ncols, 0);
    std::vector<double> seq_len(n);
    for (uint32_t i = 0; i < n; ++i) {   // DistanceFactoryAngle.h:63-94
        const sequence_t &seq = sequences.at(order[i]);
        seq_len[i] = (double)seq.size();
        int prev = -1;
        for (size_t j = 0; j < seq.size(); ++j) {
            int v = a.value(seq[j]);
            if (v < 0 || v >= (int)D) v = -1;
            if (prev != -1 && v != -1) counts[(size_t)i * ncols + (size_t)prev * D + (size_t)v] += 1;
            prev = v;
        }
    }
    Backend &be = default_backend();
    const auto t0 = std::chrono::steady_clock::now();
    be.kmer_cosine(n, ncols, counts.data(), distances.distances.data());   // :100
    // kmer_cosine writes element (i, j) at i + n j (the reference's column-major matrix); DistanceMatrix::D(i, j) reads i n + j.  The
    // matrix is NOT symmetric in its last bits — ((c_i / |c_i|) . c_j) / |c_j| is rounded differently from the (j, i) element — and
    // BioNJ reads it by rows, by columns and at (index1, index2) / (index2, index1): the orientation has to be the reference's.
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = i + 1; j < n; ++j) std::swap(distances.distances[(size_t)i * n + j], distances.distances[(size_t)j * n + i]);
    be.seconds_mldist += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const bool ml = cmdlineopts.mldist_flag || cmdlineopts.mldist_gap_flag;
    // log and exp of the reference binary are those of the glibc it is linked with statically (2.17: IBM's correctly rounded ones);
    // today's libm is within 0.52 ulp, which moved the exact final tie of 3 trees in 60.  Correctly rounded here: the long double
    // function when its result is not within 2^-9 ulp of a rounding boundary, binary128 (libquadmath) otherwise.
    auto correctly_rounded = [](long double y, __float128 (*exact)(__float128), double x) {
        const double lo = (double)(y - fabsl(y) * 0x1p-62L), hi = (double)(y + fabsl(y) * 0x1p-62L);
        return lo == hi ? lo : (double)exact((__float128)x);
    };
    auto cr_log = [&](double x) { return correctly_rounded(logl((long double)x), logq, x); };
    auto cr_exp = [&](double x) { return correctly_rounded(expl((long double)x), expq, x); };
    parallel_for((size_t)n, [&](size_t r) {   // :101-105, row by row on the host threads
        for (uint32_t c = 0; c < n; ++c) {
            double &d = distances.distances[r * n + c];
            d = -1.0 * cr_log((d * d + 0.4) / 1.4);
            if (!ml) {
                const double e = cr_exp(d);
                d = -0.5 * (5.0 * e - std::sqrt(45.0 * (e * e) - 20.0 * e)) * (1.0 / e);
            }
        }
    });
    for (uint32_t j = 0; j < n; ++j)   // :107-113: variances = distances / ((len_i + len_j) / 2), at least 1e-5
        for (uint32_t i = 0; i < n; ++i) {
            double v = 1.0 / ((seq_len[j] + seq_len[i]) / 2);
            v *= distances.D((int)i, (int)j);
            distances.V((int)i, (int)j) = std::max(v, 1e-5);
        }
    dump_distances(distances);
    return distances;
}

PhyTree *TreeNJ(const Alphabet &a, const std::map<std::string, sequence_t> &seqs, const ModelFactory *mf, bool prealigned) {
    if (seqs.size() < 2) error("cannot construct tree from < 2 sequences");
    if (prealigned) {
        std::vector<std::string> order;
        for (const auto &kv : seqs) order.push_back(kv.first);
        DistanceFactoryPrealigned df(a, mf);
        const auto tq0 = std::chrono::steady_clock::now();
        DistanceMatrix dist = df.computePwDistances(seqs, order);
        for (int i = 0; i < dist.dim; ++i) { dist.D(i, i) = 0; dist.V(i, i) = 0; }
        const auto tq1 = std::chrono::steady_clock::now();
        PhyTree *t = midpointRoot(buildNJTree(order, dist));
        if (host_switches().profile)
            fprintf(stderr, "  TreeNJ: distances %.1f ms, BioNJ + rooting %.1f ms\n", std::chrono::duration<double, std::milli>(tq1 - tq0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq1).count());
        return t;
    }
    std::vector<std::string> order;
    for (const auto &kv : seqs) order.push_back(kv.first);  // std::map key order (TreeNJ.h:34-39)
    if (!cmdlineopts.nwdist_flag) {   // DistanceFactory::getDefault (DistanceFactory.cpp:9-20): the k-mer angle distances
        DistanceMatrix dist = angleDistances(a, seqs, order);
        for (int i = 0; i < dist.dim; ++i) { dist.D(i, i) = 0; dist.V(i, i) = 0; }
        return midpointRoot(buildNJTree(order, dist));
    }
    DistanceFactoryAlign df(a, mf);
    DistanceMatrix dist = df.computePwDistances(seqs, order);
    for (int i = 0; i < dist.dim; ++i) { dist.D(i, i) = 0; dist.V(i, i) = 0; }
    PhyTree *tree = buildNJTree(order, dist);
    return midpointRoot(tree);
}

}  // namespace pgm
