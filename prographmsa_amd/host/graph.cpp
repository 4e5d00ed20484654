// graph.cpp — Graph / SequenceGraph / CleanedGraph (reference src/Graph.h, SequenceGraph.h,
// CleanedGraph.h).  Eigen's row-major sparse matrices become explicit CSR arrays, which is also
// the layout the C ABI hands to the device (include/pgm_hip.h: pgm_graph).
#include "pgm_host.h"

#include <algorithm>
#include <cmath>

namespace pgm {

static const dp_score_t MAX_EDGE_COST = 10000.0f;  // Graph.h:18

Graph::Graph(int dim) : dim_(dim), n_(2) {  // Graph.h:141-150
    sites_.assign((size_t)dim * 2, 0.0);
    e_rowptr_.assign(3, 0);
    r_rowptr_.assign(3, 0);
    fillInitialEdges();
}

Graph::Graph(int dim, const std::vector<std::vector<double>> &nodes) : dim_(dim), n_((index_t)nodes.size()) {
    sites_.assign((size_t)dim * n_, 0.0);
    for (index_t i = 1; i + 1 < n_; ++i) std::copy(nodes[i].begin(), nodes[i].end(), sites_.begin() + (size_t)dim * i);
    e_rowptr_.assign(n_ + 1, 0);
    r_rowptr_.assign(n_ + 1, 0);
    fillInitialEdges();
}

Graph::Graph(int dim, const std::vector<std::vector<double>> &nodes, const EdgeMap &edges, const RepeatMap &repeats)
    : dim_(dim), n_((index_t)nodes.size()) {
    sites_.assign((size_t)dim * n_, 0.0);
    for (index_t i = 1; i + 1 < n_; ++i) std::copy(nodes[i].begin(), nodes[i].end(), sites_.begin() + (size_t)dim * i);
    setEdgesFromMap(edges);
    setRepeatsFromMap(repeats);
}

Graph::Graph(int dim, index_t n, const double *profiles, const std::vector<EdgeRec> &edges, const std::vector<RepeatRec> &repeats)
    : dim_(dim), n_(n) {
    if (profiles) {   // (NULL: the profiles stay on the device, see devSites())
        sites_.assign((size_t)dim * n_, 0.0);
        if (n_ > 2) std::copy(profiles + dim, profiles + (size_t)dim * (n_ - 1), sites_.begin() + dim);
    }
    e_rowptr_.assign(n_ + 1, 0);
    e_col_.resize(edges.size());
    e_val_.resize(edges.size());
    for (size_t k = 0; k < edges.size(); ++k) {
        e_rowptr_[edges[k].to + 1]++;
        e_col_[k] = edges[k].from;
        e_val_[k] = std::min(edges[k].cost, MAX_EDGE_COST) - MAX_EDGE_COST;  // Graph.h:85
    }
    r_rowptr_.assign(n_ + 1, 0);
    r_col_.resize(repeats.size());
    r_units_.resize(repeats.size());
    for (size_t k = 0; k < repeats.size(); ++k) {
        r_rowptr_[repeats[k].to + 1]++;
        r_col_[k] = repeats[k].from;
        r_units_[k] = repeats[k].units;
    }
    for (index_t i = 0; i < n_; ++i) { e_rowptr_[i + 1] += e_rowptr_[i]; r_rowptr_[i + 1] += r_rowptr_[i]; }
}

Graph::Graph(int dim, index_t n, NoSites) : dim_(dim), n_(n) {
    e_rowptr_.assign(n_ + 1, 0);
    r_rowptr_.assign(n_ + 1, 0);
    fillInitialEdges();
}

void Graph::fillInitialEdges() {  // chain edges (i+1 <- i) with stored value -MAX_EDGE_COST (cost 0)
    e_rowptr_.assign(n_ + 1, 0);
    e_col_.clear();
    e_val_.clear();
    for (index_t i = 0; i < n_; ++i) {
        if (i > 0) {
            e_col_.push_back(i - 1);
            e_val_.push_back(-MAX_EDGE_COST);
        }
        e_rowptr_[i + 1] = (int32_t)e_col_.size();
    }
}

void Graph::setEdgesFromMap(const EdgeMap &edge_map) {  // keys are (to,from), ordered -> CSR directly
    e_rowptr_.assign(n_ + 1, 0);
    e_col_.clear();
    e_val_.clear();
    e_col_.reserve(edge_map.size());
    e_val_.reserve(edge_map.size());
    for (const auto &kv : edge_map) {
        dp_score_t cost = std::min(kv.second, MAX_EDGE_COST) - MAX_EDGE_COST;  // Graph.h:85
        e_rowptr_[kv.first.first + 1]++;
        e_col_.push_back(kv.first.second);
        e_val_.push_back(cost);
    }
    for (index_t i = 0; i < n_; ++i) e_rowptr_[i + 1] += e_rowptr_[i];
}

void Graph::setRepeatsFromMap(const RepeatMap &rep_map) {
    r_rowptr_.assign(n_ + 1, 0);
    r_col_.clear();
    r_units_.clear();
    for (const auto &kv : rep_map) {
        r_rowptr_[kv.first.first + 1]++;
        r_col_.push_back(kv.first.second);
        r_units_.push_back(kv.second);
    }
    for (index_t i = 0; i < n_; ++i) r_rowptr_[i + 1] += r_rowptr_[i];
}

void Graph::reset() {   // Graph.h:369-373
    if (sites_.empty()) error("Graph::reset needs the profiles on the host");
    std::fill(sites_.begin(), sites_.end(), 1.0);
    std::fill(sites_.begin(), sites_.begin() + dim_, 0.0);
    std::fill(sites_.end() - dim_, sites_.end(), 0.0);
}

void Graph::rmNodes(index_t first, index_t count) {   // Graph.h:375-426
    if (first < 1 || first + count > n_) error("Graph::rmNodes: range outside the graph");
    const index_t nn = n_ - count;
    auto gone = [&](index_t v) { return v >= first && v < first + count; };
    auto moved = [&](index_t v) { return v >= first ? v - count : v; };
    std::vector<int32_t> erp(nn + 1, 0), rrp(nn + 1, 0);
    std::vector<uint32_t> ec, rc, ru;
    std::vector<float> ev;
    for (index_t to = 0; to < n_; ++to) {
        if (gone(to)) continue;
        for (int32_t e = e_rowptr_[to]; e < e_rowptr_[to + 1]; ++e)
            if (!gone(e_col_[e])) { ec.push_back(moved(e_col_[e])); ev.push_back(e_val_[e]); }
        for (int32_t e = r_rowptr_[to]; e < r_rowptr_[to + 1]; ++e)
            if (!gone(r_col_[e])) { rc.push_back(moved(r_col_[e])); ru.push_back(r_units_[e]); }
        erp[moved(to) + 1] = (int32_t)ec.size();
        rrp[moved(to) + 1] = (int32_t)rc.size();
    }
    e_rowptr_.swap(erp); e_col_.swap(ec); e_val_.swap(ev);
    r_rowptr_.swap(rrp); r_col_.swap(rc); r_units_.swap(ru);
    if (!sites_.empty()) sites_.erase(sites_.begin() + (size_t)dim_ * first, sites_.begin() + (size_t)dim_ * (first + count));
    n_ = nn;
}

void Graph::addRepeats(const std::vector<std::vector<int>> &tr_homologies) {
    RepeatMap tredges;
    for (const std::vector<int> &h : tr_homologies) {   // getRepeatEdges (Graph.h:48-79), offset 0
        for (index_t from = 0; from != h.size(); ++from) {
            if (h[from] < 0) continue;
            index_t n_units = 0;
            bool take_next = false;
            for (index_t to = from + 1; to != h.size(); ++to) {
                if (h[to] < 0) continue;
                if (h[to] <= h[to - 1]) n_units += 1;
                if (take_next) {
                    const std::pair<index_t, index_t> key(to, from);
                    auto it = tredges.find(key);
                    if (it != tredges.end()) it->second = std::min(it->second, n_units);
                    else tredges[key] = n_units;
                    take_next = false;
                }
                if (h[to] == h[from]) take_next = true;
            }
        }
    }
    setRepeatsFromMap(tredges);
}

pgm_graph Graph::flat() const {
    pgm_graph g;
    g.n = n_;
    g.dim = (uint32_t)dim_;
    g.sites = sites_.empty() ? nullptr : sites_.data();
    g.e_rowptr = e_rowptr_.data();
    g.e_col = e_col_.data();
    g.e_val = e_val_.data();
    g.r_rowptr = r_col_.empty() ? nullptr : r_rowptr_.data();
    g.r_col = r_col_.data();
    g.r_units = r_units_.data();
    return g;
}

Graph::PredIterator::PredIterator(const Graph &g, index_t row, dp_score_t repeat_init, dp_score_t repeat_ext)
    : g_(&g), i_(g.e_rowptr_[row]), iend_(g.e_rowptr_[row + 1]), j_(g.r_rowptr_[row]), jend_(g.r_rowptr_[row + 1]),
      repeatInit_(repeat_init), repeatExt_(repeat_ext) {}

dp_score_t Graph::PredIterator::value() const {  // Graph.h:223-239
    if (i_ < iend_) {
        dp_score_t c = g_->e_val_[i_];
        if (c == 0) return (dp_score_t)INFINITY;
        return c + MAX_EDGE_COST;
    }
    index_t c = g_->r_units_[j_];
    if (c == 0) return (dp_score_t)INFINITY;
    return repeatInit_ + repeatExt_ * (c - 1);
}

// SequenceGraph(seq) (SequenceGraph.h:44-54, 101-104): one-hot columns, uniform for invalid symbols
Graph SequenceGraph(const Alphabet &a, const sequence_t &seq) {
    const int D = a.DIM;
    std::vector<std::vector<double>> nodes(seq.size() + 2, std::vector<double>(D, 0.0));
    for (size_t i = 0; i < seq.size(); ++i) {
        std::vector<double> &v = nodes[i + 1];
        if (a.isValid(seq[i])) v[a.value(seq[i])] = 1.0;
        else std::fill(v.begin(), v.end(), 1.0 / D);
    }
    return Graph(D, nodes);
}

// SequenceGraph(seq, csprofile, model) (SequenceGraph.h:111-121): the profile matrix comes from
// createProfile, including whatever it left in columns 0 and n-1 (zeros).
Graph SequenceGraphFromProfile(int dim, index_t nnodes, const std::vector<double> &sites) {
    std::vector<std::vector<double>> nodes(nnodes, std::vector<double>(dim, 0.0));
    for (index_t i = 0; i < nnodes; ++i) std::copy(sites.begin() + (size_t)dim * i, sites.begin() + (size_t)dim * (i + 1), nodes[i].begin());
    Graph g(dim, nodes);
    return g;
}

// ---------------------------------------------------------------------------------------
CleanedGraph::CleanedGraph(const Graph &original) : Graph(original) {  // CleanedGraph.h:39-146
    const index_t n = original.size();
    original_size_ = n;
    std::vector<bool> marked_fw(n, false), marked_bw(n, false);
    std::vector<index_t> mapping(n, (index_t)-1);
    dp_score_t repeatExt = cmdlineopts.repeatext_prob == 0 ? (dp_score_t)INFINITY : 0;

    marked_fw[0] = true;
    marked_bw[n - 1] = true;
    mapping[0] = 0;
    index_t newDim = 1;

    for (index_t i = n; i > 0; --i) {  // reachable from end
        index_t to = i - 1;
        if (!marked_bw[to]) continue;
        for (PredIterator from = original.getPreds(to, 0, repeatExt); from; ++from)
            if (from.value() != (dp_score_t)INFINITY) marked_bw[*from] = true;
    }
    for (index_t to = 1; to < n; ++to) {  // reachable from start
        for (PredIterator from = original.getPreds(to, 0, repeatExt); from; ++from) {
            if (from.value() != (dp_score_t)INFINITY && marked_fw[*from]) {
                marked_fw[to] = true;
                if (marked_bw[to]) mapping[to] = newDim++;
                break;
            }
        }
    }

    std::vector<int32_t> rp(newDim + 1, 0);
    std::vector<uint32_t> col;
    std::vector<float> val;
    // rows are visited in increasing `to`, mapping is monotone, so the triplets arrive CSR-ordered
    for (index_t to = 0; to < n; ++to) {
        for (int32_t k = original.e_rowptr_[to]; k < original.e_rowptr_[to + 1]; ++k) {
            index_t y = mapping[to], x = mapping[original.e_col_[k]];
            if (x != (index_t)-1 && y != (index_t)-1 && original.e_val_[k] < 0) {
                rp[y + 1]++;
                col.push_back(x);
                val.push_back(original.e_val_[k]);
            }
        }
    }
    for (index_t i = 0; i < newDim; ++i) rp[i + 1] += rp[i];
    std::vector<int32_t> rrp(newDim + 1, 0);
    std::vector<uint32_t> rcol, runits;
    for (index_t to = 0; to < n; ++to) {
        for (int32_t k = original.r_rowptr_[to]; k < original.r_rowptr_[to + 1]; ++k) {
            index_t y = mapping[to], x = mapping[original.r_col_[k]];
            if (x != (index_t)-1 && y != (index_t)-1 && original.r_units_[k] > 0) {
                rrp[y + 1]++;
                rcol.push_back(x);
                runits.push_back(original.r_units_[k]);
            }
        }
    }
    for (index_t i = 0; i < newDim; ++i) rrp[i + 1] += rrp[i];

    const bool host_sites = original.hasHostSites();   // (else: on the device, gathered there through outmapping_)
    std::vector<double> newSites(host_sites ? (size_t)dim_ * newDim : 0, 0.0);
    outmapping_.assign(newDim, 0);
    for (index_t i = 0; i < n; ++i) {
        if (mapping[i] != (index_t)-1) {
            if (host_sites) std::copy(original.col(i), original.col(i) + dim_, newSites.begin() + (size_t)dim_ * mapping[i]);
            outmapping_[mapping[i]] = i;
        }
    }
    n_ = newDim;
    sites_.swap(newSites);
    e_rowptr_.swap(rp);
    e_col_.swap(col);
    e_val_.swap(val);
    r_rowptr_.swap(rrp);
    r_col_.swap(rcol);
    r_units_.swap(runits);
}

void CleanedGraph::uncleanMapping(std::vector<index_t> &mapping) const {
    for (index_t &m : mapping)
        if (m != (index_t)-1) m = getMapping(m);
}

}  // namespace pgm
