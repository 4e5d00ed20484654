// progressive.cpp — progressive alignment driver (reference src/ProgressiveAlignment.{h,cpp}).
//
// The reference recurses strictly sequentially (ProgressiveAlignment.cpp:50-51).  The results of
// sibling subtrees are independent, so this driver walks the tree level by level ("height" = longest
// path to a leaf) and hands every ready internal node of a level to the backend in ONE batched
// alignGraphs call; the per-node host work (CleanedGraph, mergeGraphs, extend_alignment) runs on a
// few host threads.  The values produced per node are those of the reference's recursion.
#include "pgm_host.h"
#include "../csrc/pgm_pool.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <thread>

namespace pgm {

// extend_alignment (ProgressiveAlignment.h:245-264)
void parallel_for(size_t n, const std::function<void(size_t)> &fn);
// `spread`: the rows are filled on the host threads (the few nodes near the root carry hundreds of rows each)
static void extend_alignment(const Alphabet &a, ProgressiveAlignmentResult &result, const std::vector<index_t> &mapping,
                             const std::map<std::string, sequence_t> &aligned_sequences, bool spread) {
    std::vector<std::pair<const sequence_t *, sequence_t *>> rows;
    rows.reserve(aligned_sequences.size());
    for (const auto &kv : aligned_sequences) rows.emplace_back(&kv.second, &result.aligned_sequences[kv.first]);
    const index_t n = result.graph.size();
    auto fill = [&](size_t r) {
        sequence_t extended(n - 2, a.unknown());
        const sequence_t &original = *rows[r].first;
        index_t k = 0;
        for (index_t j = 1; j < n - 1; ++j) {
            if (mapping[j] != (index_t)-1) extended[j - 1] = original[k++];
            else extended[j - 1] = a.gap();
        }
        *rows[r].second = std::move(extended);
    };
    if (spread && rows.size() >= 64) parallel_for(rows.size(), fill);
    else for (size_t r = 0; r < rows.size(); ++r) fill(r);
}

// ancestral sequences and profiles (--ancestral_seqs; ProgressiveAlignment.h:289-411)
static std::string create_ancestral_seq_name(const std::map<std::string, sequence_t> &aligned_seqs) {
    std::vector<std::string> leaves;
    for (const auto &kv : aligned_seqs) if (kv.first[0] != '(') leaves.push_back(kv.first);
    std::sort(leaves.begin(), leaves.end());
    std::string s = "(";
    for (size_t i = 0; i < leaves.size(); ++i) { if (i) s += ","; s += leaves[i]; }
    return s + ")";
}
static int8_t symbol_of(const Alphabet &a, int j) {   // ALPHABET(j): the symbol whose value() is j
    static const char *aa = "ACDEFGHIKLMNPQRSTVWY";
    if (a.kind == ALPHA_AA) return sequenceFromString(a, std::string(1, aa[j]))[0];
    static const char nt[] = "TCAG";   // the 61 sense codons in TCAG order (Alphabet.cpp)
    int k = -1;
    for (int c = 0; c < 64; ++c) {
        const std::string cod = {nt[c >> 4], nt[(c >> 2) & 3], nt[c & 3]};
        if (cod == "TAA" || cod == "TAG" || cod == "TGA") continue;
        if (++k == j) return sequenceFromString(a, cod)[0];
    }
    return a.unknown();
}
// the sequence and the profile of `src`'s nodes as seen from the (new) graph of `result`: node i of the result shows column
// (col .* pi) of source node mapping[i] if it is a matched node, a gap otherwise.  prelim: src = result.graph itself.
static void ancestral_seq(const Alphabet &a, ProgressiveAlignmentResult &result, const std::string &anc_name, const Graph &src, const std::vector<index_t> *mapping,
                          const std::vector<bool> &matched, const Model &model) {
    const index_t n = result.graph.size();
    const int D = a.DIM;
    sequence_t extended(n - 2, a.unknown());
    Profile prof;
    prof.dim = D;
    std::vector<double> col((size_t)D);
    for (index_t i = 1; i < n - 1; ++i) {
        const bool on = matched[i] && (!mapping || (*mapping)[i] != (index_t)-1);
        if (!on) { extended[i - 1] = a.gap(); continue; }
        const double *g = src.col(mapping ? (*mapping)[i] : i);
        int best = 0;
        double sum = 0;
        for (int k = 0; k < D; ++k) { col[(size_t)k] = g[k] * model.pi[(size_t)k]; if (col[(size_t)k] > col[(size_t)best]) best = k; }   // maxCoeff: the first maximum
        extended[i - 1] = symbol_of(a, best);
        for (int k = 0; k < D; ++k) sum += col[(size_t)k];
        for (int k = 0; k < D; ++k) prof.data.push_back(col[(size_t)k] / sum);
        ++prof.cols;
    }
    result.aligned_sequences[anc_name] = extended;
    result.profiles[anc_name] = std::move(prof);
}

void parallel_for(size_t n, const std::function<void(size_t)> &fn) {
    static unsigned nt = []() {
        unsigned v = std::thread::hardware_concurrency();
        if (const char *e = getenv("PGM_HOST_THREADS")) v = (unsigned)atoi(e);
        return std::max(1u, std::min(v, 16u));
    }();
    static pgm_pool::Pool pool(nt);   // persistent: a level has half a dozen sections of a millisecond or two
    const std::string err = pool.run(n, nt, fn);
    if (!err.empty()) throw pgm_exception(err);
}

namespace {
struct Node {
    std::string tr_note;   // -R: "TR indels at (...): n" of this internal node
    const PhyTree *tree;
    int child[2] = {-1, -1};
    int height = 0;
    ProgressiveAlignmentResult res;
};
struct Pending {  // state of one align_progressive_results call between its two halves
    Model model, model1, model2;
    std::unique_ptr<CleanedGraph> cg1, cg2;
};
}  // namespace

static int collect(const PhyTree &t, std::vector<Node> &nodes) {
    if (!t.isLeaf() && t.n_children() != 2) error("only bifurcating trees allowed");
    int c0 = -1, c1 = -1;
    if (!t.isLeaf()) {
        c0 = collect(t[0], nodes);
        c1 = collect(t[1], nodes);
    }
    Node n;
    n.tree = &t;
    n.child[0] = c0;
    n.child[1] = c1;
    n.height = t.isLeaf() ? 0 : 1 + std::max(nodes[c0].height, nodes[c1].height);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
}

// extend_tr_homologies (ProgressiveAlignment.h:266-287)
static void extend_tr_homologies(ProgressiveAlignmentResult &result, const std::vector<index_t> &mapping, const std::vector<std::vector<int>> &tr_homologies,
                                 const std::vector<std::string> &tr_source) {
    const index_t n = result.graph.size();
    for (size_t r = 0; r < tr_homologies.size(); ++r) {
        std::vector<int> extended(n - 2, -1);
        const std::vector<int> &original = tr_homologies[r];
        index_t k = 0;
        for (index_t j = 1; j < n - 1; ++j) extended[j - 1] = mapping[j] != (index_t)-1 ? original[k++] : -1;
        result.tr_homologies.push_back(extended);
        result.tr_source.push_back(tr_source[r]);
    }
}

// --read_repeats: the T-REKS report (RepeatDetectionTReks.cpp:62-151).  Per sequence ('>' line) any number of repeats: a header
// line "Length: ... from S to E ..." (S 1-based), then the aligned repeat units, one per line, up to a line of asterisks; '-',
// blanks and tabs are gaps; the residues of the units must spell the sequence from position S on.
static std::string strip_ws(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && isspace((unsigned char)s[b])) ++b;
    while (e > b && isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}
std::map<std::string, std::vector<repeat_t>> read_repeats(const Alphabet &a, const std::string &filename, const std::map<std::string, sequence_t> &seqs) {
    std::map<std::string, std::string> seqs2;
    for (const auto &kv : seqs) seqs2[kv.first] = stringFromSequence(a, kv.second);
    std::ifstream in(filename.c_str());
    std::map<std::string, std::vector<repeat_t>> map;
    index_t n_sequences = 0, n_repeats = 0;
    std::string name, line;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '>') { name = strip_ws(line.substr(1)); ++n_sequences; }
        else if (line.compare(0, 7, "Length:") == 0) {
            const size_t from = line.find("from");
            if (from == line.npos) throw pgm_exception("format error (from)");
            const size_t to = line.find("to", from);
            if (to == line.npos) throw pgm_exception("format error (to)");
            repeat_t repeat;
            ++n_repeats;
            long start = 0;
            { std::stringstream ss; ss << line.substr(from + 4, to - from - 4); ss >> start; if (!ss || start <= 0) throw pgm_exception("format error (number)"); }
            repeat.start = (index_t)(start - 1);
            auto orig_entry = seqs2.find(name);
            if (orig_entry == seqs2.end()) throw pgm_exception("unknown sequence name: " + name);
            size_t orig = repeat.start;
            repeat.len = (index_t)-1;
            while (std::getline(in, line)) {
                line = strip_ws(line);
                if (line.compare(0, 22, "**********************") == 0) break;
                for (char &c : line) if (c == '-' || c == ' ' || c == '\n' || c == '\t' || c == '\r') c = '_';
                if (repeat.len != (index_t)-1 && line.size() != repeat.len) throw pgm_exception("repeat unit lengths differ");
                repeat.len = (index_t)line.size();
                for (index_t i = 0; i < repeat.len; ++i)
                    if (line[i] != '_') {
                        repeat.tr_hom.push_back((int)i);
                        if (orig >= orig_entry->second.size() || orig_entry->second[orig] != line[i]) throw pgm_exception("character mismatch in repeat of \"" + name + "\"");
                        ++orig;
                    }
            }
            map[name].push_back(repeat);
        }
    }
    std::cerr << "found " << n_repeats << " repeats in " << n_sequences << " sequences" << std::endl;
    return map;
}

// A resident pass on several device contexts: the guide tree is cut into subtrees — the most expensive subtree of the cut is replaced
// by its two children until there are three per worker (or only leaves are left) —, the subtrees are dealt to the workers by cost
// (longest first to the least loaded: farm_shards), and every node of a subtree belongs to its worker: leaves are built there, jobs
// run there, merged profiles stay there.  A node above the cut belongs to the worker of its more expensive child; the other child's
// profiles are copied over once (Backend::resident_import).  cost(node) = sum over the internal nodes below of the product of the
// children's mean leaf lengths: the DP cells, near enough.  `nodes` is in post-order (children before parents).
static std::vector<int> assign_owners(const std::vector<Node> &nodes, int root, int nworkers, const std::vector<double> &leaf_len) {
    const size_t n = nodes.size();
    std::vector<double> cost(n, 0.0), len_sum(n, 0.0), nleaf(n, 0.0);
    for (size_t i = 0; i < n; ++i) {
        if (nodes[i].child[0] < 0) { len_sum[i] = leaf_len[i]; nleaf[i] = 1.0; continue; }
        const int c0 = nodes[i].child[0], c1 = nodes[i].child[1];
        len_sum[i] = len_sum[c0] + len_sum[c1]; nleaf[i] = nleaf[c0] + nleaf[c1];
        cost[i] = cost[c0] + cost[c1] + (len_sum[c0] / nleaf[c0]) * (len_sum[c1] / nleaf[c1]);
    }
    std::vector<int> owner(n, 0);
    if (nworkers <= 1) return owner;
    std::vector<int> cut{root};
    std::vector<char> above(n, 0);
    while (cut.size() < 3 * (size_t)nworkers) {
        int pick = -1;
        for (size_t k = 0; k < cut.size(); ++k)
            if (nodes[cut[k]].child[0] >= 0 && (pick < 0 || cost[cut[k]] > cost[cut[(size_t)pick]])) pick = (int)k;
        if (pick < 0) break;
        const int v = cut[(size_t)pick];
        above[v] = 1;
        cut[(size_t)pick] = nodes[v].child[0];
        cut.push_back(nodes[v].child[1]);
    }
    std::vector<uint64_t> c64(cut.size());
    for (size_t k = 0; k < cut.size(); ++k) c64[k] = (uint64_t)cost[cut[k]] + 1u;
    const std::vector<std::vector<uint32_t>> shards = farm_shards(c64, nworkers);
    for (size_t w = 0; w < shards.size(); ++w)
        for (uint32_t k : shards[w]) {
            std::vector<int> stack{cut[k]};
            while (!stack.empty()) {
                const int v = stack.back(); stack.pop_back();
                owner[v] = (int)w;
                if (nodes[v].child[0] >= 0) { stack.push_back(nodes[v].child[0]); stack.push_back(nodes[v].child[1]); }
            }
        }
    for (size_t i = 0; i < n; ++i)   // post-order: the children of a node above the cut have their owners
        if (above[i]) owner[i] = cost[nodes[i].child[0]] + nleaf[nodes[i].child[0]] >= cost[nodes[i].child[1]] + nleaf[nodes[i].child[1]] ? owner[nodes[i].child[0]] : owner[nodes[i].child[1]];
    return owner;
}

// earlyRefinement (ProgressiveAlignment.h:114-243) of the nodes of one level, right after their alignment: the node's graph is
// aligned once more with the graph of every grandchild (of a leaf child: with the leaf) — the second call site of alignGraphs
// (:170), on the graphs as they are, not cleaned — and rebuilt from those alignments one descendant at a time
// (mergeGraphsIncremental); nodes no descendant maps to are dropped.  The 2-4 alignments of a node read only the node's old graph,
// so the whole level goes to the backend as ONE batch; the merges are per node, in the reference's order.
static void early_refinement(const Alphabet &a, std::vector<Node> &nodes, const std::vector<int> &level, const ModelFactory &model_factory, bool with_repeats) {
    const index_t NONE = (index_t)-1;
    struct Desc { int node; double distance, gap_distance; };
    std::vector<std::vector<Desc>> desc(level.size());
    std::vector<Model> models;
    std::vector<std::pair<size_t, size_t>> job_of;   // (position in the level, descendant)
    auto is_leaf = [&](int v) { return nodes[(size_t)v].child[0] < 0; };
    for (size_t k = 0; k < level.size(); ++k) {
        const Node &nd = nodes[(size_t)level[k]];
        if (is_leaf(nd.child[0]) && is_leaf(nd.child[1])) continue;   // (:124-125: nothing to refine)
        for (int i = 0; i < 2; ++i) {
            const int c = nd.child[i];
            const PhyTree &ct = *nodes[(size_t)c].tree;
            if (is_leaf(c)) {
                desc[k].push_back(Desc{c, nodes[(size_t)c].res.is_csprofile ? 0.0 : ct.getBranchLength(), ct.getBranchLength()});
            } else {
                for (int j = 0; j < 2; ++j) {
                    const int gc = nodes[(size_t)c].child[j];
                    const double bl = nodes[(size_t)gc].tree->getBranchLength();
                    desc[k].push_back(Desc{gc, ct.getBranchLength() + (nodes[(size_t)gc].res.is_csprofile ? 0.0 : bl), ct.getBranchLength() + bl});
                }
            }
        }
        for (size_t i = 0; i < desc[k].size(); ++i) { models.push_back(model_factory.getModel(desc[k][i].distance, desc[k][i].gap_distance)); job_of.emplace_back(k, i); }
    }
    if (job_of.empty()) return;
    std::vector<const Graph *> g1(job_of.size()), g2(job_of.size());
    std::vector<const Model *> mm(job_of.size());
    for (size_t q = 0; q < job_of.size(); ++q) {
        g1[q] = &nodes[(size_t)level[job_of[q].first]].res.graph;
        g2[q] = &nodes[(size_t)desc[job_of[q].first][job_of[q].second].node].res.graph;
        mm[q] = &models[q];
    }
    std::vector<AlignmentResult> ar = alignGraphsBatch(g1, g2, mm, {}, {});
    std::vector<size_t> first_job(level.size(), 0);
    for (size_t q = job_of.size(); q-- > 0;) first_job[job_of[q].first] = q;
    parallel_for(level.size(), [&](size_t k) {
        const size_t nres = desc[k].size();
        if (nres == 0) return;
        Node &nd = nodes[(size_t)level[k]];
        const ProgressiveAlignmentResult &old_result = nd.res;
        Graph anc_graph = old_result.graph;
        anc_graph.reset();
        std::vector<index_t> anc_mapping(anc_graph.size());
        for (index_t i = 0; i < anc_graph.size(); ++i) anc_mapping[i] = i;
        std::vector<std::vector<index_t>> mappings(nres);
        for (size_t i = 0; i < nres; ++i) {
            AlignmentResult &al = ar[first_job[k] + i];
            const Graph &dg = nodes[(size_t)desc[k][i].node].res.graph;
            for (index_t &m : al.mapping1) if (m != NONE) m = anc_mapping[m];
            AncestralResult anc = mergeGraphsIncremental(anc_graph, dg, al.mapping1, al.mapping2, models[first_job[k] + i]);
            anc_graph = anc.graph;
            mappings[i] = anc.mapping2;
            /* the old graph's nodes inside the new ancestral graph, which is a superset (:180-188) */
            std::vector<index_t> inv(anc_graph.size(), (index_t)-2);
            for (index_t j = 0; j < anc.mapping1.size(); ++j) if (anc.mapping1[j] != NONE) inv[anc.mapping1[j]] = j;
            for (index_t &m : anc_mapping) m = inv[m];
            for (size_t j = 0; j < i; ++j) {   // the earlier descendants' mappings follow (:190-198)
                std::vector<index_t> nm(anc_graph.size());
                for (index_t v = 0; v < anc_graph.size(); ++v) nm[v] = anc.mapping1[v] != NONE ? mappings[j][anc.mapping1[v]] : NONE;
                mappings[j].swap(nm);
            }
        }
        /* remove the nodes no descendant uses (:201-228) */
        auto used = [&](index_t v) { for (size_t j = 0; j < nres; ++j) if (mappings[j][v] != NONE) return true; return false; };
        for (index_t i = 0; i < anc_graph.size(); ++i) {
            if (used(i)) continue;
            for (index_t j = i + 1; j < anc_graph.size(); ++j)
                if (used(j)) {
                    anc_graph.rmNodes(i, j - i);
                    for (size_t q = 0; q < nres; ++q) mappings[q].erase(mappings[q].begin() + i, mappings[q].begin() + j);
                    --i;
                    break;
                }
        }
        ProgressiveAlignmentResult result;
        result.score = old_result.score;
        result.is_csprofile = false;
        result.n_tr_indels = old_result.n_tr_indels;
        result.graph = anc_graph;
        for (size_t i = 0; i < nres; ++i) {
            const ProgressiveAlignmentResult &dr = nodes[(size_t)desc[k][i].node].res;
            extend_alignment(a, result, mappings[i], dr.aligned_sequences, false);
            extend_tr_homologies(result, mappings[i], dr.tr_homologies, dr.tr_source);
        }
        if (with_repeats) result.graph.addRepeats(result.tr_homologies);
        nd.res = std::move(result);
    });
}

ProgressiveAlignmentResult progressive_alignment(const Alphabet &a, const std::map<std::string, sequence_t> &sequences,
                                                 const PhyTree &tree, const CSProfile *csprofile,
                                                 const ModelFactory &model_factory,
                                                 const std::map<std::string, std::vector<repeat_t>> *repeats) {
    const auto tl0 = std::chrono::steady_clock::now();
    default_backend().resident_reset();   // (merged profiles a previous pass left on the device)
    std::vector<Node> nodes;
    int root = collect(tree, nodes);
    int maxh = nodes[root].height;

    // ---- leaves (ProgressiveAlignment.cpp:17-46) ----
    std::vector<int> leaves;
    for (size_t i = 0; i < nodes.size(); ++i)
        if (nodes[i].tree->isLeaf()) leaves.push_back((int)i);
    for (int li : leaves)
        if (sequences.find(nodes[li].tree->getName()) == sequences.end()) error("unknown sequence name: %s", nodes[li].tree->getName().c_str());
    // A pass whose merged profiles stay on the device (see the level loop) builds its leaf graphs there too: the host keeps
    // their edges only (SequenceGraph's profile matrix is 160 bytes per residue: 41 MB for 256 x 1000, otherwise built here,
    // copied into the staging block and uploaded for the alignments, and once more for the merges)
    const bool resident_pass = default_backend().resident() && !host_switches().host_merge && cmdlineopts.profile_file.empty() && !cmdlineopts.ancestral_flag && !job_dump_active() &&
                               !cmdlineopts.earlyref_flag;   // (the incremental merges of an early refinement read the profiles on the host)
    const bool resident_leaves = resident_pass && !csprofile;
    // where the profiles of every node live (worker = device context): all 0 with one context
    std::vector<int> owner(nodes.size(), 0);
    if (resident_pass && default_backend().workers() > 1) {
        std::vector<double> leaf_len(nodes.size(), 0.0);
        for (int li : leaves) leaf_len[(size_t)li] = (double)sequences.at(nodes[li].tree->getName()).size();
        owner = assign_owners(nodes, root, default_backend().workers(), leaf_len);
    }
    default_backend().resident_pass = resident_pass;
    default_backend().resident_imports = 0;
    const auto tl1 = std::chrono::steady_clock::now();
    parallel_for(leaves.size(), [&](size_t k) {   // (independent leaves: a thousand SequenceGraphs are 50 ms on one thread)
        Node &nd = nodes[leaves[k]];
        auto it = sequences.find(nd.tree->getName());
        nd.res.aligned_sequences[it->first] = it->second;
        nd.res.score = 0;
        nd.res.n_tr_indels = 0;
        nd.res.is_csprofile = false;
        if (resident_leaves) {
            nd.res.graph = Graph(a.DIM, (index_t)it->second.size() + 2, Graph::NoSites());
        } else if (!csprofile) {
            nd.res.graph = SequenceGraph(a, it->second);
            if (!cmdlineopts.profile_file.empty()) {   // result.profiles[name] = sites without START / END (ProgressiveAlignment.h:73)
                Profile &pf = nd.res.profiles[it->first];
                pf.dim = a.DIM; pf.cols = nd.res.graph.size() - 2;
                pf.data.assign(nd.res.graph.col(1), nd.res.graph.col(1) + (size_t)a.DIM * pf.cols);
            }
        }
    });
    const auto tl2 = std::chrono::steady_clock::now();
    if (resident_leaves && !leaves.empty()) {
        const int nw = default_backend().workers();
        for (int w = 0; w < nw; ++w) {   // every worker builds the leaves of its subtrees
            std::vector<int> mine;
            for (int li : leaves) if (owner[(size_t)li] == w) mine.push_back(li);
            if (mine.empty()) continue;
            std::vector<uint32_t> offs(mine.size() + 1, 0);
            std::vector<const sequence_t *> seqs(mine.size());
            for (size_t s = 0; s < mine.size(); ++s) {
                seqs[s] = &sequences.at(nodes[mine[s]].tree->getName());
                offs[s + 1] = offs[s] + (uint32_t)seqs[s]->size();
            }
            std::vector<int8_t> syms(offs[mine.size()]);
            parallel_for(mine.size(), [&](size_t s) {   // (a quarter of a million residues one push_back at a time: 1 ms of the leaf stage)
                int8_t *out = syms.data() + offs[s];
                for (int8_t c : *seqs[s]) *out++ = a.isValid(c) ? (int8_t)a.value(c) : (int8_t)-1;
            });
            std::vector<const double *> dev(mine.size(), nullptr);
            if (!default_backend().resident_onehot((uint32_t)a.DIM, (uint32_t)mine.size(), syms.data(), offs.data(), dev.data(), w)) error("the backend could not build the leaf graphs on the device");
            for (size_t s = 0; s < mine.size(); ++s) nodes[mine[s]].res.graph.setDevSites(dev[s]);
        }
    }
    if (csprofile) {
        // SequenceGraph(seq, csprofile, model_factory.getModel(branch_length)) for every leaf in one
        // createProfile batch (SequenceGraph.h:111-121, CSProfile.cpp:175-225).
        if (a.kind != ALPHA_AA) error("context-specific profiles need the AA alphabet");
        const uint32_t ns = (uint32_t)leaves.size();
        std::vector<uint32_t> offs(ns + 1, 0);
        std::vector<uint64_t> out_offs(ns + 1, 0);
        std::vector<double> tau(ns), p_uniform((size_t)ns * 20), pi(20, 0.0);
        std::vector<const sequence_t *> seqs(ns);
        for (uint32_t s = 0; s < ns; ++s) {
            seqs[s] = &sequences.at(nodes[leaves[s]].tree->getName());
            offs[s + 1] = offs[s] + (uint32_t)seqs[s]->size();
            out_offs[s + 1] = out_offs[s] + (uint64_t)20 * (seqs[s]->size() + 2);
        }
        std::vector<int8_t> syms(offs[ns]);
        parallel_for(ns, [&](size_t s) {   // (a model per leaf — its branch length —: P(t) from the eigen form, 20 us each)
            int8_t *o = syms.data() + offs[s];
            for (int8_t c : *seqs[s]) *o++ = a.isValid(c) ? (int8_t)a.value(c) : (int8_t)20;
            Model m = model_factory.getModel(nodes[leaves[s]].tree->getBranchLength());
            tau[s] = m.divergence / 0.8;
            if (s == 0) pi = m.pi;   // (the same for every leaf)
            for (int i = 0; i < 20; ++i) {  // model.P * Constant(1/20)
                double acc = 0;
                for (int j = 0; j < 20; ++j) acc += m.P[i + 20 * j] * (1.0 / 20);
                p_uniform[(size_t)s * 20 + i] = acc;
            }
        });
        bool on_device = false;
        if (resident_pass) {   // the profile matrices stay where the leaf's subtree is aligned (as the one-hot leaves above)
            Backend &be = default_backend();
            on_device = true;
            for (int w = 0; w < be.workers() && on_device; ++w) {
                std::vector<uint32_t> mine;
                for (uint32_t s = 0; s < ns; ++s) if (owner[(size_t)leaves[s]] == w) mine.push_back(s);
                if (mine.empty()) continue;
                std::vector<int8_t> sy;
                std::vector<uint32_t> of(mine.size() + 1, 0);
                std::vector<double> ta(mine.size()), pu(mine.size() * 20);
                for (size_t k = 0; k < mine.size(); ++k) {
                    const uint32_t s = mine[k];
                    sy.insert(sy.end(), syms.begin() + offs[s], syms.begin() + offs[s + 1]);
                    of[k + 1] = (uint32_t)sy.size();
                    ta[k] = tau[s];
                    std::copy(p_uniform.begin() + (size_t)s * 20, p_uniform.begin() + (size_t)s * 20 + 20, pu.begin() + k * 20);
                }
                std::vector<const double *> dev(mine.size(), nullptr);
                on_device = be.csprofile_create_batch_res(*csprofile, (uint32_t)mine.size(), sy.data(), of.data(), ta.data(), pi.data(), pu.data(), dev.data(), w);
                if (!on_device) { if (w != 0) error("the backend could not build the leaf profiles on the device"); break; }
                for (size_t k = 0; k < mine.size(); ++k) {
                    Node &nd = nodes[leaves[mine[k]]];
                    nd.res.graph = Graph(20, (index_t)seqs[mine[k]]->size() + 2, Graph::NoSites());
                    nd.res.graph.setDevSites(dev[k]);
                    nd.res.is_csprofile = true;
                }
            }
            if (on_device) be.farm_leaf_workers = std::max(be.farm_leaf_workers, be.workers());
        }
        std::vector<double> out(on_device ? 0 : out_offs[ns]);
        if (!on_device) {
            // the leaves are independent (SequenceGraph.h:111-121): dealt to the device contexts by length
            Backend &be = default_backend();
            std::vector<uint64_t> cost(ns);
            for (uint32_t s = 0; s < ns; ++s) cost[s] = offs[s + 1] - offs[s];
            const std::vector<std::vector<uint32_t>> shards = farm_shards(cost, be.workers());
            if (shards.size() <= 1) {
                be.csprofile_create_batch(*csprofile, ns, syms.data(), offs.data(), tau.data(), pi.data(), p_uniform.data(), out.data(), out_offs.data(), 0);
            } else {
                farm_run(shards, [&](int w) {
                    const std::vector<uint32_t> &sh = shards[(size_t)w];
                    const uint32_t m = (uint32_t)sh.size();
                    std::vector<int8_t> sy;
                    std::vector<uint32_t> of(m + 1, 0);
                    std::vector<uint64_t> oo(m + 1, 0);
                    std::vector<double> ta(m), pu((size_t)m * 20);
                    for (uint32_t k = 0; k < m; ++k) {
                        const uint32_t s = sh[k];
                        sy.insert(sy.end(), syms.begin() + offs[s], syms.begin() + offs[s + 1]);
                        of[k + 1] = (uint32_t)sy.size();
                        oo[k + 1] = oo[k] + (out_offs[s + 1] - out_offs[s]);
                        ta[k] = tau[s];
                        std::copy(p_uniform.begin() + (size_t)s * 20, p_uniform.begin() + (size_t)s * 20 + 20, pu.begin() + (size_t)k * 20);
                    }
                    std::vector<double> o(oo[m]);
                    be.csprofile_create_batch(*csprofile, m, sy.data(), of.data(), ta.data(), pi.data(), pu.data(), o.data(), oo.data(), w);
                    for (uint32_t k = 0; k < m; ++k) std::copy(o.begin() + oo[k], o.begin() + oo[k + 1], out.begin() + out_offs[sh[k]]);
                });
            }
            be.farm_leaf_workers = std::max(be.farm_leaf_workers, (int)shards.size());
        }
        for (uint32_t s = 0; s < ns && !on_device; ++s) {
            Node &nd = nodes[leaves[s]];
            index_t nn = (index_t)((out_offs[s + 1] - out_offs[s]) / 20);
            std::vector<double> sites(out.begin() + out_offs[s], out.begin() + out_offs[s + 1]);
            nd.res.graph = SequenceGraphFromProfile(20, nn, sites);
            nd.res.is_csprofile = true;
        }
    }

    if (repeats && !repeats->empty()) {   // tandem-repeat annotation of the leaves (ProgressiveAlignment.cpp:30-38)
        for (int li : leaves) {
            Node &nd = nodes[li];
            auto it2 = repeats->find(nd.tree->getName());
            if (it2 == repeats->end()) continue;
            for (const repeat_t &rep : it2->second) {
                std::vector<int> tr_hom(nd.res.graph.size(), -1);
                if ((size_t)rep.start + 1 + rep.tr_hom.size() > tr_hom.size()) error("repeat beyond the end of sequence %s", nd.tree->getName().c_str());
                std::copy(rep.tr_hom.begin(), rep.tr_hom.end(), tr_hom.begin() + rep.start + 1);
                nd.res.tr_homologies.push_back(tr_hom);
                nd.res.tr_source.push_back(nd.tree->getName());
            }
            nd.res.graph.addRepeats(nd.res.tr_homologies);
        }
    }
    if (host_switches().profile)
        fprintf(stderr, "leaves: %zu, %.1f ms (names / owners %.2f, graphs %.2f, profiles %.2f)\n", leaves.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count(),
                std::chrono::duration<double, std::milli>(tl1 - tl0).count(), std::chrono::duration<double, std::milli>(tl2 - tl1).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl2).count());
    // ---- internal nodes, one guide-tree level per batch (ProgressiveAlignment.h:413-476) ----
    for (int h = 1; h <= maxh; ++h) {
        std::vector<int> level;
        for (size_t i = 0; i < nodes.size(); ++i)
            if (nodes[i].height == h) level.push_back((int)i);
        const size_t L = level.size();
        std::vector<Pending> pend(L);
        const auto tp0 = std::chrono::steady_clock::now();
        if (resident_pass)   // children of a node above the cut of the subtrees: the profiles of the one on another worker are copied over
            for (size_t k = 0; k < L; ++k) {
                const Node &nd = nodes[level[k]];
                for (int c = 0; c < 2; ++c) {
                    const int ch = nd.child[c];
                    Graph &g = nodes[ch].res.graph;
                    if (owner[(size_t)ch] == owner[(size_t)level[k]] || !g.devSites()) continue;
                    const double *there = default_backend().resident_import(owner[(size_t)level[k]], owner[(size_t)ch], g.devSites(), (size_t)a.DIM * g.size());
                    if (!there) error("the backend could not copy resident profiles between its workers");
                    g.setDevSites(there);
                    owner[(size_t)ch] = owner[(size_t)level[k]];
                    ++default_backend().resident_imports;
                }
            }
        parallel_for(L, [&](size_t k) {
            Node &nd = nodes[level[k]];
            const ProgressiveAlignmentResult &r1 = nodes[nd.child[0]].res, &r2 = nodes[nd.child[1]].res;
            double distance1 = (*nd.tree)[0].getBranchLength(), distance2 = (*nd.tree)[1].getBranchLength();
            double gap_distance1 = distance1, gap_distance2 = distance2;
            if (r1.is_csprofile) distance1 = 0;
            if (r2.is_csprofile) distance2 = 0;
            Pending &p = pend[k];
            p.model = model_factory.getModel(distance1 + distance2, gap_distance1 + gap_distance2);
            p.model1 = model_factory.getModel(distance1, gap_distance1);
            p.model2 = model_factory.getModel(distance2, gap_distance2);
            p.cg1.reset(new CleanedGraph(r1.graph));
            p.cg2.reset(new CleanedGraph(r2.graph));
        });
        const auto tp1 = std::chrono::steady_clock::now();
        std::vector<const Graph *> g1(L), g2(L);
        std::vector<const Model *> mm(L);
        for (size_t k = 0; k < L; ++k) { g1[k] = pend[k].cg1.get(); g2[k] = pend[k].cg2.get(); mm[k] = &pend[k].model; }
        // graphs merged on the device one level below: their profiles never left it (gathered there through the cleaned graphs' node maps)
        std::vector<pgm_site_ref> rs1, rs2;
        {
            bool any = false;
            for (size_t k = 0; k < L; ++k) any = any || pend[k].cg1->devSites() || pend[k].cg2->devSites();
            if (any) {
                rs1.assign(L, pgm_site_ref{nullptr, nullptr, 0u}); rs2.assign(L, pgm_site_ref{nullptr, nullptr, 0u});
                for (size_t k = 0; k < L; ++k) {
                    if (pend[k].cg1->devSites()) rs1[k] = pgm_site_ref{pend[k].cg1->devSites(), pend[k].cg1->nodeMap(), (uint32_t)pend[k].cg1->originalSize()};
                    if (pend[k].cg2->devSites()) rs2[k] = pgm_site_ref{pend[k].cg2->devSites(), pend[k].cg2->nodeMap(), (uint32_t)pend[k].cg2->originalSize()};
                }
            }
        }
        std::vector<int> worker_of(L, 0);
        for (size_t k = 0; k < L; ++k) worker_of[k] = owner[(size_t)level[k]];
        std::vector<AlignmentResult> ar = alignGraphsBatch(g1, g2, mm, rs1, rs2, resident_pass && default_backend().workers() > 1 ? &worker_of : nullptr);
        const auto tp2 = std::chrono::steady_clock::now();
        // mergeGraphs of the whole level: plans on host threads, the node profiles (P g products, L2 normalisation: the
        // arithmetic of the merge) in ONE device batch, then edges / Graph / extend_alignment on host threads again
        std::vector<MergePlan> plans(L);
        std::vector<std::vector<double>> profiles(L);
        // The merged profiles stay on the device when nothing on the host reads them: one device context, no --profile_out /
        // --ancestral_seqs, no job dump (Backend::resident; PGM_NO_RESIDENT=1 keeps the round trip)
        const bool resident = resident_pass;
        std::vector<const double *> dev_profiles(L, nullptr);
        parallel_for(L, [&](size_t k) {
            Node &nd = nodes[level[k]];
            Pending &p = pend[k];
            p.cg1->uncleanMapping(ar[k].mapping1);
            p.cg2->uncleanMapping(ar[k].mapping2);
            plans[k] = planMerge(nodes[nd.child[0]].res.graph, nodes[nd.child[1]].res.graph, ar[k].mapping1, ar[k].mapping2);
            if (!resident) profiles[k].assign((size_t)a.DIM * plans[k].mapping1.size(), 0.0);
        });
        const auto tq0 = std::chrono::steady_clock::now();
        bool on_device = false;
        // (a resident pass: the graphs below need the plans only, not the profiles — the device batch of the node profiles runs on a
        // thread of its own beside them)
        std::thread merge_thread;
        std::exception_ptr merge_error;
        std::vector<pgm_merge_job> mj(L);   // (filled before the thread starts: the graphs section below releases the children)
        bool all_on_device = true;
        if (!host_switches().host_merge)
            for (size_t k = 0; k < L; ++k) {
                Node &nd = nodes[level[k]];
                const Graph &ga = nodes[nd.child[0]].res.graph, &gb = nodes[nd.child[1]].res.graph;
                pgm_merge_job &j = mj[k];
                j.dim = (uint32_t)a.DIM; j.n1 = ga.size(); j.n2 = gb.size(); j.nnodes = (uint32_t)plans[k].mapping1.size();
                j.sites1 = ga.devSites() ? ga.devSites() : ga.col(0); j.sites2 = gb.devSites() ? gb.devSites() : gb.col(0);
                j.P1 = pend[k].model1.P.data(); j.P2 = pend[k].model2.P.data();
                j.k1 = plans[k].mapping1.data(); j.k2 = plans[k].mapping2.data(); j.g2_with_P1 = plans[k].g2_with_P1.data();
                j.profiles = resident ? nullptr : profiles[k].data();
                all_on_device = all_on_device && ga.devSites() && gb.devSites();
            }
        auto merge_on_device = [&]() {
        if (!host_switches().host_merge) {
            const auto tm0 = std::chrono::steady_clock::now();
            {
                // the merges of a level are independent: dealt to the device contexts by the size of the merged graph
                Backend &be = default_backend();
                std::vector<uint64_t> cost(L);
                for (size_t k = 0; k < L; ++k) cost[k] = mj[k].nnodes;
                const std::vector<std::vector<uint32_t>> shards = farm_shards(cost, be.workers());
                if (resident && be.workers() > 1) {   // every merge where its children are: one batch per worker, concurrently
                    std::vector<std::vector<uint32_t>> mine((size_t)be.workers());
                    for (size_t k = 0; k < L; ++k) mine[(size_t)worker_of[k]].push_back((uint32_t)k);
                    std::vector<char> ok(mine.size(), 1);
                    farm_run(mine, [&](int w) {
                        const std::vector<uint32_t> &sh = mine[(size_t)w];
                        std::vector<pgm_merge_job> q(sh.size());
                        std::vector<const double *> dv(sh.size(), nullptr);
                        for (size_t k = 0; k < sh.size(); ++k) q[k] = mj[sh[k]];
                        ok[(size_t)w] = be.merge_profiles_batch_res((uint32_t)q.size(), q.data(), dv.data(), w) ? 1 : 0;
                        for (size_t k = 0; k < sh.size(); ++k) dev_profiles[sh[k]] = dv[k];
                    });
                    on_device = true;
                    for (char c : ok) on_device = on_device && c;
                    if (!on_device) error("the backend could not keep the merged profiles on the device");
                } else if (resident) {
                    on_device = be.merge_profiles_batch_res((uint32_t)L, mj.data(), dev_profiles.data(), 0);
                    if (!on_device) error("the backend could not keep the merged profiles on the device");
                } else if (shards.size() <= 1) {
                    on_device = be.merge_profiles_batch((uint32_t)L, mj.data(), 0);
                } else {
                    std::vector<char> ok(shards.size(), 0);
                    farm_run(shards, [&](int w) {
                        const std::vector<uint32_t> &sh = shards[(size_t)w];
                        std::vector<pgm_merge_job> q(sh.size());
                        for (size_t k = 0; k < sh.size(); ++k) q[k] = mj[sh[k]];
                        ok[(size_t)w] = be.merge_profiles_batch((uint32_t)q.size(), q.data(), w) ? 1 : 0;
                    });
                    on_device = true;
                    for (size_t w = 0; w < shards.size(); ++w) on_device = on_device && (ok[w] || shards[w].empty());
                }
            }
            default_backend().seconds_merge_profiles += std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count();
        }
        };
        if (resident && all_on_device && !host_switches().host_merge) merge_thread = std::thread([&]() { try { merge_on_device(); } catch (...) { merge_error = std::current_exception(); } });
        else merge_on_device();
        struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } merge_joiner{merge_thread};   // (also when the section below throws)
        const auto tq1 = std::chrono::steady_clock::now();
        std::atomic<long long> ns_merge(0), ns_extend(0);
        parallel_for(L, [&](size_t k) {
            const auto tk0 = std::chrono::steady_clock::now();
            Node &nd = nodes[level[k]];
            ProgressiveAlignmentResult &r1 = nodes[nd.child[0]].res, &r2 = nodes[nd.child[1]].res;
            Pending &p = pend[k];
            ProgressiveAlignmentResult &result = nd.res;
            result.score = ar[k].score;
            result.is_csprofile = false;
            result.n_tr_indels = ar[k].n_tr_indels + r1.n_tr_indels + r2.n_tr_indels;
            if (!resident && !on_device) mergeProfilesHost(r1.graph, r2.graph, p.model1, p.model2, plans[k], profiles[k]);
            AncestralResult anc = finishMerge(r1.graph, r2.graph, plans[k], resident ? nullptr : profiles[k].data(), (*nd.tree)[0].getBranchSupport(), (*nd.tree)[1].getBranchSupport());
            result.graph = anc.graph;
            const auto tk1 = std::chrono::steady_clock::now();
            extend_alignment(a, result, anc.mapping1, r1.aligned_sequences, L == 1);
            extend_alignment(a, result, anc.mapping2, r2.aligned_sequences, L == 1);
            ns_merge += std::chrono::duration_cast<std::chrono::nanoseconds>(tk1 - tk0).count();
            ns_extend += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tk1).count();
            if (!r1.tr_homologies.empty() || !r2.tr_homologies.empty()) {   // ProgressiveAlignment.h:455-456, 468
                extend_tr_homologies(result, anc.mapping1, r1.tr_homologies, r1.tr_source);
                extend_tr_homologies(result, anc.mapping2, r2.tr_homologies, r2.tr_source);
            }
            if (!cmdlineopts.profile_file.empty() || cmdlineopts.ancestral_flag) {
                result.profiles.insert(r1.profiles.begin(), r1.profiles.end());
                result.profiles.insert(r2.profiles.begin(), r2.profiles.end());
            }
            if (cmdlineopts.ancestral_flag) {   // ProgressiveAlignment.h:458-466
                if (r1.aligned_sequences.size() > 1) ancestral_seq(a, result, create_ancestral_seq_name(r1.aligned_sequences), r1.graph, &anc.mapping1, anc.is_matched, p.model1);
                if (r2.aligned_sequences.size() > 1) ancestral_seq(a, result, create_ancestral_seq_name(r2.aligned_sequences), r2.graph, &anc.mapping2, anc.is_matched, p.model2);
                ancestral_seq(a, result, create_ancestral_seq_name(result.aligned_sequences), result.graph, nullptr, anc.is_matched, p.model);
            }
            if (cmdlineopts.repeats_flag) nd.tr_note = "TR indels at " + create_ancestral_seq_name(result.aligned_sequences) + ": " + std::to_string(ar[k].n_tr_indels);   // (:470-473)
            if (repeats && !repeats->empty()) result.graph.addRepeats(result.tr_homologies);   // (:468; with no annotation at all the merged graph has no repeat edges either way)
            // children are no longer needed (the reference copies them by value and drops them)
            if (!cmdlineopts.earlyref_flag) {   // (the reference's alignment_cache, ProgressiveAlignment.h:107-109: the parent's refinement reads them again)
                r1 = ProgressiveAlignmentResult();
                r2 = ProgressiveAlignmentResult();
            }
            p.cg1.reset();
            p.cg2.reset();
            profiles[k] = std::vector<double>();
        });
        if (merge_thread.joinable()) {
            merge_thread.join();
            if (merge_error) std::rethrow_exception(merge_error);
        }
        if (resident) for (size_t k = 0; k < L; ++k) nodes[level[k]].res.graph.setDevSites(dev_profiles[k]);
        if (cmdlineopts.earlyref_flag) early_refinement(a, nodes, level, model_factory, repeats && !repeats->empty());
        const auto tp3 = std::chrono::steady_clock::now();
        if (host_switches().profile)
            fprintf(stderr, "[%.1f ms] level %d: %zu nodes, host pre %.1f ms, alignGraphsBatch %.1f ms, host post (merge, extend) %.1f ms\n",
                    std::chrono::duration<double, std::milli>(tp3 - tl0).count(), h, L,
                    std::chrono::duration<double, std::milli>(tp1 - tp0).count(), std::chrono::duration<double, std::milli>(tp2 - tp1).count(),
                    std::chrono::duration<double, std::milli>(tp3 - tp2).count()),
            fprintf(stderr, "    post: plans %.1f ms, node profiles %.1f, edges / graphs / extend %.1f (summed over the nodes: graphs %.2f ms, extend_alignment %.2f)\n", std::chrono::duration<double, std::milli>(tq0 - tp2).count(),
                    std::chrono::duration<double, std::milli>(tq1 - tq0).count(), std::chrono::duration<double, std::milli>(tp3 - tq1).count(), ns_merge.load() / 1e6, ns_extend.load() / 1e6);
    }
    if (cmdlineopts.repeats_flag)   // the reference prints them as its recursion returns: post-order, which is the order of `nodes`
        for (const Node &nd : nodes) if (!nd.tr_note.empty()) std::cerr << nd.tr_note << std::endl;
    ProgressiveAlignmentResult out = std::move(nodes[root].res);
    if (host_switches().profile)
        fprintf(stderr, "[%.1f ms] root result taken\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
    nodes.clear();
    if (host_switches().profile)
        fprintf(stderr, "[%.1f ms] nodes released\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
    return out;
}

}  // namespace pgm
