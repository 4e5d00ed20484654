// pgm_host.h — host-side C++ mirror of the reference's progressive-alignment call surface.
//
// Dependency-free C++17 (no Eigen, no TCLAP).  Names, argument meaning and error behaviour
// follow acg-team/ProGraphMSA so that the scaffolding around the hot path reads like the
// reference; the three hot functions themselves (alignGraphs, DistanceFactoryAlign::alignPair,
// CSProfile::createProfile) are thin wrappers over the C ABI of include/pgm_hip.h.
//
// Reference files mirrored (all under /root/reference/src):
//   Alphabet.{h,cpp}  Model.h  ModelFactory.{h,cpp}  ModelFactoryWag.cpp  ModelFactoryEcm.cpp
//   Graph.h  SequenceGraph.h  CleanedGraph.h  GraphAlign.h  ProgressiveAlignment.{h,cpp}
//   PhyTree.{h,cpp}  newick.cpp  Fasta.cpp  TreeNJ.{h,cpp}  DistanceFactory*.{h,cpp}
//   CSProfile.{h,cpp}  main.{h,cpp}
#ifndef PGM_HOST_H_
#define PGM_HOST_H_

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include <functional>

#include "../../include/pgm_hip.h"

namespace pgm {

typedef double score_t;      // main.h:31
typedef float dp_score_t;    // main.h:32
typedef uint32_t index_t;    // main.h:33
typedef double distance_t;   // main.h:34

// ---------------------------------------------------------------------------------------
// cmdlineopts (main.h:37-82) with the defaults of main.cpp:37-169
struct cmdlineopts_t {
    int iters = 2;
    bool fasta_flag = false, noforcealign_flag = false, nwdist_flag = false, onlytree_flag = false;
    bool mldist_flag = false, mldist_gap_flag = false, codon_flag = false, inputorder_flag = false;
    bool ancestral_flag = false;
    bool earlyref_flag = false;       // --early_refinement (ProgressiveAlignment.h:102-110)
    std::string profile_file;   // --profile_out (main.cpp:132)
    std::string readreps_file;  // --read_repeats (main.cpp:108)
    bool repeats_flag = false;  // -R: here only the "TR indels" lines on stderr (T-REKS itself is not run: --read_repeats supplies the repeats)
    double indel_rate = 0.0093359375;
    double end_indel_prob = 0.12;
    double gapext_prob = 0.6119140625;
    double edge_halflife = 0.3;
    double altsplice_prob = 0.328125;
    double pseudo_count = 1000;
    double cutoff_dist = 2.2;
    double repeat_rate = 0.1;
    double repeatext_prob = 0.3;
    double max_dist = 2.2, min_dist = 0.05, max_pdist = 0.8, min_pdist = 0.05;
    std::string output_file, sequence_file, tree_file, cs_file;
};
extern cmdlineopts_t cmdlineopts;

// error() of debug.cpp:43-52 prints and abort()s; the mirror throws so that the C-level
// callers (and tests) can observe it.  main() maps it to exit code 2 like main.cpp:315-319.
struct pgm_exception : std::runtime_error {
    explicit pgm_exception(const std::string &m) : std::runtime_error(m) {}
};
[[noreturn]] void error(const char *fmt, ...);

// ---------------------------------------------------------------------------------------
// Alphabets (Alphabet.h:37-115).  A symbol is stored as the reference stores it in `data`:
// AA keeps the raw character, Codon keeps the codon index (61 = unknown, 62 = gap, -1 = invalid).
enum AlphabetKind { ALPHA_AA = 0, ALPHA_CODON = 1 };
struct Alphabet {
    AlphabetKind kind;
    int DIM;
    explicit Alphabet(AlphabetKind k) : kind(k), DIM(k == ALPHA_AA ? 20 : 61) {}
    int value(int8_t data) const;                  // AA::value / Codon::value
    bool isValid(int8_t data) const { int v = value(data); return v >= 0 && v < DIM; }
    int8_t gap() const;                            // ALPHABET::GAP
    int8_t unknown() const;                        // ALPHABET::X
    bool isGap(int8_t d) const { return d == gap(); }
    char asChar(int8_t data) const;
    std::string asString(int8_t data) const;
    bool stripsStart(int8_t first) const;          // main.cpp:340
    bool stripsEnd(int8_t last) const;             // main.cpp:348
};
typedef std::basic_string<int8_t> sequence_t;
sequence_t sequenceFromString(const Alphabet &a, const std::string &str);             // Alphabet.h:118
std::string stringFromSequence(const Alphabet &a, const sequence_t &seq);             // :136
std::string stringFromSequence(const Alphabet &a, const sequence_t &seq, const std::string &orig);  // :148

// ---------------------------------------------------------------------------------------
// Model (Model.h:8-24); matrices are dim x dim column-major.
struct Model {
    int dim = 0;
    std::vector<double> M, P, Q, pi;
    double delta = 0, epsilon = 0;
    distance_t distance = 0, divergence = 0;
};

// ModelFactory (ModelFactory.h:11-34) with the WAG (ModelFactoryWag.cpp) and ECM
// (ModelFactoryEcm.cpp) rate matrices.  P(t) = exp(Q t) by scaling-and-squaring instead of Eigen's
// general EigenSolver; agrees to ~1e-14.
class ModelFactory {
public:
    static ModelFactory *getDefault(const Alphabet &a);          // ModelFactory.cpp:11-36
    Model getModel(distance_t distance) const;                   // ModelFactory.h:48-67
    Model getModel(distance_t distance, distance_t gap_distance) const;  // :70-90
    double getEpsilon(distance_t) const { return cmdlineopts.gapext_prob; }
    double getDelta(distance_t distance) const;
    int dim() const { return dim_; }
    const std::vector<double> &Qmat() const { return Q_; }
    const std::vector<double> &freqs() const { return freqs_; }
    // eigen form P(d) = V diag(exp(sigma d)) V^-1 (only for reversible rate matrices: WAG)
    bool has_eigen() const { return use_eigen_; }
    const std::vector<double> &eigV() const { return V_; }
    const std::vector<double> &eigVi() const { return Vi_; }
    const std::vector<double> &eigSigma() const { return sigma_; }

private:
    ModelFactory(int dim, const std::string &qmat_file);
    static void parseDistance(distance_t distance, Model &model);  // ModelFactory.h:104-127
    void fillP(Model &model) const;
    int dim_;
    std::vector<double> freqs_, Q_, V_, Vi_, sigma_;
    bool use_eigen_ = false;
};

// ---------------------------------------------------------------------------------------
// Graph (Graph.h:20-502): profile matrix + two CSR matrices (edges float, repeats uint).
class Graph {
public:
    typedef std::map<std::pair<index_t, index_t>, dp_score_t> EdgeMap;   // key (to,from)
    typedef std::map<std::pair<index_t, index_t>, index_t> RepeatMap;

    Graph() : Graph(20) {}
    explicit Graph(int dim);                                                    // Graph.h:141-150
    Graph(int dim, const std::vector<std::vector<double>> &nodes);              // :104-120
    Graph(int dim, const std::vector<std::vector<double>> &nodes, const EdgeMap &edges,
          const RepeatMap &repeats);                                            // :122-139
    // the same from flat data (mergeGraphs' hot path): profiles dim x n column-major (columns 0 and n - 1 are taken as zero),
    // edge / repeat lists sorted by (to, from) without duplicates — what iterating the two maps above yields
    struct EdgeRec { index_t to, from; dp_score_t cost; };
    struct RepeatRec { index_t to, from, units; };
    Graph(int dim, index_t n, const double *profiles, const std::vector<EdgeRec> &edges, const std::vector<RepeatRec> &repeats);
    // a chain of n nodes whose profiles live on the device only (a leaf graph of a resident pass: setDevSites)
    struct NoSites {};
    Graph(int dim, index_t n, NoSites);

    index_t size() const { return n_; }
    int dim() const { return dim_; }
    const double *col(index_t i) const { return &sites_[(size_t)dim_ * i]; }
    const std::vector<double> &getSites() const { return sites_; }
    // A merged graph whose profiles were left on the device (Backend::merge_profiles_batch_res) has no host copy of them
    // (sites_ empty): devSites() is the dim x n matrix in HBM.  A CleanedGraph of such a graph refers to the same matrix
    // through its node mapping (nodeMap()).
    const double *devSites() const { return dev_sites_; }
    void setDevSites(const double *p) { dev_sites_ = p; }
    bool hasHostSites() const { return !sites_.empty(); }
    pgm_graph flat() const;  // view for the C ABI (valid while *this is alive and unchanged)

    // PredIterator (Graph.h:180-248)
    class PredIterator {
    public:
        PredIterator(const Graph &g, index_t row, dp_score_t repeat_init, dp_score_t repeat_ext);
        PredIterator &operator++() { if (i_ < iend_) ++i_; else ++j_; return *this; }
        bool isRepeat() const { return !(i_ < iend_); }
        index_t repeatUnits() const { return g_->r_units_[j_]; }
        explicit operator bool() const { return i_ < iend_ || j_ < jend_; }
        dp_score_t value() const;
        dp_score_t rawValue() const { return g_->e_val_[i_]; }
        index_t operator*() const { return i_ < iend_ ? g_->e_col_[i_] : g_->r_col_[j_]; }
    private:
        const Graph *g_;
        int32_t i_, iend_, j_, jend_;
        dp_score_t repeatInit_, repeatExt_;
    };
    PredIterator getPreds(index_t node, dp_score_t repeat_init, dp_score_t repeat_ext) const {
        return PredIterator(*this, node, repeat_init, repeat_ext);
    }

protected:
    void fillInitialEdges();                            // :35-46
    void setEdgesFromMap(const EdgeMap &edge_map);      // :81-90
    void setRepeatsFromMap(const RepeatMap &rep_map);   // :92-100
public:
    // tandem-repeat edges from the unit homologies of the graph's nodes (Graph.h:48-79, 458-469): tr_homology[i] = column of
    // node i + 1 inside its repeat unit, -1 outside a repeat; replaces the repeat matrix
    void addRepeats(const std::vector<std::vector<int>> &tr_homologies);
    // early refinement (Graph.h:369-426): every profile column but START / END set to ones; nodes [first, first + count) removed
    // with every edge that touches them
    void reset();
    void rmNodes(index_t first, index_t count = 1);
private:
    int dim_;
    index_t n_;
    const double *dev_sites_ = nullptr;
    std::vector<double> sites_;  // dim x n column-major
    std::vector<int32_t> e_rowptr_;
    std::vector<uint32_t> e_col_;
    std::vector<float> e_val_;
    std::vector<int32_t> r_rowptr_;
    std::vector<uint32_t> r_col_;
    std::vector<uint32_t> r_units_;
    friend class CleanedGraph;
};

// SequenceGraph (SequenceGraph.h:101-121)
Graph SequenceGraph(const Alphabet &a, const sequence_t &seq);
Graph SequenceGraphFromProfile(int dim, index_t nnodes, const std::vector<double> &sites);  // :111-121

// CleanedGraph (CleanedGraph.h:39-160)
class CleanedGraph : public Graph {
public:
    explicit CleanedGraph(const Graph &original);
    index_t getMapping(index_t i) const { return outmapping_[i]; }
    const index_t *nodeMap() const { return outmapping_.data(); }   // node of the cleaned graph -> node of the original
    index_t originalSize() const { return original_size_; }         // nodes of the original (columns of its profile matrix)
    void uncleanMapping(std::vector<index_t> &mapping) const;
private:
    std::vector<index_t> outmapping_;
    index_t original_size_ = 0;
};

// ---------------------------------------------------------------------------------------
// GraphAlign.h
struct AlignmentResult {   // GraphAlign.h:6-12
    dp_score_t score = 0;
    index_t n_tr_indels = 0;
    std::vector<index_t> mapping1, mapping2;
};
struct AncestralResult {   // GraphAlign.h:14-20
    Graph graph;
    std::vector<index_t> mapping1, mapping2;
    std::vector<bool> is_matched;
};
double averageAlignmentLength(const Graph &g);                                        // :82-96
pgm_scores DynProgScores(const Graph &g1, const Graph &g2, const Model &model);       // :98-143

// Environment switches of the host side, read ONCE per process (host_switches()): which path runs is decided at start-up and shown by
// `pgmsa --stats` ("switches"), not re-read per pass.  All of them are test / measurement aids; the defaults are the product.
struct HostSwitches {
    bool profile = false;        // PGM_HOST_PROFILE: per-stage timings on stderr
    bool host_merge = false;     // PGM_HOST_MERGE: the node profiles of mergeGraphs on the host instead of pgm_merge_profiles_batch (same bits)
    bool no_resident = false;    // PGM_NO_RESIDENT: merged profiles travel through the host between the levels
    bool host_counts = false;    // PGM_HOST_COUNTS: pair counts of an alignment on the host instead of pgm_prealigned_counts (same integers)
    bool device_mldist = false;  // PGM_DEVICE_MLDIST: ML distances by pgm_mldist_batch (last-bit differences to the host's estimator)
    std::string describe() const;   // the switches that are on, comma separated ("" = the product's defaults)
};
const HostSwitches &host_switches();

// Backend = the C ABI entry points of include/pgm_hip.h behind a context.  The product binds
// them to libpgm_hip.so (HIP kernels); tests bind the oracle.  There is no CPU fallback: when
// the HIP library cannot create a context this throws.
struct Backend {
    virtual ~Backend() {}
    virtual const char *name() const = 0;
    // Every batch call takes the worker (device context) it runs on, 0 <= worker < workers(); calls with different workers may
    // run concurrently.  The host code shards a batch's independent units over the workers (farm_shards below): jobs of a
    // guide-tree level, leaves, merges, sequence pairs — no exchange between workers, results independent of their number.
    // res1 / res2 (may be NULL): graphs whose profiles the device already holds (include/pgm_hip.h: pgm_site_ref)
    virtual void align_graphs_batch(uint32_t njobs, const pgm_graph *const *g1, const pgm_graph *const *g2,
                                    const pgm_model *const *model, const pgm_scores *scores,
                                    pgm_align_out *out, int worker = 0, const pgm_site_ref *res1 = nullptr, const pgm_site_ref *res2 = nullptr) = 0;
    // the merged graphs' profiles may stay on the device between the levels of a progressive pass (one context, nobody reads
    // them on the host): merge_profiles_batch_res leaves them there and returns their device addresses
    virtual bool resident() const { return false; }
    virtual bool resident_onehot(uint32_t, uint32_t, const int8_t *, const uint32_t *, const double **, int = 0) { return false; }   // leaf graphs built on the device
    virtual void resident_reset() {}   // start of a progressive pass: the previous pass's device-resident profiles are dead
    virtual bool merge_profiles_batch_res(uint32_t, const pgm_merge_job *, const double **, int = 0) { return false; }
    // a resident matrix of worker `src` copied to worker `dst` (a pass sharded by subtree: the parent of two subtrees needs both children's
    // profiles where it runs); the address on `dst`, valid like every resident address until resident_reset
    virtual const double *resident_import(int /*dst*/, int /*src*/, const double *, size_t /*count*/) { return nullptr; }
    // One tile of alignPair jobs in two halves (include/pgm_hip.h: pgm_nw_pairs_submit / pgm_nw_pairs_wait): a worker keeps two
    // tiles in flight.  flags & PGM_NW_REDUCED: counts = (ident, total) per pair.  Result buffers come from host_alloc.
    virtual int nw_pairs_submit(uint32_t dim, const int32_t *score, int32_t go, int32_t ge, uint32_t nseq,
                                const int8_t *syms, const uint32_t *offs, uint32_t npairs, const uint32_t *pi,
                                const uint32_t *pj, uint32_t flags, int32_t *counts, uint32_t *gaps, int worker = 0) = 0;
    virtual void nw_pairs_wait(int ticket, int worker = 0) = 0;
    virtual void *host_alloc(size_t bytes) { return malloc(bytes); }
    virtual void host_free(void *p) { free(p); }
    // number of device contexts the farms may drive, one host thread each
    virtual int workers() const { return 1; }
    // batched DistanceFactoryML::computeDistance and the pair counts of an alignment on the device;
    // false = this backend has no such kernel (the host estimator is used)
    virtual bool mldist_batch(const pgm_mldist_model &, uint32_t, const int32_t *, const uint32_t *, const double *, double *, double *, int = 0) { return false; }
    virtual bool prealigned_counts_batch(uint32_t, uint32_t, uint32_t, const int8_t *, uint32_t, const uint32_t *, const uint32_t *, int32_t *, uint32_t *, int = 0) { return false; }
    // cosine matrix of the k-mer count vectors (DistanceFactoryAngle.h:100): counts nseq x ncols row-major -> nseq x nseq column-major
    virtual void kmer_cosine(uint32_t nseq, uint32_t ncols, const int32_t *counts, double *cosine, int worker = 0) = 0;
    // node profiles of a batch of merged graphs on the device (numeric part); false = host arithmetic
    virtual bool merge_profiles_batch(uint32_t, const pgm_merge_job *, int = 0) { return false; }
    double seconds_merge_profiles = 0;
    virtual void csprofile_create_batch(const class CSProfile &lib, uint32_t nseq, const int8_t *syms, const uint32_t *offs,
                                        const double *tau, const double *pi, const double *p_uniform, double *out,
                                        const uint64_t *out_offs, int worker = 0) = 0;
    // the profile library handed to every worker ahead of the first createProfile batch (start-up: before the clocks of the stages)
    virtual void csprofile_preload(const class CSProfile &) {}
    // ... left on the device of `worker` (resident pass): dev[s] = the 20 x (len + 2) matrix of sequence s; false = not available
    virtual bool csprofile_create_batch_res(const class CSProfile &, uint32_t, const int8_t *, const uint32_t *, const double *, const double *,
                                            const double *, const double **, int = 0) { return false; }
    int farm_workers = 0, farm_tiles = 0;   // what the last all-pairs farm used (logs / --stats)
    int farm_level_workers = 0, farm_leaf_workers = 0;   // most workers a guide-tree level's jobs / the leaves' profiles were dealt to
    bool resident_pass = false; int resident_imports = 0;   // the last progressive pass kept its profiles on the devices; matrices copied between them
    uint64_t cells_aligned = 0;   // Σ (n1-2)(n2-2)
    uint64_t cells_nw = 0;        // Σ L1*L2
    double seconds_align = 0, seconds_nw = 0, seconds_mldist = 0;
};
Backend &default_backend();            // defined by exactly one backend_*.cpp linked into the program
// Units of a batch dealt to `nw` workers: longest first, each to the worker with the least load so far (ties: the lowest
// worker).  Deterministic; shard w lists its units in descending cost.  With one worker or one unit: everything to worker 0.
std::vector<std::vector<uint32_t>> farm_shards(const std::vector<uint64_t> &cost, int nw);
// runs fn(w) for every non-empty shard, worker 0 on the calling thread, the others on threads of their own
void farm_run(const std::vector<std::vector<uint32_t>> &shards, const std::function<void(int)> &fn);
void set_job_dump(const std::string &path);  // if set, every alignGraphs job is appended to this file
bool job_dump_active();
void set_dist_dump(const std::string &path); // if set, every distance matrix TreeNJ estimates is appended (dim, D, V as raw doubles)

// alignGraphs (GraphAlign.h:200-534): one job; and the batched form the scheduler uses.
AlignmentResult alignGraphs(const Graph &g1, const Graph &g2, const Model &model);
// (res1 / res2: per job, where the device holds the graphs' profiles — empty vectors or entries with dev_sites == NULL: nowhere)
std::vector<AlignmentResult> alignGraphsBatch(const std::vector<const Graph *> &g1,
                                              const std::vector<const Graph *> &g2,
                                              const std::vector<const Model *> &model,
                                              const std::vector<pgm_site_ref> &res1 = std::vector<pgm_site_ref>(),
                                              const std::vector<pgm_site_ref> &res2 = std::vector<pgm_site_ref>(),
                                              const std::vector<int> *worker_of = nullptr);   // (given: job k runs on worker (*worker_of)[k] — where its profiles are)
// mergeGraphs (GraphAlign.h:550-727)
// mergeGraphs in three parts, so that the profiles of all merges of a guide-tree level can be computed in one device batch:
//   planMerge     the "unify" walk over the two mappings (GraphAlign.h:569-620 without the arithmetic): per node of the
//                 merged graph its source nodes, whether a skipped g2 node is propagated with model1.P (:591), is_matched
//   merge profiles  host: mergeProfilesHost; device: Backend::merge_profiles_batch (bit-identical)
//   finishMerge   homologous path, inverse mappings, penalties, edge maps, Graph (:626-727)
struct MergePlan {
    std::vector<index_t> mapping1, mapping2;   // result.mapping1 / mapping2 (PGM_GAP = none)
    std::vector<bool> is_matched;
    std::vector<uint8_t> g2_with_P1;
};
MergePlan planMerge(const Graph &g1, const Graph &g2, const std::vector<index_t> &mapping1, const std::vector<index_t> &mapping2);
void mergeProfilesHost(const Graph &g1, const Graph &g2, const Model &model1, const Model &model2, const MergePlan &plan,
                       std::vector<double> &profiles);   // dim x nnodes column-major
AncestralResult finishMerge(const Graph &g1, const Graph &g2, const MergePlan &plan, const double *profiles /* NULL: they stay on the device */,
                            double support1, double support2);
AncestralResult mergeGraphs(const Graph &g1, const Graph &g2, const std::vector<index_t> &mapping1,
                            const std::vector<index_t> &mapping2, const Model &model1, const Model &model2,
                            double support1, double support2);
// GraphAlign.h:729-882: the graph of an early refinement grows by one aligned descendant at a time (one model, no penalties)
AncestralResult mergeGraphsIncremental(const Graph &anc_graph, const Graph &graph, const std::vector<index_t> &anc_mapping,
                                       const std::vector<index_t> &mapping, const Model &model);

// ---------------------------------------------------------------------------------------
// PhyTree (PhyTree.h) + newick (newick.cpp)
class PhyTree {
public:
    explicit PhyTree(std::string name = "") : parent_(nullptr), branch_length_(0), branch_support_(1), name_(std::move(name)) {}
    ~PhyTree();
    PhyTree(const PhyTree &) = delete;
    PhyTree *copy() const;
    void addChild(PhyTree *child, double branch_length = 0, double branch_support = 1);
    void pluck();
    PhyTree *pluckChild(index_t index);
    index_t indexOf() const;
    const std::string &getName() const { return name_; }
    PhyTree *getParent() { return parent_; }
    double getBranchLength() const { return branch_length_; }
    double getBranchSupport() const { return branch_support_; }
    PhyTree &operator[](int i) { return *children_[i]; }
    const PhyTree &operator[](int i) const { return *children_[i]; }
    index_t n_children() const { return (index_t)children_.size(); }
    bool isLeaf() const { return children_.empty(); }
    std::string formatNewick() const;
private:
    std::string formatNewickR() const;
    std::vector<PhyTree *> children_;
    PhyTree *parent_;
    double branch_length_, branch_support_;
    std::string name_;
};
PhyTree *midpointRoot(PhyTree *root);                      // PhyTree.cpp:60-116
std::vector<std::string> get_tree_order(const PhyTree *tree);  // PhyTree.cpp:164-182
PhyTree *parse_newick(std::istream &in);                   // newick.cpp:127-147

// Fasta.cpp
std::map<std::string, std::string> read_fasta(const std::string &file, std::vector<std::string> &order);
void write_fasta(const std::map<std::string, std::string> &aln, const std::vector<std::string> &order, std::ostream &out);

// ---------------------------------------------------------------------------------------
// CSProfile (CSProfile.{h,cpp})
class CSProfile {
public:
    explicit CSProfile(const std::string &filename);   // parser CSProfile.cpp:29-170
    int nprof() const { return nprof_; }
    int ncols() const { return ncols_; }
    const std::vector<double> &lprofiles() const { return lprofiles_; }  // [k][col][21]
    const std::vector<double> &centre() const { return centre_; }        // [k][20]
    const std::vector<double> &priors() const { return priors_; }
private:
    int nprof_ = -1, ncols_ = -1;
    std::vector<double> lprofiles_, centre_, priors_;
};

// ---------------------------------------------------------------------------------------
// ProgressiveAlignment.{h,cpp}
struct Profile { int dim = 0; index_t cols = 0; std::vector<double> data; };   // Model<A>::Profile: dim x cols, column-major
struct repeat_t { index_t len = 0, start = 0; std::vector<int> tr_hom; };        // Repeat.h
// --read_repeats: T-REKS output (RepeatDetectionTReks.cpp:62-157) against the (start / stop stripped) sequences
std::map<std::string, std::vector<repeat_t>> read_repeats(const Alphabet &a, const std::string &filename, const std::map<std::string, sequence_t> &seqs);
struct ProgressiveAlignmentResult {   // ProgressiveAlignment.h:27-37
    std::map<std::string, sequence_t> aligned_sequences;
    std::map<std::string, Profile> profiles;   // leaves always; ancestors with --ancestral_seqs (:73, :362, :410)
    std::vector<std::vector<int>> tr_homologies;   // per annotated repeat: unit column of every node of the graph (without START / END), -1 elsewhere
    std::vector<std::string> tr_source;
    Graph graph;
    score_t score = 0;
    index_t n_tr_indels = 0;
    bool is_csprofile = false;
};
// align_progressive_results (ProgressiveAlignment.h:413-476) for a whole guide-tree level at once.
// fn(0..n-1) on the host threads (PGM_HOST_THREADS, at most 16); exceptions are rethrown on the caller
void parallel_for(size_t n, const std::function<void(size_t)> &fn);

// progressive_alignment (ProgressiveAlignment.cpp:12-71): same post-order results as the reference's
// recursion; internal nodes whose children are finished are aligned together in one batch.
ProgressiveAlignmentResult progressive_alignment(const Alphabet &a, const std::map<std::string, sequence_t> &sequences,
                                                 const PhyTree &tree, const CSProfile *csprofile,
                                                 const ModelFactory &model_factory,
                                                 const std::map<std::string, std::vector<repeat_t>> *repeats = nullptr);

// ---------------------------------------------------------------------------------------
// Distances / guide tree
struct DistanceMatrix {   // DistanceFactory.h:12-18
    int dim;
    std::vector<double> distances, variances;   // dim x dim
    explicit DistanceMatrix(int d) : dim(d), distances((size_t)d * d, 0.0), variances((size_t)d * d, 0.0) {}
    double &D(int i, int j) { return distances[(size_t)i * dim + j]; }
    double &V(int i, int j) { return variances[(size_t)i * dim + j]; }
};
struct distvar_t { distance_t dist, var; };
class DistanceFactoryML {   // DistanceFactoryML.h
public:
    DistanceFactoryML(const Alphabet &a, const ModelFactory *mf) : alphabet(a), model_factory(mf) {}
    distvar_t computeDistance(const std::vector<int32_t> &counts, index_t gaps, double seqlen) const;  // :137-190
    distvar_t computeDistance(double ident, double total, const std::vector<int32_t> *counts, index_t gaps, double seqlen) const;
    // computeDistance of every pair: on the device when PGM_DEVICE_MLDIST is set, the backend has the kernel and the model is
    // in eigen form (20 states), else on host threads; fills the symmetric matrices
    void computeDistances(const int32_t *counts, const uint32_t *gaps, const std::vector<double> &seqlen,
                          const std::vector<uint32_t> &pi, const std::vector<uint32_t> &pj, DistanceMatrix &distances) const;
protected:
    distvar_t computeMLDist(const std::vector<int32_t> &counts, index_t gaps, double seqlen, double dist0, double var0) const;  // :66-135
    Alphabet alphabet;
    const ModelFactory *model_factory;
};
class DistanceFactoryAlign : public DistanceFactoryML {   // DistanceFactoryAlign.h
public:
    DistanceFactoryAlign(const Alphabet &a, const ModelFactory *mf);
    DistanceMatrix computePwDistances(const std::map<std::string, sequence_t> &sequences,
                                      const std::vector<std::string> &order);   // :29-56
    const std::vector<int32_t> &scoring_matrix() const { return scoring_matrix_; }
    int gap_open = -10, gap_extend = -2;
private:
    std::vector<int32_t> scoring_matrix_;   // (DIM+1)^2 column-major
};
// DistanceFactoryAngle<ALPHABET, 2> (DistanceFactoryAngle.h:55-131): the default initial distances (no -a): cosine of the 2-mer
// count vectors, turned into a distance
DistanceMatrix angleDistances(const Alphabet &a, const std::map<std::string, sequence_t> &sequences, const std::vector<std::string> &order);
class DistanceFactoryPrealigned : public DistanceFactoryML {   // DistanceFactoryPrealigned.h:34-90
public:
    DistanceFactoryPrealigned(const Alphabet &a, const ModelFactory *mf) : DistanceFactoryML(a, mf) {}
    DistanceMatrix computePwDistances(const std::map<std::string, sequence_t> &aligned,
                                      const std::vector<std::string> &order);
};
PhyTree *buildNJTree(std::vector<std::string> seqs_order, DistanceMatrix dist);   // TreeNJ.cpp:132-281 (no topology plan)
// TreeNJ.h:27-59: distances from an all-pairs alignment (-a, prealigned == false) or induced by an existing alignment
// (prealigned == true, the guide-tree re-estimation of main.cpp:404-430)
PhyTree *TreeNJ(const Alphabet &a, const std::map<std::string, sequence_t> &seqs, const ModelFactory *mf, bool prealigned = false);

std::string data_dir();   // directory holding wag.qmat etc.

}  // namespace pgm
#endif
