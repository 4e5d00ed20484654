// main.cpp — command-line driver mirroring the reference's main()/doAlign (src/main.cpp:32-483) for
// the flags the hot path's configurations use.  TCLAP is not available, so a minimal parser accepts
// the same spellings.  Built twice: `pgmsa` (HIP backend, the product) and, for tests only,
// `oracle/_build/pgmsa_oracle` (CPU oracle backend).
#include "pgm_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <malloc.h>
#include <unistd.h>
#include <string>
#include <thread>

using namespace pgm;

static void usage() {
    std::cerr << "USAGE: pgmsa [-f|--fasta] [-t|--tree <newick>] [-o <file>] [-T] [-I] [-a] [-m] [-M]\n"
                 "             [--codon] [-c|--cs_profile <lib>] [-i <iters>] [-g rate] [-e prob] [-E prob]\n"
                 "             [-s prob] [-A] [--early_refinement] [--ancestral_seqs] [--profile_out <file>] [-R] [--read_repeats <file>]\n"
                 "             [--dump_jobs <file>] [--dump_dist <file>] [--stats] <fasta file>\n";
}

// The backend (device contexts: the HIP runtime's start-up takes 80-400 ms) is created on a thread of its own while the
// sequences are read and the models are set up; doAlign waits for it before the clocks of the stages start.
namespace {
struct BackendStartup {
    std::thread thr;
    double seconds = 0;
    std::string err;
    void start() {
        thr = std::thread([this]() {
            const auto t0 = std::chrono::steady_clock::now();
            try { default_backend(); }
            catch (std::exception &e) { err = e.what(); }
            seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    void wait() {
        if (thr.joinable()) thr.join();
        if (!err.empty()) throw pgm_exception(err);
    }
    ~BackendStartup() { if (thr.joinable()) thr.join(); }
} g_startup;
}  // namespace

static std::string value_name(const Alphabet &a, int j) {   // ALPHABET(j).asString()
    static const char *aa = "ACDEFGHIKLMNPQRSTVWY";
    if (a.kind == ALPHA_AA) return std::string(1, aa[j]);
    static const char nt[] = "TCAG";
    int k = -1;
    for (int c = 0; c < 64; ++c) {
        const std::string cod = {nt[c >> 4], nt[(c >> 2) & 3], nt[c & 3]};
        if (cod == "TAA" || cod == "TAG" || cod == "TGA") continue;
        if (++k == j) return cod;
    }
    return "?";
}

static int doAlign(const Alphabet &a, const std::map<std::string, std::string> &seqs,
                   std::map<std::string, std::string> &out_aligned, PhyTree *&out_tree, bool stats) {
    // strip start/stop (main.cpp:332-353)
    bool any_start = false, any_end = false;
    std::map<std::string, bool> startStripped, endStripped;
    std::map<std::string, sequence_t> seqs2;
    for (const auto &kv : seqs) {
        sequence_t seq = sequenceFromString(a, kv.second);
        if (!cmdlineopts.noforcealign_flag) {
            if (!seq.empty() && a.stripsStart(seq[0])) { seq = seq.substr(1); any_start = true; startStripped[kv.first] = true; }
            else startStripped[kv.first] = false;
            if (!seq.empty() && a.stripsEnd(seq[seq.size() - 1])) { seq = seq.substr(0, seq.size() - 1); any_end = true; endStripped[kv.first] = true; }
            else endStripped[kv.first] = false;
        }
        seqs2[kv.first] = seq;
    }
    std::unique_ptr<ModelFactory> model_factory(ModelFactory::getDefault(a));
    std::unique_ptr<CSProfile> csprofile;
    if (!cmdlineopts.cs_file.empty()) csprofile.reset(new CSProfile(cmdlineopts.cs_file));

    // the device contexts (HIP runtime start-up, code object load) exist before the clocks of the stages start; init_s is the
    // time their creation took (it ran beside the set-up above), init_wait_s what of it was left to wait for here
    auto t0 = std::chrono::steady_clock::now();
    g_startup.wait();
    default_backend();
    double t_preload = 0;
    if (csprofile) {   // the profile library goes to the devices with the rest of the start-up (counted in init_s)
        const auto tp = std::chrono::steady_clock::now();
        default_backend().csprofile_preload(*csprofile);
        t_preload = std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
    }
    const double t_init = (g_startup.seconds > 0 ? g_startup.seconds : std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()) + t_preload;
    if (host_switches().profile)
        fprintf(stderr, "backend start-up %.1f ms, of which %.1f ms waited for after the set-up\n", t_init * 1e3,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    std::map<std::string, std::vector<repeat_t>> reps;   // main.cpp:367-370 (detection by T-REKS itself is not built here: --read_repeats only)
    if (!cmdlineopts.readreps_file.empty()) reps = read_repeats(a, cmdlineopts.readreps_file, seqs2);
    PhyTree *tree = nullptr;
    t0 = std::chrono::steady_clock::now();
    if (!cmdlineopts.tree_file.empty()) {
        std::ifstream ts(cmdlineopts.tree_file.c_str());
        if (!ts) error("cannot open tree file %s", cmdlineopts.tree_file.c_str());
        tree = parse_newick(ts);
    } else {
        tree = TreeNJ(a, seqs2, model_factory.get());
    }
    double t_tree = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ProgressiveAlignmentResult result, old_result;
    t0 = std::chrono::steady_clock::now();
    // further rounds of alignment followed by estimation of an improved tree from the induced pairwise distances
    // (main.cpp:404-430; the default is two such rounds when no tree is given)
    for (int i = 0; i < cmdlineopts.iters; ++i) {
        result = progressive_alignment(a, seqs2, *tree, csprofile.get(), *model_factory, &reps);
        for (auto it = result.aligned_sequences.begin(); it != result.aligned_sequences.end();)   // ancestral sequences
            if (!it->first.empty() && it->first[0] == '(') it = result.aligned_sequences.erase(it); else ++it;
        if (i > 0 && result.aligned_sequences == old_result.aligned_sequences) break;   // converged
        delete tree;
        const auto tt0 = std::chrono::steady_clock::now();
        tree = TreeNJ(a, result.aligned_sequences, model_factory.get(), true);
        if (host_switches().profile) fprintf(stderr, "guide tree from the alignment: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count());
        old_result = result;
    }
    if (!cmdlineopts.onlytree_flag) result = progressive_alignment(a, seqs2, *tree, csprofile.get(), *model_factory, &reps);
    if (host_switches().profile) fprintf(stderr, "[%.1f ms] back in main\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    double t_prog = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out_tree = tree;

    // re-insert start/stop (main.cpp:459-482)
    for (auto &kv : result.aligned_sequences) {
        sequence_t aseq = kv.second;
        if (any_start) aseq.insert(aseq.begin(), startStripped[kv.first] ? a.unknown() : a.gap());
        if (any_end) aseq.insert(aseq.end(), endStripped[kv.first] ? a.unknown() : a.gap());
        out_aligned[kv.first] = seqs.count(kv.first) ? stringFromSequence(a, aseq, seqs.at(kv.first)) : stringFromSequence(a, aseq);   // (ancestral rows have no original)
    }
    if (cmdlineopts.repeats_flag) {
        if (cmdlineopts.readreps_file.empty()) error("-R: tandem-repeat detection (T-REKS, a Java program) is not built here; supply the repeats with --read_repeats");
        std::cerr << "TR indels: " << result.n_tr_indels << std::endl;   // main.cpp:447-449
    }
    if (!cmdlineopts.profile_file.empty()) {   // write_profile (profile.h:12-31; main.cpp:451-456): default stream formatting, 6 significant digits
        std::ofstream pf(cmdlineopts.profile_file.c_str());
        for (const auto &kv : result.profiles) {
            pf << '>' << kv.first << std::endl;
            for (int j = 0; j < a.DIM; ++j) {
                pf << value_name(a, j);
                for (index_t k = 0; k < kv.second.cols; ++k) pf << '\t' << kv.second.data[(size_t)j + (size_t)a.DIM * k];
                pf << std::endl;
            }
        }
    }
    if (stats) {
        Backend &be = default_backend();
        fprintf(stderr,
                "{\"backend\": \"%s\", \"init_s\": %.6f, \"tree_s\": %.6f, \"progressive_s\": %.6f, \"align_cells\": %llu, \"align_s\": %.6f, "
                "\"nw_cells\": %llu, \"nw_s\": %.6f, \"mldist_s\": %.6f, \"merge_profiles_s\": %.6f, \"farm_workers\": %d, \"farm_tiles\": %d, \"farm_level_workers\": %d, \"farm_leaf_workers\": %d, \"resident\": %s, \"resident_imports\": %d, \"switches\": \"%s\"}\n",
                be.name(), t_init, t_tree, t_prog, (unsigned long long)be.cells_aligned, be.seconds_align,
                (unsigned long long)be.cells_nw, be.seconds_nw, be.seconds_mldist, be.seconds_merge_profiles, be.farm_workers, be.farm_tiles, be.farm_level_workers, be.farm_leaf_workers, be.resident_pass ? "true" : "false", be.resident_imports, host_switches().describe().c_str());
    }
    return 0;
}

int main(int argc, char **argv) {
    const auto t_main = std::chrono::steady_clock::now();
    // The graphs of a level are hundreds of vectors of 0.1-1 MB built and dropped by the host threads: with glibc's defaults
    // each is an mmap / munmap of its own (page faults on every reuse, the address-space lock shared by all threads), and
    // the heap is trimmed back to the system whenever its top is freed.  Keep them in the heaps instead.
    if (!getenv("PGM_MALLOC_DEFAULTS")) {
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
        mallopt(M_TOP_PAD, 64 << 20);
    }
    try {
        bool iters_set = false, stats = false, indel_set = false, edgehl_set = false, maxdist_set = false, cutdist_set = false;
        std::string dump, dist_dump;
        for (int i = 1; i < argc; ++i) {
            std::string s = argv[i];
            auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); exit(1); } return argv[++i]; };
            if (s == "-f" || s == "--fasta") cmdlineopts.fasta_flag = true;
            else if (s == "-t" || s == "--tree") cmdlineopts.tree_file = val();
            else if (s == "-o" || s == "--output") cmdlineopts.output_file = val();
            else if (s == "-T" || s == "--only_tree") cmdlineopts.onlytree_flag = true;
            else if (s == "-I" || s == "--input_order") cmdlineopts.inputorder_flag = true;
            else if (s == "-a" || s == "--nwdist") cmdlineopts.nwdist_flag = true;
            else if (s == "-m" || s == "--mldist") cmdlineopts.mldist_flag = true;
            else if (s == "-M" || s == "--mldist_gap") cmdlineopts.mldist_gap_flag = true;
            else if (s == "-A" || s == "--no_force_align") cmdlineopts.noforcealign_flag = true;
            else if (s == "--codon") cmdlineopts.codon_flag = true;
            else if (s == "-c" || s == "--cs_profile") cmdlineopts.cs_file = val();
            else if (s == "-i" || s == "--iterations") { cmdlineopts.iters = atoi(val().c_str()); iters_set = true; }
            else if (s == "-g" || s == "--indel_rate") { cmdlineopts.indel_rate = atof(val().c_str()); indel_set = true; }
            else if (s == "-e" || s == "--gap_ext") cmdlineopts.gapext_prob = atof(val().c_str());
            else if (s == "-E" || s == "--end_indel_prob") cmdlineopts.end_indel_prob = atof(val().c_str());
            else if (s == "-s" || s == "--altsplice_prob") cmdlineopts.altsplice_prob = atof(val().c_str());
            else if (s == "-l" || s == "--edge_halflife") { cmdlineopts.edge_halflife = atof(val().c_str()); edgehl_set = true; }
            else if (s == "-x" || s == "--cutoff_dist") { cmdlineopts.cutoff_dist = atof(val().c_str()); cutdist_set = true; }
            else if (s == "-d" || s == "--min_dist") cmdlineopts.min_dist = atof(val().c_str());
            else if (s == "-D" || s == "--max_dist") { cmdlineopts.max_dist = atof(val().c_str()); maxdist_set = true; }
            else if (s == "-p" || s == "--min_pdist") cmdlineopts.min_pdist = atof(val().c_str());
            else if (s == "-P" || s == "--max_pdist") cmdlineopts.max_pdist = atof(val().c_str());
            else if (s == "--ancestral_seqs") cmdlineopts.ancestral_flag = true;
            else if (s == "--early_refinement") cmdlineopts.earlyref_flag = true;
            else if (s == "--profile_out") cmdlineopts.profile_file = val();
            else if (s == "--read_repeats") cmdlineopts.readreps_file = val();
            else if (s == "-R" || s == "--repeats") cmdlineopts.repeats_flag = true;
            else if (s == "--dump_jobs") dump = val();
            else if (s == "--dump_dist") dist_dump = val();
            else if (s == "--stats") stats = true;
            else if (s == "-h" || s == "--help") { usage(); return 0; }
            else if (!s.empty() && s[0] == '-') { std::cerr << "Command line error: unknown flag " << s << std::endl; return 1; }
            else cmdlineopts.sequence_file = s;
        }
        if (cmdlineopts.sequence_file.empty()) { usage(); return 1; }
        if (cmdlineopts.codon_flag) {  // main.cpp:225-241
            if (!indel_set) cmdlineopts.indel_rate /= 2.6;
            if (!edgehl_set) cmdlineopts.edge_halflife *= 2.6;
            if (!maxdist_set) cmdlineopts.max_dist = 5.0;
            if (!cutdist_set) cmdlineopts.cutoff_dist = 5.0;
        }
        // main.cpp:243-246; this build also cannot iterate (see doAlign), so -a/-T runs behave as `-i 0`
        if (!iters_set && !cmdlineopts.tree_file.empty()) cmdlineopts.iters = 0;   // do not iterate when a guide tree is provided (main.cpp:243-246)
        if (!dump.empty()) set_job_dump(dump);
        if (!dist_dump.empty()) set_dist_dump(dist_dump);

        g_startup.start();
        parallel_for(64, [](size_t) {});   // (the driver's host threads start while the device runtime does)
        std::vector<std::string> input_order;
        std::map<std::string, std::string> seqs = read_fasta(cmdlineopts.sequence_file, input_order);
        std::ofstream custom_out;
        std::ostream *out = &std::cout;
        if (!cmdlineopts.output_file.empty()) {
            custom_out.open(cmdlineopts.output_file.c_str());
            if (!custom_out) error("error opening output file");
            out = &custom_out;
        }
        std::map<std::string, std::string> aligned;
        PhyTree *tree = nullptr;
        Alphabet a(cmdlineopts.codon_flag ? ALPHA_CODON : ALPHA_AA);
        doAlign(a, seqs, aligned, tree, stats);
        if (!cmdlineopts.onlytree_flag) {
            std::vector<std::string> order = input_order;
            if (!cmdlineopts.inputorder_flag) order = get_tree_order(tree);
            if (!cmdlineopts.fasta_flag) std::cerr << "note: Stockholm output is not built here; writing FASTA" << std::endl;
            write_fasta(aligned, order, *out);
        } else {
            *out << tree->formatNewick() << std::endl;
        }
        delete tree;
        if (host_switches().profile)
            fprintf(stderr, "main: output written %.1f ms after its start\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_main).count());
        // The output is complete.  Releasing the contexts' gigabytes of device memory and pinned blocks and shutting the HIP runtime
        // down in an orderly way takes 30-70 ms that the kernel driver spends anyway when the process is gone: leave directly
        // (PGM_FULL_EXIT=1 for runs under a profiler or a sanitizer, whose reports are written by exit handlers).
        if (!getenv("PGM_FULL_EXIT") && std::string(default_backend().name()) == "hip") {
            custom_out.close();
            std::cout.flush(); std::cerr.flush(); fflush(nullptr);
            _exit(0);
        }
    } catch (std::exception &e) {
        std::cerr << "ERROR:" << e.what() << std::endl;
        return 2;
    }
    return 0;
}
