// csprofile.cpp — loader for context-specific profile libraries (reference src/CSProfile.cpp:29-170;
// file format in SURVEY Appendix D).  createProfile itself (CSProfile.cpp:175-225) runs behind the
// C ABI (pgm_csprofile_load / pgm_csprofile_create_batch).
#include "pgm_host.h"

#include <cmath>
#include <fstream>
#include <sstream>

namespace pgm {

static const double w_center = .26236426446749105203; /* log(1.3) */
static const double beta = -.10536051565782630122;    /* log(.9) */
static const double log_2 = .69314718055994530941;

CSProfile::CSProfile(const std::string &filename) {
    const Alphabet aa(ALPHA_AA);
    std::ifstream file(filename.c_str());
    std::string line;
    if (!std::getline(file, line) || line.find("ProfileLibrary") != 0) throw pgm_exception("error opening profile library");
    while (std::getline(file, line)) {  // header
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line);
        std::string key;
        if (line.find("NPROF") == 0) { is >> key >> nprof_; if (!is || nprof_ <= 0) throw pgm_exception("parse error: " + line); }
        else if (line.find("NCOLS") == 0) { is >> key >> ncols_; if (!is || ncols_ <= 0) throw pgm_exception("parse error: " + line); }
        else if (line.find("ITERS") == 0 || line.find("LOG") == 0) {}
        else if (line.find("ContextProfile") == 0) break;
        else throw pgm_exception("parse error: " + line);
    }
    if (nprof_ <= 0 || ncols_ <= 0) throw pgm_exception("missing information in header");
    const int K = nprof_, C = ncols_, center = C / 2;
    lprofiles_.assign((size_t)K * C * 21, 0.0);
    centre_.assign((size_t)K * 20, 0.0);
    priors_.assign(K, 0.0);
    std::vector<double> weights(C);
    for (int j = -center; j <= center; ++j) weights[center + j] = std::exp(w_center + beta * std::abs(j));

    do {
        if (line.empty() || line[0] == '#') continue;
        if (line.find("ContextProfile") != 0) throw pgm_exception("parse error: " + line);
        int index = -1;
        double prior = -1;
        std::vector<double> profile((size_t)C * 20, 0.0);  // [col][a]
        bool finished = false;
        while (!finished && std::getline(file, line)) {
            if (line.empty() || line[0] == '#' || line.find("ITERS") == 0 || line.find("LOG") == 0) continue;
            std::istringstream is(line);
            std::string key;
            if (line.find("INDEX") == 0) { is >> key >> index; if (!is || index < 0 || index >= K) throw pgm_exception("parse error: " + line); }
            else if (line.find("PRIOR") == 0) { is >> key >> prior; if (!is || prior <= 0) throw pgm_exception("parse error: " + line); }
            else if (line.find("NCOLS") == 0) { int pc; is >> key >> pc; if (!is || pc != C) throw pgm_exception("parse error: " + line); }
            else if (line.find("ALPH") == 0) { int pa; is >> key >> pa; if (!is || pa != 20) throw pgm_exception("parse error: " + line); }
            else if (std::isspace((unsigned char)line[0])) {
                std::vector<int> cols(20);
                for (int i = 0; i < 20; ++i) {
                    char s = 0;
                    is >> s;
                    cols[i] = aa.value((int8_t)s);
                    if (!is || cols[i] < 0 || cols[i] >= 20) throw pgm_exception("parse error in column names");
                }
                while (std::getline(file, line)) {
                    if (line == "//") { finished = true; break; }
                    std::istringstream row(line);
                    int col = 0;
                    row >> col;
                    if (!row || col <= 0 || col > C) throw pgm_exception("parse error: invalid column number");
                    for (int i = 0; i < 20; ++i) {
                        double v;
                        row >> v;
                        if (!row || v < 0) throw pgm_exception("parse error in profile");
                        profile[(size_t)(col - 1) * 20 + cols[i]] = v;
                    }
                }
            } else throw pgm_exception("parse error: " + line);
        }
        if (index < 0 || index >= K) throw pgm_exception("parse error: invalid index");
        if (prior <= 0) throw pgm_exception("parse error: invalid prior");
        // CSProfile.cpp:157-165
        for (int c = 0; c < C; ++c) {
            double p[20], sum = 0;
            for (int a = 0; a < 20; ++a) { p[a] = std::exp(profile[(size_t)c * 20 + a] * (-log_2 / 1000.0)); }
            for (int a = 0; a < 20; ++a) sum += p[a];
            double lsum = std::log(sum);
            double *lp = &lprofiles_[((size_t)index * C + c) * 21];
            for (int a = 0; a < 20; ++a) {
                double l = std::log(p[a]) - lsum;
                if (c == center) centre_[(size_t)index * 20 + a] = std::exp(l);
                lp[a] = weights[c] * l;
            }
            lp[20] = weights[c] * 0.0;
        }
        priors_[index] = std::log(prior);
    } while (std::getline(file, line));
}

}  // namespace pgm
