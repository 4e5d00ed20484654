// alphabet.cpp — symbol encodings (reference src/Alphabet.{h,cpp}) and small utilities.
#include "pgm_host.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <unistd.h>

namespace pgm {

cmdlineopts_t cmdlineopts;

void error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw pgm_exception(buf);
}

std::string data_dir() {
    if (const char *e = getenv("PGM_DATA_DIR")) return e;
    // <repo>/prographmsa_amd/host/data relative to this shared object / executable
    Dl_info info;
    std::string base;
    if (dladdr((void *)&data_dir, &info) && info.dli_fname) base = info.dli_fname;
    char exe[4096];
    if (base.empty() || base.find('/') == std::string::npos) {
        ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (n > 0) { exe[n] = 0; base = exe; }
    }
    size_t p = base.rfind('/');
    std::string dir = (p == std::string::npos) ? "." : base.substr(0, p);
    const char *cands[] = {"/../host/data", "/data", "/../prographmsa_amd/host/data", "/../../prographmsa_amd/host/data"};
    for (const char *c : cands) {
        std::string d = dir + c;
        if (access((d + "/wag.qmat").c_str(), R_OK) == 0) return d;
    }
    return dir + "/../host/data";
}

// ---- AA ------------------------------------------------------------------------------
// value(): 20 amino acids in alphabetical one-letter order -> 0..19, other letters -> 20,
// anything else -> -1 (aa_translation_table, Alphabet.cpp:6-20).
static int aa_value(int8_t data) {
    static const char order[] = "ACDEFGHIKLMNPQRSTVWY";
    int c = (unsigned char)data;
    if (c >= 'a' && c <= 'z') c -= 32;
    if (c < 'A' || c > 'Z') return -1;
    const char *p = strchr(order, c);
    return p ? (int)(p - order) : 20;
}

// ---- Codon -----------------------------------------------------------------------------
// nucleotide code T/U=0 C=1 A=2 G=3 X=4 else -1 (dna_translation_table, Alphabet.cpp:22-36)
static int nt_value(char ch) {
    switch (ch) {
        case 'T': case 't': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'A': case 'a': return 2;
        case 'G': case 'g': return 3;
        case 'X': case 'x': return 4;
        default: return -1;
    }
}
static const int CODON_DIM = 61;
// Codon::Codon(c1,c2,c3) (Alphabet.cpp:120-155)
static int8_t codon_data(char c1, char c2, char c3) {
    auto isgap = [](char c) { return c == '_' || c == '-' || c == ' ' || c == '.'; };
    if (isgap(c1) || isgap(c2) || isgap(c3)) return CODON_DIM + 1;
    int c = nt_value(c3);
    if (c < 0) return -1;
    if (c >= 4) return CODON_DIM;
    c += 4 * nt_value(c2);
    if (c < 0) return -1;
    if (c >= 16) return CODON_DIM;
    c += 16 * nt_value(c1);
    if (c < 0) return -1;
    if (c >= 64) return CODON_DIM;
    // 64 -> 61: stop codons TAA(10) TAG(11) TGA(14) are invalid (-1)
    if (c == 10 || c == 11 || c == 14) return -1;
    return (int8_t)(c - (c > 10) - (c > 11) - (c > 14));
}
static const char *const NT = "TCAG";
static std::string codon_string(int idx) {  // codon_inv_translation_table
    int c = idx + (idx >= 10) + (idx >= 10) + (idx >= 12);  // undo the stop-codon compaction
    // idx 0..9 -> 0..9 ; 10,11 -> 12,13 ; 12.. -> 15..
    if (idx < 10) c = idx; else if (idx < 12) c = idx + 2; else c = idx + 3;
    std::string s;
    s += NT[c / 16];
    s += NT[(c / 4) % 4];
    s += NT[c % 4];
    return s;
}
static char codon_aa(int idx) {  // codon_inv3_translation_table (standard genetic code, TCAG order)
    static const char t[] = "FFLLSSSSYYCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGGX";
    return t[idx];
}

int Alphabet::value(int8_t data) const {
    if (kind == ALPHA_AA) return aa_value(data);
    if (data == CODON_DIM + 1) return -1;  // gap
    return data;
}
int8_t Alphabet::gap() const { return kind == ALPHA_AA ? (int8_t)'-' : (int8_t)(CODON_DIM + 1); }
int8_t Alphabet::unknown() const { return kind == ALPHA_AA ? (int8_t)'X' : (int8_t)CODON_DIM; }
char Alphabet::asChar(int8_t data) const {
    if (kind == ALPHA_AA) return (char)data;
    if (isGap(data)) return '-';
    if (!isValid(data)) return 'X';
    return codon_aa(data);
}
std::string Alphabet::asString(int8_t data) const {
    if (kind == ALPHA_AA) return std::string(1, (char)data);
    if (isGap(data)) return "---";
    if (!isValid(data)) return "XXX";
    return codon_string(data);
}
bool Alphabet::stripsStart(int8_t first) const {
    // AA::stripStart = 'M', Codon::stripStart = ATG (Alphabet.cpp:94-96)
    if (kind == ALPHA_AA) return first == (int8_t)'M';
    return first == codon_data('A', 'T', 'G');
}
bool Alphabet::stripsEnd(int8_t last) const {
    // AA::stripEnd = GAP (disabled), Codon::stripEnd = XXX (Alphabet.cpp:98-100)
    if (kind == ALPHA_AA) return false;
    return last == (int8_t)CODON_DIM;
}

sequence_t sequenceFromString(const Alphabet &a, const std::string &str) {
    sequence_t seq;
    if (a.kind == ALPHA_AA) {
        seq.reserve(str.size());
        for (char ch : str) {
            if (ch == '_' || ch == '-' || ch == '.' || ch == ' ') error("No support for gapped sequences (yet)");
            seq += (int8_t)ch;
        }
    } else {  // Alphabet.cpp:258-275
        seq.reserve((str.size() + 2) / 3);
        for (size_t i = 0; i + 2 < str.size(); i += 3) {
            int8_t c = codon_data(str[i], str[i + 1], str[i + 2]);
            if (c == a.gap()) error("No support for gapped sequences (yet)");
            seq += c;
        }
        if (str.size() % 3 != 0) seq += codon_data((char)-1, (char)-1, (char)-1);
    }
    return seq;
}

std::string stringFromSequence(const Alphabet &a, const sequence_t &seq) {
    std::string s;
    for (int8_t c : seq) {
        if (a.kind == ALPHA_AA) s += a.asChar(c); else s += a.asString(c);
    }
    return s;
}

std::string stringFromSequence(const Alphabet &a, const sequence_t &seq, const std::string &orig) {
    std::string s;
    s.reserve(orig.size());
    if (a.kind == ALPHA_AA) {  // Alphabet.h:148-165
        size_t j = 0;
        for (int8_t c : seq) {
            if (a.isGap(c)) s += a.asChar(c); else s += orig[j++];
        }
    } else {  // Alphabet.cpp:237-256
        size_t k = 0;
        for (int8_t c : seq) {
            if (a.isGap(c)) s += a.asString(c);
            else {
                for (size_t j = k; j < k + 3 && j < orig.size(); ++j) s += orig[j];
                k += 3;
            }
        }
    }
    return s;
}

}  // namespace pgm
