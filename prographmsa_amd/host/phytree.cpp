// phytree.cpp — guide-tree container, newick reader/writer and FASTA IO
// (reference src/PhyTree.{h,cpp}, src/newick.cpp, src/Fasta.cpp).  Host scaffolding only.
// NOTE: maxDistPairR / midpointRoot below are DERIVED FROM reference src/PhyTree.cpp:11-116 (statement-for-statement restatement,
// identifiers kept): host scaffolding outside SURVEY §8, needed only so that a FASTA / newick artefact can be produced
// for the parity tests; it earns no coverage credit.
#include "pgm_host.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <istream>
#include <sstream>

namespace pgm {

PhyTree::~PhyTree() {
    for (PhyTree *c : children_) { c->parent_ = nullptr; delete c; }
}

PhyTree *PhyTree::copy() const {
    PhyTree *out = new PhyTree(name_);
    out->branch_length_ = branch_length_;
    out->branch_support_ = branch_support_;
    for (PhyTree *c : children_) out->addChild(c->copy(), c->branch_length_, c->branch_support_);
    return out;
}

void PhyTree::addChild(PhyTree *child, double branch_length, double branch_support) {
    children_.push_back(child);
    child->parent_ = this;
    child->branch_length_ = branch_length;
    child->branch_support_ = branch_support;
}

index_t PhyTree::indexOf() const {
    for (index_t i = 0; i < parent_->children_.size(); ++i)
        if (parent_->children_[i] == this) return i;
    return (index_t)-1;
}

void PhyTree::pluck() {  // PhyTree.h:112-123
    index_t index = indexOf();
    parent_->children_.erase(parent_->children_.begin() + index);
    parent_ = nullptr;
    branch_length_ = 0;
    branch_support_ = 1;
}

PhyTree *PhyTree::pluckChild(index_t index) {  // PhyTree.h:125-135
    PhyTree *child = children_[index];
    children_.erase(children_.begin() + index);
    child->parent_ = nullptr;
    child->branch_length_ = 0;
    branch_support_ = 1;
    return child;
}

std::string PhyTree::formatNewickR() const {  // PhyTree.h:41-55 (default ostream precision: 6 s.f.)
    if (isLeaf()) return name_;
    std::stringstream ss;
    ss << "(";
    for (size_t i = 0; i < children_.size(); ++i) {
        if (i) ss << ",";
        ss << children_[i]->formatNewickR() << ":" << children_[i]->branch_length_;
    }
    ss << ")";
    return ss.str();
}
std::string PhyTree::formatNewick() const { return formatNewickR() + ";"; }

// ---- midpoint rooting (PhyTree.cpp:11-116) -------------------------------------------------
static void maxDistPairR(const PhyTree *root, std::vector<double> &distances, std::vector<const PhyTree *> &leaves,
                         const PhyTree *&max, double &max_dist) {
    if (!root->isLeaf()) {
        if (root->n_children() != 2) error("multifurcations not supported");
        std::vector<double> distances2;
        std::vector<const PhyTree *> leaves2;
        maxDistPairR(&(*root)[0], distances, leaves, max, max_dist);
        maxDistPairR(&(*root)[1], distances2, leaves2, max, max_dist);
        for (double &d : distances) d += (*root)[0].getBranchLength();
        for (double &d : distances2) d += (*root)[1].getBranchLength();
        for (size_t i = 0; i < distances.size(); ++i)
            for (size_t j = 0; j < distances2.size(); ++j)
                if (distances[i] + distances2[j] > max_dist) {
                    max_dist = distances[i] + distances2[j];
                    max = distances[i] > distances2[j] ? leaves[i] : leaves2[j];
                }
        distances.insert(distances.end(), distances2.begin(), distances2.end());
        leaves.insert(leaves.end(), leaves2.begin(), leaves2.end());
    } else {
        distances.push_back(0);
        leaves.push_back(root);
    }
}

PhyTree *midpointRoot(PhyTree *root) {
    const PhyTree *max = nullptr;
    std::vector<double> distances;
    std::vector<const PhyTree *> leaves;
    double dist = -INFINITY;
    maxDistPairR(root, distances, leaves, max, dist);
    PhyTree *current = const_cast<PhyTree *>(max);
    dist /= 2;
    while (current != root && dist - current->getBranchLength() > 0) {
        dist -= current->getBranchLength();
        current = current->getParent();
    }
    if (current == root) return root;

    PhyTree *new_root = new PhyTree("new_root");
    double current_dist = current->getBranchLength() - dist;
    double current_support = current->getBranchSupport();
    PhyTree *parent = current->getParent();
    current->pluck();
    new_root->addChild(current, dist, current_support);
    current = new_root;
    while (parent != root) {
        double new_dist = parent->getBranchLength();
        double new_support = parent->getBranchSupport();
        PhyTree *new_parent = parent->getParent();
        parent->pluck();
        current->addChild(parent, current_dist, current_support);
        current = parent;
        parent = new_parent;
        current_dist = new_dist;
        current_support = new_support;
    }
    current_dist += (*root)[0].getBranchLength();
    current_support = std::max(current_support, (*root)[0].getBranchSupport());
    PhyTree *other = root->pluckChild(0);
    current->addChild(other, current_dist, current_support);
    delete root;
    return new_root;
}

static void get_tree_order_rec(const PhyTree *tree, std::vector<std::string> &order) {
    if (tree->isLeaf()) order.push_back(tree->getName());
    else
        for (index_t i = 0; i < tree->n_children(); ++i) get_tree_order_rec(&(*tree)[i], order);
}
// (--ancestral_seqs: every internal node's name, the sorted list of its leaves in parentheses, stands between the orders of
//  its two subtrees: PhyTree.cpp:118-160)
static std::string list_to_name(const std::vector<std::string> &leaves) {
    std::vector<std::string> sorted = leaves;
    std::sort(sorted.begin(), sorted.end());
    std::string s = "(";
    bool first = true;
    for (const std::string &name : sorted)
        if (name[0] != '(') { if (!first) s += ","; first = false; s += name; }
    return s + ")";
}
static std::vector<std::string> get_tree_order_ancestral(const PhyTree *tree) {
    std::vector<std::string> order;
    if (tree->isLeaf()) { order.push_back(tree->getName()); return order; }
    for (index_t i = 0; i < tree->n_children(); ++i) {
        const std::vector<std::string> sub = get_tree_order_ancestral(&(*tree)[i]);
        const size_t pos = order.size();
        order.insert(order.end(), sub.begin(), sub.end());
        if (i != 0) order.insert(order.begin() + pos, list_to_name(order));
    }
    return order;
}
std::vector<std::string> get_tree_order(const PhyTree *tree) {
    if (cmdlineopts.ancestral_flag) return get_tree_order_ancestral(tree);
    std::vector<std::string> order;
    get_tree_order_rec(tree, order);
    return order;
}

// ---- newick (newick.cpp) ----------------------------------------------------------------------
namespace {
struct TokenBuffer {
    std::istream &in;
    std::string last;
    bool valid = false;
    explicit TokenBuffer(std::istream &i) : in(i) {}
    static bool issep(int c) { return c == ',' || c == ':' || c == '(' || c == ')' || c == ';'; }
    std::string next_token() {
        std::string token;
        while (in) {
            int c = in.get();
            if (c == EOF) break;
            if (isspace(c)) continue;
            if (issep(c)) return token + (char)c;
            token += (char)c;
            while (in && in.peek() != EOF && !isspace(in.peek()) && !issep(in.peek())) token += (char)in.get();
            return token;
        }
        throw pgm_exception("Unexpected EOF or I/O error");
    }
    std::string peek() { if (!valid) last = next_token(); valid = true; return last; }
    std::string next() { if (!valid) last = next_token(); valid = false; return last; }
};
double parse_double(const std::string &s) { std::istringstream ss(s); double out = 0; ss >> out; return out; }

PhyTree *parse_tree(TokenBuffer &buffer) {
    PhyTree *t = new PhyTree();
    std::string tok = buffer.next();
    if (tok != "(") throw pgm_exception("Unexpected token: '" + tok + "', expected: '('");
    do {
        PhyTree *child;
        tok = buffer.peek();
        if (tok == "(") child = parse_tree(buffer);
        else child = new PhyTree(buffer.next());
        tok = buffer.next();
        if (tok != ":") {  // branch support is ignored (newick.cpp:97-98) -> support 1
            tok = buffer.next();
            if (tok != ":") throw pgm_exception("Unexpected token: '" + tok + "', expected: ':'");
        }
        tok = buffer.next();
        t->addChild(child, parse_double(tok));
        tok = buffer.peek();
        if (tok == ")") { buffer.next(); break; }
        tok = buffer.next();
        if (tok != ",") throw pgm_exception("Unexpected token: '" + tok + "', expected: ','");
    } while (true);
    return t;
}
}  // namespace

PhyTree *parse_newick(std::istream &in) {
    TokenBuffer buffer(in);
    PhyTree *t = parse_tree(buffer);
    std::string tok = buffer.next();
    if (tok != ";") {
        if (tok != ":") tok = buffer.next();
        if (tok != ":") throw pgm_exception("Unexpected token: " + tok);
        buffer.next();
        tok = buffer.next();
    }
    if (tok != ";") throw pgm_exception("Unexpected token: " + tok);
    return t;
}

// ---- FASTA (Fasta.cpp:38-126) --------------------------------------------------------------------
static std::string strip(const std::string &s) {
    size_t start = s.find_first_not_of(" \t\f\v\n\r");
    if (start == std::string::npos) return "";
    size_t end = s.find_last_not_of(" \t\f\v\n\r");
    return s.substr(start, end - start + 1);
}

std::map<std::string, std::string> read_fasta(const std::string &file, std::vector<std::string> &order) {
    std::ifstream fp(file.c_str());
    if (!fp) throw pgm_exception("error opening file");
    if (fp.peek() != '>') throw pgm_exception("format error");
    std::map<std::string, std::string> result;
    while (fp && fp.peek() != EOF) {
        fp.ignore();
        std::string name, seq, line;
        std::getline(fp, name);
        name = strip(name);
        while (fp && fp.peek() != '>' && fp.peek() != EOF) {
            std::getline(fp, line);
            seq += strip(line);
        }
        if (result.count(name)) error("duplicate sequence name \"%s\"", name.c_str());
        result[name] = seq;
        order.push_back(name);
    }
    return result;
}

void write_fasta(const std::map<std::string, std::string> &aln, const std::vector<std::string> &order, std::ostream &out) {
    for (const std::string &n : order) out << ">" << n << "\n" << aln.at(n) << "\n";
}

}  // namespace pgm
