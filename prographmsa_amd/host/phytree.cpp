// phytree.cpp — guide-tree container, newick reader/writer and FASTA IO
// (reference src/PhyTree.{h,cpp}, src/newick.cpp, src/Fasta.cpp).  Host scaffolding only.
// Midpoint rooting reproduces the behaviour of reference src/PhyTree.cpp:11-116 (same longest pair among ties, same child order of
// the re-rooted tree: the newick must come out byte for byte) in this mirror's own terms; host scaffolding outside SURVEY §8.
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

// ---- midpoint rooting (behaviour of PhyTree.cpp:11-116: the newick of the re-rooted tree must come out byte for byte) ----
// Own formulation.  The leaves are numbered in depth-first order, so the leaves of a subtree are one contiguous range of the
// array `depth` (distance of the leaf to the root of the subtree being finished); an internal node adds its two branch lengths to
// its children's ranges and compares every leaf of the first child with every leaf of the second, first child's leaf outermost —
// the same visiting order, hence the same winner among equal path lengths, as the reference's recursion.  The winner of a pair is
// its deeper leaf: the walk to the midpoint starts there.
namespace {
struct LeafSpan { size_t lo, hi; };

struct Diameter {
    std::vector<double> depth;
    std::vector<const PhyTree *> leaf;
    double longest = -INFINITY;
    const PhyTree *deep_end = nullptr;

    LeafSpan finish(const PhyTree *node) {
        if (node->isLeaf()) {
            depth.push_back(0.0);
            leaf.push_back(node);
            return LeafSpan{depth.size() - 1, depth.size()};
        }
        if (node->n_children() != 2) error("multifurcations not supported");
        const LeafSpan first = finish(&(*node)[0]), second = finish(&(*node)[1]);
        const double len0 = (*node)[0].getBranchLength(), len1 = (*node)[1].getBranchLength();
        for (size_t k = first.lo; k < first.hi; ++k) depth[k] += len0;
        for (size_t k = second.lo; k < second.hi; ++k) depth[k] += len1;
        for (size_t p = first.lo; p < first.hi; ++p)
            for (size_t q = second.lo; q < second.hi; ++q) {
                const double path = depth[p] + depth[q];
                if (path > longest) { longest = path; deep_end = depth[p] > depth[q] ? leaf[p] : leaf[q]; }
            }
        return LeafSpan{first.lo, second.hi};
    }
};
}  // namespace

PhyTree *midpointRoot(PhyTree *root) {
    Diameter dia;
    dia.finish(root);
    // climb from the deeper end of the longest path until half of it is used up: the midpoint lies on the branch above `below`
    PhyTree *below = const_cast<PhyTree *>(dia.deep_end);
    double remaining = dia.longest / 2;
    while (below != root && remaining - below->getBranchLength() > 0) {
        remaining -= below->getBranchLength();
        below = below->getParent();
    }
    if (below == root) return root;

    // the nodes from the midpoint branch's upper end up to (not including) the old root, with the branch each hangs on
    struct Hop { PhyTree *node; double length, support; };
    std::vector<Hop> climb;
    for (PhyTree *n = below->getParent(); n != root; n = n->getParent()) climb.push_back(Hop{n, n->getBranchLength(), n->getBranchSupport()});

    PhyTree *fresh = new PhyTree("new_root");
    const double upper_part = below->getBranchLength() - remaining, upper_support = below->getBranchSupport();
    below->pluck();
    fresh->addChild(below, remaining, upper_support);
    // turn the climb upside down: every node on it becomes the last child of the node that was below it, on the branch that node
    // used to hang on
    PhyTree *attach_to = fresh;
    double hang_length = upper_part, hang_support = upper_support;
    for (const Hop &h : climb) {
        h.node->pluck();
        attach_to->addChild(h.node, hang_length, hang_support);
        attach_to = h.node;
        hang_length = h.length;
        hang_support = h.support;
    }
    // the old root disappears: its remaining child continues the last branch
    const PhyTree &sibling = (*root)[0];   // (the climb's top node, or `below` itself, has been plucked: one child is left)
    const double merged_length = hang_length + sibling.getBranchLength();
    const double merged_support = std::max(hang_support, sibling.getBranchSupport());
    attach_to->addChild(root->pluckChild(0), merged_length, merged_support);
    delete root;
    return fresh;
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
