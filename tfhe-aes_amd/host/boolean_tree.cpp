#include "boolean_tree.h"

#include <algorithm>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <unordered_map>

namespace tfheaes {

struct BooleanExpr::Node {
    Kind kind;
    Operand op;
    const Node *l, *r;
    uint8_t stage;
    uint32_t id;
};

namespace {
struct NodeKey {
    uint8_t kind, op;
    const void *l, *r;
    bool operator==(const NodeKey &o) const { return kind == o.kind && op == o.op && l == o.l && r == o.r; }
};
struct NodeKeyHash {
    size_t operator()(const NodeKey &k) const {
        size_t h = (size_t)k.kind * 0x9E3779B97F4A7C15ull ^ ((size_t)k.op << 8);
        h ^= reinterpret_cast<size_t>(k.l) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= reinterpret_cast<size_t>(k.r) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        return h;
    }
};
std::mutex g_mutex;
std::deque<BooleanExpr::Node> g_nodes;  // stable addresses
std::unordered_map<NodeKey, const BooleanExpr::Node *, NodeKeyHash> g_table;
}  // namespace

const char *operand_name(Operand o) {
    static const char *names[18] = {"True",    "False", "Bit0",    "NotBit0", "Bit1",    "NotBit1", "Bit2",    "NotBit2", "Bit3",
                                    "NotBit3", "Bit4",  "NotBit4", "Bit5",    "NotBit5", "Bit6",    "NotBit6", "Bit7",    "NotBit7"};
    return names[static_cast<uint8_t>(o)];
}

BooleanExpr BooleanExpr::make(Kind k, Operand op, const Node *l, const Node *r) {
    std::lock_guard<std::mutex> lock(g_mutex);
    NodeKey key{static_cast<uint8_t>(k), static_cast<uint8_t>(op), l, r};
    auto it = g_table.find(key);
    if (it != g_table.end()) return BooleanExpr(it->second);
    uint8_t st = (k == Kind::Operand) ? 0 : static_cast<uint8_t>(std::max(l->stage, r->stage) + 1);
    g_nodes.push_back(Node{k, op, l, r, st, static_cast<uint32_t>(g_nodes.size())});
    const Node *n = &g_nodes.back();
    g_table.emplace(key, n);
    return BooleanExpr(n);
}

BooleanExpr BooleanExpr::operand(Operand o) { return make(Kind::Operand, o, nullptr, nullptr); }

std::vector<BooleanExpr> BooleanExpr::from_bool_vec(const std::vector<bool> &items) {
    std::vector<BooleanExpr> out;
    out.reserve(items.size());
    for (bool b : items) out.push_back(from_bool(b));
    return out;
}

BooleanExpr::Kind BooleanExpr::kind() const { return node_->kind; }
Operand BooleanExpr::op() const { return node_->op; }
BooleanExpr BooleanExpr::lhs() const { return BooleanExpr(node_->l); }
BooleanExpr BooleanExpr::rhs() const { return BooleanExpr(node_->r); }
uint8_t BooleanExpr::stage() const { return node_->stage; }
uint32_t BooleanExpr::id() const { return node_->id; }

// The reference's Ord (boolean_tree.rs:537-562): discriminant first, then fields left to right.
int BooleanExpr::compare(const BooleanExpr &a, const BooleanExpr &b) {
    if (a.node_ == b.node_) return 0;
    const Node *x = a.node_, *y = b.node_;
    if (x->kind != y->kind) return static_cast<int>(x->kind) < static_cast<int>(y->kind) ? -1 : 1;
    if (x->kind == Kind::Operand) return static_cast<int>(x->op) < static_cast<int>(y->op) ? -1 : 1;
    if (x->kind == Kind::Mux && x->op != y->op) return static_cast<int>(x->op) < static_cast<int>(y->op) ? -1 : 1;
    int c = compare(BooleanExpr(x->l), BooleanExpr(y->l));
    if (c) return c;
    return compare(BooleanExpr(x->r), BooleanExpr(y->r));
}

BooleanExpr BooleanExpr::ordered_and(const BooleanExpr &a, const BooleanExpr &b) {
    return a < b ? make(Kind::And, Operand::True, a.node_, b.node_) : make(Kind::And, Operand::True, b.node_, a.node_);
}
BooleanExpr BooleanExpr::ordered_or(const BooleanExpr &a, const BooleanExpr &b) {
    return a < b ? make(Kind::Or, Operand::True, a.node_, b.node_) : make(Kind::Or, Operand::True, b.node_, a.node_);
}
BooleanExpr BooleanExpr::ordered_xor(const BooleanExpr &a, const BooleanExpr &b) {
    return a < b ? make(Kind::Xor, Operand::True, a.node_, b.node_) : make(Kind::Xor, Operand::True, b.node_, a.node_);
}

// Negation is pushed to the leaves (boolean_tree.rs:130-144).
BooleanExpr BooleanExpr::operator!() const {
    switch (node_->kind) {
    case Kind::Operand: return operand(!node_->op);
    case Kind::And: return ordered_or(!lhs(), !rhs());
    case Kind::Or: return ordered_and(!lhs(), !rhs());
    case Kind::Xor: return ordered_xor(!lhs(), rhs());
    default: {
        BooleanExpr nl = !lhs(), nr = !rhs();
        return make(Kind::Mux, node_->op, nl.node_, nr.node_);
    }
    }
}

// Mux simplification rules (boolean_tree.rs:148-185); mux(sel, l, r) = if sel { l } else { r }.
BooleanExpr BooleanExpr::mux_left_true(Operand sel, const BooleanExpr &right) {
    if (right.kind() == Kind::Operand && right.op() == Operand::True) return operand(Operand::True);
    if (right.kind() == Kind::Operand && right.op() == Operand::False) return operand(sel);
    return ordered_or(operand(sel), right);
}
BooleanExpr BooleanExpr::mux_left_false(Operand sel, const BooleanExpr &right) {
    if (right.kind() == Kind::Operand && right.op() == Operand::True) return operand(!sel);
    if (right.kind() == Kind::Operand && right.op() == Operand::False) return operand(Operand::False);
    return ordered_and(operand(!sel), right);
}
BooleanExpr BooleanExpr::mux_left_x(Operand sel, const BooleanExpr &left, const BooleanExpr &right) {
    if (right.kind() == Kind::Operand && right.op() == Operand::True) return mux_left_true(!sel, left);
    if (right.kind() == Kind::Operand && right.op() == Operand::False) return mux_left_false(!sel, left);
    if (left == right) return left;
    if (left == !right) return ordered_xor(operand(!sel), left);
    return make(Kind::Mux, sel, left.node_, right.node_);
}
BooleanExpr BooleanExpr::mux(Operand sel, const BooleanExpr &left, const BooleanExpr &right) {
    if (left.kind() == Kind::Operand && left.op() == Operand::True) return mux_left_true(sel, right);
    if (left.kind() == Kind::Operand && left.op() == Operand::False) return mux_left_false(sel, right);
    return mux_left_x(sel, left, right);
}

BooleanExpr BooleanExpr::reduce_mux(const std::vector<BooleanExpr> &items) {
    const size_t len = items.size();
    if (len & (len - 1)) throw std::invalid_argument("Input was not a power of 2! It was " + std::to_string(len));
    if (len == 0) return operand(Operand::True);
    if (len == 1) return items[0];
    static const Operand selectors[8] = {Operand::Bit0, Operand::Bit1, Operand::Bit2, Operand::Bit3,
                                         Operand::Bit4, Operand::Bit5, Operand::Bit6, Operand::Bit7};
    size_t levels = 0;
    while ((size_t(1) << levels) < len) levels++;
    if (levels > 8) throw std::invalid_argument("truth table longer than 256 entries");
    std::vector<BooleanExpr> acc = items;
    for (size_t k = 0; k < levels; k++) {
        std::vector<BooleanExpr> next;
        next.reserve(acc.size() / 2);
        for (size_t i = 0; i + 1 < acc.size(); i += 2) next.push_back(mux(selectors[k], acc[i], acc[i + 1]));
        acc.swap(next);
    }
    return acc[0];
}

void BooleanExpr::to_hashset(std::set<BooleanExpr> &out) const {
    if (!out.insert(*this).second) return;
    if (node_->kind != Kind::Operand) {
        lhs().to_hashset(out);
        rhs().to_hashset(out);
    }
}

bool BooleanExpr::evaluate_clear(uint8_t x) const {
    switch (node_->kind) {
    case Kind::Operand: {
        uint8_t o = static_cast<uint8_t>(node_->op);
        if (o < 2) return o == 0;
        bool v = (x >> ((o - 2) >> 1)) & 1;
        return (o & 1) ? !v : v;
    }
    case Kind::And: return lhs().evaluate_clear(x) && rhs().evaluate_clear(x);
    case Kind::Or: return lhs().evaluate_clear(x) || rhs().evaluate_clear(x);
    case Kind::Xor: return lhs().evaluate_clear(x) != rhs().evaluate_clear(x);
    default: return operand(node_->op).evaluate_clear(x) ? lhs().evaluate_clear(x) : rhs().evaluate_clear(x);
    }
}

std::string BooleanExpr::to_string() const {
    switch (node_->kind) {
    case Kind::Operand: return operand_name(node_->op);
    case Kind::And: return "And(" + lhs().to_string() + "," + rhs().to_string() + ")";
    case Kind::Or: return "Or(" + lhs().to_string() + "," + rhs().to_string() + ")";
    case Kind::Xor: return "Xor(" + lhs().to_string() + "," + rhs().to_string() + ")";
    default: return std::string("Mux(") + operand_name(node_->op) + "," + lhs().to_string() + "," + rhs().to_string() + ")";
    }
}

}  // namespace tfheaes
