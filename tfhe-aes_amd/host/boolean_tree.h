// Boolean-expression IR of the S-box circuits: host-side mirror of
// /root/reference/base/src/boolean_tree.rs (Operand :14-100, BooleanExpr :121-332, ordering and
// hashing :507-601).  Expressions are hash-consed: every structurally distinct expression has one
// node, so equality and hashing are pointer operations and `stage()` is memoised, which removes
// the per-call deep hashing the reference pays in FHEByte::sub_byte.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <vector>

namespace tfheaes {

// Same declaration order as the reference; the order is the total order used for canonical forms.
enum class Operand : uint8_t {
    True, False, Bit0, NotBit0, Bit1, NotBit1, Bit2, NotBit2, Bit3, NotBit3, Bit4, NotBit4, Bit5, NotBit5, Bit6, NotBit6, Bit7, NotBit7
};
inline Operand operator!(Operand o) { return static_cast<Operand>(static_cast<uint8_t>(o) ^ 1u); }
inline Operand operand_from_bool(bool b) { return b ? Operand::True : Operand::False; }
const char *operand_name(Operand o);

class BooleanExpr {
  public:
    enum class Kind : uint8_t { Operand = 0, And = 1, Or = 2, Xor = 3, Mux = 4 };  // = the reference's discriminants

    BooleanExpr() : node_(nullptr) {}
    static BooleanExpr operand(Operand o);
    static BooleanExpr from_bool(bool b) { return operand(operand_from_bool(b)); }
    static std::vector<BooleanExpr> from_bool_vec(const std::vector<bool> &items);
    // And/Or/Xor with operands put in canonical order; Mux with the reference's simplification rules.
    static BooleanExpr ordered_and(const BooleanExpr &a, const BooleanExpr &b);
    static BooleanExpr ordered_or(const BooleanExpr &a, const BooleanExpr &b);
    static BooleanExpr ordered_xor(const BooleanExpr &a, const BooleanExpr &b);
    static BooleanExpr mux(Operand sel, const BooleanExpr &left, const BooleanExpr &right);
    // Folds a 2^k-entry truth table with selectors Bit0..Bit(k-1); throws std::invalid_argument
    // when the length is not a power of two (the reference panics).
    static BooleanExpr reduce_mux(const std::vector<BooleanExpr> &items);

    BooleanExpr operator!() const;
    Kind kind() const;
    Operand op() const;           // Operand leaf, or the selector of a Mux
    BooleanExpr lhs() const;      // And/Or/Xor left, Mux `then`
    BooleanExpr rhs() const;      // And/Or/Xor right, Mux `else`
    uint8_t stage() const;        // height of the node; leaves are 0
    uint32_t id() const;          // dense id of the interned node
    void to_hashset(std::set<BooleanExpr> &out) const;  // all distinct sub-expressions, leaves included
    bool evaluate_clear(uint8_t x) const;               // Bit_k = bit k of x
    std::string to_string() const;

    bool operator==(const BooleanExpr &o) const { return node_ == o.node_; }
    bool operator!=(const BooleanExpr &o) const { return node_ != o.node_; }
    bool operator<(const BooleanExpr &o) const { return compare(*this, o) < 0; }
    static int compare(const BooleanExpr &a, const BooleanExpr &b);  // the reference's Ord

    struct Node;

  private:
    explicit BooleanExpr(const Node *n) : node_(n) {}
    static BooleanExpr make(Kind k, Operand op, const Node *l, const Node *r);
    static BooleanExpr mux_left_true(Operand sel, const BooleanExpr &right);
    static BooleanExpr mux_left_false(Operand sel, const BooleanExpr &right);
    static BooleanExpr mux_left_x(Operand sel, const BooleanExpr &left, const BooleanExpr &right);
    const Node *node_;
};

}  // namespace tfheaes
