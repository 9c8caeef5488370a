#include "sbox.h"

#include <set>

namespace tfheaes {

namespace {
// The AES S-box is the affine image of the multiplicative inverse in GF(2^8) (FIPS-197 5.1.1).
std::array<uint8_t, 256> build_sbox() {
    std::array<uint8_t, 256> s{};
    auto mul = [](uint8_t a, uint8_t b) {
        uint8_t r = 0;
        while (b) {
            if (b & 1) r ^= a;
            a = static_cast<uint8_t>((a << 1) ^ ((a & 0x80) ? 0x1B : 0));
            b >>= 1;
        }
        return r;
    };
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x)
            for (int y = 1; y < 256; y++)
                if (mul(static_cast<uint8_t>(x), static_cast<uint8_t>(y)) == 1) {
                    inv = static_cast<uint8_t>(y);
                    break;
                }
        uint8_t v = inv;
        for (int k = 1; k <= 4; k++) v ^= static_cast<uint8_t>((inv << k) | (inv >> (8 - k)));
        s[x] = v ^ 0x63;
    }
    return s;
}
std::array<uint8_t, 256> invert(const std::array<uint8_t, 256> &s) {
    std::array<uint8_t, 256> r{};
    for (int x = 0; x < 256; x++) r[s[x]] = static_cast<uint8_t>(x);
    return r;
}
}  // namespace

const std::array<uint8_t, 256> S_BOX_DATA = build_sbox();
const std::array<uint8_t, 256> INV_S_BOX_DATA = invert(S_BOX_DATA);

std::vector<bool> bit_x_s_box(const std::array<uint8_t, 256> &data, uint8_t position) {
    std::vector<bool> plane(256);
    for (int i = 0; i < 256; i++) plane[i] = (data[255 - i] >> position) & 1;
    return plane;
}

std::vector<BooleanExpr> generate_reduced_bool_expr(const std::array<uint8_t, 256> &data) {
    std::vector<BooleanExpr> roots;
    for (int pos = 7; pos >= 0; pos--) roots.push_back(BooleanExpr::reduce_mux(BooleanExpr::from_bool_vec(bit_x_s_box(data, static_cast<uint8_t>(pos)))));
    return roots;
}

StagedCircuit stage_exprs(const std::vector<BooleanExpr> &roots) {
    StagedCircuit c;
    c.roots = roots;
    std::set<BooleanExpr> all;  // ordered by the canonical order, so stage lists are deterministic
    for (const auto &r : roots) r.to_hashset(all);
    uint8_t top = 0;
    for (const auto &e : all) top = std::max(top, e.stage());
    c.stages.resize(top + 1);
    for (const auto &e : all) {
        c.stages[e.stage()].push_back(e);
        c.index.emplace(e.id(), c.node_count++);
    }
    return c;
}

const StagedCircuit &s_box_circuit() {
    static const StagedCircuit c = stage_exprs(generate_reduced_bool_expr(S_BOX_DATA));
    return c;
}
const StagedCircuit &inv_s_box_circuit() {
    static const StagedCircuit c = stage_exprs(generate_reduced_bool_expr(INV_S_BOX_DATA));
    return c;
}

}  // namespace tfheaes
