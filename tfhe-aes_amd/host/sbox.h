// AES S-box tables as boolean circuits: host-side mirror of /root/reference/base/src/sbox.rs
// (tables :6-43, bit_x_s_box :45-50, generate_reduced_bool_expr :53-61) plus the staged form of
// the circuits.  The reference rebuilds the staging (hash set of sub-expressions bucketed by
// height) inside every FHEByte::sub_byte call (primitive.rs:212-225); here it is built once.
#pragma once
#include <array>
#include <unordered_map>
#include <vector>

#include "boolean_tree.h"

namespace tfheaes {

extern const std::array<uint8_t, 256> S_BOX_DATA;
extern const std::array<uint8_t, 256> INV_S_BOX_DATA;

// One bit plane of the table, walked in reverse so that input x selects data[x] after reduce_mux.
std::vector<bool> bit_x_s_box(const std::array<uint8_t, 256> &data, uint8_t position);
// The 8 output-bit circuits, most significant bit first.
std::vector<BooleanExpr> generate_reduced_bool_expr(const std::array<uint8_t, 256> &data);

struct StagedCircuit {
    std::vector<BooleanExpr> roots;                // 8 outputs, MSB first
    std::vector<std::vector<BooleanExpr>> stages;  // stages[h] = distinct sub-expressions of height h, canonical order
    std::unordered_map<uint32_t, uint32_t> index;  // expression id -> dense index into a result table
    uint32_t node_count = 0;
};
StagedCircuit stage_exprs(const std::vector<BooleanExpr> &roots);
const StagedCircuit &s_box_circuit();      // the reference's S_BOX_EXPR (primitive.rs:14-15), staged
const StagedCircuit &inv_s_box_circuit();  // the reference's INV_S_BOX_EXPR (primitive.rs:16-17), staged

}  // namespace tfheaes
