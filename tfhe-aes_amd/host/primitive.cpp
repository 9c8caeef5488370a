#include "primitive.h"

#include <stdexcept>

namespace tfheaes {

Runnable::Runnable(const std::array<Ciphertext, 18> &operands, const std::vector<Ciphertext> &results, const StagedCircuit &circuit,
                   const BooleanExpr &expr)
    : expr_(expr) {
    auto result_of = [&](const BooleanExpr &e) -> const Ciphertext & { return results[circuit.index.at(e.id())]; };
    switch (expr.kind()) {
    case BooleanExpr::Kind::Operand: operands_ = {operands[static_cast<uint8_t>(expr.op())]}; break;
    case BooleanExpr::Kind::Mux: operands_ = {operands[static_cast<uint8_t>(expr.op())], result_of(expr.lhs()), result_of(expr.rhs())}; break;
    default: operands_ = {result_of(expr.lhs()), result_of(expr.rhs())}; break;
    }
}

Ciphertext Runnable::run(ServerKey &server_key) const {
    switch (expr_.kind()) {
    case BooleanExpr::Kind::Operand: return operands_[0];
    case BooleanExpr::Kind::And: return server_key.and_(operands_[0], operands_[1]);
    case BooleanExpr::Kind::Or: return server_key.or_(operands_[0], operands_[1]);
    case BooleanExpr::Kind::Xor: return server_key.xor_(operands_[0], operands_[1]);
    default: return server_key.mux(operands_[0], operands_[1], operands_[2]);
    }
}

FHEByte::FHEByte(const std::vector<bool> &bool_data, const ClientKey &client_key) {
    if (bool_data.size() != 8) throw std::invalid_argument("FHEByte has to be initialized with 8 booleans !");
    for (int i = 0; i < 8; i++) data_[i] = client_key.encrypt(bool_data[i]);
}
FHEByte FHEByte::from_u8_enc(uint8_t value, const ClientKey &client_key) {
    FHEByte b;
    for (int i = 0; i < 8; i++) b.data_[i] = client_key.encrypt((value >> (7 - i)) & 1);
    return b;
}
FHEByte FHEByte::from_u8_clear(uint8_t value, const ServerKey &server_key) {
    FHEByte b;
    for (int i = 0; i < 8; i++) b.data_[i] = server_key.trivial_encrypt((value >> (7 - i)) & 1);
    return b;
}
FHEByte FHEByte::trivial_clear(uint8_t clear_value, const ServerKey &server_key) { return from_u8_clear(clear_value, server_key); }
FHEByte FHEByte::trivial_false(const ServerKey &server_key) { return trivial_clear(0, server_key); }

std::vector<bool> FHEByte::decrypt(const ClientKey &client_key) const {
    std::vector<bool> out(8);
    for (int i = 0; i < 8; i++) out[i] = client_key.decrypt(data_[i]);
    return out;
}
uint8_t FHEByte::decrypt_to_u8(const ClientKey &client_key) const {
    uint8_t v = 0;
    for (int i = 0; i < 8; i++)
        if (client_key.decrypt(data_[i])) v |= static_cast<uint8_t>(1u << (7 - i));
    return v;
}

void FHEByte::xor_in_place(const FHEByte &rhs, ServerKey &sk) {
    for (int i = 0; i < 8; i++) sk.xor_assign(data_[i], rhs.data_[i]);
}
FHEByte FHEByte::xor_(const FHEByte &rhs, ServerKey &sk) const {
    FHEByte r = *this;
    r.xor_in_place(rhs, sk);
    return r;
}
void FHEByte::and_in_place(const FHEByte &rhs, ServerKey &sk) {
    for (int i = 0; i < 8; i++) sk.and_assign(data_[i], rhs.data_[i]);
}
FHEByte FHEByte::and_(const FHEByte &rhs, ServerKey &sk) const {
    FHEByte r = *this;
    r.and_in_place(rhs, sk);
    return r;
}
void FHEByte::or_in_place(const FHEByte &rhs, ServerKey &sk) {
    for (int i = 0; i < 8; i++) sk.or_assign(data_[i], rhs.data_[i]);
}
FHEByte FHEByte::or_(const FHEByte &rhs, ServerKey &sk) const {
    FHEByte r = *this;
    r.or_in_place(rhs, sk);
    return r;
}
void FHEByte::not_in_place(const ServerKey &sk) {
    for (auto &c : data_) sk.not_assign(c);
}
FHEByte FHEByte::not_(const ServerKey &sk) const {
    FHEByte r = *this;
    r.not_in_place(sk);
    return r;
}

// Staged evaluation of the 8 output circuits (primitive.rs:209-274): operands Bit_k / NotBit_k are
// bit k of the byte counted from the LSB, every distinct sub-expression is evaluated once, stage by
// stage, and the roots are read back MSB first.  Each stage's gates go to the ServerKey together.
FHEByte FHEByte::staged_eval(const StagedCircuit &circuit, ServerKey &sk) const {
    std::array<Ciphertext, 18> operands;
    operands[static_cast<uint8_t>(Operand::True)] = sk.trivial_encrypt(true);
    operands[static_cast<uint8_t>(Operand::False)] = sk.trivial_encrypt(false);
    for (int k = 0; k < 8; k++) {
        operands[2 + 2 * k] = data_[7 - k];
        operands[3 + 2 * k] = sk.not_(data_[7 - k]);
    }
    std::vector<Ciphertext> results(circuit.node_count);
    for (const auto &stage : circuit.stages) {
        std::vector<Runnable> runnables;
        runnables.reserve(stage.size());
        for (const auto &expr : stage) runnables.emplace_back(operands, results, circuit, expr);
        for (size_t i = 0; i < stage.size(); i++) results[circuit.index.at(stage[i].id())] = runnables[i].run(sk);
    }
    FHEByte out;
    for (int i = 0; i < 8; i++) out.data_[i] = results[circuit.index.at(circuit.roots[i].id())];
    return out;
}
FHEByte FHEByte::sub_byte(ServerKey &sk) const { return staged_eval(s_box_circuit(), sk); }
FHEByte FHEByte::inv_sub_byte(ServerKey &sk) const { return staged_eval(inv_s_box_circuit(), sk); }

void FHEByte::shift_left_in_place(unsigned shift, const ServerKey &sk) {
    if (shift > 8) shift = 8;
    for (unsigned s = 0; s < shift; s++) {
        for (int i = 0; i < 7; i++) data_[i] = data_[i + 1];
        data_[7] = sk.trivial_encrypt(false);
    }
}

// Multiplication by x in GF(2^8) (primitive.rs:349-362): shift left, then per bit
// mux(old MSB, bit ^ 0x1b-bit, bit).  The xor with a trivial bit is free; the LSB resolves without
// a bootstrap because both of its branches are trivial.
void FHEByte::mul_x_gf2_in_place(ServerKey &sk) {
    Ciphertext conditional_bit = data_[0];
    shift_left_in_place(1, sk);
    FHEByte irr_poly = trivial_clear(0x1b, sk);
    for (int i = 0; i < 8; i++) data_[i] = sk.mux(conditional_bit, sk.xor_(data_[i], irr_poly.data_[i]), data_[i]);
}
FHEByte FHEByte::mul_x_gf2(ServerKey &sk) const {
    FHEByte r = *this;
    r.mul_x_gf2_in_place(sk);
    return r;
}

}  // namespace tfheaes
