// FHEByte and Runnable: host-side mirror of /root/reference/base/src/primitive.rs:50-368 and
// /root/reference/base/src/boolean_tree.rs:605-655.
#pragma once
#include <array>
#include <vector>

#include "boolean_tree.h"
#include "sbox.h"
#include "tfhe_boolean.h"

namespace tfheaes {

// One gate of a staged circuit with its operands bound (1 for a leaf, 2 for And/Or/Xor, 3 for Mux).
// run() hands the gate to the ServerKey, which records it for the next batched stage.
class Runnable {
  public:
    Runnable(const std::array<Ciphertext, 18> &operands, const std::vector<Ciphertext> &results, const StagedCircuit &circuit,
             const BooleanExpr &expr);
    Ciphertext run(ServerKey &server_key) const;

  private:
    BooleanExpr expr_;
    std::vector<Ciphertext> operands_;
};

// An encrypted byte: 8 boolean ciphertexts, index 0 = most significant bit.
class FHEByte {
  public:
    FHEByte() = default;
    FHEByte(const std::vector<bool> &bool_data, const ClientKey &client_key);  // exactly 8 booleans, else std::invalid_argument
    static FHEByte from_u8_enc(uint8_t value, const ClientKey &client_key);
    static FHEByte from_u8_clear(uint8_t value, const ServerKey &server_key);
    static FHEByte trivial_clear(uint8_t clear_value, const ServerKey &server_key);
    static FHEByte trivial_false(const ServerKey &server_key);

    std::vector<bool> decrypt(const ClientKey &client_key) const;
    uint8_t decrypt_to_u8(const ClientKey &client_key) const;

    void xor_in_place(const FHEByte &rhs, ServerKey &server_key);
    FHEByte xor_(const FHEByte &rhs, ServerKey &server_key) const;
    void and_in_place(const FHEByte &rhs, ServerKey &server_key);
    FHEByte and_(const FHEByte &rhs, ServerKey &server_key) const;
    void or_in_place(const FHEByte &rhs, ServerKey &server_key);
    FHEByte or_(const FHEByte &rhs, ServerKey &server_key) const;
    void not_in_place(const ServerKey &server_key);
    FHEByte not_(const ServerKey &server_key) const;

    FHEByte sub_byte(ServerKey &server_key) const;
    FHEByte inv_sub_byte(ServerKey &server_key) const;
    void mul_x_gf2_in_place(ServerKey &server_key);
    FHEByte mul_x_gf2(ServerKey &server_key) const;

    const Ciphertext &bit(int i) const { return data_[i]; }  // i = 0 is the MSB
    Ciphertext &bit(int i) { return data_[i]; }

  private:
    FHEByte staged_eval(const StagedCircuit &circuit, ServerKey &server_key) const;
    void shift_left_in_place(unsigned shift, const ServerKey &server_key);
    std::array<Ciphertext, 8> data_;
};

}  // namespace tfheaes
