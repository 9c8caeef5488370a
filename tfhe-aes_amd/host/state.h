// AES-128 state and key containers and round functions over FHEBytes: host-side mirror of
// /root/reference/base/src/state.rs:9-313 and /root/reference/base/src/key_schedule.rs:7-210.
// Both hold their 16 bytes transposed, data[4*row + col]; every gate they issue goes through the
// ServerKey batch submitter, so the 16 bytes of a state (and all states in flight) share stages.
#pragma once
#include <array>
#include <vector>

#include "primitive.h"

namespace tfheaes {

class Key {
  public:
    Key() = default;
    static Key from_u128_enc(const std::array<uint8_t, 16> &big_endian, const ClientKey &client_key);
    static Key from_u128_clear(const std::array<uint8_t, 16> &big_endian, const ServerKey &server_key);
    static Key from_u8_enc(const std::array<uint8_t, 16> &data, const ClientKey &client_key);
    static Key from_u8_clear(const std::array<uint8_t, 16> &data, const ServerKey &server_key);
    void generate_next_key_in_place(uint8_t rcon, ServerKey &server_key);
    std::array<uint8_t, 16> decrypt_to_u8(const ClientKey &client_key) const;
    // Encrypted ("online") key expansion: the 11 round keys.
    std::array<Key, 11> generate_round_keys(ServerKey &server_key) const;
    std::array<FHEByte, 16> data;
};
// Clear ("offline") key expansion.
std::array<std::array<uint8_t, 16>, 11> key_expansion_clear(const std::array<uint8_t, 16> &key);

class State {
  public:
    State() = default;
    static State from_u128_enc(const std::array<uint8_t, 16> &big_endian, const ClientKey &client_key);
    static State from_u8_enc(const std::array<uint8_t, 16> &data, const ClientKey &client_key);
    void sub_bytes(ServerKey &server_key);
    void inv_sub_bytes(ServerKey &server_key);
    void shift_rows();
    void inv_shift_rows();
    void mix_columns(ServerKey &server_key);
    void inv_mix_columns(ServerKey &server_key);
    void xor_key_enc(const Key &key, ServerKey &server_key);
    void xor_key_clear(const std::array<uint8_t, 16> &key, ServerKey &server_key);
    void xor_state(const State &state, ServerKey &server_key);
    std::array<uint8_t, 16> decrypt_to_u8(const ClientKey &client_key) const;
    const std::array<FHEByte, 16> &bytes() const { return data_; }
    std::array<FHEByte, 16> &bytes() { return data_; }

  private:
    std::array<FHEByte, 16> data_;
};

// in-place 4x4 transpose of a row-major/column-major byte matrix (the six swaps of the reference)
template <typename T>
void transpose4x4(std::array<T, 16> &m) {
    for (int r = 0; r < 4; r++)
        for (int c = r + 1; c < 4; c++) std::swap(m[4 * r + c], m[4 * c + r]);
}

}  // namespace tfheaes
