#include "state.h"

#include <algorithm>

namespace tfheaes {

namespace {
const uint8_t RCON[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};

template <typename MakeByte>
std::array<FHEByte, 16> make_transposed(const std::array<uint8_t, 16> &bytes, MakeByte make) {
    std::array<FHEByte, 16> d;
    for (int i = 0; i < 16; i++) d[i] = make(bytes[i]);
    transpose4x4(d);
    return d;
}
std::array<uint8_t, 16> decrypt_transposed(const std::array<FHEByte, 16> &d, const ClientKey &ck) {
    std::array<uint8_t, 16> out;
    for (int i = 0; i < 16; i++) out[i] = d[i].decrypt_to_u8(ck);
    transpose4x4(out);
    return out;
}
}  // namespace

// ------------------------------------------------------------------ Key
Key Key::from_u8_enc(const std::array<uint8_t, 16> &bytes, const ClientKey &ck) {
    Key k;
    k.data = make_transposed(bytes, [&](uint8_t v) { return FHEByte::from_u8_enc(v, ck); });
    return k;
}
Key Key::from_u8_clear(const std::array<uint8_t, 16> &bytes, const ServerKey &sk) {
    Key k;
    k.data = make_transposed(bytes, [&](uint8_t v) { return FHEByte::from_u8_clear(v, sk); });
    return k;
}
// A u128 in the reference is consumed most significant byte first, i.e. as the same 16 bytes.
Key Key::from_u128_enc(const std::array<uint8_t, 16> &big_endian, const ClientKey &ck) { return from_u8_enc(big_endian, ck); }
Key Key::from_u128_clear(const std::array<uint8_t, 16> &big_endian, const ServerKey &sk) { return from_u8_clear(big_endian, sk); }
std::array<uint8_t, 16> Key::decrypt_to_u8(const ClientKey &ck) const { return decrypt_transposed(data, ck); }

// One round of the key schedule on the transposed key (key_schedule.rs:90-131): column 3 is
// rotated and substituted (bytes 7, 11, 15, 3), RCON enters as a trivial byte, and the four
// columns are chained left to right.
void Key::generate_next_key_in_place(uint8_t rcon, ServerKey &sk) {
    static const int rot_word[4] = {7, 11, 15, 3};
    std::array<FHEByte, 4> temp;
    for (int r = 0; r < 4; r++) temp[r] = data[rot_word[r]].sub_byte(sk);
    temp[0].xor_in_place(FHEByte::from_u8_clear(rcon, sk), sk);
    for (int r = 0; r < 4; r++) data[4 * r].xor_in_place(temp[r], sk);
    for (int c = 1; c < 4; c++)
        for (int r = 0; r < 4; r++) data[4 * r + c].xor_in_place(data[4 * r + c - 1], sk);
}

std::array<Key, 11> Key::generate_round_keys(ServerKey &sk) const {
    std::array<Key, 11> keys;
    keys[0] = *this;
    for (int i = 0; i < 10; i++) {
        keys[i + 1] = keys[i];
        keys[i + 1].generate_next_key_in_place(RCON[i], sk);
    }
    return keys;
}

std::array<std::array<uint8_t, 16>, 11> key_expansion_clear(const std::array<uint8_t, 16> &key) {
    std::array<std::array<uint8_t, 16>, 11> rk{};
    rk[0] = key;
    for (int round = 1; round <= 10; round++) {
        const auto &prev = rk[round - 1];
        uint8_t t[4] = {S_BOX_DATA[prev[13]], S_BOX_DATA[prev[14]], S_BOX_DATA[prev[15]], S_BOX_DATA[prev[12]]};
        t[0] ^= RCON[round - 1];
        for (int i = 0; i < 4; i++) rk[round][i] = prev[i] ^ t[i];
        for (int i = 4; i < 16; i++) rk[round][i] = prev[i] ^ rk[round][i - 4];
    }
    return rk;
}

// ------------------------------------------------------------------ State
State State::from_u8_enc(const std::array<uint8_t, 16> &bytes, const ClientKey &ck) {
    State s;
    s.data_ = make_transposed(bytes, [&](uint8_t v) { return FHEByte::from_u8_enc(v, ck); });
    return s;
}
State State::from_u128_enc(const std::array<uint8_t, 16> &big_endian, const ClientKey &ck) { return from_u8_enc(big_endian, ck); }
std::array<uint8_t, 16> State::decrypt_to_u8(const ClientKey &ck) const { return decrypt_transposed(data_, ck); }

void State::sub_bytes(ServerKey &sk) {
    for (auto &b : data_) b = b.sub_byte(sk);
}
void State::inv_sub_bytes(ServerKey &sk) {
    for (auto &b : data_) b = b.inv_sub_byte(sk);
}
void State::shift_rows() {
    for (int r = 1; r < 4; r++) std::rotate(data_.begin() + 4 * r, data_.begin() + 4 * r + r, data_.begin() + 4 * r + 4);
}
void State::inv_shift_rows() {
    for (int r = 1; r < 4; r++) std::rotate(data_.begin() + 4 * r, data_.begin() + 4 * r + 4 - r, data_.begin() + 4 * r + 4);
}

// MixColumns on the transposed state (state.rs:72-118).  With rows r0..r3:
//   y_i = XOR of the three other rows (two xor layers), r_i <- 2*r_i, y_i ^= r_i, y_i ^= r_{i+1}
// which gives y0 = 2 r0 ^ 3 r1 ^ r2 ^ r3.  The gate list is the reference's (512 xor + 112 mux).
void State::mix_columns(ServerKey &sk) {
    static const int first[16] = {8, 9, 10, 11, 8, 9, 10, 11, 0, 1, 2, 3, 0, 1, 2, 3};
    static const int second[16] = {4, 5, 6, 7, 0, 1, 2, 3, 12, 13, 14, 15, 8, 9, 10, 11};
    static const int last[16] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3};
    std::array<FHEByte, 16> y;
    for (int i = 0; i < 16; i++) y[i] = data_[first[i]].xor_(data_[first[i] + 4], sk);
    for (int i = 0; i < 16; i++) y[i].xor_in_place(data_[second[i]], sk);
    for (auto &b : data_) b.mul_x_gf2_in_place(sk);
    for (int i = 0; i < 16; i++) y[i].xor_in_place(data_[i], sk);
    for (int i = 0; i < 16; i++) y[i].xor_in_place(data_[last[i]], sk);
    data_ = y;
}

// InvMixColumns on the transposed state, same gate sequence as state.rs:120-215:
//   y_i = x_i ^ x_{i+1};  u = 4*(x_i ^ x_{i+2});  v = x_0^x_1^x_2^x_3;
//   x_i ^= 2*y_i ^ u ^ v ^ 8*v
void State::inv_mix_columns(ServerKey &sk) {
    std::array<FHEByte, 16> y;
    for (int i = 0; i < 16; i++) y[i] = data_[i].xor_(data_[(i + 4) % 16], sk);
    std::array<FHEByte, 8> t0;
    for (int i = 0; i < 8; i++) t0[i] = data_[i].xor_(data_[i + 8], sk);
    for (auto &b : t0) b.mul_x_gf2_in_place(sk);
    for (auto &b : t0) b.mul_x_gf2_in_place(sk);
    std::array<FHEByte, 4> t1;
    for (int i = 0; i < 4; i++) t1[i] = y[i].xor_(y[8 + i], sk);
    for (auto &b : y) b.mul_x_gf2_in_place(sk);
    for (int i = 0; i < 16; i++) y[i].xor_in_place(t0[i % 8], sk);
    for (int i = 0; i < 16; i++) data_[i].xor_in_place(y[i], sk);
    for (int i = 0; i < 16; i++) data_[i].xor_in_place(t1[i % 4], sk);
    for (int rep = 0; rep < 3; rep++)
        for (auto &b : t1) b.mul_x_gf2_in_place(sk);
    for (int i = 0; i < 16; i++) data_[i].xor_in_place(t1[i % 4], sk);
}

void State::xor_key_enc(const Key &key, ServerKey &sk) {
    for (int i = 0; i < 16; i++) data_[i].xor_in_place(key.data[i], sk);
}
void State::xor_key_clear(const std::array<uint8_t, 16> &key, ServerKey &sk) {
    std::array<uint8_t, 16> k = key;
    transpose4x4(k);
    for (int i = 0; i < 16; i++) data_[i].xor_in_place(FHEByte::trivial_clear(k[i], sk), sk);
}
void State::xor_state(const State &state, ServerKey &sk) {
    for (int i = 0; i < 16; i++) data_[i].xor_in_place(state.data_[i], sk);
}

}  // namespace tfheaes
