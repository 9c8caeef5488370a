#include "modes.h"

#include <stdexcept>

namespace tfheaes {

ECB::ECB(const std::vector<Key> &keys) : keys_(keys) {
    if (keys_.size() != 11) throw std::invalid_argument("ECB needs the 11 AES-128 round keys");
}

void ECB::encrypt(State &state, ServerKey &sk) const {
    state.xor_key_enc(keys_[0], sk);
    for (int round = 1; round < 10; round++) {
        state.sub_bytes(sk);
        state.shift_rows();
        state.mix_columns(sk);
        state.xor_key_enc(keys_[round], sk);
    }
    state.sub_bytes(sk);
    state.shift_rows();
    state.xor_key_enc(keys_[10], sk);
}

void ECB::decrypt(State &state, ServerKey &sk) const {
    state.xor_key_enc(keys_[10], sk);
    for (int round = 1; round < 10; round++) {
        state.inv_shift_rows();
        state.inv_sub_bytes(sk);
        state.xor_key_enc(keys_[10 - round], sk);
        state.inv_mix_columns(sk);
    }
    state.inv_shift_rows();
    state.inv_sub_bytes(sk);
    state.xor_key_enc(keys_[0], sk);
}

CBC::CBC(const std::vector<Key> &keys, const State &iv, uint8_t n) : ecb_(keys), iv_(iv), n_(n) {}

void CBC::encrypt(std::vector<State> &plaintext, ServerKey &sk) const {
    if (plaintext.size() < n_) throw std::invalid_argument("CBC: fewer blocks than n");
    if (n_ == 0) return;
    plaintext[0].xor_state(iv_, sk);
    ecb_.encrypt(plaintext[0], sk);
    for (size_t i = 1; i < n_; i++) {
        plaintext[i].xor_state(plaintext[i - 1], sk);
        ecb_.encrypt(plaintext[i], sk);
    }
}
void CBC::decrypt(std::vector<State> &ciphertexts, ServerKey &sk) const {
    if (ciphertexts.size() < n_) throw std::invalid_argument("CBC: fewer blocks than n");
    if (n_ == 0) return;
    for (size_t i = n_; i-- > 1;) {
        ecb_.decrypt(ciphertexts[i], sk);
        ciphertexts[i].xor_state(ciphertexts[i - 1], sk);
    }
    ecb_.decrypt(ciphertexts[0], sk);
    ciphertexts[0].xor_state(iv_, sk);
}

OFB::OFB(const std::vector<Key> &keys, const State &iv, uint8_t n) : ecb_(keys), iv_(iv), n_(n) {}

void OFB::encrypt(std::vector<State> &plaintext, ServerKey &sk) const {
    if (plaintext.size() < n_) throw std::invalid_argument("OFB: fewer blocks than n");
    State curr = iv_;
    for (size_t i = 0; i < n_; i++) {
        ecb_.encrypt(curr, sk);
        plaintext[i].xor_state(curr, sk);
    }
}
void OFB::decrypt(std::vector<State> &ciphertexts, ServerKey &sk) const { encrypt(ciphertexts, sk); }

CTR::CTR(const std::vector<Key> &keys, const std::vector<State> &counters, uint8_t n) : ecb_(keys), counters_(counters) {
    if (counters.size() != n) throw std::invalid_argument("CTR: counters.len() != n");
}
void CTR::encrypt(std::vector<State> &plaintext, ServerKey &sk) const {
    if (plaintext.size() < counters_.size()) throw std::invalid_argument("CTR: fewer blocks than counters");
    std::vector<State> enc_counters = counters_;
    for (auto &c : enc_counters) ecb_.encrypt(c, sk);
    for (size_t i = 0; i < enc_counters.size(); i++) plaintext[i].xor_state(enc_counters[i], sk);
}
void CTR::decrypt(std::vector<State> &ciphertexts, ServerKey &sk) const { encrypt(ciphertexts, sk); }

}  // namespace tfheaes
