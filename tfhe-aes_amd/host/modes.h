// Block-cipher modes over encrypted states: host-side mirror of /root/reference/modes/src/
// ecb.rs:5-51, cbc.rs:7-43, ofb.rs:7-39, ctr.rs:12-47.  Gates of all blocks handled by one call are
// recorded first and executed together, so independent blocks (ECB batches, CTR, CBC decryption)
// share every stage.
#pragma once
#include <vector>

#include "state.h"

namespace tfheaes {

class ECB {
  public:
    explicit ECB(const std::vector<Key> &keys);  // the 11 round keys
    void encrypt(State &state, ServerKey &server_key) const;
    void decrypt(State &state, ServerKey &server_key) const;

  private:
    std::vector<Key> keys_;
};

class CBC {
  public:
    CBC(const std::vector<Key> &keys, const State &iv, uint8_t n);
    void encrypt(std::vector<State> &plaintext, ServerKey &server_key) const;
    void decrypt(std::vector<State> &ciphertexts, ServerKey &server_key) const;

  private:
    ECB ecb_;
    State iv_;
    uint8_t n_;
};

class OFB {
  public:
    OFB(const std::vector<Key> &keys, const State &iv, uint8_t n);
    void encrypt(std::vector<State> &plaintext, ServerKey &server_key) const;
    void decrypt(std::vector<State> &ciphertexts, ServerKey &server_key) const;

  private:
    ECB ecb_;
    State iv_;
    uint8_t n_;
};

class CTR {
  public:
    // counters are encrypted by the client; throws std::invalid_argument unless counters.size() == n
    CTR(const std::vector<Key> &keys, const std::vector<State> &counters, uint8_t n);
    void encrypt(std::vector<State> &plaintext, ServerKey &server_key) const;
    void decrypt(std::vector<State> &ciphertexts, ServerKey &server_key) const;

  private:
    ECB ecb_;
    std::vector<State> counters_;
};

}  // namespace tfheaes
