// Host-side mirror of the tfhe::boolean API surface that `base` uses (gen_keys, ClientKey,
// ServerKey, Ciphertext; call sites /root/reference/base/src/primitive.rs:63-145,170-248,359 and
// /root/reference/base/src/boolean_tree.rs:73-95,644-654), sitting on the C ABI of the HIP engine.
//
// Same calls, same results, different execution model: a ServerKey gate does not bootstrap on the
// spot.  It allocates the output slot in the device-resident pool, records the gate together with
// its dependency level (1 + the deepest encrypted operand) and returns the handle at once.
// ServerKey::flush() - implied by every decryption - submits the recorded gates level by level, so
// every gate that is independent at a given tree height, across all bytes of a state and all blocks
// in flight, lands in one tfa_stage_submit call.  This is the batch submitter that replaces the
// rayon pool of FHEByte::sub_byte (primitive.rs:250-265).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "tfhe_aes_hip.h"

namespace tfheaes {

class ServerKey;
class ClientKey;

namespace detail {
struct SlotRef;  // owns one pool slot; returns it to the pool when the last handle dies
struct Enc;      // an encrypted value: host words until first used by a ServerKey, then a pool slot
}

// tfhe::boolean::Ciphertext: Encrypted(LWE) | Trivial(bool).  An encrypted value is either still
// on the host (fresh from ClientKey::encrypt) or a slot of a ServerKey's pool; `neg` is a pending
// free negation (ServerKey::not) folded into the next consumer.
class Ciphertext {
  public:
    Ciphertext() = default;
    static Ciphertext trivial(bool v);
    bool is_trivial() const { return kind_ == Kind::Trivial; }
    bool trivial_value() const { return value_; }

  private:
    friend class ServerKey;
    friend class ClientKey;
    enum class Kind : uint8_t { Empty, Trivial, Encrypted };
    Kind kind_ = Kind::Empty;
    bool value_ = false;
    bool neg_ = false;
    std::shared_ptr<detail::Enc> enc_;  // shared by all copies of the handle
};

class ClientKey {
  public:
    ClientKey(const tfa_params &p, uint64_t seed, std::vector<uint32_t> lwe_sk, std::vector<uint32_t> glwe_sk);
    Ciphertext encrypt(bool message) const;
    bool decrypt(const Ciphertext &ct) const;  // flushes the owning ServerKey when needed
    const tfa_params &params() const { return p_; }
    const std::vector<uint32_t> &lwe_secret_key() const { return lwe_sk_; }

  private:
    tfa_params p_;
    mutable uint64_t enc_seed_, enc_counter_ = 0;
    std::vector<uint32_t> lwe_sk_, glwe_sk_;
};

struct ServerKeyStats {
    uint64_t gates = 0, stages = 0, flushes = 0, max_stage_width = 0;
};

class ServerKey {
  public:
    // Uploads the bootstrapping and key-switching keys to `device`; throws std::runtime_error on failure.
    ServerKey(const tfa_params &p, const uint32_t *bsk, const uint32_t *ksk, int device);
    ~ServerKey();
    ServerKey(const ServerKey &) = delete;
    ServerKey &operator=(const ServerKey &) = delete;

    Ciphertext trivial_encrypt(bool v) const { return Ciphertext::trivial(v); }
    Ciphertext and_(const Ciphertext &a, const Ciphertext &b);
    Ciphertext nand(const Ciphertext &a, const Ciphertext &b);
    Ciphertext or_(const Ciphertext &a, const Ciphertext &b);
    Ciphertext nor(const Ciphertext &a, const Ciphertext &b);
    Ciphertext xor_(const Ciphertext &a, const Ciphertext &b);
    Ciphertext xnor(const Ciphertext &a, const Ciphertext &b);
    Ciphertext mux(const Ciphertext &cond, const Ciphertext &then_ct, const Ciphertext &else_ct);
    Ciphertext not_(const Ciphertext &a) const;
    void and_assign(Ciphertext &a, const Ciphertext &b) { a = and_(a, b); }
    void or_assign(Ciphertext &a, const Ciphertext &b) { a = or_(a, b); }
    void xor_assign(Ciphertext &a, const Ciphertext &b) { a = xor_(a, b); }
    void not_assign(Ciphertext &a) const { a = not_(a); }

    // Runs every recorded gate (level by level) and waits for the device.
    void flush();
    // On (the default; TFHE_AES_XOR_REBALANCE=0 turns it off at construction): before a flush, chains of xor gates whose
    // intermediate values have no live handle are re-associated into trees of least depth (same gates, same results, fewer
    // stages; tfhe_boolean.cpp: rebalance_xor_chains).  Off: gates run in the association they were recorded in.
    void set_xor_rebalance(bool on) { xor_rebalance_ = on; }
    bool xor_rebalance() const { return xor_rebalance_; }
    // LWE words of an encrypted ciphertext (flushes first); used by ClientKey::decrypt and by gathers.
    std::vector<uint32_t> read(const Ciphertext &ct);
    void read_many(const std::vector<const Ciphertext *> &cts, uint32_t *out /*[count][n+1]*/);
    void read_many_device(const std::vector<const Ciphertext *> &cts, void *d_out /*[count][n+1] on the context's GPU*/);
    Ciphertext import_lwe(const uint32_t *lwe);  // adopt n+1 host words as an encrypted ciphertext

    const tfa_params &params() const { return p_; }
    tfa_ctx *context() const { return ctx_; }
    ServerKeyStats stats() const { return stats_; }
    uint64_t pending_gates() const { return pending_count_; }
    static constexpr uint64_t kAutoFlushGates = 2000000;  // about 28 AES blocks in flight = 6.5 GB of pool

  private:
    friend struct detail::SlotRef;
    struct Resolved {
        uint32_t slot, level;
        bool neg;
    };
    Ciphertext binary(uint32_t op, const Ciphertext &a, const Ciphertext &b);
    Resolved resolve(const Ciphertext &ct);  // uploads host ciphertexts on first use
    Ciphertext record(uint32_t op, const Resolved *in, int n_in);
    uint32_t alloc_slot();
    void release_slot(uint32_t slot);
    void upload_pending();
    void auto_flush();
    void rebalance_xor_chains();

    tfa_params p_;
    tfa_ctx *ctx_ = nullptr;
    uint32_t pool_slots_ = 0, next_fresh_ = 0;
    std::vector<uint32_t> free_, deferred_free_;
    uint64_t epoch_ = 1;
    std::vector<std::vector<tfa_gate>> levels_;  // levels_[L-1] = gates at dependency level L
    uint64_t pending_count_ = 0;
    bool xor_rebalance_ = true;
    std::vector<uint32_t> upload_slots_, upload_words_;
    ServerKeyStats stats_;
    std::shared_ptr<int> alive_;  // lets slot handles outlive the key safely
};

// tfhe::boolean::gen_keys(); seed 0 = OS entropy.
struct KeyPair {
    std::unique_ptr<ClientKey> client;
    std::unique_ptr<ServerKey> server;
};
KeyPair gen_keys(uint64_t seed = 0, int device = 0);

}  // namespace tfheaes
