// C view of the host mirror (include/tfhe_aes_host.h): thin forwarding, exceptions -> error codes.
#include <cstring>
#include <stdexcept>
#include <string>

#include "modes.h"
#include "tfhe_aes_host.h"

using namespace tfheaes;

struct tfh_keys {
    KeyPair kp;
};
struct tfh_ct {
    Ciphertext ct;
};
struct tfh_byte {
    FHEByte b;
};
struct tfh_state {
    State s;
};
struct tfh_key {
    Key k;
};

static thread_local std::string g_err;
extern "C" const char *tfh_last_error(void) { return g_err.c_str(); }

#define TFH_TRY(body)                    \
    try {                                \
        body;                            \
        return 0;                        \
    } catch (const std::invalid_argument &e) { \
        g_err = e.what();                \
        return TFA_EINVAL;               \
    } catch (const std::exception &e) {  \
        g_err = e.what();                \
        return TFA_EHIP;                 \
    }

static std::array<uint8_t, 16> arr16(const uint8_t *p) {
    std::array<uint8_t, 16> a;
    std::memcpy(a.data(), p, 16);
    return a;
}
static std::vector<Key> round_keys(tfh_key *const rk[11]) {
    std::vector<Key> v;
    for (int i = 0; i < 11; i++) {
        if (!rk[i]) throw std::invalid_argument("null round key");
        v.push_back(rk[i]->k);
    }
    return v;
}

extern "C" int tfh_gen_keys(uint64_t seed, int device, tfh_keys **out) {
    if (!out) return TFA_EINVAL;
    *out = nullptr;
    TFH_TRY({
        auto *k = new tfh_keys{gen_keys(seed, device)};
        *out = k;
    })
}
extern "C" void tfh_keys_free(tfh_keys *k) { delete k; }
extern "C" tfa_ctx *tfh_context(tfh_keys *k) { return k ? k->kp.server->context() : nullptr; }
extern "C" int tfh_flush(tfh_keys *k) { TFH_TRY(k->kp.server->flush()) }
extern "C" int tfh_set_xor_rebalance(tfh_keys *k, int on) {
    if (!k) return TFA_EINVAL;
    k->kp.server->set_xor_rebalance(on != 0);
    return 0;
}
extern "C" int tfh_host_stats(tfh_keys *k, uint64_t out[4]) {
    ServerKeyStats s = k->kp.server->stats();
    out[0] = s.gates;
    out[1] = s.stages;
    out[2] = s.flushes;
    out[3] = s.max_stage_width;
    return 0;
}
extern "C" int tfh_lwe_secret_key(tfh_keys *k, uint32_t *out) {
    const auto &sk = k->kp.client->lwe_secret_key();
    std::memcpy(out, sk.data(), sk.size() * sizeof(uint32_t));
    return 0;
}

extern "C" int tfh_encrypt(tfh_keys *k, int message, tfh_ct **out) { TFH_TRY(*out = new tfh_ct{k->kp.client->encrypt(message != 0)}) }
extern "C" int tfh_trivial(tfh_keys *k, int message, tfh_ct **out) { TFH_TRY(*out = new tfh_ct{k->kp.server->trivial_encrypt(message != 0)}) }
extern "C" int tfh_decrypt(tfh_keys *k, const tfh_ct *ct, int *message) { TFH_TRY(*message = k->kp.client->decrypt(ct->ct) ? 1 : 0) }
extern "C" int tfh_gate(tfh_keys *k, uint32_t op, const tfh_ct *a, const tfh_ct *b, const tfh_ct *c, tfh_ct **out) {
    TFH_TRY({
        ServerKey &sk = *k->kp.server;
        if (!a || (op <= TFA_MUX && !b) || (op == TFA_MUX && !c)) throw std::invalid_argument("missing gate operand");
        Ciphertext r;
        switch (op) {
        case TFA_AND: r = sk.and_(a->ct, b->ct); break;
        case TFA_NAND: r = sk.nand(a->ct, b->ct); break;
        case TFA_OR: r = sk.or_(a->ct, b->ct); break;
        case TFA_NOR: r = sk.nor(a->ct, b->ct); break;
        case TFA_XOR: r = sk.xor_(a->ct, b->ct); break;
        case TFA_XNOR: r = sk.xnor(a->ct, b->ct); break;
        case TFA_MUX: r = sk.mux(a->ct, b->ct, c->ct); break;
        case TFA_NOT: r = sk.not_(a->ct); break;
        case TFA_COPY: r = a->ct; break;
        default: throw std::invalid_argument("unknown gate opcode");
        }
        *out = new tfh_ct{r};
    })
}
extern "C" int tfh_ct_is_trivial(const tfh_ct *ct) { return ct && ct->ct.is_trivial(); }
extern "C" void tfh_ct_free(tfh_ct *ct) { delete ct; }

extern "C" int tfh_byte_from_u8_enc(tfh_keys *k, uint8_t value, tfh_byte **out) { TFH_TRY(*out = new tfh_byte{FHEByte::from_u8_enc(value, *k->kp.client)}) }
extern "C" int tfh_byte_from_u8_clear(tfh_keys *k, uint8_t value, tfh_byte **out) {
    TFH_TRY(*out = new tfh_byte{FHEByte::from_u8_clear(value, *k->kp.server)})
}
extern "C" int tfh_byte_decrypt_to_u8(tfh_keys *k, const tfh_byte *b, uint8_t *out) { TFH_TRY(*out = b->b.decrypt_to_u8(*k->kp.client)) }
extern "C" int tfh_byte_binary(tfh_keys *k, uint32_t op, const tfh_byte *a, const tfh_byte *b, tfh_byte **out) {
    TFH_TRY({
        ServerKey &sk = *k->kp.server;
        FHEByte r;
        if (op == TFA_XOR)
            r = a->b.xor_(b->b, sk);
        else if (op == TFA_AND)
            r = a->b.and_(b->b, sk);
        else if (op == TFA_OR)
            r = a->b.or_(b->b, sk);
        else
            throw std::invalid_argument("FHEByte supports xor, and, or");
        *out = new tfh_byte{r};
    })
}
extern "C" int tfh_byte_not(tfh_keys *k, const tfh_byte *a, tfh_byte **out) { TFH_TRY(*out = new tfh_byte{a->b.not_(*k->kp.server)}) }
extern "C" int tfh_byte_sub_byte(tfh_keys *k, const tfh_byte *a, int inverse, tfh_byte **out) {
    TFH_TRY(*out = new tfh_byte{inverse ? a->b.inv_sub_byte(*k->kp.server) : a->b.sub_byte(*k->kp.server)})
}
extern "C" int tfh_byte_mul_x_gf2(tfh_keys *k, const tfh_byte *a, tfh_byte **out) { TFH_TRY(*out = new tfh_byte{a->b.mul_x_gf2(*k->kp.server)}) }
extern "C" void tfh_byte_free(tfh_byte *b) { delete b; }

extern "C" int tfh_key_from_u8_enc(tfh_keys *k, const uint8_t key[16], tfh_key **out) { TFH_TRY(*out = new tfh_key{Key::from_u8_enc(arr16(key), *k->kp.client)}) }
extern "C" int tfh_key_from_u8_clear(tfh_keys *k, const uint8_t key[16], tfh_key **out) {
    TFH_TRY(*out = new tfh_key{Key::from_u8_clear(arr16(key), *k->kp.server)})
}
extern "C" int tfh_key_generate_round_keys(tfh_keys *k, const tfh_key *key, tfh_key *out[11]) {
    TFH_TRY({
        auto keys = key->k.generate_round_keys(*k->kp.server);
        for (int i = 0; i < 11; i++) out[i] = new tfh_key{keys[i]};
    })
}
extern "C" int tfh_key_decrypt_to_u8(tfh_keys *k, const tfh_key *key, uint8_t out[16]) {
    TFH_TRY({
        auto v = key->k.decrypt_to_u8(*k->kp.client);
        std::memcpy(out, v.data(), 16);
    })
}
extern "C" void tfh_key_free(tfh_key *key) { delete key; }
extern "C" int tfh_key_expansion_clear(const uint8_t key[16], uint8_t out[11][16]) {
    auto rk = key_expansion_clear(arr16(key));
    for (int i = 0; i < 11; i++) std::memcpy(out[i], rk[i].data(), 16);
    return 0;
}

extern "C" int tfh_state_from_u8_enc(tfh_keys *k, const uint8_t data[16], tfh_state **out) {
    TFH_TRY(*out = new tfh_state{State::from_u8_enc(arr16(data), *k->kp.client)})
}
extern "C" int tfh_state_clone(const tfh_state *s, tfh_state **out) { TFH_TRY(*out = new tfh_state{s->s}) }
extern "C" int tfh_state_apply(tfh_keys *k, tfh_state *s, uint32_t op) {
    TFH_TRY({
        ServerKey &sk = *k->kp.server;
        switch (op) {
        case TFH_SUB_BYTES: s->s.sub_bytes(sk); break;
        case TFH_INV_SUB_BYTES: s->s.inv_sub_bytes(sk); break;
        case TFH_SHIFT_ROWS: s->s.shift_rows(); break;
        case TFH_INV_SHIFT_ROWS: s->s.inv_shift_rows(); break;
        case TFH_MIX_COLUMNS: s->s.mix_columns(sk); break;
        case TFH_INV_MIX_COLUMNS: s->s.inv_mix_columns(sk); break;
        default: throw std::invalid_argument("unknown state operation");
        }
    })
}
extern "C" int tfh_state_xor_key_enc(tfh_keys *k, tfh_state *s, const tfh_key *key) { TFH_TRY(s->s.xor_key_enc(key->k, *k->kp.server)) }
extern "C" int tfh_state_xor_key_clear(tfh_keys *k, tfh_state *s, const uint8_t key[16]) { TFH_TRY(s->s.xor_key_clear(arr16(key), *k->kp.server)) }
extern "C" int tfh_state_xor_state(tfh_keys *k, tfh_state *s, const tfh_state *other) { TFH_TRY(s->s.xor_state(other->s, *k->kp.server)) }
extern "C" int tfh_state_decrypt_to_u8(tfh_keys *k, const tfh_state *s, uint8_t out[16]) {
    TFH_TRY({
        k->kp.server->flush();
        auto v = s->s.decrypt_to_u8(*k->kp.client);
        std::memcpy(out, v.data(), 16);
    })
}
extern "C" int tfh_state_read(tfh_keys *k, const tfh_state *s, uint32_t *out) {
    TFH_TRY({
        ServerKey &sk = *k->kp.server;
        const uint32_t words = sk.params().lwe_dimension + 1;
        std::vector<const Ciphertext *> cts;
        std::vector<size_t> pos;
        for (int i = 0; i < 16; i++)
            for (int b = 0; b < 8; b++) {
                const Ciphertext &c = s->s.bytes()[i].bit(b);
                if (c.is_trivial()) {  // noiseless encoding: zero mask, body = +-1/8
                    uint32_t *o = out + (size_t)(i * 8 + b) * words;
                    std::memset(o, 0, words * sizeof(uint32_t));
                    o[words - 1] = c.trivial_value() ? (1u << 29) : (7u << 29);
                } else {
                    cts.push_back(&c);
                    pos.push_back((size_t)i * 8 + b);
                }
            }
        std::vector<uint32_t> tmp(cts.size() * words);
        sk.read_many(cts, tmp.data());
        for (size_t j = 0; j < cts.size(); j++) std::memcpy(out + pos[j] * words, tmp.data() + j * words, words * sizeof(uint32_t));
    })
}
extern "C" int tfh_states_read_device(tfh_keys *k, const tfh_state *const *states, uint32_t n, void *d_out) {
    TFH_TRY({
        std::vector<const Ciphertext *> cts;
        for (uint32_t s = 0; s < n; s++)
            for (int i = 0; i < 16; i++)
                for (int b = 0; b < 8; b++) cts.push_back(&states[s]->s.bytes()[i].bit(b));
        k->kp.server->read_many_device(cts, d_out);
    })
}
extern "C" int tfh_state_import(tfh_keys *k, const uint32_t *lwe, tfh_state **out) {
    TFH_TRY({
        ServerKey &sk = *k->kp.server;
        const uint32_t words = sk.params().lwe_dimension + 1;
        auto *st = new tfh_state{};
        for (int i = 0; i < 16; i++)
            for (int b = 0; b < 8; b++) st->s.bytes()[i].bit(b) = sk.import_lwe(lwe + (size_t)(i * 8 + b) * words);
        *out = st;
    })
}
extern "C" void tfh_state_free(tfh_state *s) { delete s; }

extern "C" int tfh_ecb(tfh_keys *k, tfh_key *const rk[11], tfh_state *const *states, uint32_t n, int decrypt) {
    TFH_TRY({
        ECB ecb(round_keys(rk));
        for (uint32_t i = 0; i < n; i++) decrypt ? ecb.decrypt(states[i]->s, *k->kp.server) : ecb.encrypt(states[i]->s, *k->kp.server);
    })
}
static std::vector<State> gather(tfh_state *const *states, uint32_t n) {
    std::vector<State> v;
    for (uint32_t i = 0; i < n; i++) v.push_back(states[i]->s);
    return v;
}
static void scatter(const std::vector<State> &v, tfh_state *const *states) {
    for (size_t i = 0; i < v.size(); i++) states[i]->s = v[i];
}
extern "C" int tfh_cbc(tfh_keys *k, tfh_key *const rk[11], const tfh_state *iv, tfh_state *const *states, uint32_t n, int decrypt) {
    TFH_TRY({
        if (n > 255) throw std::invalid_argument("block count is a u8 in the reference");
        CBC cbc(round_keys(rk), iv->s, (uint8_t)n);
        auto v = gather(states, n);
        decrypt ? cbc.decrypt(v, *k->kp.server) : cbc.encrypt(v, *k->kp.server);
        scatter(v, states);
    })
}
extern "C" int tfh_ofb(tfh_keys *k, tfh_key *const rk[11], const tfh_state *iv, tfh_state *const *states, uint32_t n, int decrypt) {
    TFH_TRY({
        if (n > 255) throw std::invalid_argument("block count is a u8 in the reference");
        OFB ofb(round_keys(rk), iv->s, (uint8_t)n);
        auto v = gather(states, n);
        decrypt ? ofb.decrypt(v, *k->kp.server) : ofb.encrypt(v, *k->kp.server);
        scatter(v, states);
    })
}
extern "C" int tfh_ctr(tfh_keys *k, tfh_key *const rk[11], tfh_state *const *counters, tfh_state *const *states, uint32_t n, int decrypt) {
    TFH_TRY({
        if (n > 255) throw std::invalid_argument("block count is a u8 in the reference");
        CTR ctr(round_keys(rk), gather(counters, n), (uint8_t)n);
        auto v = gather(states, n);
        decrypt ? ctr.decrypt(v, *k->kp.server) : ctr.encrypt(v, *k->kp.server);
        scatter(v, states);
    })
}

static long emit(const std::string &s, char *buf, size_t cap) {
    if (s.size() + 1 > cap) return (long)(s.size() + 1);
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
extern "C" long tfh_reduce_mux(const uint8_t *truth_table, uint32_t len, char *buf, size_t cap) {
    try {
        std::vector<bool> v(len);
        for (uint32_t i = 0; i < len; i++) v[i] = truth_table[i] != 0;
        return emit(BooleanExpr::reduce_mux(BooleanExpr::from_bool_vec(v)).to_string(), buf, cap);
    } catch (const std::exception &e) {
        g_err = e.what();
        return TFA_EINVAL;
    }
}
extern "C" long tfh_sbox_circuit_dump(int inverse, char *buf, size_t cap) {
    const StagedCircuit &c = inverse ? inv_s_box_circuit() : s_box_circuit();
    std::string s;
    for (size_t st = 0; st < c.stages.size(); st++)
        for (const auto &e : c.stages[st]) s += std::to_string(st) + "\t" + e.to_string() + "\n";
    for (const auto &r : c.roots) s += "root\t" + r.to_string() + "\n";
    return emit(s, buf, cap);
}
