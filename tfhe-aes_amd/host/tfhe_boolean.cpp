#include "tfhe_boolean.h"

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <stdexcept>

namespace tfheaes {

namespace detail {
struct SlotRef {
    ServerKey *owner;
    std::weak_ptr<int> alive;
    uint32_t slot;
    uint32_t level;
    uint64_t epoch;
    SlotRef(ServerKey *o, const std::shared_ptr<int> &a, uint32_t s, uint32_t l, uint64_t e) : owner(o), alive(a), slot(s), level(l), epoch(e) {}
    SlotRef(const SlotRef &) = delete;  // exactly one owner per slot: a copy would release it twice
    SlotRef &operator=(const SlotRef &) = delete;
    ~SlotRef() {
        if (!alive.expired()) owner->release_slot(slot);
    }
};
struct Enc {
    std::vector<uint32_t> words;    // n+1 host words while not yet on a device
    std::shared_ptr<SlotRef> slot;  // set on first use by a ServerKey
};
}  // namespace detail

static void check(tfa_ctx *ctx, int rc, const char *what) {
    if (rc != TFA_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + tfa_last_error(ctx));
}

Ciphertext Ciphertext::trivial(bool v) {
    Ciphertext c;
    c.kind_ = Kind::Trivial;
    c.value_ = v;
    return c;
}

// ---------------------------------------------------------------------------------------------
ClientKey::ClientKey(const tfa_params &p, uint64_t seed, std::vector<uint32_t> lwe_sk, std::vector<uint32_t> glwe_sk)
    : p_(p), enc_seed_(seed ? seed * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull : 0), lwe_sk_(std::move(lwe_sk)), glwe_sk_(std::move(glwe_sk)) {}

Ciphertext ClientKey::encrypt(bool message) const {
    Ciphertext c;
    c.kind_ = Ciphertext::Kind::Encrypted;
    c.enc_ = std::make_shared<detail::Enc>();
    c.enc_->words.resize(p_.lwe_dimension + 1);
    uint8_t bit = message ? 1 : 0;
    // a fresh stream per ciphertext; seed 0 (OS entropy) stays 0 so every call draws anew
    uint64_t s = enc_seed_ ? enc_seed_ + (++enc_counter_) * 0xD1B54A32D192ED03ull : 0;
    if (enc_seed_ && s == 0) s = 1;
    int rc = tfa_client_encrypt(&p_, lwe_sk_.data(), s, &bit, 1, c.enc_->words.data());
    if (rc != TFA_OK) throw std::runtime_error("tfa_client_encrypt failed");
    return c;
}

bool ClientKey::decrypt(const Ciphertext &ct) const {
    if (ct.kind_ == Ciphertext::Kind::Trivial) return ct.value_;
    if (ct.kind_ == Ciphertext::Kind::Empty) throw std::invalid_argument("decrypt of an empty ciphertext");
    std::vector<uint32_t> words;
    if (!ct.enc_->slot) {
        words = ct.enc_->words;
        if (ct.neg_)
            for (auto &w : words) w = 0u - w;
    } else {
        if (ct.enc_->slot->alive.expired()) throw std::invalid_argument("decrypt of a ciphertext whose ServerKey is gone");
        words = ct.enc_->slot->owner->read(ct);
    }
    uint8_t bit = 0;
    if (tfa_client_decrypt(&p_, lwe_sk_.data(), words.data(), 1, &bit) != TFA_OK) throw std::runtime_error("tfa_client_decrypt failed");
    return bit != 0;
}

// ---------------------------------------------------------------------------------------------
ServerKey::ServerKey(const tfa_params &p, const uint32_t *bsk, const uint32_t *ksk, int device) : p_(p), alive_(std::make_shared<int>(0)) {
    int rc = tfa_ctx_create(&p_, device, &ctx_);
    if (rc != TFA_OK) {
        std::string msg = ctx_ ? tfa_last_error(ctx_) : "invalid parameters or device";
        if (ctx_) tfa_ctx_destroy(ctx_);
        ctx_ = nullptr;
        throw std::runtime_error("tfa_ctx_create failed (" + std::to_string(rc) + "): " + msg);
    }
    rc = tfa_keys_upload(ctx_, bsk, ksk);
    if (rc != TFA_OK) {
        std::string msg = tfa_last_error(ctx_);
        tfa_ctx_destroy(ctx_);
        ctx_ = nullptr;
        throw std::runtime_error("tfa_keys_upload failed: " + msg);
    }
    if (const char *e = std::getenv("TFHE_AES_XOR_REBALANCE")) xor_rebalance_ = std::atoi(e) != 0;
    pool_slots_ = 1u << 17;  // 131072 slots = 424 MB: one AES block in flight (72 k slots) without a reallocation
    check(ctx_, tfa_pool_reserve(ctx_, pool_slots_), "tfa_pool_reserve");
}

ServerKey::~ServerKey() {
    alive_.reset();  // outstanding handles stop calling back
    if (ctx_) tfa_ctx_destroy(ctx_);
}

uint32_t ServerKey::alloc_slot() {
    if (!free_.empty()) {
        uint32_t s = free_.back();
        free_.pop_back();
        return s;
    }
    if (next_fresh_ == pool_slots_) {
        uint64_t grown = (uint64_t)pool_slots_ * 2;
        if (grown > 0xFFFFFFF0ull) throw std::runtime_error("ciphertext pool exhausted");
        check(ctx_, tfa_pool_reserve(ctx_, (uint32_t)grown), "tfa_pool_reserve");
        pool_slots_ = (uint32_t)grown;
    }
    return next_fresh_++;
}

// A released slot may still be read by recorded, not yet submitted gates: recycle it after the flush.
void ServerKey::release_slot(uint32_t slot) { deferred_free_.push_back(slot); }

ServerKey::Resolved ServerKey::resolve(const Ciphertext &ct) {
    if (ct.kind_ != Ciphertext::Kind::Encrypted) throw std::invalid_argument("gate operand is not an encrypted ciphertext");
    detail::Enc &e = *ct.enc_;
    if (!e.slot) {
        // first use on the device: give it a slot and queue the upload (seen by all copies of the handle)
        e.slot = std::make_shared<detail::SlotRef>(this, alive_, alloc_slot(), 0u, epoch_);
        upload_slots_.push_back(e.slot->slot);
        upload_words_.insert(upload_words_.end(), e.words.begin(), e.words.end());
        e.words.clear();
        e.words.shrink_to_fit();
    }
    if (e.slot->owner != this) throw std::invalid_argument("ciphertext belongs to another ServerKey");
    uint32_t lvl = e.slot->epoch == epoch_ ? e.slot->level : 0;
    return Resolved{e.slot->slot, lvl, ct.neg_};
}

Ciphertext ServerKey::record(uint32_t op, const Resolved *in, int n_in) {
    uint32_t lvl = 0, neg = 0;
    for (int i = 0; i < n_in; i++) {
        lvl = std::max(lvl, in[i].level);
        if (in[i].neg) neg |= 1u << i;
    }
    lvl += 1;
    auto ref = std::make_shared<detail::SlotRef>(this, alive_, alloc_slot(), lvl, epoch_);
    tfa_gate g{op, in[0].slot, n_in > 1 ? in[1].slot : 0u, n_in > 2 ? in[2].slot : 0u, ref->slot, neg};
    if (levels_.size() < lvl) levels_.resize(lvl);
    levels_[lvl - 1].push_back(g);
    pending_count_++;
    Ciphertext c;
    c.kind_ = Ciphertext::Kind::Encrypted;
    c.enc_ = std::make_shared<detail::Enc>();
    c.enc_->slot = std::move(ref);
    return c;
}

Ciphertext ServerKey::not_(const Ciphertext &a) const {
    Ciphertext c = a;
    if (c.kind_ == Ciphertext::Kind::Trivial)
        c.value_ = !c.value_;
    else if (c.kind_ == Ciphertext::Kind::Empty)
        throw std::invalid_argument("not of an empty ciphertext");
    else
        c.neg_ = !c.neg_;  // LWE negation is free: folded into the consumer's linear combination
    return c;
}

// Binary gates with tfhe-rs' treatment of trivial operands (a trivial operand never costs a bootstrap).
Ciphertext ServerKey::binary(uint32_t op, const Ciphertext &a, const Ciphertext &b) {
    const bool ta = a.is_trivial(), tb = b.is_trivial();
    if (ta && tb) {
        bool x = a.value_, y = b.value_, r = false;
        switch (op) {
        case TFA_AND: r = x && y; break;
        case TFA_NAND: r = !(x && y); break;
        case TFA_OR: r = x || y; break;
        case TFA_NOR: r = !(x || y); break;
        case TFA_XOR: r = x != y; break;
        default: r = x == y; break;
        }
        return Ciphertext::trivial(r);
    }
    if (ta || tb) {
        const Ciphertext &e = ta ? b : a;
        const bool t = ta ? a.value_ : b.value_;
        switch (op) {
        case TFA_AND: return t ? e : Ciphertext::trivial(false);
        case TFA_NAND: return t ? not_(e) : Ciphertext::trivial(true);
        case TFA_OR: return t ? Ciphertext::trivial(true) : e;
        case TFA_NOR: return t ? Ciphertext::trivial(false) : not_(e);
        case TFA_XOR: return t ? not_(e) : e;
        default: return t ? e : not_(e);
        }
    }
    auto_flush();
    Resolved in[2] = {resolve(a), resolve(b)};
    return record(op, in, 2);
}
Ciphertext ServerKey::and_(const Ciphertext &a, const Ciphertext &b) { return binary(TFA_AND, a, b); }
Ciphertext ServerKey::nand(const Ciphertext &a, const Ciphertext &b) { return binary(TFA_NAND, a, b); }
Ciphertext ServerKey::or_(const Ciphertext &a, const Ciphertext &b) { return binary(TFA_OR, a, b); }
Ciphertext ServerKey::nor(const Ciphertext &a, const Ciphertext &b) { return binary(TFA_NOR, a, b); }
Ciphertext ServerKey::xor_(const Ciphertext &a, const Ciphertext &b) { return binary(TFA_XOR, a, b); }
Ciphertext ServerKey::xnor(const Ciphertext &a, const Ciphertext &b) { return binary(TFA_XNOR, a, b); }

Ciphertext ServerKey::mux(const Ciphertext &cond, const Ciphertext &then_ct, const Ciphertext &else_ct) {
    if (cond.is_trivial()) return cond.value_ ? then_ct : else_ct;
    if (then_ct.is_trivial()) return then_ct.value_ ? or_(cond, else_ct) : and_(not_(cond), else_ct);
    if (else_ct.is_trivial()) return else_ct.value_ ? or_(then_ct, not_(cond)) : and_(cond, then_ct);
    auto_flush();
    Resolved in[3] = {resolve(cond), resolve(then_ct), resolve(else_ct)};
    return record(TFA_MUX, in, 3);
}

// Every recorded gate holds a fresh pool slot until the flush, so an unbounded recording (the command line records up to 255
// blocks before it reads a result) would size the pool by the job, not by the GPU: past kAutoFlushGates pending gates the
// recording is run before the next gate is taken.  Results do not change; only gates on either side of the cut stop sharing stages.
void ServerKey::auto_flush() {
    if (pending_count_ >= kAutoFlushGates) flush();
}

void ServerKey::upload_pending() {
    if (upload_slots_.empty()) return;
    check(ctx_, tfa_pool_write(ctx_, upload_slots_.data(), (uint32_t)upload_slots_.size(), upload_words_.data()), "tfa_pool_write");
    upload_slots_.clear();
    upload_words_.clear();
}

// Re-associates chains of xor / xnor gates whose intermediate results nobody can observe: an intermediate whose handle is gone
// (its slot is on the deferred free list) and that feeds exactly one recorded gate, itself an xor / xnor.  Such a chain
// x1 ^ x2 ^ ... ^ xn is one n-input parity: it is rebuilt with the same n - 1 xor gates, on the same slots, as the tree of least
// depth for the levels at which its operands become available (the two earliest operands are combined first), and every
// gate of the recording is levelled again.  MixColumns followed by AddRoundKey (state.rs:72-118, 217-224: five dependent xor
// layers per output bit) becomes three levels.  Operand negations and xnors are gathered into one negation at the tree's root.
// Gate counts, gate kinds and decrypted results do not change; set_xor_rebalance(false) keeps the recorded association.
void ServerKey::rebalance_xor_chains() {
    std::vector<tfa_gate> gates;
    gates.reserve(pending_count_);
    for (auto &lv : levels_) gates.insert(gates.end(), lv.begin(), lv.end());  // level order is a topological order
    const size_t n = gates.size();
    std::vector<int32_t> producer(next_fresh_, -1), user(n, -1);
    std::vector<uint32_t> uses(n, 0), level(next_fresh_, 0);
    std::vector<uint8_t> dead(next_fresh_, 0), absorbed(n, 0);
    for (uint32_t s : deferred_free_) dead[s] = 1;
    for (size_t i = 0; i < n; i++) producer[gates[i].out] = (int32_t)i;
    auto operands = [](const tfa_gate &g) { return g.op == TFA_MUX ? 3 : 2; };
    auto operand = [](const tfa_gate &g, int k) { return k == 0 ? g.in0 : k == 1 ? g.in1 : g.in2; };
    auto is_xor = [](const tfa_gate &g) { return g.op == TFA_XOR || g.op == TFA_XNOR; };
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < operands(gates[i]); k++) {
            const int32_t p = producer[operand(gates[i], k)];
            if (p >= 0) {
                uses[p]++;
                user[p] = (int32_t)i;
            }
        }
    bool any = false;
    for (size_t i = 0; i < n; i++) {
        absorbed[i] = is_xor(gates[i]) && dead[gates[i].out] && uses[i] == 1 && is_xor(gates[user[i]]);
        any |= absorbed[i] != 0;
    }
    if (!any) return;
    std::vector<std::vector<tfa_gate>> out;
    auto put = [&](const tfa_gate &g, uint32_t lvl) {
        if (out.size() < lvl) out.resize(lvl);
        out[lvl - 1].push_back(g);
        level[g.out] = lvl;
    };
    struct Item {
        uint32_t level, seq, slot;
        bool operator>(const Item &o) const { return level != o.level ? level > o.level : seq > o.seq; }
    };
    std::vector<Item> heap;
    std::vector<uint32_t> spare, stack;
    for (size_t i = 0; i < n; i++) {
        if (absorbed[i]) continue;
        const tfa_gate &g = gates[i];
        bool root = false;
        if (is_xor(g))
            for (int k = 0; k < 2; k++) {
                const int32_t p = producer[operand(g, k)];
                root |= p >= 0 && absorbed[p];
            }
        if (!root) {
            uint32_t lvl = 0;
            for (int k = 0; k < operands(g); k++) lvl = std::max(lvl, level[operand(g, k)]);
            put(g, lvl + 1);
            continue;
        }
        heap.clear();
        spare.clear();
        stack.assign(1, (uint32_t)i);
        uint32_t parity = 0, seq = 0;
        while (!stack.empty()) {
            const tfa_gate &x = gates[stack.back()];
            stack.pop_back();
            parity ^= (x.op == TFA_XNOR ? 1u : 0u) ^ (x.neg & 1u) ^ ((x.neg >> 1) & 1u);
            for (int k = 0; k < 2; k++) {
                const uint32_t s = operand(x, k);
                const int32_t p = producer[s];
                if (p >= 0 && absorbed[p]) {
                    spare.push_back(s);
                    stack.push_back((uint32_t)p);
                } else {
                    heap.push_back(Item{level[s], seq++, s});
                }
            }
        }
        std::make_heap(heap.begin(), heap.end(), std::greater<Item>());
        auto pop = [&]() {
            std::pop_heap(heap.begin(), heap.end(), std::greater<Item>());
            Item it = heap.back();
            heap.pop_back();
            return it;
        };
        while (heap.size() > 2) {
            const Item a = pop(), b = pop();
            const uint32_t slot = spare.back(), lvl = std::max(a.level, b.level) + 1;
            spare.pop_back();
            put(tfa_gate{TFA_XOR, a.slot, b.slot, 0u, slot, 0u}, lvl);
            heap.push_back(Item{lvl, seq++, slot});
            std::push_heap(heap.begin(), heap.end(), std::greater<Item>());
        }
        const Item a = pop(), b = pop();
        put(tfa_gate{TFA_XOR, a.slot, b.slot, 0u, g.out, parity}, std::max(a.level, b.level) + 1);
    }
    levels_.swap(out);
}

void ServerKey::flush() {
    upload_pending();
    if (xor_rebalance_ && pending_count_ > 0) rebalance_xor_chains();
    for (auto &lv : levels_) {
        if (lv.empty()) continue;
        check(ctx_, tfa_stage_submit(ctx_, lv.data(), (uint32_t)lv.size()), "tfa_stage_submit");
        stats_.gates += lv.size();
        stats_.stages++;
        stats_.max_stage_width = std::max<uint64_t>(stats_.max_stage_width, lv.size());
    }
    check(ctx_, tfa_sync(ctx_), "tfa_sync");
    levels_.clear();
    pending_count_ = 0;
    epoch_++;
    stats_.flushes++;
    free_.insert(free_.end(), deferred_free_.begin(), deferred_free_.end());
    deferred_free_.clear();
}

void ServerKey::read_many(const std::vector<const Ciphertext *> &cts, uint32_t *out) {
    const uint32_t words = p_.lwe_dimension + 1;
    std::vector<uint32_t> slots;
    std::vector<Resolved> res;
    for (auto *c : cts) res.push_back(resolve(*c));
    flush();
    for (auto &r : res) slots.push_back(r.slot);
    if (!slots.empty()) check(ctx_, tfa_pool_read(ctx_, slots.data(), (uint32_t)slots.size(), out), "tfa_pool_read");
    for (size_t i = 0; i < res.size(); i++)
        if (res[i].neg)
            for (uint32_t w = 0; w < words; w++) out[i * words + w] = 0u - out[i * words + w];
}

void ServerKey::read_many_device(const std::vector<const Ciphertext *> &cts, void *d_out) {
    const uint32_t words = p_.lwe_dimension + 1;
    std::vector<Ciphertext> noiseless;  // trivial ciphertexts leave as noiseless encryptions: zero mask, body = +-1/8
    noiseless.reserve(cts.size());
    std::vector<Resolved> res;
    for (auto *c : cts) {
        if (c->is_trivial()) {
            std::vector<uint32_t> row(words, 0u);
            row[words - 1] = c->trivial_value() ? (1u << 29) : (7u << 29);
            noiseless.push_back(import_lwe(row.data()));
            res.push_back(resolve(noiseless.back()));
        } else {
            res.push_back(resolve(*c));
        }
    }
    flush();
    std::vector<uint32_t> slots;
    std::vector<uint8_t> neg;
    for (auto &r : res) {
        slots.push_back(r.slot);
        neg.push_back(r.neg ? 1 : 0);
    }
    if (!slots.empty()) check(ctx_, tfa_pool_read_device(ctx_, slots.data(), neg.data(), (uint32_t)slots.size(), d_out), "tfa_pool_read_device");
}

std::vector<uint32_t> ServerKey::read(const Ciphertext &ct) {
    std::vector<uint32_t> out(p_.lwe_dimension + 1);
    read_many({&ct}, out.data());
    return out;
}

Ciphertext ServerKey::import_lwe(const uint32_t *lwe) {
    Ciphertext c;
    c.kind_ = Ciphertext::Kind::Encrypted;
    c.enc_ = std::make_shared<detail::Enc>();
    c.enc_->words.assign(lwe, lwe + p_.lwe_dimension + 1);
    return c;
}

KeyPair gen_keys(uint64_t seed, int device) {
    tfa_params p;
    tfa_default_params(&p);
    std::vector<uint32_t> lwe_sk(p.lwe_dimension), glwe_sk((size_t)p.glwe_dimension * p.polynomial_size);
    std::vector<uint32_t> bsk(tfa_bsk_words(&p)), ksk(tfa_ksk_words(&p));
    if (tfa_client_keygen(&p, seed, lwe_sk.data(), glwe_sk.data(), bsk.data(), ksk.data()) != TFA_OK)
        throw std::runtime_error("tfa_client_keygen failed");
    KeyPair kp;
    kp.server = std::make_unique<ServerKey>(p, bsk.data(), ksk.data(), device);
    kp.client = std::make_unique<ClientKey>(p, seed, std::move(lwe_sk), std::move(glwe_sk));
    return kp;
}

}  // namespace tfheaes
