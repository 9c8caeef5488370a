// Command-line driver with the flags and printed lines of the reference binary
// (/root/reference/tfhe-aes/src/main.rs:15-32 flags, :127-400 the four mode tests): random blocks are
// encrypted and decrypted homomorphically on the GPU and compared with a clear AES-128 (the reference compares with the
// `aes` crate, main.rs:136-141; here the clear cipher is the one of host/sbox.cpp's tables, which tests/test_aes_clear.py
// pins to FIPS-197 and SP 800-38A).  Additional lines: key generation time and the engine's gate / stage counts.
//   tfhe-aes-hip -k <32 hex> -i <32 hex> [-n outputs] [-x] [-m ECB|CBC|CTR|OFB] [--seed S] [--device D]
// --seed / --device have no counterpart in the reference (its keys come from OS entropy, it has no GPU).
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../host/modes.h"
#include "../host/sbox.h"
#include "../host/state.h"
#include "../host/tfhe_boolean.h"

using namespace tfheaes;
using Block = std::array<uint8_t, 16>;
using Clock = std::chrono::steady_clock;

static double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

static Block parse_hex_16(const std::string &s, const char *what) {
    if (s.size() != 32) throw std::invalid_argument(std::string("Invalid ") + what + " format: must be 32 hex characters (16 bytes), it is currently " +
                                                    std::to_string(s.size()) + " characters.");
    Block b{};
    for (int i = 0; i < 16; i++) {
        unsigned v;
        if (sscanf(s.c_str() + 2 * i, "%2x", &v) != 1) throw std::invalid_argument(std::string("Invalid ") + what + " format: failed to decode hex");
        b[i] = (uint8_t)v;
    }
    return b;
}

// clear AES-128 from the same tables the encrypted circuits are generated from
static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x & 0x80) ? 0x1b : 0)); }
static Block aes_encrypt_clear(const Block &in, const std::array<Block, 11> &rk) {
    Block s = in;
    auto add = [&](int r) {
        for (int i = 0; i < 16; i++) s[i] ^= rk[r][i];
    };
    add(0);
    for (int r = 1; r <= 10; r++) {
        for (auto &b : s) b = S_BOX_DATA[b];
        Block t = s;
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++) s[4 * c + row] = t[4 * ((c + row) & 3) + row];  // ShiftRows, column-major bytes
        if (r < 10) {
            for (int c = 0; c < 4; c++) {
                uint8_t *col = &s[4 * c];
                const uint8_t a0 = col[0], a1 = col[1], a2 = col[2], a3 = col[3], all = a0 ^ a1 ^ a2 ^ a3;
                col[0] = a0 ^ all ^ xtime(a0 ^ a1);
                col[1] = a1 ^ all ^ xtime(a1 ^ a2);
                col[2] = a2 ^ all ^ xtime(a2 ^ a3);
                col[3] = a3 ^ all ^ xtime(a3 ^ a0);
            }
        }
        add(r);
    }
    return s;
}
static Block xor_blocks(const Block &a, const Block &b) {
    Block r;
    for (int i = 0; i < 16; i++) r[i] = a[i] ^ b[i];
    return r;
}
// counter k of the reference (main.rs:443-466): the IV's first 8 bytes as nonce, the last 8 a big-endian count from 0
static Block add_counter(const Block &iv, uint64_t k) {
    Block r = iv;
    for (int i = 15; i >= 8; i--) {
        r[i] = (uint8_t)k;
        k >>= 8;
    }
    return r;
}

static std::vector<Key> key_expansion(const Block &key, bool offline, ServerKey &sk, const ClientKey &ck) {
    // the two lines of tfhe-aes/src/main.rs:397-415
    printf("---Key Expansion (%s)---\n", offline ? "offline" : "online");
    std::vector<Key> keys;
    const auto t0 = Clock::now();
    if (offline) {
        for (const auto &k : key_expansion_clear(key)) keys.push_back(Key::from_u8_enc(k, ck));
    } else {
        const auto rk = Key::from_u8_enc(key, ck).generate_round_keys(sk);
        sk.flush();
        keys.assign(rk.begin(), rk.end());
    }
    printf("AES key expansion took: %.3fs\n", seconds_since(t0));
    return keys;
}
static std::vector<State> to_states(const std::vector<Block> &blocks, const ClientKey &ck) {
    std::vector<State> v;
    for (const auto &b : blocks) v.push_back(State::from_u8_enc(b, ck));
    return v;
}
static void expect_equal(const std::vector<State> &got, const std::vector<Block> &want, const ClientKey &ck, const char *what) {
    for (size_t i = 0; i < got.size(); i++)
        if (got[i].decrypt_to_u8(ck) != want[i]) throw std::runtime_error(std::string(what) + ": block " + std::to_string(i) + " differs from the clear AES");
}

int main(int argc, char **argv) {
    int n = 1, device = 0;
    uint64_t seed = 0;
    std::string iv_s, key_s, mode = "ECB";
    bool offline = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) {
                fprintf(stderr, "error: a value is required for '%s'\n", a.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-n" || a == "--number-of-outputs") n = atoi(val().c_str());
        else if (a == "-i" || a == "--iv") iv_s = val();
        else if (a == "-k" || a == "--key") key_s = val();
        else if (a == "-x" || a == "--key-expansion-offline") offline = true;
        else if (a == "-m" || a == "--mode") mode = val();
        else if (a == "--seed") seed = strtoull(val().c_str(), nullptr, 0);
        else if (a == "--device") device = atoi(val().c_str());
        else {
            fprintf(stderr, "Usage: %s -k <KEY> -i <IV> [-n <NUMBER_OF_OUTPUTS>] [-x] [-m <MODE>] [--seed S] [--device D]\n", argv[0]);
            return a == "-h" || a == "--help" ? 0 : 2;
        }
    }
    if (key_s.empty() || iv_s.empty() || n < 1 || n > 255) {
        fprintf(stderr, "error: --key and --iv are required, --number-of-outputs is 1..255\n");
        return 2;
    }
    printf("Number of Outputs: %d\nIV: %s\nKey: %s\nKey Expansion Offline: %s\nMode: %s\n", n, iv_s.c_str(), key_s.c_str(), offline ? "true" : "false",
           mode.c_str());
    try {
        const Block key = parse_hex_16(key_s, "key"), iv = parse_hex_16(iv_s, "IV");
        if (mode != "ECB" && mode != "CBC" && mode != "CTR" && mode != "OFB") throw std::invalid_argument("Invalid Mode format: Invalid mode: " + mode);
        std::mt19937_64 rng(seed ? seed : std::random_device{}());
        std::vector<Block> blocks(n);
        for (auto &b : blocks)
            for (auto &x : b) x = (uint8_t)rng();
        const auto rk_clear = key_expansion_clear(key);

        // expected ciphertexts from the clear AES
        std::vector<Block> expected(n);
        Block chain = iv;
        for (int i = 0; i < n; i++) {
            if (mode == "ECB") expected[i] = aes_encrypt_clear(blocks[i], rk_clear);
            else if (mode == "CBC") chain = expected[i] = aes_encrypt_clear(xor_blocks(blocks[i], chain), rk_clear);
            else if (mode == "OFB") {
                chain = aes_encrypt_clear(chain, rk_clear);
                expected[i] = xor_blocks(blocks[i], chain);
            } else expected[i] = xor_blocks(blocks[i], aes_encrypt_clear(add_counter(iv, (uint64_t)i), rk_clear));
        }

        auto t0 = Clock::now();
        KeyPair kp = gen_keys(seed, device);
        printf("Key generation and upload: %.3fs\n", seconds_since(t0));
        ServerKey &sk = *kp.server;
        const ClientKey &ck = *kp.client;
        // banners as the reference prints them: "---Testing ECB mode---" (main.rs:134), "Testing CBC mode" (:206, :270, :336)
        if (mode == "ECB")
            printf("---Testing ECB mode---\n");
        else
            printf("Testing %s mode\n", mode.c_str());
        const std::vector<Key> keys = key_expansion(key, offline, sk, ck);
        printf("---Begin Encryption---\n");
        t0 = Clock::now();
        std::vector<State> states = to_states(blocks, ck);
        printf("Conversion to FHE Time Taken: %.3fs\n", seconds_since(t0));

        std::vector<State> counters;
        if (mode == "CTR")
            for (int i = 0; i < n; i++) counters.push_back(State::from_u8_enc(add_counter(iv, (uint64_t)i), ck));
        const State iv_state = State::from_u8_enc(iv, ck);
        auto run = [&](bool decrypt) {
            if (mode == "ECB") {
                ECB ecb(keys);
                for (auto &s : states) decrypt ? ecb.decrypt(s, sk) : ecb.encrypt(s, sk);  // recorded together, run in shared stages
            } else if (mode == "CBC") {
                CBC m(keys, iv_state, (uint8_t)n);
                decrypt ? m.decrypt(states, sk) : m.encrypt(states, sk);
            } else if (mode == "OFB") {
                OFB m(keys, iv_state, (uint8_t)n);
                decrypt ? m.decrypt(states, sk) : m.encrypt(states, sk);
            } else {
                CTR m(keys, counters, (uint8_t)n);
                decrypt ? m.decrypt(states, sk) : m.encrypt(states, sk);
            }
            sk.flush();
        };
        t0 = Clock::now();
        run(false);
        printf("AES of #%d outputs computed in: %.3fs\n", n, seconds_since(t0));
        expect_equal(states, expected, ck, "encryption");
        printf("---Decryption---\n");
        t0 = Clock::now();
        run(true);
        printf("AES of #%d outputs decrypted in: %.3fs\n", n, seconds_since(t0));
        expect_equal(states, blocks, ck, "decryption");
        const ServerKeyStats st = sk.stats();
        printf("%s mode test passed\n", mode.c_str());
        printf("engine: %llu gates in %llu stages, widest %llu\n", (unsigned long long)st.gates, (unsigned long long)st.stages,
               (unsigned long long)st.max_stage_width);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
