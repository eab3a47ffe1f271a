// synth_oracle.hpp — closed-form membership test of the SYNTHETIC 9-mer index (DESIGN.md
// "Synthetic workload").  TEST INFRASTRUCTURE ONLY (see umgap_oracle.hpp).  This is an
// independent restatement of the spec the device-side generator (umgap_amd/csrc/synth.hip)
// follows, so the oracle can answer `get(key)` for an 8 G-key index without holding a table:
//
//   M = 20^9;  pi(i) = (a*i + b) mod M   (a odd, a mod 5 != 0, a < 2^25  => bijection on [0,M))
//   key_i   = the 9 base-20 digits of pi(i), most significant first, each written as a letter
//             of "ACDEFGHIKLMNPQRSTVWY", packed 5 bit/letter (code = letter-'A'),
//             key45 = sum code_j << (5*(8-j))
//   member  : key is a member iff all 9 letters are in the alphabet and pi^-1(x) < n_keys
//   value   : taxon_of_row[ pi^-1(x) mod n_tax ]   (row = TSV row of the synthetic taxonomy)
#pragma once
#include <cstdint>

struct SynthIndexParams {
    uint64_t n_keys;
    uint64_t n_tax;
    uint64_t mul_a;
    uint64_t add_b;
};

static const uint64_t SYNTH_M = 512000000000ull;  // 20^9
static const char SYNTH_ALPHABET[21] = "ACDEFGHIKLMNPQRSTVWY";

inline uint64_t synth_modinv(uint64_t a, uint64_t m) {
    __int128 t = 0, newt = 1, r = m, newr = a;
    while (newr != 0) {
        __int128 q = r / newr;
        __int128 tmp = t - q * newt; t = newt; newt = tmp;
        tmp = r - q * newr; r = newr; newr = tmp;
    }
    if (t < 0) t += m;
    return (uint64_t)t;
}

inline uint64_t synth_key_of_member(const SynthIndexParams& p, uint64_t i) {
    uint64_t x = (uint64_t)(((unsigned __int128)p.mul_a * i + p.add_b) % SYNTH_M);
    uint64_t key = 0;
    for (int j = 8; j >= 0; j--) {  // least significant digit is the LAST letter
        uint64_t d = x % 20; x /= 20;
        key |= (uint64_t)(SYNTH_ALPHABET[d] - 'A') << (5 * (8 - j));
    }
    return key;
}

inline bool synth_index_member(const SynthIndexParams& p, uint64_t ainv, uint64_t key45, uint64_t* member) {
    if (key45 >> 45) return false;
    uint64_t x = 0;
    for (int j = 0; j < 9; j++) {
        unsigned code = (unsigned)((key45 >> (5 * (8 - j))) & 31);
        int d = -1;
        for (int k = 0; k < 20; k++)
            if ((unsigned)(SYNTH_ALPHABET[k] - 'A') == code) { d = k; break; }
        if (d < 0) return false;
        x = x * 20 + (uint64_t)d;
    }
    uint64_t y = (x + SYNTH_M - (p.add_b % SYNTH_M)) % SYNTH_M;
    uint64_t i = (uint64_t)(((unsigned __int128)ainv * y) % SYNTH_M);
    if (i >= p.n_keys) return false;
    *member = i;
    return true;
}
