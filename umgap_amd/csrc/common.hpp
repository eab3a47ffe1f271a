// common.hpp — error plumbing and the HBM table layout shared by every translation unit of
// libumgap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/umgap_hip.h"

namespace umgap {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define UMGAP_HIP(expr)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return ::umgap::fail(UMGAP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
// inside a chunk loop whose tail synchronises the lanes before it returns: record the error in `rc` and leave the LOOP, not the
// function (copies into the caller's arrays and the pinned staging may still be in flight)
#define UMGAP_HIP_BREAK(expr)                                                                                   \
    {                                                                                                          \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) {                                                                               \
            rc = ::umgap::fail(UMGAP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));           \
            break;                                                                                             \
        }                                                                                                      \
    }


// Selects `device` after checking that a GPU exists at all.
int use_device(int device);

// ---------------------------------------------------------------------------------------------
// 9-mer table layout in HBM.
//
//   key45 (5 bit/residue)  --mix45: bijection on [0,2^45)-->  h
//   A = h >> 28   (17 bits: superblock)      R = h & 0x0FFFFFFF  (28 bits: remainder, stored)
//   home bucket = A*m + (R*m >> 28)          nb = m << 17 buckets of 64 B = 8 slots of 8 B
//   slot (u64): [63:32] taxon  [31] "this bucket overflowed" (slot 0 only)  [30:28] displacement
//               [27:0] R.   Empty slot = all ones (displacement 7 never matches a probe).
//   A key that does not fit its home bucket goes to bucket home+d (mod nb), d <= 6, and every
//   bucket it skipped carries the overflow flag, so a probe walks on only past flagged buckets:
//   a hit or a miss costs ONE 64-byte sector unless the home bucket overflowed at build time.
//   (bucket, d, R) determines h, hence the key: exact match, no false positives.
//   The very few keys (~1e-5 at load 0.6, fewer at the default 0.5) whose chain would exceed MAX_DISP live in the STASH: a
//   sorted array of (home bucket << 28 | R) -> taxon, binary-searched only by probes that walked
//   the whole chain through flagged buckets.  This keeps the load factor (default 0.5 = 16 B/key: 2.3 % of the
//   buckets overflow, 1.014 requests per lookup; 0.6 = 13.3 B/key: 6.6 %, 1.044) independent of the table size.
// ---------------------------------------------------------------------------------------------
static constexpr uint64_t MASK45 = (1ull << 45) - 1;
static constexpr uint32_t R_BITS = 28;
static constexpr uint32_t R_MASK = (1u << R_BITS) - 1;
static constexpr uint32_t A_BITS = 17;
static constexpr uint32_t SLOTS = 8;
static constexpr uint32_t MAX_DISP = 6;
static constexpr uint32_t OVF_BIT = 0x80000000u;
static constexpr uint32_t TAG_MASK = 0x7FFFFFFFu;
static constexpr uint64_t EMPTY_SLOT = ~0ull;
static constexpr uint64_t MIX_C1 = 0x1B0373871FADull;  // odd, < 2^45
static constexpr uint64_t MIX_C2 = 0x0D6E8FEB8665ull;  // odd, < 2^45

__host__ __device__ inline uint64_t mix45(uint64_t k) {
    uint64_t h = k & MASK45;
    h ^= h >> 23;
    h = (h * MIX_C1) & MASK45;
    h ^= h >> 21;
    h = (h * MIX_C2) & MASK45;
    h ^= h >> 24;
    return h;
}
// home bucket and 28-bit tag of a key
__host__ __device__ inline void probe_of_key(uint64_t key45, uint32_t m, uint32_t& bucket, uint32_t& tag) {
    uint64_t h = mix45(key45);
    uint32_t a = (uint32_t)(h >> R_BITS);
    uint32_t r = (uint32_t)h & R_MASK;
    bucket = a * m + (uint32_t)(((uint64_t)r * m) >> R_BITS);
    tag = r;
}

struct TableView {
    const uint64_t* slots;  // device pointer to slot 0 of bucket `first_bucket`
    uint32_t m;             // sub-buckets per superblock
    uint32_t nb;            // total buckets of the whole table = m << 17
    uint32_t first_bucket;  // shard: first bucket resident here
    uint32_t n_resident;    // shard: buckets resident (home range + halo)
    uint32_t home_lo;       // shard owns home buckets [home_lo, home_hi)
    uint32_t home_hi;
    const uint64_t* stash_keys;  // sorted (home bucket << 28 | R); whole stash on every shard
    const uint32_t* stash_vals;
    uint32_t n_stash;
};

// "UMGAPIX1" file header (little endian), followed by nb*64 bytes of buckets.
struct IndexFileHeader {
    char magic[8];
    uint32_t version;   // 1
    uint32_t kind;      // 9 = 9-mer table; 10 = 9-mer table in walk order (values = window numbers; the walk array follows the stash)
    uint64_t n_keys;
    uint32_t m;
    uint32_t nb;
    uint64_t mix_c1, mix_c2;
    uint32_t r_bits, slots;
    uint64_t n_stash;  // followed (after the buckets) by n_stash x {u64 key, u64 taxon}
    uint64_t reserved;  // kind 10: entries of the walk array (u64 each); 0 otherwise
};

}  // namespace umgap

struct umgap_var_index;  // index_var.hip

struct umgap_index {
    int device = 0;
    int kind = 9;                      // 9 = packed 9-mer table, 2 = variable-length (byte string) table, 10 = 9-mer table in walk order (index_walk.hip)
    uint64_t* d_walk = nullptr;        // kind 10: E[p] = taxon | (low 31 key bits | continues << 31) << 32 for window p of the proteome; the table's values are positions
    uint64_t n_walk = 0;
    uint32_t walk_shift = 0;           // the table holds p >> walk_shift (walk_ref.hpp: 1 from 2^32 - 2^22 entries on, 2 from 2^33 on)
    bool walk_zero_absent = false;     // an entry whose value is 0 is a 9-mer of the proteome that is no key (umgap_walk_builder_new_deferred)
    uint32_t n_walk_exc = 0;           // keys whose group holds another entry with their low bits: sorted keys -> positions
    uint64_t* d_walk_exc_keys = nullptr;
    uint64_t* d_walk_exc_pos = nullptr;
    umgap_var_index* var = nullptr;    // kind 2
    uint64_t* d_slots = nullptr;  // nb * 8 slots (whole table) or the shard's range
    uint64_t n_keys = 0;
    uint64_t* d_stash_keys = nullptr;
    uint32_t* d_stash_vals = nullptr;
    uint32_t n_stash = 0;
    umgap::TableView view{};
    hipStream_t stream = nullptr;
    // pinned staging for the host-pointer entry points
    void* h_stage[3] = {nullptr, nullptr, nullptr};
    void* d_stage[3] = {nullptr, nullptr, nullptr};
    size_t stage_cap = 0;  // keys per staging buffer
    // the upload lane (copies only: taxonomy.hpp) and the probe lane of the host-pointer look-ups (chunks c+1, c+2 upload while
    // chunk c is probed)
    hipStream_t up_stream = nullptr, run_stream = nullptr;
    // non-null: the slot values are not taxon ids but codes bound to that taxonomy (bound.hpp); only the bound
    // fused kernel understands them
    const void* bound_to = nullptr;
    // grow-only device scratch of umgap_classify_tryp_dev (tryptic.hip): [0] sized by the bytes of a call, [1] by its pieces
    // ([2], [3]: the same pair for the second lane of a call that is cut into chunks of reads; the lanes and their events are
    //  created with the first such call)
    void* tryp_scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t tryp_scratch_bytes[4] = {0, 0, 0, 0};
    uint64_t* tryp_host = nullptr;     // 8 page-locked words the digest's kernels report to (two round trips per call without a staged copy)
    hipStream_t tryp_lane[2] = {nullptr, nullptr};
    hipEvent_t tryp_ev_agg[2] = {nullptr, nullptr}, tryp_ev_done[2] = {nullptr, nullptr};
};
