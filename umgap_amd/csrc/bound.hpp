// bound.hpp -- the value code of a table that has been BOUND to a taxonomy (umgap_index_bind_taxonomy).
//
// Unbound, a slot's 32-bit value is the NCBI taxon id and the aggregate starts by gathering {pre, last} of every
// distinct surviving taxon from a 28 MB by-id table: random 8-byte reads that miss L2 and count against the same
// request ceiling as the probes (about 6 per read on the benchmark's reads, 15 with -l 1).  Bound, the value IS the
// interval, as far as 32 bits allow:
//     [31:10] pre + 1                   (preorder index; 0 is kept for "miss", so at most 2^22 - 2 reachable taxa)
//     [ 9: 0] code of size = last - pre (subtree size - 1): exact below 32, else 5 exponent + 5 mantissa bits
// size_lo(code) <= size <= size_hi(code).  The LCA* closed form needs, for a survivor x and the deepest survivor M
// (pre[x] < pre[M]), whether x is an ancestor of M, i.e. whether pre[M] <= last[x]:  pre[M] <= pre[x] + size_lo says
// yes, pre[M] > pre[x] + size_hi says no, and only a pre[M] inside the rounding gap (3 % of the subtree size at most)
// needs the exact last[x] (one 4-byte gather).  Equal values <=> equal taxa, so counting, the -l filter and
// seedextend work on the codes unchanged.  Binding is a reversible in-place pass over the table.
#pragma once
#include <cstdint>

namespace umgap {
namespace bound {

constexpr uint32_t MAX_REACH = (1u << 22) - 2;

__host__ __device__ inline uint32_t encode(uint32_t pre, uint32_t size) {
    uint32_t code;
    if (size < 32) code = size;
    else {
        uint32_t k = 31u - (uint32_t)__builtin_clz(size), shift = k - 5;    // size >> shift in [32, 64)
        code = ((shift + 1) << 5) | ((size >> shift) - 32);
    }
    return ((pre + 1) << 10) | code;
}
__host__ __device__ inline uint32_t pre_of(uint32_t v) { return (v >> 10) - 1; }
__host__ __device__ inline void size_bounds(uint32_t v, uint32_t& lo, uint32_t& hi) {
    const uint32_t code = v & 1023u, eb = code >> 5;
    if (eb == 0) { lo = hi = code; return; }
    const uint32_t shift = eb - 1;
    lo = (32u + (code & 31u)) << shift;
    hi = lo + (1u << shift) - 1;
}

}  // namespace bound
}  // namespace umgap
