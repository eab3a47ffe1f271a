// window_locate.hpp — which fragment does window g of the flat window list belong to?  (gfx950 library internal)
// out_offsets[p] = number of windows of the fragments before p (n_prot + 1 entries, non-decreasing, out_offsets[0] = 0).
// A wave walks a CONTIGUOUS range of windows, 64 at a time.  One binary search over the offsets at the start of the range (the
// same loads for every lane) finds the fragment of its first window; from then on the next 64 boundaries go to LDS, one per lane,
// every lane finds its own fragment among them with seven LDS reads, and the cursor moves on by the number of boundaries passed
// (carrying the byte offset and first window of its fragment along).
// (The first version searched the whole offset list per window -- 25 dependent global loads each, a latency chain that, not the
// probes, set the pace of umgap_index_lookup_prot9_dev: 29 ms per 840 M windows.)
#pragma once
#include <cstdint>

namespace umgap {

struct WindowCursor {
    size_t p0;       // largest p with out_offsets[p] <= base of the next step (wave-uniform)
    uint64_t oo0;    // out_offsets[p0]
    uint64_t off0;   // offsets[p0] (where fragment p0's bytes start)
};

__device__ __forceinline__ size_t window_search(const uint64_t* __restrict__ out_offsets, size_t lo, size_t n_prot, size_t g) {
    size_t hi = n_prot;     // largest p in [lo, n_prot) with out_offsets[p] <= g
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) >> 1;
        if (out_offsets[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ WindowCursor window_cursor(const uint64_t* __restrict__ out_offsets, const uint64_t* __restrict__ offsets, size_t n_prot, size_t first_window) {
    const size_t p = window_search(out_offsets, 0, n_prot, first_window);
    return WindowCursor{p, out_offsets[p], offsets[p]};
}

// One step: windows [base, base + 64), base = the cursor's position.  Returns where the lane's window starts in the byte array
// (offsets[p] + (base + lane - out_offsets[p]) for its fragment p) and its number w inside the fragment, and moves the cursor to
// base + 64.  Every lane of the wave calls it; `tmp`: 3 x 64 words of LDS private to the wave.  The boundaries of the next 64
// fragments (window count and byte offset) are fetched together, one per lane, and everything after that is LDS: no load depends on
// another one.
// BYTES_IN_LDS: the byte offsets of the 64 fragments travel with their window counts (no load depends on another one: the walk
// kernel, whose chain of round trips is the longer one, gains 7 %); without it the lane fetches offsets[p] once it knows p (the
// plain kernel measured 17.6 ms that way against 21 ms with the extra stream).
template <bool BYTES_IN_LDS>
__device__ __forceinline__ uint64_t locate_window(WindowCursor& C, const uint64_t* __restrict__ out_offsets, const uint64_t* __restrict__ offsets, size_t n_prot,
                                                  size_t base, uint32_t lane, uint32_t* tmp, uint64_t& w) {
    const size_t p0 = C.p0;
    // boundary k = fragment p0 + 1 + k: its first window as a distance from base (1 .. 64 exact, 65 = further away or past the list)
    const size_t bi = p0 + 1 + lane;
    const uint64_t b = bi <= n_prot ? out_offsets[bi] : ~0ull;
    const uint64_t bo = (BYTES_IN_LDS && bi < n_prot) ? offsets[bi] : 0ull;
    const uint32_t rel = (b < base || b - base > 64ull) ? (b < base ? 0u : 65u) : (uint32_t)(b - base);
    uint32_t* trel = tmp;
    uint32_t* tlo = tmp + 64;
    uint32_t* thi = tmp + 128;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    trel[lane] = rel;
    if (BYTES_IN_LDS) {
        tlo[lane] = (uint32_t)bo;
        thi[lane] = (uint32_t)(bo >> 32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t last_rel = trel[63];
    uint64_t start;
    if (last_rel <= 64u) {
        // 64 fragments or more start inside these 64 windows (runs of fragments without windows): the long way
        const size_t p = window_search(out_offsets, p0, n_prot, base + lane);
        w = (base + lane) - out_offsets[p];
        start = offsets[p] + w;
        const size_t q = window_search(out_offsets, p0, n_prot, base + 64);
        C = WindowCursor{q, out_offsets[q], offsets[q]};
    } else {
        // number of boundaries at distance <= x: first k with trel[k] > x (trel is non-decreasing)
        auto count_le = [&](uint32_t x) {
            uint32_t l2 = 0, h2 = 64;
#pragma unroll
            for (int s = 0; s < 7; s++) {
                if (l2 < h2) {
                    const uint32_t mid = (l2 + h2) >> 1;
                    if (trel[mid] <= x) l2 = mid + 1; else h2 = mid;
                }
            }
            return l2;
        };
        const uint32_t k = count_le(lane);           // the lane's fragment is p0 + k; k = 0: the cursor's own
        const uint64_t frag_first = k ? base + trel[k - 1] : C.oo0;                       // its first window
        w = (base + lane) - frag_first;
        if (BYTES_IN_LDS) start = (k ? ((uint64_t)thi[k - 1] << 32) | tlo[k - 1] : C.off0) + w;
        else start = offsets[p0 + k < n_prot ? p0 + k : n_prot - 1] + w;
        const uint32_t kn = count_le(64u);           // fragments that start at or before the next base
        if (kn) C = WindowCursor{p0 + kn, base + trel[kn - 1], BYTES_IN_LDS ? ((uint64_t)thi[kn - 1] << 32) | tlo[kn - 1] : 0ull};
    }
    return start;
}

}  // namespace umgap
