// seedextend.hpp — the per-record state machine of `umgap seedextend`, unranked mode, for records held in device
// memory (gfx950 library internal; classify.hip has the LDS-streaming twin used inside the fused kernels).
#pragma once
#include "common.hpp"

namespace umgap {

// src/commands/seedextend.rs:99-148 restated per record, in place.  The reference appends a 0 sentinel (:99) and collects
// (start, end) seeds which it slices out of the record (:169-173); here everything outside the seeds is overwritten with 0
// (taxa2agg drops zeros, src/commands/taxa2agg.rs:169).  Returns the number of taxa the reference prints for the record,
// or UMGAP_NONE where it panics -- a seed with start > end, "slice index starts at s but ends at e", only reachable with
// --min-seed-size < 2 -- in which case the whole record is zeroed (nothing of it is printed).
__device__ __forceinline__ uint32_t seedextend_record(uint32_t* __restrict__ t, uint32_t n, uint32_t min_seed, uint32_t max_gap) {
    auto at = [&](uint32_t i) -> uint32_t { return i < n ? t[i] : 0u; };  // index n = the sentinel
    const uint32_t len = n + 1;
    uint32_t cursor = 0, nkept = 0;
    bool panicked = false;
    auto emit = [&](uint32_t s, uint32_t e) {  // seeds.push((s, e))
        if (s > e) { panicked = true; return; }
        if (e > n) e = n;                      // the sentinel itself is never stored
        for (uint32_t i = cursor; i < s; i++) t[i] = 0;
        nkept += e - s;
        cursor = e > cursor ? e : cursor;
    };
    uint32_t start = 0, end = 1;
    uint32_t last_tid = at(start), same_tid = 1, same_max = 1;
    while (end < len) {
        uint32_t cur = at(end);
        if (last_tid == cur) { same_tid += 1; end += 1; continue; }                 // :109-113
        if (last_tid == 0 && same_tid > max_gap) {                                  // :116-127
            if (same_max >= min_seed) emit(start, end - same_tid);
            start = end;
            last_tid = cur;
            same_tid = 1;
            same_max = 1;
            end += 1;
            continue;
        }
        if (last_tid == 0 && (end - start) == same_tid) {                           // :130-134
            end += 1;
            start = end;
            continue;
        }
        if (last_tid != 0) same_max = same_max > same_tid ? same_max : same_tid;    // :137-139
        last_tid = cur;
        same_tid = 1;
        end += 1;
    }
    if (same_max >= min_seed) {                                                     // :143-148
        if (last_tid == 0) end -= same_tid;
        emit(start, end);
    }
    if (panicked) cursor = 0;
    for (uint32_t i = cursor; i < n; i++) t[i] = 0;
    return panicked ? UMGAP_NONE : nkept;
}

}  // namespace umgap
