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

// Ranked mode (`seedextend -r taxa.tsv`, src/commands/seedextend.rs:150-163): of the record's extended seeds only the one with the
// highest score stays -- the LAST of equal ones (Iterator::max_by_key) -- where a seed's score is the sum over its taxa of
// TaxonList::score (12 under a named rank, None otherwise) or `penalty` for None (and for ids the list does not hold).
// scores[id] = that score, negative = None; n_scores = the list's length.  Same return value as seedextend_record.
__device__ __forceinline__ uint32_t seedextend_record_ranked(uint32_t* __restrict__ t, uint32_t n, uint32_t min_seed, uint32_t max_gap,
                                                             const int8_t* __restrict__ scores, uint32_t n_scores, uint32_t penalty) {
    auto at = [&](uint32_t i) -> uint32_t { return i < n ? t[i] : 0u; };  // index n = the sentinel
    const uint32_t len = n + 1;
    bool any = false;
    uint32_t bs = 0, be = 0;
    uint64_t best = 0;
    auto emit = [&](uint32_t s, uint32_t e) {
        // .skip(s).take(e - s): with s > e (a seed the scanner pushes behind a short leading zero-run when min_seed < 2) the
        // subtraction wraps in a release build and take() covers every taxon from s on, the sentinel included.
        uint64_t sc = 0;
        const uint32_t stop = s > e ? len : e;
        for (uint32_t j = s; j < stop && j < len; j++) {
            const uint32_t v = at(j);
            const int32_t q = v < n_scores ? (int32_t)scores[v] : -1;
            sc += q < 0 ? (uint64_t)penalty : (uint64_t)q;
        }
        if (!any || sc >= best) { any = true; best = sc; bs = s; be = e; }
    };
    uint32_t start = 0, end = 1;
    uint32_t last_tid = at(start), same_tid = 1, same_max = 1;
    while (end < len) {
        uint32_t cur = at(end);
        if (last_tid == cur) { same_tid += 1; end += 1; continue; }
        if (last_tid == 0 && same_tid > max_gap) {
            if (same_max >= min_seed) emit(start, end - same_tid);
            start = end;
            last_tid = cur;
            same_tid = 1;
            same_max = 1;
            end += 1;
            continue;
        }
        if (last_tid == 0 && (end - start) == same_tid) {
            end += 1;
            start = end;
            continue;
        }
        if (last_tid != 0) same_max = same_max > same_tid ? same_max : same_tid;
        last_tid = cur;
        same_tid = 1;
        end += 1;
    }
    if (same_max >= min_seed) {
        if (last_tid == 0) end -= same_tid;
        emit(start, end);
    }
    const bool panicked = any && (bs > be || be > len);       // taxons[start..end] of the chosen seed
    if (!any || panicked) { bs = 0; be = 0; }
    if (be > n) be = n;
    for (uint32_t i = 0; i < bs && i < n; i++) t[i] = 0;
    for (uint32_t i = be; i < n; i++) t[i] = 0;
    return panicked ? UMGAP_NONE : be - bs;
}

}  // namespace umgap
