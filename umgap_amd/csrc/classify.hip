// classify.hip — the fused 9-mer high-precision pipeline in ONE kernel (gfx950, wave64):
//     packed 9-mers -> table probe -> [seedextend per fragment] -> per-read LCA* -> snapped taxon
// i.e. `prot2kmer2lca -o | seedextend -gG -sS | uniq -d / | taxa2agg -lL -a lca*`
// (scripts/umgap-analyse.sh:299-310) for reads whose k-mers are already packed; the hits never
// leave the CU.  Reference code per stage: src/commands/pept2lca.rs:93 (get), src/commands/
// seedextend.rs:99-148, src/commands/taxa2agg.rs:161-180 + src/tree/lca.rs:32-40.
//
// Why fuse: the probe is bound by the chip's random-request rate (~50 G 64-byte requests/s, see
// DESIGN.md) and leaves the ALUs idle; the aggregate is instruction/latency bound.  In one kernel
// the aggregate of some waves runs under the memory waits of the others, and the 4 B/hit round
// trip through HBM disappears.
//
// A wave owns a group of reads at a time (stride = k-mers per read, 84 -> 12 reads = 1008 keys):
// probe -> taxa into the wave's LDS strip -> seedextend on all the group's fragments in parallel
// lanes -> per read: LDS hash-count, leaders gather {pre,last}, O(1) range-min LCA (aggregate.hip
// has the commentary).  k_classify9_general is the same pipeline for MRTL and the hybrids, k_classify_prot9 takes the
// reads as protein fragments (rolling keys from LDS) instead of packed k-mers.
#include "agg_general.hpp"
#include "agg_group.hpp"
#include "bound.hpp"
#include "probe_common.hpp"
#include "prot_limits.hpp"
#include "shard_wire.hpp"
#include "walk_ref.hpp"

using namespace umgap;
using namespace umgap::agg;
using namespace umgap::probe;

namespace {

// seedextend state machine over one fragment held in LDS (unranked mode; pipeline.hip has the
// device-memory twin with the line-by-line citations).  Streaming form: the reference's scanner
// consumes the elements t[1..n] (t[n] = the 0 sentinel) strictly in order -- its one irregular
// move, "don't start with a missing taxon" (src/commands/seedextend.rs:130-134), merely skips
// element `end` and restarts right after it -- so the next element is prefetched while the
// current one is processed (the LDS read latency is the cost of this stage).
// Returns true when the reference would panic on the record: a seed with start > end, sliced at src/commands/seedextend.rs:169-173
// (only reachable with --min-seed-size < 2).
__device__ __forceinline__ bool seedextend_lds(uint32_t* t, uint32_t n, uint32_t min_seed, uint32_t max_gap) {
    uint32_t cursor = 0;
    bool panicked = false;
    auto emit = [&](uint32_t s, uint32_t e) {   // seeds.push((s, e)): everything before s and after the last e is dropped
        if (s > e) { panicked = true; return; }
        if (e > n) e = n;
        for (uint32_t i = cursor; i < s; i++) t[i] = 0;
        cursor = e > cursor ? e : cursor;
    };
    if (n == 0) return false;
    uint32_t start = 0, last_tid = t[0], same_tid = 1, same_max = 1;
    uint32_t nxt = n > 1 ? t[1] : 0u;
    for (uint32_t end = 1; end <= n; end++) {
        const uint32_t cur = nxt;                       // t[end], or the sentinel at end == n
        nxt = end + 1 < n ? t[end + 1] : 0u;
        if (last_tid == cur) { same_tid += 1; continue; }                                   // :109-113
        if (last_tid == 0 && same_tid > max_gap) {                                          // :116-127
            if (same_max >= min_seed) emit(start, end - same_tid);
            start = end; last_tid = cur; same_tid = 1; same_max = 1;
            continue;
        }
        if (last_tid == 0 && (end - start) == same_tid) { start = end + 1; continue; }      // :130-134
        if (last_tid != 0) same_max = same_max > same_tid ? same_max : same_tid;            // :137-139
        last_tid = cur; same_tid = 1;
    }
    if (same_max >= min_seed) {                                                             // :143-148
        uint32_t end = n + 1;
        if (last_tid == 0) end -= same_tid;
        emit(start, end);
    }
    for (uint32_t i = cursor; i < n; i++) t[i] = 0;
    return panicked;
}

// (lca_star_from_lds / lca_star_bound_from_lds, the per-read LCA* over a strip in LDS, live in agg_group.hpp)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr uint32_t STRIP = 1024;  // taxa per wave held in LDS between the stages (4 KB)

// Probe stage of a group: keys kp[0..nk) (nk <= STRIP) -> taxa (0 = miss) in the wave's LDS strip; chunks of 64 keys,
// two chunks (8 sectors per quad) in flight.
__device__ __forceinline__ void probe_group_to_strip(const TableView& T, const uint64_t* __restrict__ kp, uint32_t nk, uint32_t* wtaxa, uint32_t lane) {
    const uint32_t q = lane >> 2, j = lane & 3;
    for (uint32_t c = 0; c * 64 < nk; c += 2) {
        ProbeBatch B[2];
        uint32_t res[2][4];
        const bool two = (c + 1) * 64 < nk;
        {
            uint32_t i = c * 64 + lane, local = 0, tag = 0, valid = 0;
            if (i < nk) valid = locate(T, __builtin_nontemporal_load(kp + i), local, tag) ? 1u : 0u;
            probe_issue(T, local, tag | (valid << 31), lane, B[0]);
        }
        if (two) {
            uint32_t i = (c + 1) * 64 + lane, local = 0, tag = 0, valid = 0;
            if (i < nk) valid = locate(T, __builtin_nontemporal_load(kp + i), local, tag) ? 1u : 0u;
            probe_issue(T, local, tag | (valid << 31), lane, B[1]);
        } else probe_skip(B[1]);
        probe_eval_multi<2>(T, B, 0u, lane, res);
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t i0 = c * 64 + 16 * r + q, i1 = i0 + 64;
                if (i0 < nk) wtaxa[i0] = res[0][r];
                if (two && i1 < nk) wtaxa[i1] = res[1][r];
            }
        }
    }
}


// Walk-order index (round 5): packed 9-mers whose neighbours are consecutive windows of a protein -- the real input of pept2lca is
// exactly that (prot2kmer prints every window of a fragment, /root/reference/src/commands/prot2kmer.rs) -- are answered by ONE probe per
// run plus the walk-array entries behind it, as umgap_classify_prot9_dev does for residues.
// Phase A: a lane owns a run of at most WALK_R consecutive keys of one fragment (runs never cross a fragment of the read).  The run's
// first key is its anchor: a quad probe gives its entry number p (or says the key is absent: answered), and E[p .. p + WALK_R) are
// fetched in one trip.  Along those entries the PROTEOME's own windows are rebuilt: entry p + k continues entry p + k - 1 (its flag)
// means window p + k is the next window of the same protein, so its key is the previous one shifted by a residue -- the entry's low
// five key bits.  Where that rebuilt key equals the read's key k, the entry is the key's entry (exact: all 45 bits compared) and its
// taxon the answer; where it does not -- a changed residue in the read -- the key is left for phase B, and the walk goes ON: the read's
// windows behind the changed residue meet the proteome's again, no second anchor needed.  (The first version re-anchored at every
// break, step by step: as many requests, but sixteen dependent probe + fetch round trips per pass -- 26 ms per 10 M pairs at 3 %
// divergence beside 17 ms over the plain table.)
// Phase B: the keys phase A left open (marked in the strip) are probed like any key, two batches of 64 strip slots at a time; a batch
// none of whose slots is open costs nothing.  A 9-mer of the proteome that is no key (deferred values) has taxon 0: a miss.
constexpr uint32_t WALK_R = 16;
constexpr uint32_t WALK_OPEN = 0xFFFFFFFFu;        // strip marker: not answered by a walk (no taxon id, no interval code is all ones)
__device__ __forceinline__ void walk_group_to_strip(const TableView& T, const HitSource& H, const uint64_t* __restrict__ kp, uint32_t nk, uint32_t flen, uint32_t* wtaxa,
                                                    uint32_t* list_words, uint32_t lane) {
    WalkRef W;
    W.E = H.walk_E; W.n = H.walk_n; W.shift = H.walk_shift; W.n_exc = H.walk_n_exc; W.exc_keys = H.walk_exc_keys; W.exc_pos = H.walk_exc_pos;
    const uint32_t rpf = (flen + WALK_R - 1) / WALK_R;            // runs per fragment
    const uint32_t n_runs = (nk / flen) * rpf;                     // (nk is a multiple of flen)
    const uint32_t row = lane >> 4, k = lane & 15u;
    for (uint32_t run0 = 0; run0 < n_runs; run0 += 64) {           // wave-uniform
        // ---- A1, lane = run: the run's anchor (its first key) probed, the entry number behind it ----
        const uint32_t run = run0 + lane;
        const uint32_t f = run / rpf, jr = run - f * rpf;
        const uint32_t first = f * flen + jr * WALK_R;
        const uint32_t n_mine = run < n_runs ? min(WALK_R, flen - jr * WALK_R) : 0u;
        const uint64_t key0 = n_mine ? __builtin_nontemporal_load(kp + first) : ~0ull;
        uint32_t local = 0, tag = 0, valid = 0;
        if (n_mine) valid = locate(T, key0, local, tag) ? 1u : 0u;
        uint32_t pr[4];
        if (H.walk_debug & 4u) { pr[0] = pr[1] = pr[2] = pr[3] = (uint32_t)(key0 >> 7) & 0xFFFFFu; }
        else {
            ProbeBatch B;
            probe_issue(T, local, tag | (valid << 31), lane, B);
            probe_eval(T, B, UMGAP_NONE, lane, pr);
        }
        // the answer to lane L's key sits, quad-uniform, in quad L & 15 as round L >> 4
        const int src = 4 * (int)(lane & 15u);
        const uint32_t v0 = (uint32_t)__shfl((int)pr[0], src, 64), v1 = (uint32_t)__shfl((int)pr[1], src, 64),
                       v2 = (uint32_t)__shfl((int)pr[2], src, 64), v3 = (uint32_t)__shfl((int)pr[3], src, 64);
        const uint32_t r4 = lane >> 4;
        const uint32_t q32 = r4 == 0 ? v0 : (r4 == 1 ? v1 : (r4 == 2 ? v2 : v3));
        const uint64_t pos = (n_mine && valid) ? walk_position(W, q32, key0) : WALK_NOPOS;
        // ---- A2, TRANSPOSED: a row of 16 lanes = one run, lane k of the row = its k-th key; four runs per step.  The keys and the
        // entries of a run are 128 contiguous bytes each: a step's two loads touch ~12 lines, not 128 as with a lane per run (sixteen
        // strided 8-byte loads per lane: the vector cache looked up a line per key and per entry -- 5.9 of the kernel's 12.7 ms went to
        // the key loads alone, UMGAP_WALK_DEBUG=31), and the steps are independent: their loads are in flight together.
        // The proteome's key at entry p + k: K_k = (key0 << 5k | sum over j = 1..k of r_j << 5(k - j)) mod 2^45, r_j = the low five key
        // bits of entry p + j -- the residues of the nine lanes up to k, fetched with row shifts ----
        const uint32_t steps = (min(64u, n_runs - run0) + 3u) >> 2;
        // four steps at a time: their eight loads are issued together, branch-free (a lane with nothing to fetch reads kp[0]), and
        // only then waited for -- with a load under its own `if` the compiler ended every step in a wait, eighteen round trips per pass
        constexpr uint32_t UB = 4;
        for (uint32_t s0 = 0; s0 < steps; s0 += UB) {             // wave-uniform (steps <= 16: 4 (s0 + u) + row <= 63)
            uint32_t r_first[UB];
            uint64_t r_key0[UB], key[UB], e[UB];
            bool active[UB], has[UB];
#pragma unroll
            for (uint32_t u = 0; u < UB; u++) {
                const int from = (int)(4u * (s0 + u) + row);
                r_first[u] = (uint32_t)__shfl((int)first, from, 64);
                const uint32_t r_n = (uint32_t)__shfl((int)n_mine, from, 64);
                const uint64_t r_pos = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(pos >> 32), from, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)pos, from, 64);
                r_key0[u] = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(key0 >> 32), from, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)key0, from, 64);
                active[u] = k < r_n;
                has[u] = r_pos != WALK_NOPOS;
                const bool fetch = !(H.walk_debug & 2u) && active[u] && has[u] && r_pos + k < W.n;
                const uint64_t* kq = active[u] ? kp + r_first[u] + k : kp;
                const uint64_t* eq = fetch ? W.E + r_pos + k : kp;
                key[u] = __builtin_nontemporal_load(kq);
                e[u] = *eq;
                if (!active[u]) key[u] = ~0ull;
                if (!fetch) e[u] = WALK_PAD_ENTRY;
            }
#pragma unroll
            for (uint32_t u = 0; u < UB; u++) {
                const uint32_t ey = (uint32_t)(e[u] >> 32);
                // entries p + 1 .. p + k all continue their predecessor: the row's bits 1..k of the ballot
                const uint64_t cm = __ballot(k == 0u || (ey >> 31));
                const uint32_t rb = (uint32_t)(cm >> (16u * row)) & 0xFFFFu;
                const bool chain = has[u] && ((~rb) & ((2u << k) - 2u)) == 0u;
                const uint32_t r = k == 0u ? 0u : (ey & 31u);
                // the nine residues up to lane k, by doubling: pairs, fours, eight + one (row shifts, zero from beyond the row's start)
#define UMGAP_ROW_SHR(V, N) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(V), 0x110 + (N), 0xF, 0xF, true)
                const uint32_t p2 = r | (UMGAP_ROW_SHR(r, 1) << 5);
                const uint32_t p4 = p2 | (UMGAP_ROW_SHR(p2, 2) << 10);
                const uint32_t s4 = UMGAP_ROW_SHR(p4, 4), r8 = UMGAP_ROW_SHR(r, 8);
#undef UMGAP_ROW_SHR
                const uint64_t acc = ((uint64_t)((s4 >> 12) | (r8 << 8)) << 32) | (uint64_t)(p4 | (s4 << 20));
                const uint64_t K = (k <= 8u ? (r_key0[u] << (5u * k)) & MASK45 : 0ull) | acc;
                if (active[u]) wtaxa[r_first[u] + k] = k == 0u ? (has[u] ? (uint32_t)e[u] : 0u) : ((chain && K == key[u]) ? (uint32_t)e[u] : WALK_OPEN);
            }
        }
    }
    wave_lds_sync();
    // ---- phase B: the open slots, COMPACTED (16-bit strip indices in the aggregate stage's count tables, which are idle and all zero now:
    // 512 words = 1024 items), probed 128 at a time -- dense batches and as few dependent round trips as the open keys need, instead of
    // one sparse batch per 128 slots of the strip ----
    uint16_t* list = reinterpret_cast<uint16_t*>(list_words);
    uint32_t U = 0;                                                // open slots so far (wave-uniform)
    for (uint32_t c = 0; c * 64 < nk; c++) {                      // wave-uniform
        const uint32_t i = c * 64 + lane;
        const bool open = i < nk && wtaxa[i] == WALK_OPEN;
        const uint64_t m = __ballot(open);
        if (open) list[U + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
        U += (uint32_t)__builtin_popcountll(m);
    }
    if (U == 0) return;
    wave_lds_sync();
    if (H.walk_debug & 1u) U = 0;
    for (uint32_t i0 = 0; i0 < U; i0 += 128) {                    // wave-uniform
        ProbeBatch B[2];
        uint32_t pr[2][4], item[2];
        uint64_t bkey[2];
        const bool two = i0 + 64 < U;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h == 1 && !two) { probe_skip(B[1]); item[1] = WALK_OPEN; bkey[1] = ~0ull; break; }   // wave-uniform
            const uint32_t i = i0 + 64 * h + lane;
            item[h] = i < U ? (uint32_t)list[i] : WALK_OPEN;
            uint32_t local = 0, tag = 0, valid = 0;
            bkey[h] = item[h] != WALK_OPEN ? __builtin_nontemporal_load(kp + item[h]) : ~0ull;
            if (item[h] != WALK_OPEN) valid = locate(T, bkey[h], local, tag) ? 1u : 0u;
            probe_issue(T, local, tag | (valid << 31), lane, B[h]);
        }
        probe_eval_multi<2>(T, B, UMGAP_NONE, lane, pr);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h == 1 && !two) break;                             // wave-uniform
            const int src = 4 * (int)(lane & 15u);
            const uint32_t v0 = (uint32_t)__shfl((int)pr[h][0], src, 64), v1 = (uint32_t)__shfl((int)pr[h][1], src, 64),
                           v2 = (uint32_t)__shfl((int)pr[h][2], src, 64), v3 = (uint32_t)__shfl((int)pr[h][3], src, 64);
            const uint32_t r4 = lane >> 4;
            const uint32_t q32 = r4 == 0 ? v0 : (r4 == 1 ? v1 : (r4 == 2 ? v2 : v3));
            if (item[h] != WALK_OPEN) wtaxa[item[h]] = walk_taxon(W, q32, bkey[h], 0u);
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 8; i++) list_words[lane + 64 * i] = 0u;    // the count tables go back to all zero
    wave_lds_sync();
}

// Sharded mode (shard.hip): the hits of a group were looked up by the shards that own the keys and came back over the
// exchange.  The group is one RUN of the piece (shard_wire.hpp): the wave walks its keys in rounds of 64, recomputes owner and
// rank of every key exactly as the pack did and finds the key's value at base[run][owner] + rank of the returned buffer (3 or 4
// bytes per slot, all-ones = miss; a key that was not representable or whose send region overflowed is a miss) -- the values
// of a run lie in `world` contiguous stretches, so a round's loads coalesce.  Keys this rank owns itself never went anywhere:
// they are probed here, against the rank's shard, with the hash already in hand.
// The strip receives the keys [want_lo, want_hi) of the run (the whole run in the first pass; one read of it in the large-scratch
// pass, which still has to walk the run from its start to know the ranks).
__device__ __forceinline__ void gather_run_to_strip(const TableView& T, const HitSource& H, size_t run, size_t run_first_key, uint32_t want_lo, uint32_t want_hi,
                                                    uint32_t* wtaxa, uint32_t lane) {
    const uint64_t* kp = H.keys + run_first_key;
    if (H.world == 1 && H.self == 0) { probe_group_to_strip(T, kp + want_lo, want_hi - want_lo, wtaxa, lane); return; }   // nobody else holds anything
    const uint32_t q = lane >> 2, j = lane & 3;
    const double wn = (double)H.world / (double)T.nb;
    const uint32_t base = lane < H.world ? H.bases[run * H.world + lane] : 0u;
    uint32_t seen = 0;
    for (uint32_t i0 = 0; i0 < want_hi; i0 += 256) {     // four rounds: four independent key loads, then four independent value loads, per lane
        uint64_t key[4];
        wire::Routed R[4];
        uint32_t val[4];
        uint64_t slot[4];
        bool wanted[4], local[4], travelled[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + 64 * u + lane;
            key[u] = i < want_hi ? __builtin_nontemporal_load(kp + i) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + 64 * u + lane;
            R[u] = wire::route(key[u], i < want_hi, T.m, T.nb, H.world, wn);
            const uint32_t rank = wire::run_rank(R[u].owner, H.world, lane, seen);
            const uint32_t b0 = (uint32_t)__shfl((int)base, (int)(R[u].owner & 63u), 64);
            wanted[u] = i >= want_lo && i < want_hi;
            local[u] = wanted[u] && R[u].owner == H.self;
            const uint64_t pos = (uint64_t)b0 + rank;
            travelled[u] = wanted[u] && R[u].owner != wire::NO_OWNER && R[u].owner != H.self && pos < H.cap;
            slot[u] = (uint64_t)R[u].owner * H.cap + pos;
        }
        // the four values in ONE round trip: every lane loads (slot 0 where it has nothing to fetch) and the format is told apart
        // outside the four loads -- a load under `travelled` or under the per-slot format test is followed by its own wait
#pragma unroll
        for (int u = 0; u < 4; u++) if (!travelled[u]) slot[u] = 0;
        if (H.value_bytes == 3) {
            uint32_t w0[4], w1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t* w = H.ret + (size_t)(slot[u] >> 2) * 3 + (((uint32_t)slot[u] & 3u) >> 1);
                w0[u] = w[0];
                w1[u] = w[1];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = (uint32_t)slot[u] & 3u;
                const uint32_t v = (i == 0 ? w0[u] : (i == 1 ? ((w0[u] >> 24) | (w1[u] << 8)) : (i == 2 ? ((w0[u] >> 16) | (w1[u] << 16)) : (w1[u] >> 8)))) & 0xFFFFFFu;
                val[u] = (travelled[u] && v != 0xFFFFFFu) ? v : 0u;
            }
        } else {
            uint32_t w0[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w0[u] = H.ret[slot[u]];
#pragma unroll
            for (int u = 0; u < 4; u++) val[u] = (travelled[u] && w0[u] != 0xFFFFFFFFu) ? w0[u] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = i0 + 64 * u + lane; if (wanted[u] && !local[u]) wtaxa[i - want_lo] = val[u]; }
        // the rank's own keys: probed here, two rounds in flight (keys that travelled take no part: no request is issued for them)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint64_t m0 = __ballot(local[2 * h]), m1 = __ballot(local[2 * h + 1]);
            if (!(m0 | m1)) continue;               // wave-uniform
            ProbeBatch B[2];
            uint32_t res[2][4];
            probe_issue(T, R[2 * h].bucket - T.home_lo, R[2 * h].tag | ((local[2 * h] ? 1u : 0u) << 31), lane, B[0]);
            probe_issue(T, R[2 * h + 1].bucket - T.home_lo, R[2 * h + 1].tag | ((local[2 * h + 1] ? 1u : 0u) << 31), lane, B[1]);
            probe_eval_multi<2>(T, B, 0u, lane, res);
            if (j == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if ((m0 >> (16 * r + q)) & 1ull) wtaxa[i0 + 128 * h + 16 * r + q - want_lo] = res[0][r];
                    if ((m1 >> (16 * r + q)) & 1ull) wtaxa[i0 + 128 * h + 64 + 16 * r + q - want_lo] = res[1][r];
                }
            }
        }
    }
}

// A wave owns a GROUP of G = min(STRIP / stride, 64 / frags) reads at a time: all their keys are
// probed (chunks of 64, two chunks in flight) into the LDS strip, seedextend then runs on all
// G * frags fragments at once (one lane each), and the G reads are aggregated one after another.
// 94 VGPRs would allow 5 waves/SIMD; 4 measured faster (18.8 vs 19.6 ms): the request rate is the ceiling and a fifth
// wave only adds contention.
template <bool BOUND, int GATHER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_classify9(TableView T, TaxView V, int method, const uint64_t* __restrict__ keys,
                                                   size_t n_reads, uint32_t stride, uint32_t frags, uint32_t min_seed, uint32_t max_gap,
                                                   float lower_bound, uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status, HitSource H) {
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ LcaScratch scratch[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* wtaxa = lds_taxa[wv];
    LcaScratch& S = scratch[wv];
    uint32_t* wkeys = S.hkeys;
    uint32_t* wcnt = S.hcnt;
#pragma unroll
    for (int i = 0; i < 4; i++) { wkeys[lane + 64 * i] = 0u; wcnt[lane + 64 * i] = 0u; }
    const bool tree_like = method != UMGAP_METHOD_RMQ;
    uint32_t G = STRIP / stride;
    if (frags && G * frags > 64) G = 64 / frags;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_groups = (n_reads + G - 1) / G;
    for (size_t g = wave0; g < n_groups; g += nwaves) {
        const size_t r0 = g * G;
        const uint32_t nr = (uint32_t)min((size_t)G, n_reads - r0);
        const uint32_t nk = nr * stride;                   // keys of this group (<= STRIP)
        if (GATHER == SRC_GATHER) gather_run_to_strip(T, H, g, r0 * stride, 0u, nk, wtaxa, lane);
        else if (GATHER == SRC_WALK) walk_group_to_strip(T, H, keys + r0 * stride, nk, frags ? stride / frags : stride, wtaxa, S.hkeys, lane);
        else probe_group_to_strip(T, keys + r0 * stride, nk, wtaxa, lane);
        wave_lds_sync();
        // ---- seedextend per fragment (one lane per fragment, nr * frags <= 64) ----
        uint64_t se_panic = 0;   // lanes (= fragments) on which the reference's seedextend panics
        if (frags && !(GATHER == SRC_WALK && (H.walk_debug & 8u))) {
            const uint32_t flen = stride / frags;
            bool pk = false;
            if (lane < nr * frags) pk = seedextend_lds(wtaxa + lane * flen, flen, min_seed, max_gap);
            se_panic = __ballot(pk);
            wave_lds_sync();
        }
        if (GATHER == SRC_WALK && (H.walk_debug & 16u)) { if (lane < nr) { out_taxon[r0 + lane] = wtaxa[lane * stride]; out_status[r0 + lane] = 0; } continue; }
        // ---- aggregate the group: taxonomy round trips once per group, not once per read (agg_group.hpp) ----
        lca_group<BOUND>(V, S, tree_like, lower_bound, wtaxa, stride, nr, lane, se_panic, frags, r0, out_taxon, out_status,
                         [&](uint32_t k, uint32_t& res_taxon, uint32_t& res_status) {
                             if (BOUND) lca_star_bound_from_lds(V, wtaxa + k * stride, stride, lower_bound, wkeys, wcnt, lane, res_taxon, res_status);
                             else lca_star_from_lds(V, tree_like, wtaxa + k * stride, stride, lower_bound, wkeys, wcnt, lane, res_taxon, res_status);
                         });
        wave_lds_sync();
    }
}

// The same fused pipeline for the other strategies -- MRTL, tree-hybrid (the `taxa2agg` default, used by the
// high-sensitivity pipeline, scripts/umgap-analyse.sh:283-288) and rmq-hybrid: probe and seedextend as above, then every
// read of the group goes through general_read (agg_general.hpp) with the LDS strip as its source.  A read with more
// than CAP distinct taxa is deferred to a second launch over the `todo` list with a larger scratch (one read per group).
template <int CAP, int HCAP, bool BOUND, int GATHER>
__global__ __launch_bounds__(256) void k_classify9_general(TableView T, TaxView V, int method, int strategy, float factor,
                                                           const uint64_t* __restrict__ keys, size_t n_reads, uint32_t stride, uint32_t frags,
                                                           uint32_t min_seed, uint32_t max_gap, float lower_bound,
                                                           uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status,
                                                           const uint32_t* __restrict__ todo, const uint32_t* __restrict__ todo_count,
                                                           uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count, HitSource H) {
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ Scratch<CAP, HCAP> scratch[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* wtaxa = lds_taxa[wv];
    Scratch<CAP, HCAP>& S = scratch[wv];
    for (uint32_t i = lane; i < HCAP; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
    wave_lds_sync();
    uint32_t G = STRIP / stride;
    if (frags && G * frags > 64) G = 64 / frags;
    if (todo) G = 1;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_groups = todo ? (size_t)*todo_count : (n_reads + G - 1) / G;
    for (size_t g = wave0; g < n_groups; g += nwaves) {
        const size_t r0 = todo ? (size_t)todo[g] : g * G;
        const uint32_t nr = todo ? 1u : (uint32_t)min((size_t)G, n_reads - r0);
        const uint32_t nk = nr * stride;
        if (GATHER == SRC_WALK) walk_group_to_strip(T, H, keys + r0 * stride, nk, frags ? stride / frags : stride, wtaxa, S.hkeys, lane);
        else if (GATHER == SRC_GATHER) {
            // the runs of the exchange are the groups of the FIRST pass (H.run keys): a deferred read is found inside its run
            const size_t per_run = H.run / stride, run = r0 / per_run;
            const uint32_t lo = (uint32_t)(r0 - run * per_run) * stride;
            gather_run_to_strip(T, H, run, run * per_run * stride, lo, lo + nk, wtaxa, lane);
        } else probe_group_to_strip(T, keys + r0 * stride, nk, wtaxa, lane);
        wave_lds_sync();
        uint64_t se_panic = 0;
        if (frags) {
            const uint32_t flen = stride / frags;
            bool pk = false;
            if (lane < nr * frags) pk = seedextend_lds(wtaxa + lane * flen, flen, min_seed, max_gap);
            se_panic = __ballot(pk);
            wave_lds_sync();
        }
        for (uint32_t k = 0; k < nr; k++) {
            uint32_t res_taxon, res_status;
            bool ok = general_read<false, CAP, HCAP, BOUND>(V, S, method, strategy, factor, lower_bound, wtaxa + k * stride, nullptr, stride, lane,
                                                     res_taxon, res_status);
            if (se_panic && ((se_panic >> (k * frags)) & ((frags >= 64 ? 0ull : (1ull << frags)) - 1ull))) { ok = true; res_taxon = 0; res_status = UMGAP_READ_PANIC; }
            wave_lds_sync();
            if (lane == 0) {
                if (ok) { out_taxon[r0 + k] = res_taxon; out_status[r0 + k] = res_status; }
                else if (defer) defer[atomicAdd(defer_count, 1u)] = (uint32_t)(r0 + k);
                else { out_taxon[r0 + k] = 0; out_status[r0 + k] = UMGAP_READ_UNSUPPORTED; }
            }
        }
        wave_lds_sync();
    }
}

// First pass of the same pipeline with the aggregate batched over the group (agg_group.hpp): the taxonomy round trips are
// paid once per group and phase instead of once per read.  Reads with more than 64 distinct taxa go to the `defer` list, which
// k_classify9_general<128, 512> (one read at a time, large scratch) then works through.
template <bool BOUND, int GATHER>
__global__ __launch_bounds__(256) void k_classify9_group(TableView T, TaxView V, int method, int strategy, float factor,
                                                         const uint64_t* __restrict__ keys, size_t n_reads, uint32_t stride, uint32_t frags,
                                                         uint32_t min_seed, uint32_t max_gap, float lower_bound,
                                                         uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status,
                                                         uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count, HitSource H) {
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ GroupScratch scratch[4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* wtaxa = lds_taxa[wv];
    GroupScratch& S = scratch[wv];
    for (uint32_t i = lane; i < 256; i += 64) { S.hkeys[i] = 0u; S.hcnt[i] = 0u; }
    wave_lds_sync();
    uint32_t G = STRIP / stride;
    if (frags && G * frags > 64) G = 64 / frags;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_groups = (n_reads + G - 1) / G;
    for (size_t g = wave0; g < n_groups; g += nwaves) {
        const size_t r0 = g * G;
        const uint32_t nr = (uint32_t)min((size_t)G, n_reads - r0);
        const uint32_t nk = nr * stride;
        if (GATHER == SRC_GATHER) gather_run_to_strip(T, H, g, r0 * stride, 0u, nk, wtaxa, lane);
        else if (GATHER == SRC_WALK) walk_group_to_strip(T, H, keys + r0 * stride, nk, frags ? stride / frags : stride, wtaxa, S.hkeys, lane);
        else probe_group_to_strip(T, keys + r0 * stride, nk, wtaxa, lane);
        wave_lds_sync();
        uint64_t se_panic = 0;
        if (frags) {
            const uint32_t flen = stride / frags;
            bool pk = false;
            if (lane < nr * frags) pk = seedextend_lds(wtaxa + lane * flen, flen, min_seed, max_gap);
            se_panic = __ballot(pk);
            wave_lds_sync();
        }
        general_group<BOUND>(V, S, method, strategy, factor, lower_bound, wtaxa, stride, nr, lane, se_panic, frags, r0, out_taxon, out_status, defer, defer_count);
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// Surface B: the reads arrive as protein fragments (what FGSpp / `translate` emit and
// `prot2kmer2lca` consumes, ~1.2 B per k-mer instead of 8 B): read r owns fragments
// [r*F, (r+1)*F) of a CSR byte array.  A wave owns a group of up to 8 reads: their bytes are staged in LDS
// (all loads in flight together), the group's 9-mer windows (src/commands/prot2kmer2lca.rs:172-179: all len-8
// windows of a fragment, none if it is shorter than 9) form one flat list that is packed from LDS and probed in
// chunks of 128 (two batches in flight) into a dense strip, then seedextend per fragment and LCA* per read as in
// k_classify9.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t PROT_GROUP = STRIP / PROT_MAX_WINDOWS;  // 8 reads (limits per read: prot_limits.hpp)

// per-wave aggregate scratch of k_classify_prot9: LCA* wants the count table only (its LcaView lies over the byte staging
// area), the other strategies the group scratch of agg_group.hpp (whose count table comes first)
template <bool GENERAL>
struct ProtScratch {
    uint32_t keys[256], cnt[256];
};
template <>
struct ProtScratch<true> {
    uint32_t keys[256], cnt[256];
    uint32_t reads[GROUP_VIEW_READ_WORDS];   // the per-read arrays of the group scratch; its survivor arrays lie over the byte staging area
};
// the large-scratch pass (BIG): one read per wave, up to 128 distinct taxa (general_read, agg_general.hpp)
using ProtBigScratch = Scratch<128, 512>;

// GENERAL = false: LCA* (the kernel of umgap_classify_prot9_dev, 4 workgroups per CU; 3 for GENERAL).  GENERAL = true: MRTL and the hybrids
// through general_group -- the reads of a group sit at a fixed stride of 128 in the strip instead of back to back.
// WALK (a kind-10 index, index_walk.hip): the table's values are window numbers of the proteome and the taxa live in the walk array E.
// A lane's run of consecutive windows is then answered by ONE probe -- its first window -- plus the entries E[p + 1 ..] behind the
// position p the probe returned, as long as each entry continues its predecessor and carries the window's low key bits (by
// induction its key equals the window's: exact); where the chain breaks, that window is probed and the walk goes on from there.
struct WalkView {
    WalkRef W;                     // the walk array and how the table's values address it (walk_ref.hpp)
    const uint32_t* gate = nullptr;   // non-null: three words {offsets bad, unfit for the short variant, unfit for the LONG one} (records.hip):
                                      // the short variant runs when none is set, the LONG one when only word 1 is
};

// BIG (round 4): the large-scratch pass behind the GENERAL variants -- a read with more than 64 distinct taxa left after the count
// filter (MRTL, hybrids) is put on the `defer` list by the first pass and worked through here, one read per wave (LONG geometry:
// every read the first pass took fits it), with general_read's 128-taxon scratch; `todo` = that list, *defer_count its length.
template <bool BOUND, bool GENERAL, bool WALK, bool LONG, bool BIG = false>
__global__ __launch_bounds__(256, BIG ? 2 : (GENERAL ? 3 : 4)) void k_classify_prot9(TableView T, TaxView V, int method, int strategy, float factor,
                                                                      const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ foff, size_t n_reads,
                                                                      uint32_t F, uint32_t do_seed, uint32_t min_seed, uint32_t max_gap, float lower_bound,
                                                                      uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status, WalkView WK,
                                                                      uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count, const uint32_t* __restrict__ todo) {
    static_assert(!BIG || (GENERAL && LONG), "the large-scratch pass is a GENERAL, LONG kernel");
    __shared__ uint32_t lds_taxa[4][STRIP];
    __shared__ std::conditional_t<BIG, ProtBigScratch, ProtScratch<GENERAL>> lds_scratch[4];
    // per-read limits of this variant (prot_limits.hpp); either way a group fills the strip and the byte staging area exactly
    constexpr uint32_t LIM_W = LONG ? PROT_LONG_WINDOWS : PROT_MAX_WINDOWS, LIM_B = LONG ? PROT_LONG_BYTES : PROT_MAX_BYTES, GRP = STRIP / LIM_W;
    static_assert(GRP * LIM_B == PROT_GROUP * PROT_MAX_BYTES, "both variants use the same staging area");
    // a read is staged at the byte alignment it has in global memory (whole dwords move: a quarter of the load / store instructions
    // of a byte copy), so its region is four bytes longer than the longest read
    constexpr uint32_t LIM_BS = LIM_B + 4;
    static_assert(GRP * LIM_BS <= PROT_GROUP * (PROT_MAX_BYTES + 4), "the staging area holds the regions of both variants");
    __shared__ uint32_t lds_fstart[4][64], lds_fwin[4][64], lds_fbyte[4][64], lds_total[4][PROT_GROUP];
    __shared__ __attribute__((aligned(16))) uint8_t lds_bytes[4][PROT_GROUP * (PROT_MAX_BYTES + 4)];   // also the LCA scratch of the aggregate stage (u32 arrays)
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 2, j = lane & 3;
    if (WK.gate && !BIG) {     // (the large-scratch pass needs no gate: its list is empty when no first pass ran)
        const uint32_t g0 = WK.gate[0], g1 = WK.gate[1], g2 = WK.gate[2];
        if (LONG ? (g0 != 0u || g1 == 0u || g2 != 0u) : ((g0 | g1) != 0u)) return;
    }
    uint32_t* wtaxa = lds_taxa[wv];
    uint32_t *wkeys, *wcnt;
    if constexpr (BIG) {
        wkeys = lds_scratch[wv].hkeys;      // (512 words: also the item list of the WALK variant's second phase)
        wcnt = lds_scratch[wv].hcnt;
        for (uint32_t i = lane; i < 512u; i += 64) { wkeys[i] = 0u; wcnt[i] = 0u; }
    } else {
        wkeys = lds_scratch[wv].keys;
        wcnt = lds_scratch[wv].cnt;
    }
    uint8_t* wbytes = lds_bytes[wv];
    uint32_t* wfstart = lds_fstart[wv];
    uint32_t* wfwin = lds_fwin[wv];
    uint32_t* wfbyte = lds_fbyte[wv];
    uint32_t* wowner = lds_taxa[wv];   // the strip is dead until the first probe results arrive: its head holds the lane -> run table
    if constexpr (!BIG) {
#pragma unroll
        for (int i = 0; i < 4; i++) { wkeys[lane + 64 * i] = 0u; wcnt[lane + 64 * i] = 0u; }
    }
    const bool tree_like = method != UMGAP_METHOD_RMQ;
    // group size: GRP reads (<= LIM_W windows each, so the group fits the strip), at most 64 fragments (one lane each)
    uint32_t G = BIG ? 1u : GRP;
    if (G * F > 64) G = 64 / F;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_groups = BIG ? (size_t)*defer_count : (n_reads + G - 1) / G;
    // group layout: lane f = fragment f of the group (read k = f / F).  The four fragment offsets a lane needs are fetched ONE GROUP
    // AHEAD (they arrive while the group before is worked on: one dependent round trip less per group)
    const uint32_t k_of = lane / F;
    uint64_t ahead[4] = {0, 0, 0, 0};
    auto fetch_offsets = [&](size_t r0x, uint32_t nfrx, uint64_t (&o)[4]) {
        if (lane < nfrx) {
            const uint64_t* fo = foff + r0x * F;
            o[0] = fo[lane]; o[1] = fo[lane + 1]; o[2] = fo[k_of * F]; o[3] = fo[k_of * F + F];
        }
    };
    if constexpr (!BIG) {
        if (wave0 < n_groups) fetch_offsets(wave0 * G, (uint32_t)min((size_t)G, n_reads - wave0 * G) * F, ahead);
    }
    for (size_t grp = wave0; grp < n_groups; grp += nwaves) {
        const size_t r0 = BIG ? (size_t)todo[grp] : grp * G;
        const uint32_t nr = BIG ? 1u : (uint32_t)min((size_t)G, n_reads - r0);
        const uint32_t nfr = nr * F;                       // fragments of the group, lane f owns fragment f
        uint64_t off4[4] = {ahead[0], ahead[1], ahead[2], ahead[3]};
        if constexpr (BIG) fetch_offsets(r0, nfr, off4);
        else if (grp + nwaves < n_groups) fetch_offsets((grp + nwaves) * G, (uint32_t)min((size_t)G, n_reads - (grp + nwaves) * G) * F, ahead);
        uint32_t my_len = 0, my_rel = 0, rd_bytes = 0;
        uint64_t rd_base = 0;
        if (lane < nfr) {
            const uint64_t b0 = off4[0], b1 = off4[1];
            rd_base = off4[2];
            const uint64_t rb = off4[3] - rd_base;
            rd_bytes = rb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rb;
            my_len = (b1 - b0) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(b1 - b0);
            my_rel = (uint32_t)(b0 - rd_base);
        }
        uint32_t my_win = my_len >= 9 ? my_len - 8 : 0u;
        // windows of the lane's read (segmented sum over its F fragments; every lane takes part in the shuffles)
        uint32_t rd_win = 0;
        for (uint32_t t = 0; t < F; t++) {
            uint32_t wsrc = (uint32_t)__shfl((int)my_win, (int)((k_of * F + t) & 63), 64);
            uint32_t add = wsrc > LIM_W ? LIM_W + 1 : wsrc;
            rd_win = rd_win + add > LIM_W + 1 ? LIM_W + 1 : rd_win + add;
        }
        const bool unsupported = lane < nfr && (rd_bytes > LIM_B || rd_win > LIM_W);
        if (unsupported || lane >= nfr) my_win = 0;
        // dense strip layout: exclusive prefix sum of the window counts
        uint32_t incl = my_win;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += t; }
        uint32_t my_start = incl - my_win;
        const uint32_t W = (uint32_t)__shfl((int)incl, 63, 64);          // windows of the group (<= STRIP)
        if (GENERAL) {                                                   // read k starts at k * 128 (every lane shuffles)
            const uint32_t rd_start = (uint32_t)__shfl((int)my_start, (int)((k_of * F) & 63), 64);
            my_start = k_of * LIM_W + (my_start - rd_start);
        }
        wfstart[lane] = lane < nfr ? my_start : 0xFFFFFFFFu;
        wfwin[lane] = my_win;
        wfbyte[lane] = k_of * LIM_BS + ((uint32_t)(uintptr_t)(bytes + rd_base) & 3u) + my_rel;
        if (lane < nfr && lane % F == 0) lds_total[wv][k_of] = unsupported ? 0xFFFFFFFFu : rd_win;
        // ---- stage the reads' bytes (region k * LIM_BS per read; the loads of all reads are in flight together) ----
        for (uint32_t k = 0; k < nr; k++) {
            const uint32_t src = k * F;
            const uint32_t nb_lo = (uint32_t)__shfl((int)rd_bytes, (int)src, 64);
            const uint32_t b_lo = (uint32_t)__shfl((int)(uint32_t)rd_base, (int)src, 64), b_hi = (uint32_t)__shfl((int)(uint32_t)(rd_base >> 32), (int)src, 64);
            const bool skip = (bool)__shfl((int)unsupported, (int)src, 64);
            if (skip) continue;   // wave-uniform
            const uint8_t* gp = bytes + (((uint64_t)b_hi << 32) | b_lo);
            const uint32_t a = (uint32_t)(uintptr_t)gp & 3u;               // (the dwords around the read's ends lie in the words its own bytes lie in)
            const uint32_t* gw = reinterpret_cast<const uint32_t*>(gp - a);
            uint32_t* lw = reinterpret_cast<uint32_t*>(wbytes + k * LIM_BS);
            const uint32_t nd = (nb_lo + a + 3u) >> 2;
            for (uint32_t i = lane; i < nd; i += 64) lw[i] = gw[i];
        }
        wave_lds_sync();
        // ---- probe: every lane walks a run of R consecutive windows of ONE fragment with a rolling key ----
        // (one LDS byte and a shift per window instead of nine bytes and a search for the owning fragment; batch t = the
        // t-th window of every lane; runs never cross a fragment because lanes are dealt out per fragment.)
        uint32_t R = (W + 63) / 64, my_lanes, lanes_incl, n_lanes;
        if (R == 0) R = 1;
        for (;;) {   // smallest run length whose lanes fit the wave (wave-uniform loop; usually 1-2 trips)
            my_lanes = (my_win + R - 1) / R;
            lanes_incl = my_lanes;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)lanes_incl, o, 64); if (lane >= (uint32_t)o) lanes_incl += t; }
            n_lanes = (uint32_t)__shfl((int)lanes_incl, 63, 64);
            if (n_lanes <= 64) break;
            R++;
        }
        {
            const uint32_t first_lane = lanes_incl - my_lanes;
            for (uint32_t x = 0; x < my_lanes; x++) wowner[first_lane + x] = lane | (x << 8);   // fragment id, run number
        }
        wave_lds_sync();
        uint32_t run_cnt = 0, sb = 0, poison = 0, bo = 0;
        uint64_t key = 0;
        if (lane < n_lanes) {
            const uint32_t ow = wowner[lane], f = ow & 0xFFu, w0 = (ow >> 8) * R;
            const uint32_t fw = wfwin[f];
            run_cnt = fw - w0 < R ? fw - w0 : R;
            sb = wfstart[f] + w0;
            bo = wfbyte[f] + w0;
        }
        wave_lds_sync();   // the table is read; the strip may be written now
        if (GENERAL) {     // the gaps between the reads must read "no hit"
#pragma unroll
            for (uint32_t i = 0; i < STRIP; i += 64) wtaxa[i + lane] = 0u;
            wave_lds_sync();
        }
        if constexpr (WALK) {
            // ---- phase 1: the first window of every run is its ANCHOR: one quad-probe batch (lane = key), then the walk ----
            uint32_t open_mask = 0;                                // windows of the run (bit t) that phase 1 leaves to phase 2 (a run has at most 16)
            {
                uint64_t akey = 0;
                uint32_t apoison = 0;
                if (run_cnt) {
#pragma unroll
                    for (int c9 = 0; c9 < 9; c9++) {
                        const uint32_t code = (uint32_t)wbytes[bo + c9] - 'A';
                        if (code >= 26u) apoison = c9 + 1;        // that many windows, this one included, hold the byte
                        akey = (akey << 5) | (code & 31u);
                    }
                }
                uint32_t valid = 0, loc = 0, tg = 0;
                if (run_cnt && apoison == 0) valid = locate(T, akey, loc, tg) ? 1u : 0u;
                ProbeBatch B;
                uint32_t pr[4];
                probe_issue(T, loc, tg | (valid << 31), lane, B);
                probe_eval(T, B, NONE, lane, pr);
                // the answer to lane L's key sits, quad-uniform, in quad L & 15 as round L >> 4
                const int src = 4 * (int)(lane & 15u);
                const uint32_t v0 = (uint32_t)__shfl((int)pr[0], src, 64), v1 = (uint32_t)__shfl((int)pr[1], src, 64),
                               v2 = (uint32_t)__shfl((int)pr[2], src, 64), v3 = (uint32_t)__shfl((int)pr[3], src, 64);
                const uint32_t r4 = lane >> 4;
                const uint32_t q32 = r4 == 0 ? v0 : (r4 == 1 ? v1 : (r4 == 2 ? v2 : v3));
                const WalkRef& WR = WK.W;
                const uint64_t pos = (run_cnt && valid) ? walk_position(WR, q32, akey) : WALK_NOPOS;
                if (run_cnt) {
                    // The walk (round 5: it goes ON behind a window that does not match).  Along the entries pos, pos + 1, ... the PROTEOME's own
                    // windows are rebuilt: entry p + t continues entry p + t - 1 (its flag) means window p + t is the next window of the same
                    // protein, so its key is the previous one shifted by the entry's low five key bits.  Where that key equals the read's
                    // rolling key (all 45 bits: exact) the entry's taxon is the answer; where not -- a changed residue in the read: nine windows
                    // -- the window is left to phase 2 and the walk continues: behind the changed residue the read meets the proteome again.
                    // (Rounds 3-4 stopped the walk at the first mismatch and probed every window behind it.)  A window with a byte outside
                    // 'A'..'Z' is no key: answered here.
                    uint64_t wkey = akey, K = akey;
                    uint32_t wpoison = apoison;
                    bool chain = pos != WALK_NOPOS;
                    for (uint32_t c0 = 0; c0 < run_cnt; c0 += 16) {
                        uint64_t e[16];
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) e[k] = (chain && c0 + k < run_cnt && pos + c0 + k < WR.n) ? WR.E[pos + c0 + k] : WALK_PAD_ENTRY;   // (past the end: continues nothing)
#pragma unroll
                        for (uint32_t k = 0; k < 16; k++) {
                            const uint32_t t = c0 + k;
                            if (t < run_cnt) {
                                if (t == 0) wtaxa[sb] = chain ? (uint32_t)e[0] : 0u;     // the anchor was probed (or holds no key): answered
                                else {                                                // the next residue enters, the oldest leaves
                                    const uint32_t code = (uint32_t)wbytes[bo + t + 8] - 'A';
                                    wkey = ((wkey << 5) | (code & 31u)) & MASK45;
                                    wpoison = code >= 26u ? 9u : (wpoison ? wpoison - 1u : 0u);
                                    const uint32_t ey = (uint32_t)(e[k] >> 32);
                                    chain = chain && (ey >> 31);
                                    K = ((K << 5) | (uint64_t)(ey & 31u)) & MASK45;
                                    if (wpoison) wtaxa[sb + t] = 0u;
                                    else if (chain && K == wkey) wtaxa[sb + t] = (uint32_t)e[k];
                                    else open_mask |= 1u << (t & 31u);
                                }
                            }
                        }
                    }
                }
            }
            // ---- phase 2: what the walks did not reach, COMPACTED: the unanswered windows of all lanes form one list (strip index | byte
            // offset << 10, in the count tables of the aggregate stage, which are idle and all zero now) and are probed 128 at a time,
            // lane = list item -- dense batches instead of one sparse step per window of the longest hole; a hit takes its taxon from
            // the walk array ----
            {
                const uint32_t mine_n = (uint32_t)__builtin_popcount(open_mask);
                uint32_t incl2 = mine_n;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t tt = (uint32_t)__shfl_up((int)incl2, o, 64); if (lane >= (uint32_t)o) incl2 += tt; }
                const uint32_t my_first = incl2 - mine_n, U = (uint32_t)__shfl((int)incl2, 63, 64);       // U = unanswered windows of the group (wave-uniform)
                uint32_t* list = wkeys;                                  // 512 words: keys[256] and cnt[256] lie back to back
                for (uint32_t l0 = 0; l0 < U; l0 += 512) {               // wave-uniform; one trip unless more than half the strip is unanswered
                    {
                        uint32_t m = open_mask, at = my_first;
                        while (m) {
                            const uint32_t t = (uint32_t)__builtin_ctz(m);
                            m &= m - 1u;
                            if (at >= l0 && at < l0 + 512) list[at - l0] = (sb + t) | ((bo + t) << 10);
                            at++;
                        }
                    }
                    wave_lds_sync();
                    const uint32_t n_here = U - l0 < 512u ? U - l0 : 512u;
                    for (uint32_t i0 = 0; i0 < n_here; i0 += 128) {
                        ProbeBatch B[2];
                        uint32_t pr[2][4], item[2];
                        uint64_t ikey[2] = {0, 0};
                        const bool two = i0 + 64 < n_here;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (h == 1 && !two) { probe_skip(B[1]); item[1] = NONE; break; }   // wave-uniform
                            const uint32_t i = i0 + 64 * h + lane;
                            item[h] = i < n_here ? list[i] : NONE;
                            uint32_t valid = 0, loc = 0, tg = 0;
                            if (item[h] != NONE) {
                                const uint32_t byte0 = item[h] >> 10;
                                uint64_t k9 = 0;
                                bool ok = true;
#pragma unroll
                                for (int c9 = 0; c9 < 9; c9++) {
                                    const uint32_t code = (uint32_t)wbytes[byte0 + c9] - 'A';
                                    ok = ok && code < 26u;
                                    k9 = (k9 << 5) | (code & 31u);
                                }
                                if (ok) valid = locate(T, k9, loc, tg) ? 1u : 0u;
                                ikey[h] = k9;
                            }
                            probe_issue(T, loc, tg | (valid << 31), lane, B[h]);
                        }
                        probe_eval_multi<2>(T, B, NONE, lane, pr);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (h == 1 && !two) break;                     // wave-uniform
                            // the answer to lane L's item sits, quad-uniform, in quad L & 15 as round L >> 4
                            const int src = 4 * (int)(lane & 15u);
                            const uint32_t v0 = (uint32_t)__shfl((int)pr[h][0], src, 64), v1 = (uint32_t)__shfl((int)pr[h][1], src, 64),
                                           v2 = (uint32_t)__shfl((int)pr[h][2], src, 64), v3 = (uint32_t)__shfl((int)pr[h][3], src, 64);
                            const uint32_t r4 = lane >> 4;
                            const uint32_t q32 = r4 == 0 ? v0 : (r4 == 1 ? v1 : (r4 == 2 ? v2 : v3));
                            if (item[h] != NONE) wtaxa[item[h] & 1023u] = walk_taxon(WK.W, q32, ikey[h], 0u);
                        }
                    }
                    wave_lds_sync();
                }
                if (U) {                                                  // the count tables go back to all zero
#pragma unroll
                    for (int i = 0; i < 8; i++) list[lane + 64 * i] = 0u;
                    wave_lds_sync();
                }
            }
        } else {
            for (uint32_t t = 0; t < R; t += 2) {
                ProbeBatch B[2];
                uint32_t pr[2][4];
                const bool two = t + 1 < R;
    #pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (h == 1 && !two) { probe_skip(B[1]); break; }   // wave-uniform
                    const uint32_t tt = t + h;
                    const bool active = tt < run_cnt;
                    uint32_t valid = 0, loc = 0, tg = 0;
                    if (tt == 0) {            // wave-uniform: the first window of every run is read whole
                        if (active) {
    #pragma unroll
                            for (int c9 = 0; c9 < 9; c9++) {
                                const uint32_t code = (uint32_t)wbytes[bo + c9] - 'A';
                                if (code >= 26u) poison = c9 + 1;             // that many windows, this one included, hold the byte
                                key = (key << 5) | (code & 31u);
                            }
                        }
                    } else if (active) {      // the next residue enters, the oldest leaves
                        const uint32_t code = (uint32_t)wbytes[bo + tt + 8] - 'A';
                        key = ((key << 5) | (code & 31u)) & MASK45;
                        poison = code >= 26u ? 9u : (poison ? poison - 1u : 0u);
                    }
                    if (active && poison == 0) valid = locate(T, key, loc, tg) ? 1u : 0u;
                    probe_issue(T, loc, tg | (valid << 31), lane, B[h]);
                }
                probe_eval_multi<2>(T, B, 0u, lane, pr);
                // the result of round x belongs to the window of lane 16x + q: fetch its strip position (every lane shuffles)
    #pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t sidx = (t + h < run_cnt) ? sb + t + h : NONE;
    #pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t dst = (uint32_t)__shfl((int)sidx, 16 * x + (int)q, 64);
                        if (j == 0 && dst != NONE && (h == 0 || two)) wtaxa[dst] = pr[h][x];
                    }
                }
            }
        }
        wave_lds_sync();
        // ---- seedextend: all fragments of the group at once, one lane each ----
        uint64_t se_panic = 0;
        if (do_seed) {
            bool pk = false;
            if (lane < nfr && my_win) pk = seedextend_lds(wtaxa + my_start, my_win, min_seed, max_gap);
            se_panic = __ballot(pk);
            wave_lds_sync();
        }
        if constexpr (LONG) {
            // ---- LONG reads: the hits that are left (taxa2agg drops the zeros anyway, src/commands/taxa2agg.rs:169) are moved to the
            // front of the read's part of the strip -- for the general aggregate to k * 128, where it expects read k -- and the read goes
            // on with at most PROT_MAX_WINDOWS of them; a read with more is reported like an over-long one.  Moving towards lower
            // addresses, read by read and 64 hits at a time, never overwrites a hit that is still to be read. ----
            for (uint32_t k = 0; k < nr; k++) {
                const uint32_t total = lds_total[wv][k];
                if (total == 0xFFFFFFFFu) continue;                    // wave-uniform
                const uint32_t src = GENERAL ? k * LIM_W : wfstart[k * F], dst = GENERAL ? k * PROT_MAX_WINDOWS : src;
                uint32_t cnt = 0;
                for (uint32_t c0 = 0; c0 < total; c0 += 64) {
                    const uint32_t v = c0 + lane < total ? wtaxa[src + c0 + lane] : 0u;
                    const uint64_t m = __ballot(v != 0u);
                    wave_lds_sync();                                    // every lane holds its hit before the slots below it are rewritten
                    const uint32_t at = cnt + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (v != 0u && at < PROT_MAX_WINDOWS) wtaxa[dst + at] = v;
                    cnt += (uint32_t)__builtin_popcountll(m);
                    wave_lds_sync();
                }
                if (GENERAL) {                                          // the general aggregate reads all 128 slots of a read
                    for (uint32_t i = (cnt < PROT_MAX_WINDOWS ? cnt : PROT_MAX_WINDOWS) + lane; i < PROT_MAX_WINDOWS; i += 64) wtaxa[dst + i] = 0u;
                }
                if (lane == 0) lds_total[wv][k] = cnt > PROT_MAX_WINDOWS ? 0xFFFFFFFFu : cnt;
                wave_lds_sync();
            }
        }
        if constexpr (BIG) {
            // ---- the large-scratch pass: the one read of the wave through general_read (up to 128 distinct taxa) ----
            const uint32_t total = lds_total[wv][0];
            uint32_t res_taxon = 0, res_status = UMGAP_READ_UNSUPPORTED;
            bool ok = false;
            if (total != 0xFFFFFFFFu)
                ok = general_read<false, 128, 512, BOUND>(V, lds_scratch[wv], method, strategy, factor, lower_bound, wtaxa, nullptr, PROT_MAX_WINDOWS, lane, res_taxon, res_status);
            if (!ok) { res_taxon = 0; res_status = UMGAP_READ_UNSUPPORTED; }
            if (se_panic & ((F >= 64 ? 0ull : (1ull << F)) - 1ull)) { res_taxon = 0; res_status = UMGAP_READ_PANIC; }
            wave_lds_sync();
            if (lane == 0) { out_taxon[r0] = res_taxon; out_status[r0] = res_status; }
            wave_lds_sync();
            continue;
        } else if constexpr (GENERAL) {
            // ---- aggregate, any strategy: reads at stride 128; a read with more than 64 distinct surviving taxa goes to the `defer`
            // list (the large-scratch pass behind this kernel: BIG), over-long ones are reported UMGAP_READ_UNSUPPORTED ----
            static_assert(GROUP_VIEW_SURVIVOR_WORDS * 4 <= PROT_GROUP * PROT_MAX_BYTES, "the group scratch must fit the staging area");
            GroupView GV = group_view_over(wkeys, wcnt, reinterpret_cast<uint32_t*>(wbytes), lds_scratch[wv].reads);
            const uint64_t unsupported_reads = __ballot(lane < nr && lds_total[wv][lane] == 0xFFFFFFFFu);
            general_group<BOUND>(V, GV, method, strategy, factor, lower_bound, wtaxa, PROT_MAX_WINDOWS, nr, lane, se_panic, F, r0, out_taxon, out_status, defer, defer_count,
                                 unsupported_reads);
            wave_lds_sync();
            continue;
        }
        // ---- aggregate: the group's reads together (agg_group.hpp); its scratch lies over the byte staging area, which is dead now ----
        static_assert(lca_view_words(PROT_GROUP * PROT_MAX_WINDOWS / 8, PROT_GROUP) * 4 <= PROT_GROUP * PROT_MAX_BYTES, "the LCA scratch must fit the staging area");
        const LcaView LV = lca_view_over(wkeys, wcnt, reinterpret_cast<uint32_t*>(wbytes), PROT_GROUP * PROT_MAX_WINDOWS / 8, PROT_GROUP);
        lca_group_spans<BOUND>(V, LV, tree_like, lower_bound, wtaxa,
                               [&](uint32_t k, uint32_t& start, uint32_t& n) {
                                   const uint32_t total = lds_total[wv][k];
                                   start = wfstart[k * F];
                                   n = total == 0xFFFFFFFFu ? 0u : total;
                                   return total != 0xFFFFFFFFu;     // false: more than 128 windows or 384 bytes (use lookup_prot9 + aggregate for such reads)
                               },
                               nr, lane, se_panic, F, r0, out_taxon, out_status,
                               [&](uint32_t k, uint32_t& res_taxon, uint32_t& res_status) {
                                   if (BOUND) lca_star_bound_from_lds(V, wtaxa + wfstart[k * F], lds_total[wv][k], lower_bound, wkeys, wcnt, lane, res_taxon, res_status);
                                   else lca_star_from_lds(V, tree_like, wtaxa + wfstart[k * F], lds_total[wv][k], lower_bound, wkeys, wcnt, lane, res_taxon, res_status);
                               },
                               4u);
        wave_lds_sync();
    }
}

}  // namespace

namespace umgap {
// seedextend + aggregate of reads whose hits came back over the exchange (shard.hip): the fused kernels with the probe stage
// replaced by a gather through `perm`.  Same argument rules as umgap_classify9[_strategy]_dev, checked by the caller.
int launch_classify9_from_hits(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only, const HitSource& H,
                               size_t n_reads, uint32_t kmers_per_read, uint32_t fragments_per_read, uint32_t min_seed_size, uint32_t max_gap_size,
                               uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t s) {
    if (!n_reads) return UMGAP_OK;
    const uint32_t G = classify9_group_reads(kmers_per_read, fragments_per_read);
    if (H.run != G * kmers_per_read) return fail(UMGAP_ERR_INVALID, "launch_classify9_from_hits: the exchange was packed with another run length");
    // one group per wave, as many workgroups as groups: short-lived workgroups, so this piece's aggregate shares the compute
    // units with the probe and the pack of the pieces behind it
    const size_t g = ((n_reads + G - 1) / G + 3) / 4;   // (a capped, persistent grid and 4 instead of 8 pieces measured the same: 25.2-25.9 ms)
    TaxView V = view_of(tx, ranked_only);
    const TableView shard = ix->view;    // the rank's own keys (perm = LOCAL) are probed by these kernels
    if (strategy == UMGAP_STRATEGY_LCA_STAR) {
        hipLaunchKernelGGL((k_classify9<false, SRC_GATHER>), dim3((unsigned)g), dim3(256), 0, s, shard, V, method, (const uint64_t*)nullptr, n_reads, kmers_per_read,
                           fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, H);
    } else {
        if (n_reads >= 0xFFFFFFFFull) return fail(UMGAP_ERR_INVALID, "at most 2^32 - 2 reads per call");
        uint32_t* d_defer = defer_list_for(tx, s, (n_reads + 1) * sizeof(uint32_t));
        if (!d_defer) return UMGAP_ERR_NOMEM;
        uint32_t* d_count = d_defer;
        uint32_t* d_list = d_defer + 1;
        UMGAP_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL((k_classify9_group<false, SRC_GATHER>), dim3((unsigned)g), dim3(256), 0, s, shard, V, method, strategy, factor, (const uint64_t*)nullptr,
                           n_reads, kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status,
                           d_list, d_count, H);
        hipLaunchKernelGGL((k_classify9_general<128, 512, false, SRC_GATHER>), dim3(256), dim3(256), 0, s, shard, V, method, strategy, factor, (const uint64_t*)nullptr,
                           n_reads, kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status,
                           (const uint32_t*)d_list, (const uint32_t*)d_count, (uint32_t*)nullptr, (uint32_t*)nullptr, H);
    }
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}
}  // namespace umgap

namespace {
HitSource walk_source(const umgap_index* ix) {
    HitSource H;
    H.walk_E = ix->d_walk;
    H.walk_n = ix->n_walk;
    H.walk_shift = ix->walk_shift;
    H.walk_n_exc = ix->n_walk_exc;
    H.walk_exc_keys = ix->d_walk_exc_keys;
    H.walk_exc_pos = ix->d_walk_exc_pos;
    if (const char* e = getenv("UMGAP_WALK_DEBUG")) H.walk_debug = (uint32_t)atoi(e);
    return H;
}
}  // namespace

extern "C" int umgap_classify9_dev(umgap_index* ix, umgap_taxonomy* tx, int method, float lower_bound, int ranked_only,
                                   const uint64_t* d_keys45, size_t n_reads, uint32_t kmers_per_read, uint32_t fragments_per_read,
                                   uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* d_out_taxon, uint32_t* d_out_status,
                                   void* stream) {
    if (!ix || !tx || (n_reads && (!d_keys45 || !d_out_taxon || !d_out_status))) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: null argument");
    if (kmers_per_read == 0 || kmers_per_read > 128) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: kmers_per_read must be in 1..128 (use lookup9 + aggregate for longer reads)");
    if (fragments_per_read && (kmers_per_read % fragments_per_read) != 0) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: kmers_per_read must be a multiple of fragments_per_read");
    if (fragments_per_read > 64) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: at most 64 fragments per read");
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: needs a 9-mer index");
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: index and taxonomy live on different devices");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (!n_reads) return UMGAP_OK;
    uint32_t G = 1024u / kmers_per_read;
    if (fragments_per_read && G * fragments_per_read > 64) G = 64 / fragments_per_read;
    size_t groups = (n_reads + G - 1) / G;
    size_t g = (groups + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;     // measured: 8 workgroups per CU (two resident rounds) beats 4 and 16
    if (ix->bound_to && ix->bound_to != tx) return fail(UMGAP_ERR_INVALID, "umgap_classify9_dev: the index is bound to another taxonomy");
    if (ix->kind == 10) {      // walk order: one probe per run of neighbouring 9-mers, the rest from the walk array (walk_group_to_strip)
        const HitSource H = walk_source(ix);
        if (ix->bound_to)
            hipLaunchKernelGGL((k_classify9<true, SRC_WALK>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ix->view, view_of(tx, ranked_only), method, d_keys45, n_reads,
                               kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, H);
        else
            hipLaunchKernelGGL((k_classify9<false, SRC_WALK>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ix->view, view_of(tx, ranked_only), method, d_keys45, n_reads,
                               kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, H);
    } else if (ix->bound_to)
        hipLaunchKernelGGL((k_classify9<true, SRC_PROBE>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ix->view, view_of(tx, ranked_only), method, d_keys45, n_reads,
                           kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, HitSource{});
    else
        hipLaunchKernelGGL((k_classify9<false, SRC_PROBE>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ix->view, view_of(tx, ranked_only), method, d_keys45, n_reads,
                           kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, HitSource{});
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

extern "C" int umgap_classify9_strategy_dev(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                            int ranked_only, const uint64_t* d_keys45, size_t n_reads, uint32_t kmers_per_read,
                                            uint32_t fragments_per_read, uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* d_out_taxon,
                                            uint32_t* d_out_status, void* stream) {
    if (strategy == UMGAP_STRATEGY_LCA_STAR)
        return umgap_classify9_dev(ix, tx, method, lower_bound, ranked_only, d_keys45, n_reads, kmers_per_read, fragments_per_read, min_seed_size,
                                   max_gap_size, d_out_taxon, d_out_status, stream);
    if (!ix || !tx || (n_reads && (!d_keys45 || !d_out_taxon || !d_out_status))) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: null argument");
    if (ix->bound_to && ix->bound_to != tx) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: the index is bound to another taxonomy");
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: needs a 9-mer index");
    if (kmers_per_read == 0 || kmers_per_read > 128) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: kmers_per_read must be in 1..128 (use lookup9 + aggregate for longer reads)");
    if (fragments_per_read && (kmers_per_read % fragments_per_read) != 0) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: kmers_per_read must be a multiple of fragments_per_read");
    if (fragments_per_read > 64) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: at most 64 fragments per read");
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (strategy != UMGAP_STRATEGY_HYBRID && strategy != UMGAP_STRATEGY_MRTL) return fail(UMGAP_ERR_INVALID, "unknown strategy");
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        return fail(UMGAP_ERR_INVALID, "Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");  // src/commands/taxa2agg.rs:134-138
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: index and taxonomy live on different devices");
    if (n_reads >= 0xFFFFFFFFull) return fail(UMGAP_ERR_INVALID, "umgap_classify9_strategy_dev: at most 2^32 - 2 reads per call");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (!n_reads) return UMGAP_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* d_defer = defer_list_for(tx, s, (n_reads + 1) * sizeof(uint32_t));
    if (!d_defer) return UMGAP_ERR_NOMEM;
    uint32_t* d_count = d_defer;
    uint32_t* d_list = d_defer + 1;
    UMGAP_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
    uint32_t G = 1024u / kmers_per_read;
    if (fragments_per_read && G * fragments_per_read > 64) G = 64 / fragments_per_read;
    size_t g = ((n_reads + G - 1) / G + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    TaxView V = view_of(tx, ranked_only);
    // first pass, then the reads with more than 64 distinct taxa (at most 128: kmers_per_read <= 128)
    const HitSource HS = ix->kind == 10 ? walk_source(ix) : HitSource{};
#define UMGAP_LAUNCH_GENERAL(BOUND, SRC)                                                                                                         \
    hipLaunchKernelGGL((k_classify9_group<BOUND, SRC>), dim3((unsigned)g), dim3(256), 0, s, ix->view, V, method, strategy, factor, d_keys45,           \
                       n_reads, kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status,           \
                       d_list, d_count, HS);                                                                                                       \
    hipLaunchKernelGGL((k_classify9_general<128, 512, BOUND, SRC>), dim3(256), dim3(256), 0, s, ix->view, V, method, strategy, factor, d_keys45, n_reads, \
                       kmers_per_read, fragments_per_read, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status,                    \
                       (const uint32_t*)d_list, (const uint32_t*)d_count, (uint32_t*)nullptr, (uint32_t*)nullptr, HS)
    if (ix->kind == 10) { if (ix->bound_to) { UMGAP_LAUNCH_GENERAL(true, SRC_WALK); } else { UMGAP_LAUNCH_GENERAL(false, SRC_WALK); } }
    else if (ix->bound_to) { UMGAP_LAUNCH_GENERAL(true, SRC_PROBE); } else { UMGAP_LAUNCH_GENERAL(false, SRC_PROBE); }
#undef UMGAP_LAUNCH_GENERAL
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}

extern "C" int umgap_classify_prot9_dev(umgap_index* ix, umgap_taxonomy* tx, int method, float lower_bound, int ranked_only,
                                        const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads, uint32_t fragments_per_read,
                                        int seedextend, uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* d_out_taxon,
                                        uint32_t* d_out_status, void* stream) {
    return umgap_classify_prot9_strategy_dev(ix, tx, method, UMGAP_STRATEGY_LCA_STAR, 0.0f, lower_bound, ranked_only, d_bytes, d_frag_offsets, n_reads,
                                             fragments_per_read, seedextend, min_seed_size, max_gap_size, d_out_taxon, d_out_status, stream);
}

extern "C" int umgap_classify_prot9_strategy_dev(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                                 int ranked_only, const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads,
                                                 uint32_t fragments_per_read, int seedextend, uint32_t min_seed_size, uint32_t max_gap_size,
                                                 uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream) {
    // reads of more than PROT_SHORT_FRAGS fragments (a pair in six frames: 12) take the LONG variant: 512 windows / 1536 residues per read
    return umgap::classify_prot9_gated(ix, tx, method, strategy, factor, lower_bound, ranked_only, d_bytes, d_frag_offsets, n_reads, fragments_per_read, seedextend,
                                       min_seed_size, max_gap_size, d_out_taxon, d_out_status, nullptr, fragments_per_read > PROT_SHORT_FRAGS ? 1 : 0, stream);
}
// `gate` (device, three words, nullable) and `long_variant`: prot_limits.hpp
int umgap::classify_prot9_gated(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound,
                                int ranked_only, const uint8_t* d_bytes, const uint64_t* d_frag_offsets, size_t n_reads,
                                uint32_t fragments_per_read, int seedextend, uint32_t min_seed_size, uint32_t max_gap_size,
                                uint32_t* d_out_taxon, uint32_t* d_out_status, const uint32_t* gate, int long_variant, void* stream) {
    if (!ix || !tx || (n_reads && (!d_bytes || !d_frag_offsets || !d_out_taxon || !d_out_status))) return fail(UMGAP_ERR_INVALID, "umgap_classify_prot9_dev: null argument");
    if (ix->bound_to && ix->bound_to != tx) return fail(UMGAP_ERR_INVALID, "umgap_classify_prot9_dev: the index is bound to another taxonomy");
    if (ix->kind != 9 && ix->kind != 10) return fail(UMGAP_ERR_INVALID, "umgap_classify_prot9_dev: needs a 9-mer index");
    if (fragments_per_read == 0 || fragments_per_read > (long_variant ? PROT_LONG_FRAGS : PROT_MAX_FRAGS))
        return fail(UMGAP_ERR_INVALID, "umgap_classify_prot9_dev: fragments_per_read must be in 1..32");
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (strategy != UMGAP_STRATEGY_LCA_STAR && strategy != UMGAP_STRATEGY_HYBRID && strategy != UMGAP_STRATEGY_MRTL) return fail(UMGAP_ERR_INVALID, "unknown strategy");
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_classify_prot9_dev: index and taxonomy live on different devices");
    int rc = use_device(ix->device);
    if (rc) return rc;
    if (!n_reads) return UMGAP_OK;
    uint32_t G = long_variant ? STRIP / PROT_LONG_WINDOWS : STRIP / PROT_MAX_WINDOWS;
    if (G * fragments_per_read > 64) G = 64 / fragments_per_read;
    size_t g = ((n_reads + G - 1) / G + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    const bool general = strategy != UMGAP_STRATEGY_LCA_STAR, bound = ix->bound_to != nullptr, walk = ix->kind == 10;
    WalkView WK;
    WK.W = walk_ref_of(ix);
    WK.gate = gate;
    // MRTL / hybrids: the reads with more than 64 distinct taxa go to a list (one per stream, shared by the variants of a gated pair:
    // at most one of them runs) that the large-scratch pass works through behind the first
    uint32_t *d_count = nullptr, *d_list = nullptr;
    if (general) {
        if (n_reads >= 0xFFFFFFFFull) return fail(UMGAP_ERR_INVALID, "umgap_classify_prot9_dev: at most 2^32 - 2 reads per call");
        uint32_t* d_defer = defer_list_for(tx, (hipStream_t)stream, (n_reads + 1) * sizeof(uint32_t));
        if (!d_defer) return UMGAP_ERR_NOMEM;
        d_count = d_defer;
        d_list = d_defer + 1;
        // (a gated pair: cleared once, before its first launch -- which is the LONG one when the reads have too many fragments for the short)
        if (!gate || !long_variant || fragments_per_read > PROT_MAX_FRAGS) UMGAP_HIP(hipMemsetAsync(d_count, 0, sizeof(uint32_t), (hipStream_t)stream));
    }
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ix->view, view_of(tx, ranked_only), method, strategy, factor, d_bytes,
                           d_frag_offsets, n_reads, fragments_per_read, seedextend ? 1u : 0u, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, WK,
                           d_list, d_count, (const uint32_t*)nullptr);
    };
    auto launch_big = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, ix->view, view_of(tx, ranked_only), method, strategy, factor, d_bytes,
                           d_frag_offsets, n_reads, fragments_per_read, seedextend ? 1u : 0u, min_seed_size, max_gap_size, lower_bound, d_out_taxon, d_out_status, WK,
                           (uint32_t*)nullptr, d_count, (const uint32_t*)d_list);
    };
#define UMGAP_PROT9(LONGV)                                                                                                                      \
    if (walk) {                                                                                                                                \
        if (general) { if (bound) launch(k_classify_prot9<true, true, true, LONGV>); else launch(k_classify_prot9<false, true, true, LONGV>); }      \
        else { if (bound) launch(k_classify_prot9<true, false, true, LONGV>); else launch(k_classify_prot9<false, false, true, LONGV>); }            \
    }                                                                                                                                          \
    else if (general) { if (bound) launch(k_classify_prot9<true, true, false, LONGV>); else launch(k_classify_prot9<false, true, false, LONGV>); }   \
    else { if (bound) launch(k_classify_prot9<true, false, false, LONGV>); else launch(k_classify_prot9<false, false, false, LONGV>); }
    if (long_variant) { UMGAP_PROT9(true) } else { UMGAP_PROT9(false) }
#undef UMGAP_PROT9
    // the large-scratch pass: behind an ungated launch, or behind the LONG launch of a gated pair (the second of the two)
    if (general && (!gate || long_variant)) {
        if (walk) { if (bound) launch_big(k_classify_prot9<true, true, true, true, true>); else launch_big(k_classify_prot9<false, true, true, true, true>); }
        else { if (bound) launch_big(k_classify_prot9<true, true, false, true, true>); else launch_big(k_classify_prot9<false, true, false, true, true>); }
    }
    UMGAP_HIP(hipGetLastError());
    return UMGAP_OK;
}
