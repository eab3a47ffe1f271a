// agg_group.hpp -- the general aggregation (MRTL, tree-hybrid, rmq-hybrid) for a GROUP of reads held by one wave in LDS
// (the fused kernels of classify.hip), with the taxonomy round trips batched over the group (round 2).
//
// general_read (agg_general.hpp) takes one read at a time and pays 2 (MRTL) to 4 (hybrids) DEPENDENT cache round trips per
// read -- {pre,last} of the survivors, depth + range minimum of the adjacent pairs, their parents, the snap of the answer --
// so a group of 12 reads sat through ~30-50 of them one after another (70 % of the wave cycles in SQ_WAIT_ANY,
// profiles/r01_v8_pmc_classify.json).  Here every phase runs over all reads of the group before the next one starts:
//
//   P1  per read, LDS only : count + filter (lds_count), survivors compacted IN PLACE over the strip
//   P2  whole group        : {pre,last} of every survivor                                   1 round trip
//   P3a per read, LDS only : errors / empties, sort by preorder index; MRTL finishes here
//   P3b whole group        : depth of every survivor, adjacent-pair range minima (hybrids)  1 round trip
//                            parents of the minima (+ subtree ends for rmq-hybrid)          1 round trip
//   P3c per read, LDS only : the greedy descent (tree-hybrid) / the candidate scores (rmq-hybrid)
//   P4  whole group        : snap of the answers, one lane per read, results stored coalesced   1 round trip
//
// The per-read arithmetic (sums in `nearest ancestor first` order, the f32 division of src/tree/mix.rs:57, the products of
// src/rmq/mix.rs:52-54, ties to the smallest preorder index) is that of general_read, statement for statement: the two must
// give identical answers (tests/test_gpu_parity.py compares them bit for bit).  A group whose survivors exceed GCAP is
// processed in several sub-batches; a read with more than 64 distinct taxa is handed to the large-scratch pass as before.
#pragma once
#include "agg_general.hpp"

namespace umgap {
namespace agg {

constexpr uint32_t GCAP = 152;     // survivors (distinct filtered taxa) of one sub-batch
constexpr uint32_t GREADS = 64;    // reads of one sub-batch (one lane per read in P4)
constexpr uint32_t ST_PENDING = 0x100u, ST_DEFER = 0x101u;   // internal per-read states (not UMGAP_READ_* values)

struct GroupScratch {
    uint32_t hkeys[256], hcnt[256];   // the per-read hash table of lds_count (clean between reads)
    uint32_t last[GCAP];
    float cnt[GCAP];
    uint32_t dep[GCAP], adj[GCAP], lca[GCAP];
    uint32_t roff[GREADS + 1], rans[GREADS], rstat[GREADS];
    uint8_t rid[GCAP];                 // read (within the sub-batch) of every survivor
};

// Aggregates reads [0, nr) of a group whose taxa sit in the wave's strip (read k at wtaxa[k*stride .. +stride), stride <= 128,
// 0 = no hit).  The strip is consumed.  Results go to out_taxon/out_status[first_read + k]; a read that needs the
// large-scratch pass is appended to `defer` (or flagged UMGAP_READ_UNSUPPORTED when there is no list).  se_panic: bit
// (k * frags + f) set = seedextend panicked on fragment f of read k (classify.hip).  unsupported: bit k set = the caller could
// not stage read k (its strip is empty); reported as UMGAP_READ_UNSUPPORTED.  too_long: bit k set = read k did not fit the strip
// (left empty): it goes to the defer list like a read with too many distinct taxa.
// The same scratch as pointers, for a kernel that lays the arrays over LDS it no longer needs (k_classify_prot9).
struct GroupView {
    uint32_t *hkeys, *hcnt, *last;
    float* cnt;
    uint32_t *dep, *adj, *lca, *roff, *rans, *rstat;
    uint8_t* rid;
};
constexpr uint32_t GROUP_VIEW_SURVIVOR_WORDS = 5 * GCAP;   // last, cnt, dep, adj, lca
// survivor arrays over `mem` (GROUP_VIEW_SURVIVOR_WORDS words), per-read arrays over `reads` (3 * GREADS + 1 words + GCAP bytes)
__device__ __forceinline__ GroupView group_view_over(uint32_t* hkeys, uint32_t* hcnt, uint32_t* mem, uint32_t* reads) {
    GroupView G;
    G.hkeys = hkeys; G.hcnt = hcnt;
    G.last = mem; G.cnt = reinterpret_cast<float*>(mem + GCAP); G.dep = mem + 2 * GCAP; G.adj = mem + 3 * GCAP; G.lca = mem + 4 * GCAP;
    G.roff = reads; G.rans = reads + GREADS + 1; G.rstat = reads + 2 * GREADS + 1;
    G.rid = reinterpret_cast<uint8_t*>(reads + 3 * GREADS + 1);
    return G;
}
constexpr uint32_t GROUP_VIEW_READ_WORDS = 3 * GREADS + 1 + (GCAP + 3) / 4;

// MODE: 0 = method and strategy as passed (run-time); 1 = MRTL, 2 = tree hybrid, 3 = rmq hybrid fixed at compile time -- a kernel
// instantiated per mode carries one strategy's code and registers instead of all three (aggregate_group.hip).
template <bool BOUND, int MODE = 0, class GS>
__device__ __forceinline__ void general_group(const TaxView& V, GS& G, int method, int strategy, float factor, float lower_bound,
                                              uint32_t* wtaxa, uint32_t stride, uint32_t nr, uint32_t lane, uint64_t se_panic, uint32_t frags,
                                              size_t first_read, uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status,
                                              uint32_t* __restrict__ defer, uint32_t* __restrict__ defer_count, uint64_t unsupported = 0ull,
                                              uint64_t too_long = 0ull) {
    if (MODE == 1) { method = UMGAP_METHOD_RMQ; strategy = UMGAP_STRATEGY_MRTL; }
    else if (MODE == 2) { method = UMGAP_METHOD_TREE; strategy = UMGAP_STRATEGY_HYBRID; }
    else if (MODE == 3) { method = UMGAP_METHOD_RMQ; strategy = UMGAP_STRATEGY_HYBRID; }
    const bool tree_like = method != UMGAP_METHOD_RMQ || strategy == UMGAP_STRATEGY_MRTL;
    uint32_t k0 = 0;
    while (k0 < nr) {
        // ---- P1: count + filter, survivors compacted over the strip from `base` on -----------------------------------------
        const uint32_t base = k0 * stride;
        uint32_t w = 0, k = k0;
        if (lane == 0) G.roff[0] = 0;
        // Short reads (stride <= 16: the tryptic pipelines, CSR reads of a few hits): 64 / stride reads per pass, one lane per item.
        // The multiplicity of an item is the number of equal items in its read -- stride - 1 shuffles and compares instead of the
        // hash table below, whose ~100 instructions per READ are spent on a wave that is 75-95 % idle at this length -- and the
        // item is its taxon's leader when no equal item precedes it.  Same survivors, counts and bookkeeping as the loop below.
        const uint32_t per_pass = stride <= 16 ? 64u / stride : 0u;
        const uint32_t r_in = per_pass ? lane / stride : 0u, j_in = per_pass ? lane - r_in * stride : 0u;
        while (per_pass && k < nr && k - k0 < GREADS) {
            const uint32_t rp = min(per_pass, min(nr - k, GREADS - (k - k0)));
            const bool active = r_in < rp;
            const uint32_t t = active ? wtaxa[(k + r_in) * stride + j_in] : 0u;
            uint32_t mult = t != 0u ? 1u : 0u;
            bool lead = t != 0u;
            for (uint32_t d = 1; d < stride; d++) {
                uint32_t jo = j_in + d;
                if (jo >= stride) jo -= stride;
                const uint32_t o = (uint32_t)__shfl((int)t, (int)(lane - j_in + jo), 64);    // another item of the same read
                if (t != 0u && o == t) { mult++; if (jo < j_in) lead = false; }
            }
            const bool keep = lead && (float)mult >= lower_bound;                            // agg::filter keeps freq >= lower_bound
            const uint64_t b = __ballot(keep);
            const uint32_t n_pass = (uint32_t)__builtin_popcountll(b);
            if (w + n_pass > GCAP) break;        // (k > k0 here: a pass holds at most 64 survivors) these reads open the next sub-batch
            const uint32_t bit0 = r_in * stride;                                             // first bit of the lane's read
            const uint32_t seg = (uint32_t)(b >> bit0) & ((1u << stride) - 1u);
            const uint32_t before = (uint32_t)__builtin_popcountll(b & ((1ull << bit0) - 1ull));
            wave_sync();                         // every lane holds its taxon in a register before the strip is overwritten
            if (keep) {
                const uint32_t i = w + before + (uint32_t)__builtin_popcount(seg & ((1u << j_in) - 1u));
                wtaxa[base + i] = t;
                G.cnt[i] = (float)mult;
                G.rid[i] = (uint8_t)(k - k0 + r_in);
            }
            if (active && j_in == 0) { G.rstat[k - k0 + r_in] = ST_PENDING; G.roff[k - k0 + r_in + 1] = w + before + (uint32_t)__builtin_popcount(seg); }
            w += n_pass;
            wave_sync();
            k += rp;
        }
        while (!per_pass && k < nr && k - k0 < GREADS) {
            const uint32_t* rt = wtaxa + k * stride;
            const uint32_t t0 = lane < stride ? rt[lane] : 0u, t1 = 64 + lane < stride ? rt[64 + lane] : 0u;
            bool keep0, keep1;
            float c0, c1;
            lds_count(G.hkeys, G.hcnt, t0, t1, lower_bound, keep0, keep1, c0, c1);
            const uint64_t b0 = __ballot(keep0), b1 = __ballot(keep1);
            const uint32_t n0 = (uint32_t)__builtin_popcountll(b0), ns = n0 + (uint32_t)__builtin_popcountll(b1);
            if (ns > 64) {                      // more distinct taxa than one lane each: the large-scratch pass takes the read
                if (lane == 0) { G.rstat[k - k0] = ST_DEFER; G.roff[k - k0 + 1] = w; }
                k++;
                continue;
            }
            if (w + ns > GCAP) {
                if (k == k0) {                   // cannot happen (ns <= 64 <= GCAP), kept as a guard against an endless loop
                    if (lane == 0) { G.rstat[0] = ST_DEFER; G.roff[1] = w; }
                    k++;
                }
                break;                           // the read opens the next sub-batch (its taxa are untouched)
            }
            wave_sync();                         // every lane holds its taxa in registers before the strip is overwritten
            if (keep0) { const uint32_t i = w + (uint32_t)__builtin_popcountll(b0 & lanemask_lt(lane)); wtaxa[base + i] = t0; G.cnt[i] = c0; G.rid[i] = (uint8_t)(k - k0); }
            if (keep1) { const uint32_t i = w + n0 + (uint32_t)__builtin_popcountll(b1 & lanemask_lt(lane)); wtaxa[base + i] = t1; G.cnt[i] = c1; G.rid[i] = (uint8_t)(k - k0); }
            w += ns;
            if (lane == 0) { G.rstat[k - k0] = ST_PENDING; G.roff[k - k0 + 1] = w; }
            wave_sync();
            k++;
        }
        const uint32_t nb = k - k0;              // reads of this sub-batch
        wave_sync();
        // ---- P2: {pre, last} of every survivor, one round trip; the strip now holds preorder indexes -----------------------
        bool any_err = false;
        for (uint32_t eb = 0; eb < w; eb += 64) {
            const uint32_t e = eb + lane;
            uint64_t errkey = NOERR;
            if (e < w) {
                const uint32_t t = wtaxa[base + e];
                Hit h{NONE, NONE};
                if (BOUND) {
                    uint32_t lo, hi;
                    bound::size_bounds(t, lo, hi);
                    h.pre = bound::pre_of(t);
                    h.last = lo == hi ? h.pre + lo : V.last_of_pre[h.pre];
                } else h = resolve(V, tree_like, t, errkey);
                wtaxa[base + e] = h.pre;
                G.last[e] = h.last;
                G.dep[e] = (uint32_t)errkey;      // free until P3b: the element's error key (NOERR = all ones)
                G.adj[e] = (uint32_t)(errkey >> 32);
            }
            any_err = any_err || __any(errkey != NOERR);
        }
        wave_sync();
        // ---- P3a: errors, empties, sort by preorder index, MRTL -- LDS only, and FLAT: one lane per read for the per-read
        // decisions, one lane per survivor of the whole sub-batch (up to GCAP = 3 per lane) for the sort and the MRTL scores,
        // instead of twelve passes of a mostly idle wave (a read has ~15-25 survivors) ------------------------------------------
        if (lane < nb && G.rstat[lane] == ST_PENDING) {
            const uint32_t o0 = G.roff[lane], o1 = G.roff[lane + 1];
            uint64_t ek = NOERR;
            if (any_err)
                for (uint32_t e = o0; e < o1; e++) { const uint64_t k2 = ((uint64_t)G.adj[e] << 32) | G.dep[e]; ek = k2 < ek ? k2 : ek; }
            if (ek != NOERR) { G.rstat[lane] = ((ek >> 32) & 1) ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON; G.rans[lane] = (uint32_t)ek; }
            else if (o1 == o0) { G.rstat[lane] = UMGAP_READ_EMPTY; G.rans[lane] = 1; }   // literal "1" (src/commands/taxa2agg.rs:174-175)
        }
        wave_sync();
        {
            // sort every read's survivors by preorder index (distinct taxa => distinct keys): rank by counting inside the read,
            // all elements first read into registers, then scattered back in place
            constexpr uint32_t EPL = (GCAP + 63) / 64;
            uint32_t sp[EPL], sl[EPL], sdst[EPL];
            float sc[EPL];
#pragma unroll
            for (uint32_t u = 0; u < EPL; u++) {
                const uint32_t e = u * 64 + lane;
                sdst[u] = NONE;
                sp[u] = sl[u] = 0u;
                sc[u] = 0.0f;
                if (e < w) {
                    const uint32_t r = G.rid[e], o0 = G.roff[r], o1 = G.roff[r + 1];
                    sp[u] = wtaxa[base + e];
                    sl[u] = G.last[e];
                    sc[u] = G.cnt[e];
                    uint32_t rank = 0;
                    for (uint32_t f = o0; f < o1; f++) rank += wtaxa[base + f] < sp[u];
                    sdst[u] = o0 + rank;
                }
            }
            wave_sync();
#pragma unroll
            for (uint32_t u = 0; u < EPL; u++)
                if (sdst[u] != NONE) {
                    wtaxa[base + sdst[u]] = sp[u];
                    G.last[sdst[u]] = sl[u];
                    G.cnt[sdst[u]] = sc[u];
                    G.adj[sdst[u]] = NONE;        // "no next element in this read" until P3b fills the inner ones
                }                                 // (rid needs no move: an element stays inside its read)
            wave_sync();
        }
        if (strategy == UMGAP_STRATEGY_MRTL) {
            // score(x) = count(x) + the counts of its ancestors among the read's survivors, nearest first (src/rmq/rtl.rs:42-49)
            for (uint32_t eb = 0; eb < w; eb += 64) {
                const uint32_t e = eb + lane;
                if (e < w) {
                    const uint32_t o0 = G.roff[G.rid[e]], pi = wtaxa[base + e];
                    float score = G.cnt[e];
                    for (uint32_t j = e; j-- > o0;)
                        if (G.last[j] >= pi) score = __fadd_rn(score, G.cnt[j]);
                    G.dep[e] = __float_as_uint(score);
                }
            }
            wave_sync();
            if (lane < nb && G.rstat[lane] == ST_PENDING) {   // argmax, the first of several maxima (= smallest preorder index)
                const uint32_t o0 = G.roff[lane], o1 = G.roff[lane + 1];
                float best = -__builtin_inff();
                uint32_t be = o0;
                for (uint32_t e = o0; e < o1; e++) { const float sc2 = __uint_as_float(G.dep[e]); if (sc2 > best) { best = sc2; be = e; } }
                G.rans[lane] = wtaxa[base + be];
            }
        }
        wave_sync();
        if (strategy != UMGAP_STRATEGY_MRTL) {
            // ---- P3b: depth of every survivor and the LCA of every adjacent pair of a read, whole sub-batch at once ----------
            // element e has a successor in its read unless it is the read's last one; the reads' ends are in roff
            for (uint32_t eb = 0; eb < w; eb += 64) {
                const uint32_t e = eb + lane;
                uint32_t u = 0, v = 0, du = 0, lu = 0;
                bool inner = false, anc = false;
                uint64_t m = 0;
                if (e < w && G.rstat[G.rid[e]] == ST_PENDING) {          // a read that ended in an error holds unresolved elements (NONE):
                    u = wtaxa[base + e];                                  // none of them may reach the taxonomy tables
                    lu = G.last[e];
                    inner = G.roff[G.rid[e] + 1u] != e + 1;              // e is not the last element of its read
                    if (u == NONE) inner = false;
                    if (u != NONE) du = V.depth[u];
                    if (inner) {
                        v = wtaxa[base + e + 1];
                        anc = lu >= v;                                    // u is an ancestor of v
                        if (!anc) m = lane_argmin_depth(V, u + 1, v);     // LCA = parent of a shallowest node in (pre[u], pre[v]]
                    }
                }
                uint32_t node = u, nlast = lu;
                if (inner && !anc) {
                    node = V.parent_pre[(uint32_t)m];
                    if (method == UMGAP_METHOD_RMQ) nlast = V.last_of_pre[node];   // rmq::mix weighs the LCA candidates by their subtree
                }
                if (e < w && u != NONE && G.rstat[G.rid[e]] == ST_PENDING) {
                    if (inner) { G.lca[e] = node; G.adj[e] = anc ? du : (uint32_t)(m >> 32) - 1u; }
                    G.dep[e] = method == UMGAP_METHOD_RMQ ? nlast : du;  // tree::mix needs the depths, rmq::mix the candidates' subtree ends
                }
            }
            wave_sync();
            // ---- P3c: per read, LDS only ------------------------------------------------------------------------------------
            // a read with a single survivor is that survivor (both mixes reduce to it): all of them at once, one lane per read
            if (lane < nb && G.rstat[lane] == ST_PENDING && G.roff[lane + 1] - G.roff[lane] == 1u) G.rans[lane] = wtaxa[base + G.roff[lane]];
            wave_sync();
            // tree::mix on SHORT lists (at most SMALL survivors: what a tryptic read or a filtered 9-mer read leaves), one lane per READ
            // (round 4): the same greedy descent as the wave-wide form below, over registers.  The sums are sums of whole counts, exact
            // in f32 in any order, so both forms give the same bits.  (The wave-wide form spent ~200 instructions of the whole wave on
            // every read: 3.7 of the 6.3 ms of the tree-hybrid tryptic preset per 10 M reads of four peptides.)
            constexpr uint32_t SMALL = 8;
            if (method == UMGAP_METHOD_TREE) {
                if (lane < nb && G.rstat[lane] == ST_PENDING) {
                    const uint32_t o0 = G.roff[lane], D = G.roff[lane + 1] - o0;
                    if (D >= 2u && D <= SMALL) {
                        float pfx[SMALL];
                        uint32_t adjv[SMALL], depv[SMALL];
                        float run = 0.0f;
#pragma unroll
                        for (uint32_t k = 0; k < SMALL; k++) {
                            if (k < D) run = __fadd_rn(run, G.cnt[o0 + k]);
                            pfx[k] = run;
                            adjv[k] = k + 1 < D ? G.adj[o0 + k] : NONE;
                            depv[k] = k < D ? G.dep[o0 + k] : 0u;
                        }
                        uint32_t i = 0, j = D - 1, ans = NONE;
                        float wcur = pfx[SMALL - 1];
                        for (;;) {
                            // suffix minima of adj over [k, j) and the first k in [i, j) whose own depth differs from it: the branching
                            uint32_t suf[SMALL], br = NONE;
                            uint32_t sm = NONE;
#pragma unroll
                            for (int k = (int)SMALL - 1; k >= 0; k--) {
                                const bool inr = (uint32_t)k >= i && (uint32_t)k < j;
                                if (inr) sm = adjv[k] < sm ? adjv[k] : sm;
                                suf[k] = inr ? sm : NONE;
                                if (inr && depv[k] != sm) br = (uint32_t)k;
                            }
                            if (br == NONE) { ans = wtaxa[base + o0 + j]; break; }           // a pure chain down to the last element
                            i = br;
                            uint32_t mm = NONE;
#pragma unroll
                            for (uint32_t k = 0; k < SMALL; k++) if (k == i) mm = suf[k];
                            // children: a child ends at every k in [i, j) with adj == mm, and at j; the heaviest (first of equal ones) wins
                            uint32_t node = NONE, st = i, best_s = i, best_e = NONE;
                            float best_w = -1.0f, before = 0.0f;          // before = the sum in front of the child that starts at st
#pragma unroll
                            for (uint32_t k = 0; k < SMALL; k++) if (k + 1 == i) before = pfx[k];      // (no runtime index into the registers)
#pragma unroll
                            for (uint32_t k = 0; k < SMALL; k++) {
                                const bool inr = k >= i && k < j;
                                const bool cut = inr && adjv[k] == mm;
                                if (cut && node == NONE) node = G.lca[o0 + k];
                                if (cut || k == j) {
                                    const float seg = __fsub_rn(pfx[k], before);
                                    if (seg > best_w) { best_w = seg; best_e = k; best_s = st; }
                                    st = k + 1;
                                    before = pfx[k];
                                }
                            }
                            if (__fdiv_rn(best_w, wcur) < factor) { ans = node; break; }     // src/tree/mix.rs:57-59
                            wcur = best_w;
                            i = best_s;
                            j = best_e;
                        }
                        G.rans[lane] = ans;
                    }
                }
                wave_sync();
            } else {
                // rmq::mix on short lists, one lane per read as well: every candidate (the hits and the LCAs of adjacent hits) against
                // every hit; sums of whole counts, the two products and their sum rounded as in the wave-wide form; ties to the smallest
                // preorder index
                if (lane < nb && G.rstat[lane] == ST_PENDING) {
                    const uint32_t o0 = G.roff[lane], D = G.roff[lane + 1] - o0;
                    if (D >= 2u && D <= SMALL) {
                        const uint32_t* bp = wtaxa + base + o0;
                        const float one_minus = __fsub_rn(1.0f, factor);
                        float best = -__builtin_inff();
                        uint32_t best_p = NONE;
                        for (uint32_t cnd = 0; cnd < 2 * D - 1; cnd++) {
                            const uint32_t pc = cnd < D ? bp[cnd] : G.lca[o0 + cnd - D];
                            const uint32_t lc = cnd < D ? G.last[o0 + cnd] : G.dep[o0 + cnd - D];
                            float lw = 0.0f, rw = 0.0f;
                            for (uint32_t q = 0; q < D; q++) {
                                const uint32_t pk = bp[q];
                                if (pc <= pk && pk <= lc) lw = __fadd_rn(lw, G.cnt[o0 + q]);
                                if (pk <= pc && pc <= G.last[o0 + q]) rw = __fadd_rn(rw, G.cnt[o0 + q]);
                            }
                            const float score = __fadd_rn(__fmul_rn(lw, factor), __fmul_rn(rw, one_minus));
                            if (score > best || (score == best && pc < best_p)) { best = score; best_p = pc; }
                        }
                        G.rans[lane] = best_p;
                    }
                }
                wave_sync();
            }
            for (uint32_t r = 0; r < nb; r++) {
                if (G.rstat[r] != ST_PENDING) continue;
                const uint32_t o0 = G.roff[r], D = G.roff[r + 1] - o0;
                if (D == 1u) continue;
                if (D <= SMALL) continue;                                         // done above, one lane per read
                const uint32_t* bpre = wtaxa + base + o0;
                uint32_t ans = NONE;
                if (method == UMGAP_METHOD_TREE) {
                    // tree::mix, one lane per list element (the body of general_read's CAP <= 64 path)
                    const uint32_t kx = lane;
                    float pfx = kx < D ? G.cnt[o0 + kx] : 0.0f;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(pfx, o, 64); if (lane >= (uint32_t)o) pfx = __fadd_rn(pfx, t); }
                    const uint32_t adj_k = kx + 1 < D ? G.adj[o0 + kx] : NONE;
                    const uint32_t dep_k = kx < D ? G.dep[o0 + kx] : 0u;
                    uint32_t i = 0, j = D - 1;
                    float wcur = __shfl(pfx, (int)(D - 1), 64);
                    for (;;) {
                        bool inr = kx >= i && kx < j;
                        uint32_t suf = inr ? adj_k : NONE;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            uint32_t t = (uint32_t)__shfl_down((int)suf, o, 64);
                            if (lane + (uint32_t)o < 64) suf = t < suf ? t : suf;
                        }
                        const uint64_t branch = __ballot(inr && dep_k != suf);
                        if (!branch) { ans = bpre[j]; break; }               // a pure chain down to the last element
                        i = (uint32_t)__builtin_ctzll(branch);
                        const uint32_t mm = (uint32_t)__shfl((int)suf, (int)i, 64);
                        inr = kx >= i && kx < j;
                        const uint64_t eq = __ballot(inr && adj_k == mm);
                        const uint32_t node = G.lca[o0 + __builtin_ctzll(eq)];
                        const uint64_t ends = eq | (1ull << j);              // a child ends at every cut and at j
                        const bool is_end = (ends >> kx) & 1ull;
                        const uint64_t below = ends & lanemask_lt(kx);
                        const uint32_t st = below ? 64u - (uint32_t)__builtin_clzll(below) : i;   // where this child starts
                        const float before_raw = __shfl(pfx, (int)((st - 1u) & 63u), 64);
                        const float seg = is_end ? __fsub_rn(pfx, st ? before_raw : 0.0f) : -1.0f;
                        const float best_w = wave_maxf(seg);
                        const uint32_t best_e = wave_min(is_end && seg == best_w ? kx : NONE);   // first maximum
                        const uint32_t best_s = (uint32_t)__shfl((int)st, (int)best_e, 64);
                        if (__fdiv_rn(best_w, wcur) < factor) { ans = node; break; }  // src/tree/mix.rs:57-59
                        wcur = best_w;
                        i = best_s;
                        j = best_e;
                    }
                } else {
                    // rmq::mix: candidates = the hits and the LCAs of adjacent hits; weights by interval tests (src/rmq/mix.rs:52-54)
                    const float one_minus = __fsub_rn(1.0f, factor);
                    float best = -__builtin_inff();
                    uint32_t best_p = NONE;
                    const uint32_t n_cand = 2 * D - 1;
                    for (uint32_t cb = 0; cb < n_cand; cb += 64) {
                        const uint32_t cnd = cb + lane;
                        if (cnd < n_cand) {
                            const uint32_t pc = cnd < D ? bpre[cnd] : G.lca[o0 + cnd - D];
                            const uint32_t lc = cnd < D ? G.last[o0 + cnd] : G.dep[o0 + cnd - D];
                            float lw = 0.0f, rw = 0.0f;
                            for (uint32_t q = 0; q < D; q++) {
                                const uint32_t pk = bpre[q];
                                if (pc <= pk && pk <= lc) lw = __fadd_rn(lw, G.cnt[o0 + q]);              // hit below-or-at the candidate
                                if (pk <= pc && pc <= G.last[o0 + q]) rw = __fadd_rn(rw, G.cnt[o0 + q]);  // hit above-or-at the candidate
                            }
                            const float score = __fadd_rn(__fmul_rn(lw, factor), __fmul_rn(rw, one_minus));
                            if (score > best || (score == best && pc < best_p)) { best = score; best_p = pc; }
                        }
                    }
                    const float m = wave_maxf(best);
                    ans = wave_min(best == m ? best_p : NONE);
                }
                if (lane == 0) G.rans[r] = ans;
            }
            wave_sync();
        }
        // ---- P4: snap + store, one lane per read -----------------------------------------------------------------------------
        if (lane < nb) {
            const size_t read = first_read + k0 + lane;
            uint32_t st = G.rstat[lane], tx = G.rans[lane];
            if (st == ST_PENDING) {
                tx = V.snap_of_pre[tx];            // snapping[aggregate] (src/commands/taxa2agg.rs:178)
                st = tx == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
            }
            const uint32_t kk = k0 + lane;
            const bool sp = se_panic && ((se_panic >> (kk * frags)) & ((frags >= 64 ? 0ull : (1ull << frags)) - 1ull));
            if (sp) { st = UMGAP_READ_PANIC; tx = 0; }
            if ((unsupported >> kk) & 1ull) { st = UMGAP_READ_UNSUPPORTED; tx = 0; }
            if ((too_long >> kk) & 1ull) st = ST_DEFER;
            if (st == ST_DEFER) {
                if (defer) defer[atomicAdd(defer_count, 1u)] = (uint32_t)read;
                else { out_taxon[read] = 0; out_status[read] = UMGAP_READ_UNSUPPORTED; }
            } else {
                out_taxon[read] = tx;
                out_status[read] = st;
            }
        }
        wave_sync();
        k0 = k;
    }
}

// LCA* of one read whose taxa sit in LDS at rt[0..n) (n <= 128), shared by both fused kernels.
__device__ __forceinline__ void lca_star_from_lds(const TaxView& V, bool tree_like, const uint32_t* rt, uint32_t n, float lower_bound,
                                                  uint32_t* wkeys, uint32_t* wcnt, uint32_t lane, uint32_t& res_taxon, uint32_t& res_status) {
    uint32_t t0 = lane < n ? rt[lane] : 0u, t1 = 64 + lane < n ? rt[64 + lane] : 0u;
    bool keep0, keep1;
    float c0, c1;
    lds_count(wkeys, wcnt, t0, t1, lower_bound, keep0, keep1, c0, c1);
    uint64_t errkey = NOERR;
    Hit h0{NONE, NONE}, h1{NONE, NONE};
    if (keep0) h0 = resolve(V, tree_like, t0, errkey);
    if (keep1) h1 = resolve(V, tree_like, t1, errkey);
    bool any = false;
    uint32_t maxpre = 0;
    if (h0.pre != NONE) { any = true; maxpre = h0.pre; }
    if (h1.pre != NONE) { any = true; maxpre = max(maxpre, h1.pre); }
    if (__any(errkey != NOERR)) {
        uint64_t ek = wave_min64(errkey);
        res_status = ((ek >> 32) & 1) ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON;
        res_taxon = (uint32_t)ek;
    } else if (!__any(any)) {
        res_status = UMGAP_READ_EMPTY;   // literal "1" (src/commands/taxa2agg.rs:174-175)
        res_taxon = 1;
    } else {
        const uint32_t M = wave_max(any ? maxpre : 0u);
        uint32_t hp = NONE;
        if (h0.pre != NONE && h0.last < M) hp = h0.pre;
        if (h1.pre != NONE && h1.last < M) hp = min(hp, h1.pre);
        hp = wave_min(hp);
        if (hp != NONE) res_taxon = V.psnap_of_pre[wave_argmin_depth(V, hp + 1, M, lane)];
        else res_taxon = V.snap_of_pre[M];
        res_status = res_taxon == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
    }
}

// The same for a table BOUND to the taxonomy (bound.hpp): the strip holds {pre + 1, size code} instead of taxon ids, so
// no by-id gather is needed; only a survivor whose subtree end falls inside the code's rounding gap fetches it.
__device__ __forceinline__ void lca_star_bound_from_lds(const TaxView& V, const uint32_t* rt, uint32_t n, float lower_bound,
                                                        uint32_t* wkeys, uint32_t* wcnt, uint32_t lane, uint32_t& res_taxon, uint32_t& res_status) {
    uint32_t t0 = lane < n ? rt[lane] : 0u, t1 = 64 + lane < n ? rt[64 + lane] : 0u;
    bool keep0, keep1;
    float c0, c1;
    lds_count(wkeys, wcnt, t0, t1, lower_bound, keep0, keep1, c0, c1);
    if (!__any(keep0 || keep1)) {
        res_status = UMGAP_READ_EMPTY;   // literal "1" (src/commands/taxa2agg.rs:174-175)
        res_taxon = 1;
        return;
    }
    const uint32_t p0 = keep0 ? bound::pre_of(t0) : 0u, p1 = keep1 ? bound::pre_of(t1) : 0u;
    const uint32_t M = wave_max(max(p0, p1));
    auto not_ancestor_of_M = [&](uint32_t v, uint32_t p) -> bool {   // last[x] < pre[M] ?
        uint32_t lo, hi;
        bound::size_bounds(v, lo, hi);
        if (M > p + hi) return true;
        if (M <= p + lo) return false;
        return V.last_of_pre[p] < M;                                  // inside the rounding gap: the exact subtree end
    };
    uint32_t hp = NONE;
    if (keep0 && not_ancestor_of_M(t0, p0)) hp = p0;
    if (keep1 && not_ancestor_of_M(t1, p1)) hp = min(hp, p1);
    hp = wave_min(hp);
    if (hp != NONE) res_taxon = V.psnap_of_pre[wave_argmin_depth(V, hp + 1, M, lane)];
    else res_taxon = V.snap_of_pre[M];
    res_status = res_taxon == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// LCA* for a group of reads, the same way: the closed form (M = the surviving taxon of largest
// preorder index, h' = the smallest-preorder survivor that is not an ancestor of M; answer M if there is none, else LCA(h', M))
// needs {pre,last} of the survivors (unbound table), one range minimum and one snap per read -- three dependent round trips
// per read in lca_star_from_lds, here three per GROUP: survivors of all reads resolved together, then one lane per read
// for the range minimum (lane_argmin_depth) and the snap.
// ------------------------------------------------------------------------------------------------------------------------
struct LcaScratch {
    uint32_t hkeys[256], hcnt[256];
    uint32_t last[GCAP], ehi[GCAP];   // subtree end of a survivor -- or, for a taxon that did not resolve (pre = NONE), its error key (low, high)
    uint32_t roff[GREADS + 1], rM[GREADS], rhp[GREADS], rstat[GREADS], rans[GREADS];
};
// the same as pointers with run-time limits, so that a kernel can lay the arrays over LDS it has to spare (k_classify_prot9
// re-uses the staging area of the reads' bytes, which is dead once the windows are probed)
struct LcaView {
    uint32_t *hkeys, *hcnt;           // 256 each, all zero between reads
    uint32_t *last, *ehi;             // cap each
    uint32_t *roff, *rM, *rhp, *rstat, *rans;   // reads + 1, reads, reads, reads, reads
    uint32_t cap, reads;              // reads <= 64
};
__device__ __forceinline__ LcaView view_of(LcaScratch& S) {
    return LcaView{S.hkeys, S.hcnt, S.last, S.ehi, S.roff, S.rM, S.rhp, S.rstat, S.rans, GCAP, GREADS};
}
// words of LDS a view over `cap` survivors and `reads` reads needs besides the two hash arrays
__host__ __device__ constexpr uint32_t lca_view_words(uint32_t cap, uint32_t reads) { return 2 * cap + 5 * reads + 1; }
__device__ __forceinline__ LcaView lca_view_over(uint32_t* hkeys, uint32_t* hcnt, uint32_t* mem, uint32_t cap, uint32_t reads) {
    return LcaView{hkeys, hcnt, mem, mem + cap, mem + 2 * cap, mem + 2 * cap + reads + 1, mem + 2 * cap + 2 * reads + 1,
                   mem + 2 * cap + 3 * reads + 1, mem + 2 * cap + 4 * reads + 1, cap, reads};
}

// one_read(k, res_taxon, res_status): the per-read path (lca_star[_bound]_from_lds) for a read with more than 64 surviving
// taxa; its part of the strip is still untouched when it is called.
// span(k, start, n): read k of the group owns wtaxa[start .. start + n), n <= 128; the reads lie one after another (start is
// non-decreasing in k and a read does not overlap the next), which is what lets the survivors be compacted in place.
// span returns false for a read the kernel cannot take (surface B: too many windows): it gets `skip_status` and no answer.
template <bool BOUND, class Span, class OneRead>
__device__ __forceinline__ void lca_group_spans(const TaxView& V, const LcaView& G, bool tree_like, float lower_bound, uint32_t* wtaxa, Span span,
                                                uint32_t nr, uint32_t lane, uint64_t se_panic, uint32_t frags, size_t first_read,
                                                uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status, OneRead one_read,
                                                uint32_t skip_status = UMGAP_READ_UNSUPPORTED, uint32_t fixed_stride = 0) {
    // fixed_stride (1..16, else 0): read k owns wtaxa[k * fixed_stride .. + fixed_stride) -- short reads, counted 64 / stride to a pass
    auto se_panicked = [&](uint32_t k) -> bool {
        return se_panic && ((se_panic >> (k * frags)) & ((frags >= 64 ? 0ull : (1ull << frags)) - 1ull));
    };
    uint32_t k0 = 0;
    while (k0 < nr) {
        // ---- count + filter per read (LDS only); survivors compacted in place over the strip from `base` on ----
        uint32_t base = 0, n_first = 0;
        const bool first_ok = span(k0, base, n_first);
        uint32_t w = 0, k = k0;
        bool big = false;                       // the sub-batch ends in front of a read that takes the per-read path
        if (lane == 0) G.roff[0] = 0;
        // Short reads at a fixed stride (the tryptic pipelines, CSR reads of a few hits): 64 / stride reads per pass, one lane per item,
        // as in general_group -- an item's multiplicity is the number of equal items in its read (stride - 1 shuffles), the item leads
        // its taxon when no equal item precedes it.  Same survivors in the same order as the per-read loop below.
        const uint32_t per_pass = fixed_stride >= 1 && fixed_stride <= 16 ? 64u / fixed_stride : 0u;
        const uint32_t r_in = per_pass ? lane / fixed_stride : 0u, j_in = per_pass ? lane - r_in * fixed_stride : 0u;
        while (per_pass && k < nr && k - k0 < G.reads) {
            const uint32_t rp = min(per_pass, min(nr - k, G.reads - (k - k0)));
            const bool active = r_in < rp;
            const uint32_t t = active ? wtaxa[(k + r_in) * fixed_stride + j_in] : 0u;
            uint32_t mult = t != 0u ? 1u : 0u;
            bool lead = t != 0u;
            for (uint32_t d = 1; d < fixed_stride; d++) {
                uint32_t jo = j_in + d;
                if (jo >= fixed_stride) jo -= fixed_stride;
                const uint32_t o = (uint32_t)__shfl((int)t, (int)(lane - j_in + jo), 64);    // another item of the same read
                if (t != 0u && o == t) { mult++; if (jo < j_in) lead = false; }
            }
            const bool keep = lead && (float)mult >= lower_bound;                            // agg::filter keeps freq >= lower_bound
            const uint64_t b = __ballot(keep);
            const uint32_t n_pass = (uint32_t)__builtin_popcountll(b);
            if (w + n_pass > G.cap) break;       // (k > k0 here: a pass holds at most 64 survivors <= cap) these reads open the next sub-batch
            const uint32_t bit0 = r_in * fixed_stride;                                       // first bit of the lane's read
            const uint32_t seg = (uint32_t)(b >> bit0) & ((1u << fixed_stride) - 1u);
            const uint32_t before = (uint32_t)__builtin_popcountll(b & ((1ull << bit0) - 1ull));
            wave_sync();                         // every lane holds its taxon in a register before the strip is overwritten
            if (keep) wtaxa[base + w + before + (uint32_t)__builtin_popcount(seg & ((1u << j_in) - 1u))] = t;
            if (active && j_in == 0) { G.rM[k - k0 + r_in] = 0; G.roff[k - k0 + r_in + 1] = w + before + (uint32_t)__builtin_popcount(seg); }
            w += n_pass;
            wave_sync();
            k += rp;
        }
        while (!per_pass && k < nr && k - k0 < G.reads) {
            uint32_t start = 0, stride = 0;
            const bool ok = k == k0 ? first_ok : span(k, start, stride);
            if (k == k0) { start = base; stride = n_first; }
            if (!ok) {                          // not representable: no survivors, marked below (wave-uniform)
                if (lane == 0) { G.roff[k - k0 + 1] = w; G.rM[k - k0] = NONE; }
                k++;
                continue;
            }
            if (lane == 0) G.rM[k - k0] = 0;
            const uint32_t* rt = wtaxa + start;
            const uint32_t t0 = lane < stride ? rt[lane] : 0u, t1 = 64 + lane < stride ? rt[64 + lane] : 0u;
            // A read out of one or two organisms (round 5): its hits hold at most two distinct taxa -- found with four ballots, counted with
            // popcounts, no hash table, no LDS round trip (the table's three phases are a chain of LDS atomics per read).  Wave-uniform.
            const uint64_t nz0 = __ballot(t0 != 0u), nz1 = __ballot(t1 != 0u);
            uint32_t few = 0, T1 = 0, T2 = 0, cn1 = 0, cn2 = 0;       // few: 1 = at most two distinct taxa, counted here
            if (!(nz0 | nz1)) few = 1;
            else {
                T1 = nz0 ? (uint32_t)__builtin_amdgcn_readlane((int)t0, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(nz0)))
                         : (uint32_t)__builtin_amdgcn_readlane((int)t1, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(nz1)));
                const uint64_t a0 = __ballot(t0 == T1), a1 = __ballot(t1 == T1), r0 = nz0 & ~a0, r1 = nz1 & ~a1;
                cn1 = (uint32_t)__builtin_popcountll(a0) + (uint32_t)__builtin_popcountll(a1);
                if (!(r0 | r1)) few = 1;
                else {
                    T2 = r0 ? (uint32_t)__builtin_amdgcn_readlane((int)t0, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(r0)))
                            : (uint32_t)__builtin_amdgcn_readlane((int)t1, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(r1)));
                    const uint64_t c0m = __ballot(t0 == T2), c1m = __ballot(t1 == T2);
                    cn2 = (uint32_t)__builtin_popcountll(c0m) + (uint32_t)__builtin_popcountll(c1m);
                    if (!((r0 & ~c0m) | (r1 & ~c1m))) few = 1;
                }
            }
            if (few) {
                const bool k1 = cn1 != 0u && (float)cn1 >= lower_bound, k2 = cn2 != 0u && (float)cn2 >= lower_bound;     // agg::filter keeps freq >= lower_bound
                const uint32_t ns = (k1 ? 1u : 0u) + (k2 ? 1u : 0u);
                if (w + ns > G.cap) break;
                wave_sync();                     // every lane holds its taxa in registers before the strip is overwritten
                if (lane == 0) {
                    if (k1) wtaxa[base + w] = T1;
                    if (k2) wtaxa[base + w + (k1 ? 1u : 0u)] = T2;
                    G.roff[k - k0 + 1] = w + ns;
                }
                w += ns;
                wave_sync();
                k++;
                continue;
            }
            bool keep0, keep1;
            float c0, c1;
            lds_count(G.hkeys, G.hcnt, t0, t1, lower_bound, keep0, keep1, c0, c1);
            const uint64_t b0 = __ballot(keep0), b1 = __ballot(keep1);
            const uint32_t n0 = (uint32_t)__builtin_popcountll(b0), ns = n0 + (uint32_t)__builtin_popcountll(b1);
            if (ns > 64) { big = true; break; }
            if (w + ns > G.cap) break;
            wave_sync();                         // every lane holds its taxa in registers before the strip is overwritten
            if (keep0) wtaxa[base + w + (uint32_t)__builtin_popcountll(b0 & lanemask_lt(lane))] = t0;
            if (keep1) wtaxa[base + w + n0 + (uint32_t)__builtin_popcountll(b1 & lanemask_lt(lane))] = t1;
            w += ns;
            if (lane == 0) G.roff[k - k0 + 1] = w;
            wave_sync();
            k++;
        }
        const uint32_t nb = k - k0;
        wave_sync();
        // ---- {pre, last} of every survivor: one round trip (a bound table's codes carry them) ----
        bool any_err = false;
        if (!BOUND) {
            for (uint32_t eb = 0; eb < w; eb += 64) {
                const uint32_t e = eb + lane;
                uint64_t errkey = NOERR;
                if (e < w) {
                    const Hit h = resolve(V, tree_like, wtaxa[base + e], errkey);
                    wtaxa[base + e] = h.pre;                               // NONE: the taxon did not resolve
                    G.last[e] = h.pre != NONE ? h.last : (uint32_t)errkey;
                    G.ehi[e] = (uint32_t)(errkey >> 32);
                }
                any_err = any_err || __any(errkey != NOERR);
            }
            wave_sync();
        }
        // ---- per read: M = the deepest-in-preorder survivor, h' = the first survivor that is not its ancestor (LDS / registers) ----
        // reads of at most 16 survivors each (short reads; long ones out of a few organisms): one lane per read walks its own survivors,
        // all reads of the sub-batch at once, instead of two wave-wide reductions per read one read after another
        const bool lane_per_read = per_pass || !__any(lane < nb && G.roff[lane + 1] - G.roff[lane] > 16u);
        if (lane_per_read && lane < nb) {
            const uint32_t o0 = G.roff[lane], D = G.roff[lane + 1] - o0;
            uint32_t st = ST_PENDING, M = 0, hp = NONE, ans = 0;
            if (G.rM[lane] == NONE) { st = skip_status; ans = 0; }
            else if (D == 0) { st = UMGAP_READ_EMPTY; ans = 1; }                 // literal "1" (src/commands/taxa2agg.rs:174-175)
            else if (BOUND) {
                for (uint32_t e = 0; e < D; e++) { const uint32_t p = bound::pre_of(wtaxa[base + o0 + e]); M = p > M ? p : M; }
                for (uint32_t e = 0; e < D; e++) {
                    const uint32_t v = wtaxa[base + o0 + e], p = bound::pre_of(v);
                    uint32_t lo, hi;
                    bound::size_bounds(v, lo, hi);
                    bool not_anc = false;
                    if (M > p + hi) not_anc = true;
                    else if (M > p + lo) not_anc = V.last_of_pre[p] < M;          // inside the rounding gap: the exact subtree end
                    if (not_anc && p < hp) hp = p;
                }
            } else {
                uint64_t ek = NOERR;
                if (any_err)
                    for (uint32_t e = 0; e < D; e++)
                        if (wtaxa[base + o0 + e] == NONE) { const uint64_t k2 = ((uint64_t)G.ehi[o0 + e] << 32) | G.last[o0 + e]; ek = k2 < ek ? k2 : ek; }
                if (ek != NOERR) {
                    st = ((ek >> 32) & 1) ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON;
                    ans = (uint32_t)ek;
                } else {
                    for (uint32_t e = 0; e < D; e++) { const uint32_t p = wtaxa[base + o0 + e]; M = p > M ? p : M; }
                    for (uint32_t e = 0; e < D; e++) { const uint32_t p = wtaxa[base + o0 + e]; if (G.last[o0 + e] < M && p < hp) hp = p; }
                }
            }
            G.rstat[lane] = st; G.rans[lane] = ans; G.rM[lane] = M; G.rhp[lane] = hp;
        }
        for (uint32_t r = 0; r < nb && !lane_per_read; r++) {
            const uint32_t o0 = G.roff[r], D = G.roff[r + 1] - o0;        // D <= 64: one lane per survivor
            uint32_t st = ST_PENDING, M = 0, hp = NONE, ans = 0;
            if (G.rM[r] == NONE) { st = skip_status; ans = 0; }          // a read span() refused
            else if (D == 0) { st = UMGAP_READ_EMPTY; ans = 1; }         // literal "1" (src/commands/taxa2agg.rs:174-175)
            else if (BOUND && D == 1) {                                  // one surviving taxon (a read out of one organism): it is the answer's M
                M = bound::pre_of(wtaxa[base + o0]);
            } else if (BOUND) {
                const uint32_t v = lane < D ? wtaxa[base + o0 + lane] : 0u;
                const uint32_t p = lane < D ? bound::pre_of(v) : 0u;
                M = wave_max(p);
                bool not_anc = false;
                if (lane < D) {
                    uint32_t lo, hi;
                    bound::size_bounds(v, lo, hi);
                    if (M > p + hi) not_anc = true;
                    else if (M > p + lo) not_anc = V.last_of_pre[p] < M;  // inside the rounding gap: the exact subtree end
                }
                hp = wave_min(not_anc ? p : NONE);
            } else {
                const uint32_t p = lane < D ? wtaxa[base + o0 + lane] : 0u;
                const uint32_t l = lane < D ? G.last[o0 + lane] : 0u;
                uint64_t ek = NOERR;
                if (any_err) ek = wave_min64((lane < D && p == NONE) ? (((uint64_t)G.ehi[o0 + lane] << 32) | l) : NOERR);
                if (ek != NOERR) {
                    st = ((ek >> 32) & 1) ? UMGAP_READ_PANIC : UMGAP_READ_UNKNOWN_TAXON;
                    ans = (uint32_t)ek;
                } else {
                    M = wave_max(lane < D ? p : 0u);
                    hp = wave_min((lane < D && l < M) ? p : NONE);
                }
            }
            if (lane == 0) { G.rstat[r] = st; G.rans[r] = ans; G.rM[r] = M; G.rhp[r] = hp; }
        }
        wave_sync();
        // ---- one lane per read: the range minimum (one round trip), the snap (one round trip), the results stored together ----
        if (lane < nb) {
            const size_t read = first_read + k0 + lane;
            uint32_t st = G.rstat[lane], tx = G.rans[lane];
            if (st == ST_PENDING) {
                const uint32_t M = G.rM[lane], hp = G.rhp[lane];
                if (hp != NONE) tx = V.psnap_of_pre[(uint32_t)lane_argmin_depth(V, hp + 1, M)];   // LCA(h', M) = parent of a shallowest node in (pre[h'], pre[M]]
                else tx = V.snap_of_pre[M];
                st = tx == NONE ? UMGAP_READ_PANIC : UMGAP_READ_OK;
            }
            if (se_panicked(k0 + lane)) { st = UMGAP_READ_PANIC; tx = 0; }
            out_taxon[read] = tx;
            out_status[read] = st;
        }
        wave_sync();
        k0 = k;
        if (big) {
            uint32_t res_taxon, res_status;
            one_read(k0, res_taxon, res_status);
            if (se_panicked(k0)) { res_status = UMGAP_READ_PANIC; res_taxon = 0; }
            if (lane == 0) { out_taxon[first_read + k0] = res_taxon; out_status[first_read + k0] = res_status; }
            wave_sync();
            k0++;
        }
    }
}

// the fixed-stride form (packed k-mers: read k owns wtaxa[k * stride .. + stride))
template <bool BOUND, class OneRead>
__device__ __forceinline__ void lca_group(const TaxView& V, LcaScratch& G, bool tree_like, float lower_bound, uint32_t* wtaxa, uint32_t stride,
                                          uint32_t nr, uint32_t lane, uint64_t se_panic, uint32_t frags, size_t first_read,
                                          uint32_t* __restrict__ out_taxon, uint32_t* __restrict__ out_status, OneRead one_read) {
    lca_group_spans<BOUND>(V, view_of(G), tree_like, lower_bound, wtaxa, [stride](uint32_t k, uint32_t& start, uint32_t& n) { start = k * stride; n = stride; return true; },
                           nr, lane, se_panic, frags, first_read, out_taxon, out_status, one_read, UMGAP_READ_UNSUPPORTED, stride <= 16u ? stride : 0u);
}

}  // namespace agg
}  // namespace umgap
