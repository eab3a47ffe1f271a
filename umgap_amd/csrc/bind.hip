// bind.hip — binding a 9-mer table to a taxonomy: the slot values (and the stash's) are rewritten in place from NCBI
// taxon ids to the {preorder index, subtree size code} of bound.hpp, and back.  See bound.hpp for why.
#include "bound.hpp"
#include "common.hpp"
#include "taxonomy.hpp"
#include "var_index.hpp"

using namespace umgap;

namespace {

constexpr uint64_t EMPTY64 = ~0ull;

// dry run: values that have no interval (id beyond the list, absent or unreachable taxon, or 0)
__global__ __launch_bounds__(256) void k_bind_check(const uint64_t* __restrict__ slots, size_t nslots, const uint32_t* __restrict__ stash_vals,
                                                    uint32_t n_stash, const uint2* __restrict__ by_id, uint32_t len, unsigned long long* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (size_t k = i; k < nslots + n_stash; k += stride) {
        uint32_t t;
        if (k < nslots) { uint64_t s = slots[k]; if (s == EMPTY64) continue; t = (uint32_t)(s >> 32); }
        else t = stash_vals[k - nslots];
        if (t >= len || by_id[t].x == UMGAP_NONE) mine++;
    }
    if (mine) atomicAdd(bad, mine);
}

__global__ __launch_bounds__(256) void k_bind(uint64_t* __restrict__ slots, size_t nslots, uint32_t* __restrict__ stash_vals, uint32_t n_stash,
                                              const uint2* __restrict__ by_id) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < nslots + n_stash; k += stride) {
        if (k < nslots) {
            uint64_t s = slots[k];
            if (s == EMPTY64) continue;
            uint2 pl = by_id[(uint32_t)(s >> 32)];
            slots[k] = ((uint64_t)bound::encode(pl.x, pl.y - pl.x) << 32) | (s & 0xFFFFFFFFull);
        } else {
            uint2 pl = by_id[stash_vals[k - nslots]];
            stash_vals[k - nslots] = bound::encode(pl.x, pl.y - pl.x);
        }
    }
}

__global__ __launch_bounds__(256) void k_unbind(uint64_t* __restrict__ slots, size_t nslots, uint32_t* __restrict__ stash_vals, uint32_t n_stash,
                                                const uint32_t* __restrict__ id_of_pre) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < nslots + n_stash; k += stride) {
        if (k < nslots) {
            uint64_t s = slots[k];
            if (s == EMPTY64) continue;
            slots[k] = ((uint64_t)id_of_pre[bound::pre_of((uint32_t)(s >> 32))] << 32) | (s & 0xFFFFFFFFull);
        } else {
            stash_vals[k - nslots] = id_of_pre[bound::pre_of(stash_vals[k - nslots])];
        }
    }
}

// a walk-order index (kind 10, index_walk.hip) keeps its taxa in the walk array: entry = taxon | (low key bits | continues << 31) << 32;
// entries of windows that hold no 9-mer (low key bits all ones) carry no taxon
__device__ __forceinline__ bool walk_entry_has_taxon(uint64_t e) { return ((uint32_t)(e >> 32) & 0x7FFFFFFFu) != 0x7FFFFFFFu; }
__global__ __launch_bounds__(256) void k_bind_check_walk(const uint64_t* __restrict__ E, size_t n, const uint2* __restrict__ by_id, uint32_t len, unsigned long long* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (size_t k = i; k < n; k += stride) {
        const uint64_t e = E[k];
        if (!walk_entry_has_taxon(e)) continue;
        const uint32_t t = (uint32_t)e;
        if (t >= len || by_id[t].x == UMGAP_NONE) mine++;
    }
    if (mine) atomicAdd(bad, mine);
}
__global__ __launch_bounds__(256) void k_bind_walk(uint64_t* __restrict__ E, size_t n, const uint2* __restrict__ by_id) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < n; k += stride) {
        const uint64_t e = E[k];
        if (!walk_entry_has_taxon(e)) continue;
        const uint2 pl = by_id[(uint32_t)e];
        E[k] = (e & 0xFFFFFFFF00000000ull) | bound::encode(pl.x, pl.y - pl.x);
    }
}
__global__ __launch_bounds__(256) void k_unbind_walk(uint64_t* __restrict__ E, size_t n, const uint32_t* __restrict__ id_of_pre) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < n; k += stride) {
        const uint64_t e = E[k];
        if (!walk_entry_has_taxon(e)) continue;
        E[k] = (e & 0xFFFFFFFF00000000ull) | id_of_pre[bound::pre_of((uint32_t)e)];
    }
}

// the packed tryptic table (index_varp.hip): the taxon is word x of every HEADER slot (w carries VARP_MARK) and of the stash entries
__global__ __launch_bounds__(256) void k_bind_check_varp(const u32x4* __restrict__ slots, size_t nslots, const VarpStashEntry* __restrict__ stash, uint32_t n_stash,
                                                         const uint2* __restrict__ by_id, uint32_t len, unsigned long long* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (size_t k = i; k < nslots + n_stash; k += stride) {
        uint32_t t;
        if (k < nslots) { const u32x4 e = slots[k]; if (!(e.w & VARP_MARK)) continue; t = e.x; }
        else t = stash[k - nslots].taxon;
        if (t >= len || by_id[t].x == UMGAP_NONE) mine++;
    }
    if (mine) atomicAdd(bad, mine);
}
template <bool BIND>
__global__ __launch_bounds__(256) void k_rebind_varp(u32x4* __restrict__ slots, size_t nslots, VarpStashEntry* __restrict__ stash, uint32_t n_stash,
                                                     const uint2* __restrict__ by_id, const uint32_t* __restrict__ id_of_pre) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto turn = [&](uint32_t v) -> uint32_t {
        if (BIND) { const uint2 pl = by_id[v]; return bound::encode(pl.x, pl.y - pl.x); }
        return id_of_pre[bound::pre_of(v)];
    };
    for (size_t k = i; k < nslots + n_stash; k += stride) {
        if (k < nslots) {
            uint32_t* e = reinterpret_cast<uint32_t*>(slots + k);
            if (e[3] & VARP_MARK) e[0] = turn(e[0]);
        } else stash[k - nslots].taxon = turn(stash[k - nslots].taxon);
    }
}

}  // namespace

extern "C" {

int umgap_index_is_bound(const umgap_index* ix) { return ix && ix->bound_to ? 1 : 0; }

int umgap_index_bind_taxonomy(umgap_index* ix, umgap_taxonomy* tx) {
    if (!ix || !tx) return fail(UMGAP_ERR_INVALID, "umgap_index_bind_taxonomy: null argument");
    const bool varp = ix->kind == 2 && ix->var && ix->var->packed();
    if (ix->kind != 9 && ix->kind != 10 && !varp)
        return fail(UMGAP_ERR_INVALID, "umgap_index_bind_taxonomy: only the 9-mer tables and the packed peptide table (every key a word over 'A'..'Z') can be bound");
    const bool walk = ix->kind == 10;
    if (ix->bound_to) return fail(UMGAP_ERR_INVALID, "umgap_index_bind_taxonomy: the index is already bound");
    if (ix->device != tx->device) return fail(UMGAP_ERR_INVALID, "umgap_index_bind_taxonomy: index and taxonomy live on different devices");
    if (tx->n_reach > bound::MAX_REACH) return fail(UMGAP_ERR_INVALID, "umgap_index_bind_taxonomy: more than 2^22 - 2 reachable taxa");
    int rc = use_device(ix->device);
    if (rc) return rc;
    // the slots are rewritten in place: nothing launched earlier on ANY stream of this device (a classify on the
    // caller's stream, say) may still be probing them
    UMGAP_HIP(hipDeviceSynchronize());
    const size_t nslots = varp ? (size_t)(ix->var->nb + VAR_MAX_DISP) * VAR_SLOTS : (size_t)ix->view.n_resident * SLOTS;
    unsigned long long* d_bad = nullptr;
    UMGAP_HIP(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), ix->stream);
    unsigned long long bad = 0;
    if (e == hipSuccess) {
        if (varp) hipLaunchKernelGGL(k_bind_check_varp, dim3(256 * 8), dim3(256), 0, ix->stream, ix->var->d_slots, nslots, ix->var->d_pstash, ix->var->n_stash, tx->d_by_id, tx->len, d_bad);
        else if (walk) hipLaunchKernelGGL(k_bind_check_walk, dim3(256 * 8), dim3(256), 0, ix->stream, ix->d_walk, (size_t)ix->n_walk, tx->d_by_id, tx->len, d_bad);
        else hipLaunchKernelGGL(k_bind_check, dim3(256 * 8), dim3(256), 0, ix->stream, ix->d_slots, nslots, ix->d_stash_vals, ix->n_stash, tx->d_by_id, tx->len, d_bad);
        e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, ix->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("umgap_index_bind_taxonomy: ") + hipGetErrorString(e));
    if (bad) return fail(UMGAP_ERR_BUILD, "umgap_index_bind_taxonomy: " + std::to_string(bad) + " table values are not taxa of this taxonomy (the unbound kernels report them per read)");
    if (varp) hipLaunchKernelGGL(k_rebind_varp<true>, dim3(256 * 8), dim3(256), 0, ix->stream, ix->var->d_slots, nslots, ix->var->d_pstash, ix->var->n_stash, tx->d_by_id, tx->d_id_of_pre);
    else if (walk) hipLaunchKernelGGL(k_bind_walk, dim3(256 * 8), dim3(256), 0, ix->stream, ix->d_walk, (size_t)ix->n_walk, tx->d_by_id);   // the table holds window numbers: untouched
    else hipLaunchKernelGGL(k_bind, dim3(256 * 8), dim3(256), 0, ix->stream, ix->d_slots, nslots, ix->d_stash_vals, ix->n_stash, tx->d_by_id);
    UMGAP_HIP(hipGetLastError());
    UMGAP_HIP(hipStreamSynchronize(ix->stream));
    ix->bound_to = tx;
    tx->bound_indexes.push_back(ix);
    return UMGAP_OK;
}

int umgap_index_unbind_taxonomy(umgap_index* ix, umgap_taxonomy* tx) {
    if (!ix || !tx) return fail(UMGAP_ERR_INVALID, "umgap_index_unbind_taxonomy: null argument");
    if (!ix->bound_to) return UMGAP_OK;
    if (ix->bound_to != tx) return fail(UMGAP_ERR_INVALID, "umgap_index_unbind_taxonomy: bound to another taxonomy");
    int rc = use_device(ix->device);
    if (rc) return rc;
    UMGAP_HIP(hipDeviceSynchronize());   // as in bind: no kernel of any stream may be reading the slots
    const bool varp = ix->kind == 2 && ix->var && ix->var->packed();
    const size_t nslots = varp ? (size_t)(ix->var->nb + VAR_MAX_DISP) * VAR_SLOTS : (size_t)ix->view.n_resident * SLOTS;
    if (varp) hipLaunchKernelGGL(k_rebind_varp<false>, dim3(256 * 8), dim3(256), 0, ix->stream, ix->var->d_slots, nslots, ix->var->d_pstash, ix->var->n_stash, tx->d_by_id, tx->d_id_of_pre);
    else if (ix->kind == 10) hipLaunchKernelGGL(k_unbind_walk, dim3(256 * 8), dim3(256), 0, ix->stream, ix->d_walk, (size_t)ix->n_walk, tx->d_id_of_pre);
    else hipLaunchKernelGGL(k_unbind, dim3(256 * 8), dim3(256), 0, ix->stream, ix->d_slots, nslots, ix->d_stash_vals, ix->n_stash, tx->d_id_of_pre);
    UMGAP_HIP(hipGetLastError());
    UMGAP_HIP(hipStreamSynchronize(ix->stream));
    ix->bound_to = nullptr;
    for (size_t i = 0; i < tx->bound_indexes.size(); i++)
        if (tx->bound_indexes[i] == ix) { tx->bound_indexes.erase(tx->bound_indexes.begin() + (long)i); break; }
    return UMGAP_OK;
}

}  // extern "C"
