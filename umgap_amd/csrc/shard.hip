// shard.hip — the 9-mer index sharded by bucket range over the GPUs of one node, behind the C ABI (round 2).
//
// What it replaces: scripts/umgap-analyse.sh:257-264 keeps ONE `umgap prot2kmer2lca -m -o -s <socket>` process holding the
// whole ~100 GB index and every sample streams its peptides through that socket.  Here rank r (one process per GPU, or one
// thread per GPU inside `umgap`) holds the home buckets [nb*r/N, nb*(r+1)/N) (+ halo) in its HBM, classifies its OWN reads and
// asks the owners for the k-mers it does not hold:
//
//   pack      keys -> N regions by owner shard, 6 bytes per key (shard_wire.hpp), + bases[run][owner]          k_shard_pack
//   exchange  all-to-all of the regions over RCCL send/recv groups (xGMI)                                      comm_alltoall
//   probe     the received keys against the local shard (the quad probe of probe_common.hpp), 3 or 4 B/value  k_shard_probe
//   exchange  all-to-all of the values back
//   classify  the hits of a run of reads from their `world` contiguous stretches into LDS -> [seedextend] -> aggregate   classify.hip (GATHER)
//
// A batch is cut into pieces of whole reads that flow through three stage streams (pack + key exchange | probe | value
// exchange + aggregate) over four buffer sets: while piece t is packed and its keys travel, piece t-1 is probed and piece
// t-2 is aggregated.  The exchanges are
// issued in the order A(0) A(1) R(0) A(2) R(1) ... on every rank (RCCL runs one communicator's operations in issue
// order), so a return exchange never waits behind the key exchange of the piece after it.  Regions have a fixed capacity
// (the mixer spreads keys evenly: expected K/N per region, `slack` on top), which keeps every message the same size on
// every rank with no host round trip; a key that does not fit raises a sticky flag (umgap_sharded_check) and is answered
// with a miss.  Reads never move; the taxonomy is replicated.
//
// Transports: RCCL (librccl.so.1 resolved at run time with dlopen, so the library loads without it; inside a PyTorch
// process that is PyTorch's own copy) and a LOCAL hub -- all ranks are threads of one process on one GPU, the all-to-all is
// device-to-device copies ordered with events -- which lets the whole N-rank protocol be checked on a single-GPU box.
// And an IPC hub: the ranks are separate PROCESSES (one per GPU, or several sharing one GPU) that map each other's send
// buffers with hipIpc memory handles and copy their share out of them; a block of POSIX shared memory carries the handles and
// the barrier.  No RCCL involved: it is how the process-per-rank protocol is run where RCCL refuses (two ranks on one GPU).
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "probe_common.hpp"
#include "shard_wire.hpp"
#include "taxonomy.hpp"

namespace umgap {
int launch_classify9_from_hits(umgap_index* ix, umgap_taxonomy* tx, int method, int strategy, float factor, float lower_bound, int ranked_only, const HitSource& H,
                               size_t n_reads, uint32_t kmers_per_read, uint32_t fragments_per_read, uint32_t min_seed_size, uint32_t max_gap_size,
                               uint32_t* d_out_taxon, uint32_t* d_out_status, hipStream_t s);
}

using namespace umgap;
using namespace umgap::probe;

// ------------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.lib) break;
        }
        if (!R.lib) { R.error = std::string("RCCL is not loadable: ") + dlerror(); return; }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(R.lib, n);
            if (!p && R.error.empty()) R.error = std::string("RCCL lacks ") + n;
            return p;
        };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.Send = (decltype(R.Send))sym("ncclSend");
        R.Recv = (decltype(R.Recv))sym("ncclRecv");
        R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
        R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    });
    return &R;
}

// the pointers above carry the prototypes of <rccl/rccl.h>
static_assert(std::is_same<decltype(Rccl::Send), decltype(&ncclSend)>::value && std::is_same<decltype(Rccl::Recv), decltype(&ncclRecv)>::value &&
                  std::is_same<decltype(Rccl::AllReduce), decltype(&ncclAllReduce)>::value && std::is_same<decltype(Rccl::AllGather), decltype(&ncclAllGather)>::value &&
                  std::is_same<decltype(Rccl::CommInitRank), decltype(&ncclCommInitRank)>::value &&
                  std::is_same<decltype(Rccl::GetUniqueId), decltype(&ncclGetUniqueId)>::value,
              "RCCL binding out of step with rccl.h");

#define UMGAP_NCCL(expr)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r__ = (expr);                                                                                \
        if (r__ != ncclSuccess) return ::umgap::fail(UMGAP_ERR_DEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(r__)); \
    } while (0)

// all ranks of a LOCAL communicator: threads of this process sharing one GPU
struct LocalHub {
    int world = 0, device = 0, refs = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> send;
    std::vector<hipEvent_t> ready, done;
    std::vector<uint32_t> vals;
    std::vector<std::array<uint8_t, 192>> blobs;
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

// the control block of an IPC communicator, in POSIX shared memory (rank 0 creates and initialises it)
struct IpcCtl {
    std::atomic<uint32_t> ready;                  // 1: the block is initialised
    std::atomic<uint32_t> creator_pid;            // rank 0's process: a block whose creator is gone (killed before everybody was attached) is stale
    std::atomic<uint32_t> arrived, generation;    // the barrier
    std::atomic<uint32_t> failed;                 // a rank gave up: every barrier from here on fails
    struct Slot {
        hipIpcMemHandle_t mem;      // the allocation the rank's send buffer lies in
        uint64_t offset, alloc_bytes, alloc_base;
        uint32_t val;
        uint32_t pad;
        uint8_t blob[192];          // comm_allgather: a rank's contribution (an exported allocation, a layout stamp)
    } slot[64];
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "the barrier words are shared between processes");

// one rank's end of it
struct IpcHub {
    IpcCtl* ctl = nullptr;
    int world = 0, rank = 0;
    double timeout_s = 120.0;
    std::vector<std::map<std::string, void*>> opened;          // per peer: handle (+ base, size) -> where it is mapped here
    int barrier() {
        const uint32_t g = ctl->generation.load(std::memory_order_acquire);
        if (ctl->failed.load(std::memory_order_acquire)) return fail(UMGAP_ERR_DEVICE, "ipc communicator: another rank has failed");
        if (ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            ctl->arrived.store(0, std::memory_order_relaxed);
            ctl->generation.fetch_add(1, std::memory_order_release);
            return UMGAP_OK;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; ctl->generation.load(std::memory_order_acquire) == g; spins++) {
            if (ctl->failed.load(std::memory_order_acquire)) return fail(UMGAP_ERR_DEVICE, "ipc communicator: another rank has failed");
            if (spins < 2000) { sched_yield(); continue; }
            usleep(50);
            if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                ctl->failed.store(1, std::memory_order_release);
                return fail(UMGAP_ERR_DEVICE, "ipc communicator: rank " + std::to_string(rank) + " waited " + std::to_string((int)timeout_s) + " s at a barrier (UMGAP_IPC_TIMEOUT_S)");
            }
        }
        return UMGAP_OK;
    }
    // where peer p's published buffer is mapped in this process
    int peer_buffer(int p, const char** out) {
        const IpcCtl::Slot& S = ctl->slot[p];
        std::string key((const char*)&S.mem, sizeof(S.mem));
        key.append((const char*)&S.alloc_base, 8).append((const char*)&S.alloc_bytes, 8);
        auto it = opened[p].find(key);
        if (it == opened[p].end()) {
            // the peer publishes another allocation: what was mapped of its earlier ones (freed by now, as a rule: the self-test's buffers)
            // goes once a few have piled up, so that a long-lived communicator does not keep every buffer it ever saw pinned
            if (opened[p].size() >= 4) {
                for (auto& kv : opened[p]) (void)hipIpcCloseMemHandle(kv.second);
                opened[p].clear();
            }
            void* ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, S.mem, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(p) + "'s buffer): " + hipGetErrorString(e));
            it = opened[p].emplace(key, ptr).first;
        }
        *out = (const char*)it->second + S.offset;
        return UMGAP_OK;
    }
    int publish(const void* send) {
        void* base = nullptr;
        size_t size = 0;
        UMGAP_HIP(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)send));
        // (asked for every time: an allocation that was freed and made again at the same address has another handle, and the peers tell
        // mapped allocations apart by the handle)
        hipIpcMemHandle_t h;
        UMGAP_HIP(hipIpcGetMemHandle(&h, base));
        IpcCtl::Slot& S = ctl->slot[rank];
        S.mem = h;
        S.offset = (uint64_t)((const char*)send - (const char*)base);
        S.alloc_bytes = size;
        S.alloc_base = (uint64_t)(uintptr_t)base;
        return UMGAP_OK;
    }
};

}  // namespace

struct umgap_comm {
    int world = 1, rank = 0, device = 0;
    ncclComm_t nccl = nullptr;   // RCCL transport
    LocalHub* hub = nullptr;     // local transport
    IpcHub* ipc = nullptr;       // process-per-rank transport over hipIpc handles
};

namespace {

// every rank sends `bytes` to every OTHER rank: recv region p <- rank p's send region `rank`.  A rank's own share never
// travels (nor is it copied): the kernels read / write region `rank` in place (k_shard_probe).
// `with_self` (umgap_comm_selftest only) moves region `rank` too, through the same transport.
int comm_alltoall(umgap_comm* c, const void* send, void* recv, size_t bytes, hipStream_t s, bool with_self = false) {
    if (c->world == 1 && !with_self) return UMGAP_OK;
    if (c->hub) {
        LocalHub& H = *c->hub;
        H.send[c->rank] = send;
        UMGAP_HIP(hipEventRecord(H.ready[c->rank], s));
        H.barrier();
        for (int p = 0; p < c->world; p++) {
            if (p == c->rank && !with_self) continue;
            UMGAP_HIP(hipStreamWaitEvent(s, H.ready[p], 0));
            UMGAP_HIP(hipMemcpyAsync((char*)recv + (size_t)p * bytes, (const char*)H.send[p] + (size_t)c->rank * bytes, bytes, hipMemcpyDeviceToDevice, s));
        }
        UMGAP_HIP(hipEventRecord(H.done[c->rank], s));
        H.barrier();
        for (int p = 0; p < c->world; p++) UMGAP_HIP(hipStreamWaitEvent(s, H.done[p], 0));   // a send buffer is rewritten only after every rank has read it
        H.barrier();   // nobody re-records an event before every wait on it is enqueued
        return UMGAP_OK;
    }
    if (c->ipc) {
        // host-synchronised: the send buffer is complete before anybody reads it, and everybody has read it before it is rewritten
        IpcHub& H = *c->ipc;
        auto gave_up = [&](int code) { H.ctl->failed.store(1, std::memory_order_release); return code; };   // the other ranks leave their barriers
        auto hip_ok = [&](hipError_t e, const char* what) { return e == hipSuccess ? UMGAP_OK : gave_up(fail(UMGAP_ERR_DEVICE, std::string("ipc exchange, ") + what + ": " + hipGetErrorString(e))); };
        int rc = H.publish(send);
        if (rc) return gave_up(rc);
        if ((rc = hip_ok(hipStreamSynchronize(s), "the send side"))) return rc;
        if ((rc = H.barrier())) return rc;
        for (int p = 0; p < c->world; p++) {
            if (p == c->rank && !with_self) continue;
            const char* from = (const char*)send;
            if (p != c->rank && (rc = H.peer_buffer(p, &from))) return gave_up(rc);
            if ((rc = hip_ok(hipMemcpyAsync((char*)recv + (size_t)p * bytes, from + (size_t)c->rank * bytes, bytes, hipMemcpyDeviceToDevice, s), "copy from a peer"))) return rc;
        }
        if ((rc = hip_ok(hipStreamSynchronize(s), "the copies"))) return rc;
        return H.barrier();
    }
    Rccl* R = rccl();
    UMGAP_NCCL(R->GroupStart());
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank && !with_self) continue;
        UMGAP_NCCL(R->Send((const char*)send + (size_t)p * bytes, bytes, ncclUint8, p, c->nccl, s));
        UMGAP_NCCL(R->Recv((char*)recv + (size_t)p * bytes, bytes, ncclUint8, p, c->nccl, s));
    }
    UMGAP_NCCL(R->GroupEnd());
    return UMGAP_OK;
}

// max over all ranks of a host value (set-up time only; synchronises)
int comm_max_u32(umgap_comm* c, uint32_t* v) {
    if (c->hub) {
        LocalHub& H = *c->hub;
        H.vals[c->rank] = *v;
        H.barrier();
        uint32_t m = 0;
        for (uint32_t x : H.vals) m = x > m ? x : m;
        H.barrier();
        *v = m;
        return UMGAP_OK;
    }
    if (c->ipc) {
        IpcHub& H = *c->ipc;
        H.ctl->slot[c->rank].val = *v;
        int rc = H.barrier();
        if (rc) return rc;
        uint32_t m = 0;
        for (int p = 0; p < c->world; p++) m = std::max(m, H.ctl->slot[p].val);
        if ((rc = H.barrier())) return rc;
        *v = m;
        return UMGAP_OK;
    }
    uint32_t* d = nullptr;
    UMGAP_HIP(hipMalloc((void**)&d, 4));
    hipError_t e = hipMemcpy(d, v, 4, hipMemcpyHostToDevice);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = rccl()->AllReduce(d, d, 1, ncclUint32, ncclMax, c->nccl, nullptr);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(v, d, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (r != ncclSuccess) return fail(UMGAP_ERR_DEVICE, std::string("ncclAllReduce: ") + rccl()->GetErrorString(r));
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("comm_max_u32: ") + hipGetErrorString(e));
    return UMGAP_OK;
}

constexpr size_t BLOB_BYTES = 192;
// every rank's `bytes` (<= BLOB_BYTES) to every rank: all[p * bytes ..] = rank p's.  Set-up time only; synchronises.
int comm_allgather(umgap_comm* c, const void* mine, void* all, size_t bytes) {
    if (bytes > BLOB_BYTES) return fail(UMGAP_ERR_INVALID, "comm_allgather: contribution too large");
    if (c->hub) {
        LocalHub& H = *c->hub;
        memcpy(H.blobs[c->rank].data(), mine, bytes);
        H.barrier();
        for (int p = 0; p < c->world; p++) memcpy((char*)all + (size_t)p * bytes, H.blobs[p].data(), bytes);
        H.barrier();
        return UMGAP_OK;
    }
    if (c->ipc) {
        IpcHub& H = *c->ipc;
        memcpy(H.ctl->slot[c->rank].blob, mine, bytes);
        int rc = H.barrier();
        if (rc) return rc;
        for (int p = 0; p < c->world; p++) memcpy((char*)all + (size_t)p * bytes, H.ctl->slot[p].blob, bytes);
        return H.barrier();
    }
    uint8_t* d = nullptr;
    const size_t W = (size_t)c->world;
    UMGAP_HIP(hipMalloc((void**)&d, (W + 1) * bytes));
    hipError_t e = hipMemcpy(d + W * bytes, mine, bytes, hipMemcpyHostToDevice);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = rccl()->AllGather(d + W * bytes, d, bytes, ncclUint8, c->nccl, nullptr);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(all, d, W * bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (r != ncclSuccess) return fail(UMGAP_ERR_DEVICE, std::string("ncclAllGather: ") + rccl()->GetErrorString(r));
    if (e != hipSuccess) return fail(UMGAP_ERR_DEVICE, std::string("comm_allgather: ") + hipGetErrorString(e));
    return UMGAP_OK;
}
// (set-up / tear-down only)
int comm_barrier(umgap_comm* c) {
    if (c->world == 1) return UMGAP_OK;
    if (c->hub) { c->hub->barrier(); return UMGAP_OK; }
    if (c->ipc) return c->ipc->barrier();
    uint32_t v = 0;
    return comm_max_u32(c, &v);
}

// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------
// keys -> per-owner regions of `send` (6 bytes per key: h = mix45(key) split 32 + 16) and bases[run][owner] = first slot of the
// run's keys for that owner (shard_wire.hpp: the slot of a key is base + its rank among the run's keys of the same owner, which
// the consumer recomputes; nothing per key is stored).  A wave takes one run (at most 1024 keys = 16 rounds of 64) and counts per
// owner with ballots; the four waves of a workgroup add their counts up in LDS and ONE global atomic per owner reserves the
// workgroup's range.  Keys with bits above 45 (not a packed 9-mer) are not sent, nor are the rank's own (local_owner).
// put mode, the end of a kernel whose stores went into a peer's memory: 1 = nothing (the kernel's own end-of-kernel release and the signal
// kernel behind it order the data before the counter), 2 = a release fence at system scope per wave (L2 write-back), 3 = release + acquire
__device__ __forceinline__ void put_release(uint32_t put) {
    if (put == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    else if (put == 3) __threadfence_system();
}
constexpr int PACK_ROUNDS = 16;
__global__ __launch_bounds__(256) void k_shard_pack(const uint64_t* __restrict__ keys, size_t n, uint32_t run_len, uint32_t m, uint32_t nb, uint32_t world,
                                                    double world_over_nb, uint64_t cap, const uint64_t* __restrict__ key_dst, uint32_t* __restrict__ bases,
                                                    unsigned long long* __restrict__ cursor, uint32_t* __restrict__ overflow, uint32_t local_owner, uint32_t put) {
    __shared__ uint32_t wcnt[4][64];
    __shared__ uint32_t wbase[4][64];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // key_dst[o]: where the region of owner o starts -- in this rank's send buffer (the exchange then moves it), or, with peer-written
    // keys (put mode), in the OWNER's receive buffer, mapped here: the store below is the exchange.  Lane o holds the pointer of owner o.
    const uint64_t my_dst = lane < world ? key_dst[lane] : 0ull;
    const size_t n_runs = (n + run_len - 1) / run_len;
    for (size_t run0 = (size_t)blockIdx.x * 4; run0 < n_runs; run0 += (size_t)gridDim.x * 4) {   // the same trip count for the four waves
        const size_t run = run0 + wv, k0 = run * run_len;
        const uint32_t len = run < n_runs ? (uint32_t)min((size_t)run_len, n - k0) : 0u;
        uint64_t h[PACK_ROUNDS];
        uint32_t meta[PACK_ROUNDS];   // owner | rank << 8, all-ones = not sent
        uint32_t seen = 0;
        // the run's keys, all loads in flight together (inside the loop below each load was followed by its own wait: 16 dependent
        // round trips per run)
#pragma unroll
        for (int r = 0; r < PACK_ROUNDS; r++) {
            const uint32_t i = 64u * r + lane;
            h[r] = i < len ? __builtin_nontemporal_load(keys + k0 + i) : 0ull;
        }
#pragma unroll
        for (int r = 0; r < PACK_ROUNDS; r++) {
            meta[r] = 0xFFFFFFFFu;
            if (64u * r >= len) continue;   // wave-uniform
            const uint32_t i = 64u * r + lane;
            const uint64_t key = h[r];
            const wire::Routed R = wire::route(key, i < len, m, nb, world, world_over_nb);
            const uint32_t owner = R.owner == local_owner ? wire::NO_OWNER : R.owner;   // this rank's own key: probed where it is consumed
            const uint32_t rank = wire::run_rank(owner, world, lane, seen);
            h[r] = R.h;
            if (owner != wire::NO_OWNER) meta[r] = owner | (rank << 8);
        }
        wcnt[wv][lane] = seen;
        __syncthreads();
        if (wv == 0 && lane < world) {
            const uint32_t c0 = wcnt[0][lane], c1 = wcnt[1][lane], c2 = wcnt[2][lane], c3 = wcnt[3][lane];
            const uint32_t total = c0 + c1 + c2 + c3;
            unsigned long long b = total ? atomicAdd(&cursor[lane], (unsigned long long)total) : 0ull;
            auto sat = [](unsigned long long v) { return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; };
            wbase[0][lane] = sat(b);
            wbase[1][lane] = sat(b + c0);
            wbase[2][lane] = sat(b + c0 + c1);
            wbase[3][lane] = sat(b + c0 + c1 + c2);
        }
        __syncthreads();
        const uint32_t base = lane < world ? wbase[wv][lane] : 0u;
        if (run < n_runs && lane < world) bases[run * world + lane] = base;
#pragma unroll
        for (int r = 0; r < PACK_ROUNDS; r++) {
            if (64u * r >= len) continue;   // wave-uniform
            const uint32_t owner = meta[r] & 0xFFu;
            const uint32_t b0 = (uint32_t)__shfl((int)base, (int)(owner & 63u), 64);
            const uint32_t dlo = (uint32_t)__shfl((int)(uint32_t)my_dst, (int)(owner & 63u), 64), dhi = (uint32_t)__shfl((int)(uint32_t)(my_dst >> 32), (int)(owner & 63u), 64);
            if (meta[r] == 0xFFFFFFFFu) continue;
            const uint64_t pos = (uint64_t)b0 + (meta[r] >> 8);
            if (pos < cap) {
                uint8_t* region = reinterpret_cast<uint8_t*>(((uint64_t)dhi << 32) | dlo);
                __builtin_nontemporal_store((uint32_t)h[r], reinterpret_cast<uint32_t*>(region) + pos);           // (nt: index.hip wave_probe64)
                __builtin_nontemporal_store((uint16_t)(h[r] >> 32), reinterpret_cast<uint16_t*>(region + cap * 4) + pos);
            } else {
                atomicExch(overflow, 1u);   // sticky: umgap_sharded_check reports it; the consumer answers the key with a miss
            }
        }
        __syncthreads();   // wcnt / wbase are rewritten by the next trip
    }
    // peer-written keys: what this wave stored is at the peers before the kernel is through (the signal that follows the kernel on its
    // stream then tells them)
    put_release(put);
}

// A region is followed by a trailer: how many of its slots hold keys (the probe answers nothing beyond; the first version padded the
// unused tail of every region with "no key" marks, a kernel of its own per piece).  One lane per owner.
constexpr uint64_t REGION_TRAILER = 64;
__host__ __device__ inline uint64_t region_bytes(uint64_t cap) { return cap * 6 + REGION_TRAILER; }
__global__ __launch_bounds__(64) void k_shard_trailer(const uint64_t* __restrict__ key_dst, const unsigned long long* __restrict__ cursor, uint64_t cap, uint32_t world) {
    const uint32_t o = threadIdx.x;
    if (o >= world) return;
    const unsigned long long filled = cursor[o] < cap ? cursor[o] : cap;
    *reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(key_dst[o]) + cap * 6) = filled;
}

// ---- peer-written exchange (put mode): flags instead of copies ------------------------------------------------------------------
// Every rank keeps, in its own exchange block, four rows of counters per buffer set, one counter per peer; a peer writes them through
// its mapping of the block (hipIpc), the rank polls them where they lie.  Values are piece numbers + 1 and only grow:
//   KEYS_READY[set][p]  peer p's pack of piece t has written its keys into my receive region p             (p -> me, after its pack)
//   KEYS_FREE [set][p]  owner p's probe of piece t has read the keys I wrote into its receive region        (p -> me, after its probe)
//   VALS_READY[set][p]  owner p's probe of piece t has written my values into region p of my gather buffer  (p -> me, after its probe)
//   VALS_FREE [set][p]  requester p's aggregate of piece t has read the values I wrote into its buffer      (p -> me, after its aggregate)
// A stage that needs piece t of every peer waits (k_put_wait, one lane per peer, on the stage's own stream) for counter >= t + 1; a
// stage that overwrites what piece t - NSETS left waits for the matching FREE counter >= t - NSETS + 1.  Every wait is for work that
// every rank enqueues EARLIER in its (identical) issue order than the waiting stage, so the waits cannot form a cycle -- even if a
// rank's streams shared one hardware queue.  A wait gives up after `timeout` ticks of the 100 MHz clock and raises a sticky flag
// (umgap_sharded_check); the grid always drains.
constexpr int FLAG_KEYS_READY = 0, FLAG_KEYS_FREE = 1, FLAG_VALS_READY = 2, FLAG_VALS_FREE = 3, FLAG_KINDS = 4;
struct PutPeers { uint64_t base[64]; };      // rank p's exchange block as mapped in this process (own block at [self])

__global__ __launch_bounds__(64) void k_put_signal(PutPeers P, uint32_t world, uint32_t self, uint64_t off_a, uint64_t off_b, uint64_t value) {
    const uint32_t p = threadIdx.x;
    if (p >= world || p == self) return;
    __threadfence_system();
    unsigned long long* a = reinterpret_cast<unsigned long long*>(P.base[p] + off_a) + self;
    __hip_atomic_store(a, (unsigned long long)value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (off_b) {
        unsigned long long* b = reinterpret_cast<unsigned long long*>(P.base[p] + off_b) + self;
        __hip_atomic_store(b, (unsigned long long)value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the same behind a pack: first the trailer of every region (k_shard_trailer), the owner's copy of it through the mapping
__global__ __launch_bounds__(64) void k_put_keys_ready(PutPeers P, const uint64_t* __restrict__ key_dst, const unsigned long long* __restrict__ cursor, uint64_t cap,
                                                       uint32_t world, uint32_t self, uint64_t off, uint64_t value) {
    const uint32_t p = threadIdx.x;
    if (p >= world) return;
    const unsigned long long filled = cursor[p] < cap ? cursor[p] : cap;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(key_dst[p]) + cap * 6), filled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p == self) return;
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(P.base[p] + off) + self, (unsigned long long)value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_put_wait(const unsigned long long* __restrict__ flags_a, uint64_t want_a, const unsigned long long* __restrict__ flags_b,
                                                 uint64_t want_b, uint32_t world, uint32_t self, uint64_t timeout, uint32_t* __restrict__ timed_out) {
    const uint32_t p = threadIdx.x;
    if (p < world && p != self) {
        const uint64_t t0 = wall_clock64();
        for (;;) {
            const bool a_ok = !flags_a || __hip_atomic_load(flags_a + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want_a;
            const bool b_ok = !flags_b || __hip_atomic_load(flags_b + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want_b;
            if (a_ok && b_ok) break;
            if (wall_clock64() - t0 > timeout) { atomicExch(timed_out, 1u); break; }
            __builtin_amdgcn_s_sleep(64);
        }
    }
    __threadfence_system();
}

// keys (world regions x cap slots) -> one value per slot.  Region `self` is this rank's own share: read from the send buffer
// it was packed into and answered straight into the buffer the returned values are gathered from -- no copy, no wire.
// One wave step = 128 consecutive slots (cap is a multiple of 128, so they share a region): two probe batches in flight; the
// results are packed in the wave's LDS strip and leave as whole 16-byte-per-lane stores.  The grid is PERSISTENT and small
// (three workgroups per CU = 12 waves = ~1500 sectors in flight per CU, several times what the request ceiling needs): the
// hardware dispatches one grid of a queue at a time, so a grid of 200 000 short workgroups keeps every other stream's kernel
// waiting until its last workgroup is placed (a pack "beside" such a probe made no progress at all: r02_sharded_timeline_*),
// while a resident grid that leaves part of each CU's wave slots free lets pack, aggregate and RCCL's copies run next to it.
template <int VB>
__global__ __launch_bounds__(256) void k_shard_probe(TableView T, const uint8_t* __restrict__ recv, const uint8_t* __restrict__ own_send, uint32_t self,
                                                     uint32_t skip_self, uint64_t cap, uint64_t n_slots, const uint64_t* __restrict__ val_dst, uint32_t put) {
    __shared__ uint32_t lds_out[4][128];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 2, j = lane & 3;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    constexpr uint32_t MISS = VB == 3 ? 0xFFFFFFu : 0xFFFFFFFFu;
    for (uint64_t s0 = wave * 128; s0 < n_slots; s0 += nwaves * 128) {
    const uint64_t region = s0 / cap, pos0 = s0 - region * cap;
    if (skip_self && region == self) continue;     // the rank's own keys were not packed (they are probed by the consuming kernel)
    const uint8_t* rb = (region == self ? own_send : recv) + region * region_bytes(cap);
    const uint64_t filled = *reinterpret_cast<const unsigned long long*>(rb + cap * 6);     // the region's trailer (wave-uniform)
    if (pos0 >= filled) {        // nothing but unused slots: their values are never read (no run has a base up here)
        continue;
    }
    const uint32_t* lo = reinterpret_cast<const uint32_t*>(rb) + pos0;
    const uint16_t* hi = reinterpret_cast<const uint16_t*>(rb + cap * 4) + pos0;
    ProbeBatch B[2];
    uint32_t res[2][4];
    // the four key words of both batches in ONE round trip: a load the compiler can sink under `hh < 2^13` becomes a dependent trip
    // of its own, and its wait drains the other batch's bucket loads (the first version: five dependent trips per step, not two)
    uint32_t kl[2], kh[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        kl[b] = __builtin_nontemporal_load(lo + 64 * b + lane);
        kh[b] = __builtin_nontemporal_load(hi + 64 * b + lane);
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const uint64_t h = ((uint64_t)(kh[b] & 0x1FFFu) << 32) | kl[b];
        const uint32_t a = (uint32_t)(h >> R_BITS);
        const uint32_t tag = (uint32_t)h & R_MASK;
        const uint32_t bkt = a * T.m + (uint32_t)(((uint64_t)tag * T.m) >> R_BITS);
        const uint32_t valid = (pos0 + 64 * b + lane < filled && kh[b] < (1u << 13) && bkt >= T.home_lo && bkt < T.home_hi) ? 1u : 0u;   // no key in the slot, or a key this shard does not own: a miss
        probe_issue(T, valid ? bkt - T.home_lo : 0u, tag | (valid << 31), lane, B[b]);
    }
    __builtin_amdgcn_sched_barrier(0);   // all eight sector loads are out before the first is evaluated (the scheduler otherwise starts on batch 0 with half of batch 1 unissued)
    probe_eval_multi<2>(T, B, MISS, lane, res);
    uint32_t* wl = lds_out[wv];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t s = 64 * b + 16 * r;                  // slot (within the wave's 128) of quad 0 of this round
            if (VB == 4) {
                if (j == 0) wl[s + q] = res[b][r];
            } else {
                // four 24-bit values -> three words: quad i = q & 3 of a group of four builds word i from its value and the next quad's
                const uint32_t t = res[b][r] & 0xFFFFFFu, tn = (uint32_t)__shfl_down((int)t, 4, 64);
                const uint32_t i = q & 3;
                const uint32_t w = i == 0 ? (t | (tn << 24)) : (i == 1 ? ((t >> 8) | (tn << 16)) : ((t >> 16) | (tn << 8)));
                if (j == 0 && i < 3) wl[(s >> 2) * 3 + (q >> 2) * 3 + i] = w;
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 128 slots = 96 (VB 3) or 128 (VB 4) words = 24 or 32 lanes x 16 bytes, contiguous
    // val_dst[region]: where slot 0 of the region's values goes -- this rank's value send buffer (region `region`), its own gather buffer
    // (region == self), or, with peer-written values (put mode), region `self` of the REQUESTER's gather buffer, mapped here
    uint32_t* dst = reinterpret_cast<uint32_t*>(val_dst[region]) + (VB == 4 ? pos0 : (pos0 >> 2) * 3);
    if (lane < (VB == 4 ? 32u : 24u)) __builtin_nontemporal_store(reinterpret_cast<const u32x4*>(wl)[lane], reinterpret_cast<u32x4*>(dst) + lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    }
    put_release(put);
}

// the plain sharded get(): out[i] = value of key i, or miss_value.  A wave takes one run of the piece (shard_wire.hpp) in rounds
// of 64 keys: the values of the keys that travelled are read at base + rank, the rank's own keys are probed here (the quad probe,
// all lanes take part).
__global__ __launch_bounds__(256) void k_shard_unpermute(TableView T, HitSource H, size_t n, uint32_t miss_value, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, q = lane >> 2, j = lane & 3;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t mr = wire::miss_raw(H.value_bytes);
    const double wn = (double)H.world / (double)T.nb;
    const size_t n_runs = (n + H.run - 1) / H.run;
    for (size_t run = wave; run < n_runs; run += nwaves) {
        const size_t k0 = run * H.run;
        const uint32_t len = (uint32_t)min((size_t)H.run, n - k0);
        const uint32_t base = lane < H.world ? H.bases[run * H.world + lane] : 0u;
        uint32_t seen = 0;
        for (uint32_t i0 = 0; i0 < len; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint64_t key = i < len ? __builtin_nontemporal_load(H.keys + k0 + i) : 0ull;
            const wire::Routed R = wire::route(key, i < len, T.m, T.nb, H.world, wn);
            const uint32_t rank = wire::run_rank(R.owner, H.world, lane, seen);
            const uint32_t b0 = (uint32_t)__shfl((int)base, (int)(R.owner & 63u), 64);
            const bool is_local = i < len && R.owner == H.self;
            if (i < len && !is_local) {
                uint32_t v = mr;
                const uint64_t pos = (uint64_t)b0 + rank;
                if (R.owner != wire::NO_OWNER && pos < H.cap) v = wire::load_raw(H.ret, (uint64_t)R.owner * H.cap + pos, H.value_bytes);
                out[k0 + i] = v == mr ? miss_value : v;
            }
            const uint64_t lm = __ballot(is_local);
            if (lm) {   // wave-uniform
                uint32_t res[4];
                wave_probe64_vals(T, R.bucket - T.home_lo, R.tag | ((is_local ? 1u : 0u) << 31), miss_value, lane, res);
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if ((lm >> (16 * r + q)) & 1ull) out[k0 + i0 + 16 * r + q] = res[r];
                }
            }
        }
    }
}

// largest value stored in the shard (decides 3- or 4-byte values on the wire)
__global__ __launch_bounds__(256) void k_max_value(const uint64_t* __restrict__ slots, size_t n_slots, const uint32_t* __restrict__ stash_vals,
                                                   uint32_t n_stash, uint32_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t m = 0;
    for (size_t k = i; k < n_slots + n_stash; k += step) {
        uint32_t v;
        if (k < n_slots) { const uint64_t s = slots[k]; if (s == EMPTY_SLOT) continue; v = (uint32_t)(s >> 32); }
        else v = stash_vals[k - n_slots];
        m = v > m ? v : m;
    }
    if (m) atomicMax(out, m);
}

constexpr int NSETS = 4;    // exchange buffer sets: piece t uses set t % NSETS
constexpr int NPERM = 8;    // run-base buffers (piece t uses t % NPERM): a pack never waits for the aggregate NSETS pieces back

struct BufSet {
    hipEvent_t filled = nullptr, packed = nullptr, probed = nullptr;   // piece using this set: packed / keys exchanged / probe finished
    uint8_t *send_keys = nullptr, *recv_keys = nullptr;
    uint32_t *ret_send = nullptr, *ret_recv = nullptr;
    unsigned long long* cursor = nullptr;
    uint64_t *key_dst = nullptr, *val_dst = nullptr;   // [world] device tables: where the pack puts owner o's keys / the probe region r's values
};

}  // namespace

struct umgap_sharded {
    umgap_index* ix = nullptr;
    umgap_taxonomy* tx = nullptr;
    umgap_comm* comm = nullptr;
    int device = 0;
    uint64_t max_keys = 0, piece_keys = 0, cap = 0;
    uint32_t pieces = 1, value_bytes = 4;
    void* block = nullptr;
    size_t block_bytes = 0;
    BufSet set[NSETS];
    uint32_t* bases[NPERM] = {};         // [run][world] first slots of a piece's runs (shard_wire.hpp)
    uint64_t bases_runs = 0;             // runs a bases buffer holds
    hipEvent_t aggregated[NPERM] = {};   // piece using this buffer: aggregate (or un-permute) finished
    // one stream per STAGE: key exchange | pack + probe | value exchange + aggregate.  Pieces flow through them in order: the
    // probes -- each of which saturates the chip's random-request path -- run back to back with the packs between them, the
    // keys of the pieces ahead travel meanwhile and the aggregate of the pieces behind runs underneath.  (One stream per
    // PIECE, the first version, marched three pieces through their stages in lockstep: profiles/r02_sharded_timeline_lockstep.txt.)
    hipStream_t s_keys = nullptr, s_probe = nullptr, s_agg = nullptr;
    hipEvent_t ev_done[3] = {};
    uint64_t piece_no = 0;               // pieces issued so far (event slots are reused cyclically)
    bool local_bypass = true;            // the rank's own keys are probed by the consuming kernel (UMGAP_SHARD_NO_LOCAL=1 sends them through pack / probe / gather like the rest)
    uint32_t* d_overflow = nullptr;
    // peer-written exchange (put mode): the pack stores keys into the owners' receive buffers, the probe stores values into the requesters'
    // gather buffers, through hipIpc mappings of the peers' exchange blocks; counters in those blocks replace the all-to-all (k_put_wait)
    bool put = false;
    PutPeers peers = {};                 // every rank's block as mapped here
    std::vector<void*> peer_maps;        // what hipIpcOpenMemHandle returned (closed in umgap_sharded_free)
    unsigned long long* flags = nullptr; // [FLAG_KINDS][NSETS][64] in this rank's block
    uint64_t flags_off = 0;              // ... and where that is inside the block (the same on every rank)
    uint32_t put_fence = 1;              // what a kernel that stored into a peer's memory ends with (put_release)
    uint32_t* d_put_timeout = nullptr;
    uint64_t put_timeout_ticks = 0;
    hipEvent_t ev_in = nullptr, ev_probe0 = nullptr, ev_probe1 = nullptr;
    bool probe_timed = false;
    // staging of the host-pointer umgap_sharded_lookup9 (made on first use)
    hipStream_t h_stream = nullptr;
    uint64_t* d_hkeys = nullptr;
    uint32_t* d_hout = nullptr;
};

namespace {

// row `kind` of the counters of the set piece t uses: in this rank's block / as an offset into any rank's block
inline const unsigned long long* put_flags(const umgap_sharded* sh, int kind, uint64_t t) { return sh->flags + ((size_t)kind * NSETS + t % NSETS) * 64; }
inline uint64_t put_flags_off(const umgap_sharded* sh, int kind, uint64_t t) { return sh->flags_off + ((uint64_t)kind * NSETS + t % NSETS) * 64 * 8; }

int sharded_args_ok(umgap_sharded* sh, const char* who) {
    if (!sh) return fail(UMGAP_ERR_INVALID, std::string(who) + ": null handle");
    return use_device(sh->device);
}

// one piece, first half: pack -> pad -> key exchange -> local probe, all on the set's stream
// The stages of piece t (global piece number; set t % NSETS, run bases t % NPERM).
//   pack  (s_keys)  overwrites send_keys of its set, last read by the key exchange (earlier on s_keys) and the probe of piece
//                   t - NSETS (waited for), and the run bases, last read by the aggregate of piece t - NPERM;
//   keys  (s_keys)  send_keys -> recv_keys over the wire, after the pack;
//   probe (s_probe) after the keys; overwrites the value buffers of its set, last read by the value exchange and the
//                   aggregate of piece t - NSETS.
// The pack and the key exchange of piece t + LOOK are issued before the probe of piece t: they run (on s_keys) while pieces
// t .. t + LOOK - 1 are probed.
constexpr uint64_t LOOK = 2;

int stage_pack(umgap_sharded* sh, uint64_t t, const uint64_t* d_keys, size_t nk, uint32_t run_len) {
    BufSet& S = sh->set[t % NSETS];
    const uint32_t W = (uint32_t)sh->comm->world;
    const TableView& T = sh->ix->view;
    hipStream_t s = sh->s_keys;                                               // the key exchange of piece t - NSETS is earlier on this stream
    if (t >= NSETS) UMGAP_HIP(hipStreamWaitEvent(s, S.probed, 0));          // the probe of piece t - NSETS has read its own share of send_keys
    if (t >= NPERM) UMGAP_HIP(hipStreamWaitEvent(s, sh->aggregated[t % NPERM], 0));
    if (W == 1 && sh->local_bypass) {      // nobody to send to: every key is this rank's own and the consumer probes it where it stands
        UMGAP_HIP(hipEventRecord(S.filled, s));
        return UMGAP_OK;
    }
    if (sh->put && W > 1 && t >= NSETS) {  // the owners' probes of piece t - NSETS have read what this rank wrote into their receive buffers
        hipLaunchKernelGGL(k_put_wait, dim3(1), dim3(64), 0, s, put_flags(sh, FLAG_KEYS_FREE, t), t - NSETS + 1, (const unsigned long long*)nullptr, 0ull, W,
                           (uint32_t)sh->comm->rank, sh->put_timeout_ticks, sh->d_put_timeout);
    }
    UMGAP_HIP(hipMemsetAsync(S.cursor, 0, W * sizeof(unsigned long long), s));
    const size_t n_runs = (nk + run_len - 1) / run_len;
    if (n_runs > sh->bases_runs) return fail(UMGAP_ERR_INVALID, "sharded lookup: runs of " + std::to_string(run_len) + " k-mers do not fit the run table of this group");
    unsigned g = (unsigned)std::min<size_t>((n_runs + 3) / 4, 256 * 8);   // persistent: a grid of short workgroups holds up the other streams' kernels (k_shard_probe)
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_shard_pack, dim3(g), dim3(256), 0, s, d_keys, nk, run_len, T.m, T.nb, W, (double)W / (double)T.nb, sh->cap, S.key_dst, sh->bases[t % NPERM],
                       S.cursor, sh->d_overflow, sh->local_bypass ? (uint32_t)sh->comm->rank : wire::NO_SELF, sh->put ? sh->put_fence : 0u);
    if (sh->put && W > 1) hipLaunchKernelGGL(k_put_keys_ready, dim3(1), dim3(64), 0, s, sh->peers, S.key_dst, S.cursor, sh->cap, W, (uint32_t)sh->comm->rank, put_flags_off(sh, FLAG_KEYS_READY, t), t + 1);
    else hipLaunchKernelGGL(k_shard_trailer, dim3(1), dim3(64), 0, s, S.key_dst, S.cursor, sh->cap, W);
    UMGAP_HIP(hipGetLastError());
    UMGAP_HIP(hipEventRecord(S.filled, s));
    return UMGAP_OK;
}
int stage_keys(umgap_sharded* sh, uint64_t t) {
    BufSet& S = sh->set[t % NSETS];
    UMGAP_HIP(hipStreamWaitEvent(sh->s_keys, S.filled, 0));
    if (t >= NSETS) UMGAP_HIP(hipStreamWaitEvent(sh->s_keys, S.probed, 0));   // recv_keys was read by the probe of piece t - NSETS
    if (!sh->put) {                          // (put mode: the pack's stores were the exchange; the probe waits for the peers' counters)
        int rc = comm_alltoall(sh->comm, S.send_keys, S.recv_keys, (size_t)region_bytes(sh->cap), sh->s_keys);
        if (rc) return rc;
    }
    UMGAP_HIP(hipEventRecord(S.packed, sh->s_keys));
    return UMGAP_OK;
}
int stage_probe(umgap_sharded* sh, uint64_t t, bool time_probe) {
    BufSet& S = sh->set[t % NSETS];
    const uint32_t W = (uint32_t)sh->comm->world;
    const TableView& T = sh->ix->view;
    hipStream_t s = sh->s_probe;
    UMGAP_HIP(hipStreamWaitEvent(s, S.packed, 0));
    if (t >= NSETS) UMGAP_HIP(hipStreamWaitEvent(s, sh->aggregated[(t - NSETS) % NPERM], 0));
    const uint64_t n_slots = sh->cap * W;
    const unsigned gp = (unsigned)std::min<uint64_t>((n_slots / 128 + 3) / 4, 256 * 3);   // resident grid, 3 workgroups per CU (2: 33.4 ms per step, 3: 30.3, 4: 31.3, 8: 31.2)
    const uint32_t self = (uint32_t)sh->comm->rank;
    if (sh->put && W > 1) {
        // every peer's keys of piece t have arrived, and every requester has read the values of piece t - NSETS out of its gather buffer
        hipLaunchKernelGGL(k_put_wait, dim3(1), dim3(64), 0, s, put_flags(sh, FLAG_KEYS_READY, t), t + 1, t >= NSETS ? put_flags(sh, FLAG_VALS_FREE, t) : (const unsigned long long*)nullptr,
                           t >= NSETS ? t - NSETS + 1 : 0ull, W, self, sh->put_timeout_ticks, sh->d_put_timeout);
    }
    if (time_probe) UMGAP_HIP(hipEventRecord(sh->ev_probe0, s));
    const uint8_t* own_keys = sh->put ? S.recv_keys : S.send_keys;       // put mode: the rank's own region is packed into its receive buffer like a peer's
    if (sh->value_bytes == 3) hipLaunchKernelGGL(k_shard_probe<3>, dim3(gp), dim3(256), 0, s, T, S.recv_keys, own_keys, self, sh->local_bypass ? 1u : 0u, sh->cap, n_slots, S.val_dst, sh->put ? sh->put_fence : 0u);
    else hipLaunchKernelGGL(k_shard_probe<4>, dim3(gp), dim3(256), 0, s, T, S.recv_keys, own_keys, self, sh->local_bypass ? 1u : 0u, sh->cap, n_slots, S.val_dst, sh->put ? sh->put_fence : 0u);
    if (sh->put && W > 1) hipLaunchKernelGGL(k_put_signal, dim3(1), dim3(64), 0, s, sh->peers, W, self, put_flags_off(sh, FLAG_KEYS_FREE, t), put_flags_off(sh, FLAG_VALS_READY, t), t + 1);
    UMGAP_HIP(hipGetLastError());
    UMGAP_HIP(hipEventRecord(S.probed, s));
    if (time_probe) { UMGAP_HIP(hipEventRecord(sh->ev_probe1, s)); sh->probe_timed = true; }
    return UMGAP_OK;
}

// Piece t, second half, on s_agg: the values travel back; the caller then enqueues its consumer (aggregate / un-permute) on
// s_agg and calls piece_done.
int piece_back(umgap_sharded* sh, uint64_t t) {
    BufSet& S = sh->set[t % NSETS];
    UMGAP_HIP(hipStreamWaitEvent(sh->s_agg, S.probed, 0));
    if (sh->put) {                           // the owners' probes of piece t have written this rank's values where the consumer reads them
        if (sh->comm->world > 1) {
            hipLaunchKernelGGL(k_put_wait, dim3(1), dim3(64), 0, sh->s_agg, put_flags(sh, FLAG_VALS_READY, t), t + 1, (const unsigned long long*)nullptr, 0ull, (uint32_t)sh->comm->world,
                               (uint32_t)sh->comm->rank, sh->put_timeout_ticks, sh->d_put_timeout);
            UMGAP_HIP(hipGetLastError());
        }
        return UMGAP_OK;
    }
    return comm_alltoall(sh->comm, S.ret_send, S.ret_recv, (size_t)sh->cap * sh->value_bytes, sh->s_agg);
}
int piece_done(umgap_sharded* sh, uint64_t t) {
    UMGAP_HIP(hipEventRecord(sh->aggregated[t % NPERM], sh->s_agg));
    if (sh->put && sh->comm->world > 1) {    // ... and tells them when it has read them
        hipLaunchKernelGGL(k_put_signal, dim3(1), dim3(64), 0, sh->s_agg, sh->peers, (uint32_t)sh->comm->world, (uint32_t)sh->comm->rank, put_flags_off(sh, FLAG_VALS_FREE, t), 0ull, t + 1);
        UMGAP_HIP(hipGetLastError());
    }
    return UMGAP_OK;
}

constexpr uint32_t LOOKUP_RUN = 1024;    // keys per run of the plain get()

HitSource hit_source(umgap_sharded* sh, uint64_t t, const uint64_t* d_piece_keys, uint32_t run_len) {
    HitSource H;
    H.ret = sh->set[t % NSETS].ret_recv;
    H.bases = sh->bases[t % NPERM];
    H.keys = d_piece_keys;
    H.value_bytes = sh->value_bytes;
    H.world = (uint32_t)sh->comm->world;
    H.self = sh->local_bypass ? (uint32_t)sh->comm->rank : wire::NO_SELF;
    H.run = run_len;
    H.cap = sh->cap;
    return H;
}

int fork_streams(umgap_sharded* sh, hipStream_t caller) {
    UMGAP_HIP(hipEventRecord(sh->ev_in, caller));
    for (hipStream_t s : {sh->s_keys, sh->s_probe, sh->s_agg}) UMGAP_HIP(hipStreamWaitEvent(s, sh->ev_in, 0));
    return UMGAP_OK;
}
int join_streams(umgap_sharded* sh, hipStream_t caller) {
    int i = 0;
    for (hipStream_t s : {sh->s_keys, sh->s_probe, sh->s_agg}) {
        UMGAP_HIP(hipEventRecord(sh->ev_done[i], s));
        UMGAP_HIP(hipStreamWaitEvent(caller, sh->ev_done[i], 0));
        i++;
    }
    return UMGAP_OK;
}

// Put mode or not, and the tables the kernels store through.  Collective (every rank of the communicator is inside
// umgap_sharded_create).  Put mode: the default between the processes of an IPC communicator (UMGAP_SHARD_PUT=0 keeps the
// host-synchronised copies), opt-in (UMGAP_SHARD_PUT=1) under RCCL -- the same peer mappings over xGMI, never run -- and between
// the thread-ranks of a local communicator.
struct PutBlob {
    hipIpcMemHandle_t mem;
    uint64_t block, block_bytes, cap, flags_off;
    uint32_t world, value_bytes, want_put, pid;
    int32_t device, pad;
};
static_assert(sizeof(PutBlob) <= BLOB_BYTES, "the contribution of a rank fits comm_allgather");

#define PUT_TRACE(...) do { if (getenv("UMGAP_SHARD_DEBUG")) { fprintf(stderr, "[rank %d] ", sh->comm->rank); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
int put_setup(umgap_sharded* sh) {
    umgap_comm* c = sh->comm;
    PUT_TRACE("put_setup: block %zu MB", sh->block_bytes >> 20);
    const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank;
    const char* env = getenv("UMGAP_SHARD_PUT");
    bool want = c->ipc ? !(env && env[0] == '0') : (env && env[0] == '1');
    double timeout_s = 30.0;                                   // how long a stage waits for a peer's counter before it gives up
    if (const char* t = getenv("UMGAP_PUT_TIMEOUT_S")) { const double v = atof(t); if (v > 0) timeout_s = v; }
    sh->put_timeout_ticks = (uint64_t)(timeout_s * 1e8);      // wall_clock64: 100 MHz
    if (const char* f = getenv("UMGAP_PUT_FENCE")) { const int v = atoi(f); if (v >= 1 && v <= 3) sh->put_fence = (uint32_t)v; }
    std::vector<PutBlob> all(W);
    if (W > 1) {
        PutBlob mine;
        memset(&mine, 0, sizeof(mine));
        mine.block = (uint64_t)(uintptr_t)sh->block;
        mine.block_bytes = sh->block_bytes;
        mine.cap = sh->cap;
        mine.flags_off = sh->flags_off;
        mine.world = W;
        mine.value_bytes = sh->value_bytes;
        mine.want_put = want ? 1u : 0u;
        mine.pid = (uint32_t)getpid();
        mine.device = sh->device;
        if (want && !c->hub && hipIpcGetMemHandle(&mine.mem, sh->block) != hipSuccess) { (void)hipGetLastError(); mine.want_put = 0; }
        PUT_TRACE("allgather ...");
        int rc = comm_allgather(c, &mine, all.data(), sizeof(PutBlob));
        PUT_TRACE("allgather done rc %d", rc);
        if (rc) return rc;
        for (uint32_t p = 0; p < W; p++) {
            if (all[p].block_bytes != sh->block_bytes || all[p].cap != sh->cap || all[p].flags_off != sh->flags_off || all[p].value_bytes != sh->value_bytes)
                return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: rank " + std::to_string(p) + " laid its exchange buffers out differently (max_keys_per_call, pieces and slack must agree on every rank)");
            want = want && all[p].want_put;                 // everybody, or nobody
        }
        // Ranks that are threads of ONE process on ONE device share that process's hardware queues (the runtime folds its streams onto
        // four per priority): a stage that spins on a peer's counter can sit in front of the very kernel that would raise it -- seen as
        // a stall until the time-out with three thread-ranks on one GPU.  Processes have queues of their own; thread-ranks keep the
        // hub's copies unless every rank drives a device of its own.
        for (uint32_t p = 0; p < W && want; p++)
            for (uint32_t q = p + 1; q < W; q++)
                if (all[p].pid == all[q].pid && all[p].device == all[q].device) want = false;
    }
    sh->put = want;
    sh->peers.base[me] = (uint64_t)(uintptr_t)sh->block;
    int rc = UMGAP_OK;
    if (sh->put && W > 1) {
        // one rank at a time: two processes importing each other's allocation at the same moment were seen to sit in hipIpcOpenMemHandle
        // for good (11 GB blocks, both directions at once)
        for (uint32_t turn = 0; turn < W; turn++) {
          if (turn == me)
            for (uint32_t p = 0; p < W && rc == UMGAP_OK; p++) {
            if (p == me) continue;
            if (c->hub || all[p].pid == (uint32_t)getpid()) { sh->peers.base[p] = all[p].block; continue; }   // a thread of this process: its pointer is ours
            void* ptr = nullptr;
            PUT_TRACE("open rank %u's block ...", p);
            const hipError_t e = hipIpcOpenMemHandle(&ptr, all[p].mem, hipIpcMemLazyEnablePeerAccess);
            PUT_TRACE("open -> %d %p", (int)e, ptr);
            if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(UMGAP_ERR_DEVICE, std::string("umgap_sharded_create: hipIpcOpenMemHandle (rank ") + std::to_string(p) + "'s exchange block): " + hipGetErrorString(e)); break; }
            sh->peer_maps.push_back(ptr);
            sh->peers.base[p] = (uint64_t)(uintptr_t)ptr;
            }
          const int brc = comm_barrier(c);
          if (brc) return brc;
        }
        // every rank has mapped every block, or the create fails on EVERY rank (a rank that went on alone would wait for the others
        // in its first collective)
        uint32_t bad = rc ? 1u : 0u;
        const int rc2 = comm_max_u32(c, &bad);
        PUT_TRACE("agreed: bad %u", bad);
        if (rc2) return rc2;
        if (bad) {
            for (void* m : sh->peer_maps) (void)hipIpcCloseMemHandle(m);
            sh->peer_maps.clear();
            sh->put = false;
            return rc ? rc : fail(UMGAP_ERR_DEVICE, "umgap_sharded_create: another rank could not map the exchange blocks (hipIpcOpenMemHandle); UMGAP_SHARD_PUT=0 keeps the copies");
        }
    }
    // the tables: where the pack stores owner o's keys, where the probe stores region r's values
    std::vector<uint64_t> tab(2 * (size_t)W);
    for (auto& S : sh->set) {
        const uint64_t my_off_recv = (uint64_t)((char*)S.recv_keys - (char*)sh->block), my_off_ret = (uint64_t)((char*)S.ret_recv - (char*)sh->block);
        const uint64_t vwords = sh->value_bytes == 4 ? sh->cap : (sh->cap >> 2) * 3;     // words a region's values take
        for (uint32_t o = 0; o < W; o++) {
            if (sh->put) {
                tab[o] = sh->peers.base[o] + my_off_recv + (uint64_t)me * region_bytes(sh->cap);          // region `me` of rank o's receive buffer
                tab[W + o] = sh->peers.base[o] + my_off_ret + (uint64_t)me * vwords * 4;        // region `me` of rank o's gather buffer
            } else {
                tab[o] = (uint64_t)(uintptr_t)S.send_keys + (uint64_t)o * region_bytes(sh->cap);
                tab[W + o] = o == me ? (uint64_t)(uintptr_t)S.ret_recv + (uint64_t)me * vwords * 4 : (uint64_t)(uintptr_t)S.ret_send + (uint64_t)o * vwords * 4;
            }
        }
        UMGAP_HIP(hipMemcpy(S.key_dst, tab.data(), W * 8, hipMemcpyHostToDevice));
        UMGAP_HIP(hipMemcpy(S.val_dst, tab.data() + W, W * 8, hipMemcpyHostToDevice));
    }
    PUT_TRACE("tables written, put %d", (int)sh->put);
    return UMGAP_OK;
}

// put mode: nobody frees a block a peer may still write into
void put_teardown(umgap_sharded* sh) {
    if (!sh->put || sh->comm->world == 1) return;
    const bool first = comm_barrier(sh->comm) == UMGAP_OK;     // every rank has drained its streams (the caller synchronised the device)
    for (void* m : sh->peer_maps) (void)hipIpcCloseMemHandle(m);
    sh->peer_maps.clear();
    if (first) (void)comm_barrier(sh->comm);                   // ... and has let go of the others' blocks
}

}  // namespace

// ==================================================================================================================
// C ABI
// ==================================================================================================================
extern "C" {

int umgap_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(UMGAP_ERR_INVALID, "umgap_comm_unique_id: null argument");
    Rccl* R = rccl();
    if (!R->error.empty()) return fail(UMGAP_ERR_DEVICE, R->error);
    ncclUniqueId id;
    UMGAP_NCCL(R->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return UMGAP_OK;
}

int umgap_comm_init_rank(const uint8_t* id128, int world, int rank, int device, umgap_comm** out) {
    if (!id128 || !out) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_rank: null argument");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_rank: need 0 <= rank < world <= 64");
    int rc = use_device(device);
    if (rc) return rc;
    Rccl* R = rccl();
    if (!R->error.empty()) return fail(UMGAP_ERR_DEVICE, R->error);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    umgap_comm* c = new umgap_comm();
    c->world = world;
    c->rank = rank;
    c->device = device;
    ncclResult_t r = R->CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(UMGAP_ERR_DEVICE, std::string("ncclCommInitRank: ") + R->GetErrorString(r)); }
    *out = c;
    return UMGAP_OK;
}

int umgap_comm_init_local(int world, int device, umgap_comm** out_ranks) {
    if (!out_ranks) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_local: null argument");
    if (world < 1 || world > 64) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_local: need 1 <= world <= 64");
    int rc = use_device(device);
    if (rc) return rc;
    LocalHub* H = new LocalHub();
    H->world = world;
    H->device = device;
    H->refs = world;
    H->send.assign(world, nullptr);
    H->vals.assign(world, 0);
    H->blobs.resize(world);
    H->ready.assign(world, nullptr);
    H->done.assign(world, nullptr);
    for (int r = 0; r < world; r++) {
        if (hipEventCreateWithFlags(&H->ready[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&H->done[r], hipEventDisableTiming) != hipSuccess) {
            delete H;
            return fail(UMGAP_ERR_DEVICE, "umgap_comm_init_local: hipEventCreate failed");
        }
    }
    for (int r = 0; r < world; r++) {
        umgap_comm* c = new umgap_comm();
        c->world = world;
        c->rank = r;
        c->device = device;
        c->hub = H;
        out_ranks[r] = c;
    }
    return UMGAP_OK;
}

int umgap_comm_init_ipc(const char* name, int world, int rank, int device, umgap_comm** out) {
    if (!name || !out) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_ipc: null argument");
    if (name[0] != '/' || strchr(name + 1, '/') || strlen(name) < 2 || strlen(name) > 200) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_ipc: the name must look like \"/something\" (shm_open)");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(UMGAP_ERR_INVALID, "umgap_comm_init_ipc: need 0 <= rank < world <= 64");
    int rc = use_device(device);
    if (rc) return rc;
    double timeout_s = 120.0;
    if (const char* t = getenv("UMGAP_IPC_TIMEOUT_S")) { const double v = atof(t); if (v > 0) timeout_s = v; }
    const auto t0 = std::chrono::steady_clock::now();
    auto waited = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    int fd = -1;
    // a block whose creator is dead was left behind by a rank 0 that was killed before everybody had attached (the name is unlinked at
    // that point): nobody may rendezvous on it
    auto stale = [&](int f) {
        struct stat st;
        if (fstat(f, &st) != 0 || (size_t)st.st_size < sizeof(IpcCtl)) return false;      // still being made
        void* m = mmap(nullptr, sizeof(IpcCtl), PROT_READ, MAP_SHARED, f, 0);
        if (m == MAP_FAILED) return false;
        const uint32_t pid = static_cast<IpcCtl*>(m)->creator_pid.load(std::memory_order_acquire), ready = static_cast<IpcCtl*>(m)->ready.load(std::memory_order_acquire);
        munmap(m, sizeof(IpcCtl));
        return ready && pid && kill((pid_t)pid, 0) != 0 && errno == ESRCH;
    };
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 && errno == EEXIST) {
            const int old = shm_open(name, O_RDWR, 0600);
            const bool dead = old >= 0 && stale(old);
            if (old >= 0) close(old);
            if (dead) { shm_unlink(name); fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600); }
            else errno = EEXIST;
        }
        if (fd < 0) return fail(UMGAP_ERR_IO, std::string("umgap_comm_init_ipc: shm_open(") + name + "): " + strerror(errno));
        if (ftruncate(fd, sizeof(IpcCtl)) != 0) { const int e = errno; close(fd); shm_unlink(name); return fail(UMGAP_ERR_IO, std::string("umgap_comm_init_ipc: ftruncate: ") + strerror(e)); }
    } else {
        for (;;) {     // until rank 0 has created it at its full size
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(IpcCtl) && !stale(fd)) break;
            if (fd >= 0) close(fd);
            if (waited() > timeout_s) return fail(UMGAP_ERR_IO, std::string("umgap_comm_init_ipc: rank 0 did not create ") + name + " (UMGAP_IPC_TIMEOUT_S)");
            usleep(1000);
        }
    }
    void* mem = mmap(nullptr, sizeof(IpcCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { if (rank == 0) shm_unlink(name); return fail(UMGAP_ERR_IO, std::string("umgap_comm_init_ipc: mmap: ") + strerror(errno)); }
    IpcCtl* ctl = static_cast<IpcCtl*>(mem);       // (a fresh shm object is zero-filled: counters 0, nobody failed)
    if (rank == 0) { ctl->creator_pid.store((uint32_t)getpid(), std::memory_order_release); ctl->ready.store(1, std::memory_order_release); }
    else
        while (!ctl->ready.load(std::memory_order_acquire)) {
            if (waited() > timeout_s) { munmap(mem, sizeof(IpcCtl)); return fail(UMGAP_ERR_IO, "umgap_comm_init_ipc: rank 0 did not initialise the control block"); }
            usleep(200);
        }
    IpcHub* H = new IpcHub();
    H->ctl = ctl;
    H->world = world;
    H->rank = rank;
    H->timeout_s = timeout_s;
    H->opened.resize(world);
    rc = H->barrier();                              // everybody is attached: the name can go (nothing is left behind by a crash later)
    if (rank == 0) shm_unlink(name);
    if (rc) { munmap(mem, sizeof(IpcCtl)); delete H; return rc; }
    umgap_comm* c = new umgap_comm();
    c->world = world;
    c->rank = rank;
    c->device = device;
    c->ipc = H;
    *out = c;
    return UMGAP_OK;
}

int umgap_comm_world(const umgap_comm* c) { return c ? c->world : 0; }
int umgap_comm_rank(const umgap_comm* c) { return c ? c->rank : -1; }

// byte i of the message rank `from` sends to rank `to`
static inline uint8_t selftest_byte(int from, int to, size_t i) { return (uint8_t)(from * 131 + to * 31 + i * 7 + (i >> 8)); }

int umgap_comm_selftest(umgap_comm* c, size_t bytes_per_peer) {
    if (!c || !bytes_per_peer) return fail(UMGAP_ERR_INVALID, "umgap_comm_selftest: null argument");
    int rc = use_device(c->device);
    if (rc) return rc;
    const size_t B = bytes_per_peer, W = (size_t)c->world;
    std::vector<uint8_t> h(W * B);
    for (size_t p = 0; p < W; p++)
        for (size_t i = 0; i < B; i++) h[p * B + i] = selftest_byte(c->rank, (int)p, i);
    uint8_t *d_send = nullptr, *d_recv = nullptr;
    hipStream_t s = nullptr;
    auto done = [&](int code) {
        if (s) (void)hipStreamDestroy(s);
        (void)hipFree(d_send);
        (void)hipFree(d_recv);
        return code;
    };
    if (hipMalloc((void**)&d_send, W * B) != hipSuccess || hipMalloc((void**)&d_recv, W * B) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
        return done(fail(UMGAP_ERR_NOMEM, "umgap_comm_selftest: allocation failed"));
    if (hipMemcpy(d_send, h.data(), W * B, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_recv, 0, W * B) != hipSuccess)
        return done(fail(UMGAP_ERR_DEVICE, "umgap_comm_selftest: copy in failed"));
    rc = comm_alltoall(c, d_send, d_recv, B, s, /*with_self=*/true);
    if (rc) return done(rc);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h.data(), d_recv, W * B, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(UMGAP_ERR_DEVICE, "umgap_comm_selftest: the exchange did not complete"));
    for (size_t p = 0; p < W; p++)
        for (size_t i = 0; i < B; i++)
            if (h[p * B + i] != selftest_byte((int)p, c->rank, i))
                return done(fail(UMGAP_ERR_DEVICE, "umgap_comm_selftest: rank " + std::to_string(c->rank) + " got a wrong byte " + std::to_string(i) + " from rank " + std::to_string(p)));
    return done(UMGAP_OK);
}

void umgap_comm_free(umgap_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
    if (c->ipc) {
        for (auto& peer : c->ipc->opened)
            for (auto& kv : peer) (void)hipIpcCloseMemHandle(kv.second);
        munmap(c->ipc->ctl, sizeof(IpcCtl));
        delete c->ipc;
    }
    if (c->hub) {
        bool last;
        { std::lock_guard<std::mutex> g(c->hub->m); last = --c->hub->refs == 0; }
        if (last) {
            for (auto e : c->hub->ready) if (e) (void)hipEventDestroy(e);
            for (auto e : c->hub->done) if (e) (void)hipEventDestroy(e);
            delete c->hub;
        }
    }
    delete c;
}

int umgap_sharded_create(umgap_index* shard, umgap_taxonomy* tx, umgap_comm* comm, uint64_t max_keys_per_call, uint32_t pieces, double slack,
                         umgap_sharded** out) {
    if (!shard || !comm || !out) return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: null argument");
    if (shard->kind != 9) return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: needs a 9-mer index");
    if (shard->bound_to) return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: the index is bound to a taxonomy; shards exchange taxon ids");
    if (shard->device != comm->device || (tx && tx->device != shard->device)) return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: index, taxonomy and communicator must live on one device");
    const uint32_t W = (uint32_t)comm->world, nb = shard->view.nb;
    const uint32_t lo = (uint32_t)((uint64_t)nb * comm->rank / W), hi = (uint32_t)((uint64_t)nb * (comm->rank + 1) / W);
    if (shard->view.home_lo != lo || shard->view.home_hi != hi)
        return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: the index handle is not shard " + std::to_string(comm->rank) + " of " + std::to_string(W) +
                                           " (open it with umgap_index_open_shard / umgap_index_set_shard)");
    if (max_keys_per_call == 0) return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: max_keys_per_call must be positive");
    if (pieces == 0) pieces = 8;
    if (pieces > 64) pieces = 64;
    if (!(slack >= 0.0) || slack > 4.0) slack = 0.03;
    int rc = use_device(shard->device);
    if (rc) return rc;
    umgap_sharded* sh = new umgap_sharded();
    sh->ix = shard;
    sh->tx = tx;
    sh->comm = comm;
    sh->device = shard->device;
    sh->max_keys = max_keys_per_call;
    sh->pieces = pieces;
    sh->piece_keys = (max_keys_per_call + pieces - 1) / pieces;
    { const char* nl = getenv("UMGAP_SHARD_NO_LOCAL"); sh->local_bypass = !(nl && nl[0] == '1'); }
    // a region receives piece_keys / W keys on average; binomial spread ~ sqrt of that, so slack + 8 sigma + one tile
    const double mean = (double)sh->piece_keys / W;
    uint64_t cap = (uint64_t)(mean * (1.0 + slack) + 8.0 * std::sqrt(mean + 1.0)) + 4096;
    cap = (cap + 127) & ~(uint64_t)127;
    if (cap * W >= 0xFFFFFFFFull) { delete sh; return fail(UMGAP_ERR_INVALID, "umgap_sharded_create: piece too large (raise `pieces`)"); }
    sh->cap = cap;
    // 3-byte values when every value on every shard fits (all-ones is the miss marker)
    uint32_t* d_max = nullptr;
    hipError_t e = hipMalloc((void**)&d_max, 4);
    uint32_t vmax = 0;
    if (e == hipSuccess) e = hipMemset(d_max, 0, 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_max_value, dim3(256 * 8), dim3(256), 0, nullptr, shard->d_slots, (size_t)shard->view.n_resident * SLOTS, shard->d_stash_vals, shard->n_stash, d_max);
        e = hipMemcpy(&vmax, d_max, 4, hipMemcpyDeviceToHost);
    }
    if (d_max) (void)hipFree(d_max);
    if (e != hipSuccess) { delete sh; return fail(UMGAP_ERR_DEVICE, std::string("umgap_sharded_create: ") + hipGetErrorString(e)); }
    rc = comm_max_u32(comm, &vmax);
    if (rc) { delete sh; return rc; }
    sh->value_bytes = vmax < 0xFFFFFFu ? 3 : 4;
    // one block: per set {send keys, recv keys, values out, values back, cursor}, then the run bases
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // (the shortest run the fused kernels form is 33 k-mers: one read of 33..64 fragments of one k-mer each; stage_pack checks)
    sh->bases_runs = sh->piece_keys / 32 + 2;
    const size_t b_keys = pad(region_bytes(cap) * W), b_vals = pad(cap * W * 4 + 64), b_bases = pad(sh->bases_runs * W * 4), b_cur = pad(W * 8), b_tab = pad(W * 8);
    const size_t per_set = 2 * b_keys + 2 * b_vals + b_cur + 2 * b_tab;
    const size_t b_flags = (size_t)FLAG_KINDS * NSETS * 64 * 8;
    sh->block_bytes = NSETS * per_set + NPERM * b_bases + 256 + b_flags;
    // The peers map this block (put mode; the IPC communicator's copies).  hipIpcOpenMemHandle of an allocation whose size has bit 31 set
    // never comes back with the HIP / ROCr that PyTorch 2.10+rocm7.0 bundles (dmabuf IPC: the importer spins in recvmsg inside
    // rocr::core::Runtime::IPCClientImport; /opt/rocm 7.2's copies open the same sizes at once -- tools/ipc_open_probe.py,
    // tools/native/ipc_open_probe.hip).  So: whole 2 MB pages, and a size whose low 32 bits stay below 2^31.
    sh->block_bytes = (sh->block_bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (sh->block_bytes & 0x80000000ull) sh->block_bytes = (sh->block_bytes + 0xFFFFFFFFull) & ~0xFFFFFFFFull;
    e = hipMalloc(&sh->block, sh->block_bytes);
    if (e != hipSuccess) { delete sh; return fail(UMGAP_ERR_NOMEM, std::string("umgap_sharded_create: hipMalloc of the exchange buffers (") + std::to_string(sh->block_bytes >> 20) + " MiB): " + hipGetErrorString(e)); }
    char* p = (char*)sh->block;
    for (auto& S : sh->set) {
        S.send_keys = (uint8_t*)p; p += b_keys;
        S.recv_keys = (uint8_t*)p; p += b_keys;
        S.ret_send = (uint32_t*)p; p += b_vals;
        S.ret_recv = (uint32_t*)p; p += b_vals;
        S.cursor = (unsigned long long*)p; p += b_cur;
        S.key_dst = (uint64_t*)p; p += b_tab;
        S.val_dst = (uint64_t*)p; p += b_tab;
        if (hipEventCreateWithFlags(&S.filled, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S.packed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S.probed, hipEventDisableTiming) != hipSuccess) {
            umgap_sharded_free(sh);
            return fail(UMGAP_ERR_DEVICE, "umgap_sharded_create: event creation failed");
        }
    }
    for (int i = 0; i < NPERM; i++) {
        sh->bases[i] = (uint32_t*)p; p += b_bases;
        if (hipEventCreateWithFlags(&sh->aggregated[i], hipEventDisableTiming) != hipSuccess) { umgap_sharded_free(sh); return fail(UMGAP_ERR_DEVICE, "umgap_sharded_create: event creation failed"); }
    }
    sh->d_overflow = (uint32_t*)p;
    sh->d_put_timeout = sh->d_overflow + 1;
    p += 256;
    sh->flags = (unsigned long long*)p;
    sh->flags_off = (uint64_t)(p - (char*)sh->block);
    e = hipMemset(sh->d_overflow, 0, 256 + b_flags);
    if (e == hipSuccess) e = hipDeviceSynchronize();     // the counters are zero before any peer learns where they are
    if (e != hipSuccess) { umgap_sharded_free(sh); return fail(UMGAP_ERR_DEVICE, std::string("umgap_sharded_create: ") + hipGetErrorString(e)); }
    if ((rc = put_setup(sh))) { umgap_sharded_free(sh); return rc; }
    {
        // One hardware queue per stage.  Streams of one priority share a small pool of hardware queues (four by default, the null stream
        // and the caller's among them): the kernel trace of round 5 showed the pack and the probe stream on ONE queue -- they never
        // overlapped, and in put mode a pack that waits for a peer's counter held up the probes behind it.  Streams of different
        // priorities come from different pools, so each stage gets a queue of its own: probe high (the stage that saturates the
        // request path), pack normal, aggregate low.  UMGAP_SHARD_STREAM_PRIO="k,p,a" overrides (0 0 0: the shared pool).
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);        // lo = numerically largest = lowest priority
        int prio[3] = {0, hi, lo};
        if (const char* pe = getenv("UMGAP_SHARD_STREAM_PRIO")) { int a, b, c; if (sscanf(pe, "%d,%d,%d", &a, &b, &c) == 3) { prio[0] = a; prio[1] = b; prio[2] = c; } }
        int k = 0;
        for (hipStream_t* st : {&sh->s_keys, &sh->s_probe, &sh->s_agg}) {
            const int pr = std::max(hi, std::min(lo, prio[k++]));
            if (e == hipSuccess) e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, pr);
        }
    }
    for (int i = 0; i < 3; i++)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&sh->ev_probe0);
    if (e == hipSuccess) e = hipEventCreate(&sh->ev_probe1);
    if (e != hipSuccess) { umgap_sharded_free(sh); return fail(UMGAP_ERR_DEVICE, std::string("umgap_sharded_create: ") + hipGetErrorString(e)); }
    *out = sh;
    return UMGAP_OK;
}

void umgap_sharded_free(umgap_sharded* sh) {
    if (!sh) return;
    (void)hipSetDevice(sh->device);
    (void)hipDeviceSynchronize();
    put_teardown(sh);
    for (auto& S : sh->set) {
        if (S.filled) (void)hipEventDestroy(S.filled);
        if (S.packed) (void)hipEventDestroy(S.packed);
        if (S.probed) (void)hipEventDestroy(S.probed);
    }
    for (auto ev : sh->aggregated) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : sh->ev_done) if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t st : {sh->s_keys, sh->s_probe, sh->s_agg}) if (st) (void)hipStreamDestroy(st);
    if (sh->ev_in) (void)hipEventDestroy(sh->ev_in);
    if (sh->ev_probe0) (void)hipEventDestroy(sh->ev_probe0);
    if (sh->ev_probe1) (void)hipEventDestroy(sh->ev_probe1);
    if (sh->block) (void)hipFree(sh->block);
    if (sh->h_stream) (void)hipStreamDestroy(sh->h_stream);
    if (sh->d_hkeys) (void)hipFree(sh->d_hkeys);
    if (sh->d_hout) (void)hipFree(sh->d_hout);
    delete sh;
}

uint32_t umgap_sharded_value_bytes(const umgap_sharded* sh) { return sh ? sh->value_bytes : 0; }
uint64_t umgap_sharded_region_slots(const umgap_sharded* sh) { return sh ? sh->cap : 0; }
uint32_t umgap_sharded_pieces(const umgap_sharded* sh) { return sh ? sh->pieces : 0; }
int umgap_sharded_transport(const umgap_sharded* sh) { return sh && sh->put ? 1 : 0; }

int umgap_sharded_lookup9_dev(umgap_sharded* sh, const uint64_t* d_keys45, size_t n, uint32_t miss_value, uint32_t* d_out_taxa, void* stream) {
    int rc = sharded_args_ok(sh, "umgap_sharded_lookup9_dev");
    if (rc) return rc;
    if (n && (!d_keys45 || !d_out_taxa)) return fail(UMGAP_ERR_INVALID, "umgap_sharded_lookup9_dev: null argument");
    if (n > sh->max_keys) return fail(UMGAP_ERR_INVALID, "umgap_sharded_lookup9_dev: more keys than the max_keys_per_call the group was created with");
    hipStream_t caller = (hipStream_t)stream;
    // every rank takes part in every exchange, even with nothing to look up: the piece count is fixed
    const size_t per = sh->piece_keys, np = sh->pieces;
    if ((rc = fork_streams(sh, caller))) return rc;
    const uint64_t first = sh->piece_no;
    sh->piece_no += np;
    auto pack_piece = [&](size_t t) -> int {
        const size_t lo = std::min(n, t * per), hi = std::min(n, (t + 1) * per);
        int r = stage_pack(sh, first + t, d_keys45 + lo, hi - lo, LOOKUP_RUN);
        return r ? r : stage_keys(sh, first + t);
    };
    for (size_t t = 0; t < std::min<size_t>(LOOK, np); t++) if ((rc = pack_piece(t))) return rc;
    for (size_t t = 0; t <= np; t++) {
        if (t < np) {
            if (t + LOOK < np && (rc = pack_piece(t + LOOK))) return rc;
            if ((rc = stage_probe(sh, first + t, t == 0))) return rc;
        }
        if (t >= 1) {
            const size_t c = t - 1, lo = std::min(n, c * per), hi = std::min(n, (c + 1) * per);
            const uint64_t pc = first + c;
            if ((rc = piece_back(sh, pc))) return rc;
            if (hi > lo) {
                unsigned g = (unsigned)std::min<size_t>(((hi - lo + LOOKUP_RUN - 1) / LOOKUP_RUN + 3) / 4, 2048);   // a wave per run
                hipLaunchKernelGGL(k_shard_unpermute, dim3(g), dim3(256), 0, sh->s_agg, sh->ix->view, hit_source(sh, pc, d_keys45 + lo, LOOKUP_RUN), hi - lo, miss_value,
                                   d_out_taxa + lo);
                UMGAP_HIP(hipGetLastError());
            }
            if ((rc = piece_done(sh, pc))) return rc;
        }
    }
    return join_streams(sh, caller);
}

int umgap_sharded_lookup9(umgap_sharded* sh, const uint64_t* keys45, size_t n, uint32_t miss_value, uint32_t* out_taxa) {
    int rc = sharded_args_ok(sh, "umgap_sharded_lookup9");
    if (rc) return rc;
    if (n && (!keys45 || !out_taxa)) return fail(UMGAP_ERR_INVALID, "umgap_sharded_lookup9: null argument");
    if (n > sh->max_keys) return fail(UMGAP_ERR_INVALID, "umgap_sharded_lookup9: more keys than the max_keys_per_call the group was created with");
    if (!sh->h_stream) {
        UMGAP_HIP(hipStreamCreateWithFlags(&sh->h_stream, hipStreamNonBlocking));
        UMGAP_HIP(hipMalloc((void**)&sh->d_hkeys, sh->max_keys * 8));
        UMGAP_HIP(hipMalloc((void**)&sh->d_hout, sh->max_keys * 4));
    }
    if (n) UMGAP_HIP(hipMemcpyAsync(sh->d_hkeys, keys45, n * 8, hipMemcpyHostToDevice, sh->h_stream));
    rc = umgap_sharded_lookup9_dev(sh, sh->d_hkeys, n, miss_value, sh->d_hout, sh->h_stream);   // collective even when n == 0
    if (rc) return rc;
    if (n) UMGAP_HIP(hipMemcpyAsync(out_taxa, sh->d_hout, n * 4, hipMemcpyDeviceToHost, sh->h_stream));
    UMGAP_HIP(hipStreamSynchronize(sh->h_stream));
    return UMGAP_OK;
}

int umgap_sharded_classify9_dev(umgap_sharded* sh, int method, int strategy, float factor, float lower_bound, int ranked_only,
                                const uint64_t* d_keys45, size_t n_reads, uint32_t kmers_per_read, uint32_t fragments_per_read,
                                uint32_t min_seed_size, uint32_t max_gap_size, uint32_t* d_out_taxon, uint32_t* d_out_status, void* stream) {
    int rc = sharded_args_ok(sh, "umgap_sharded_classify9_dev");
    if (rc) return rc;
    if (!sh->tx) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: the group was created without a taxonomy");
    if (n_reads && (!d_keys45 || !d_out_taxon || !d_out_status)) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: null argument");
    if (kmers_per_read == 0 || kmers_per_read > 128) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: kmers_per_read must be in 1..128");
    if (fragments_per_read && (kmers_per_read % fragments_per_read) != 0) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: kmers_per_read must be a multiple of fragments_per_read");
    if (fragments_per_read > 64) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: at most 64 fragments per read");
    if (method != UMGAP_METHOD_TREE && method != UMGAP_METHOD_RMQ) return fail(UMGAP_ERR_INVALID, "unknown method");
    if (strategy < 0 || strategy > 2) return fail(UMGAP_ERR_INVALID, "unknown strategy");
    if (method == UMGAP_METHOD_TREE && strategy == UMGAP_STRATEGY_MRTL)
        return fail(UMGAP_ERR_INVALID, "Invalid invocation: Tree and MaximumRootToLeafPath cannot be combined");  // src/commands/taxa2agg.rs:134-138
    if ((uint64_t)n_reads * kmers_per_read > sh->max_keys) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: more keys than the max_keys_per_call the group was created with");
    hipStream_t caller = (hipStream_t)stream;
    const size_t per = sh->piece_keys / kmers_per_read, np = sh->pieces;   // whole reads per piece
    if (per == 0) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: a piece cannot hold one read (lower `pieces`)");
    if ((uint64_t)per * np < n_reads) return fail(UMGAP_ERR_INVALID, "umgap_sharded_classify9_dev: the reads do not fit the pieces (create the group with a max_keys_per_call that is a multiple of pieces * kmers_per_read)");
    const uint32_t run_len = classify9_group_reads(kmers_per_read, fragments_per_read) * kmers_per_read;   // the consumer's group (classify.hip)
    if ((rc = fork_streams(sh, caller))) return rc;
    const uint64_t first = sh->piece_no;
    sh->piece_no += np;
    auto pack_piece = [&](size_t t) -> int {
        const size_t lo = std::min(n_reads, t * per), hi = std::min(n_reads, (t + 1) * per);
        int r = stage_pack(sh, first + t, d_keys45 + lo * kmers_per_read, (hi - lo) * kmers_per_read, run_len);
        return r ? r : stage_keys(sh, first + t);
    };
    for (size_t t = 0; t < std::min<size_t>(LOOK, np); t++) if ((rc = pack_piece(t))) return rc;
    for (size_t t = 0; t <= np; t++) {
        if (t < np) {
            if (t + LOOK < np && (rc = pack_piece(t + LOOK))) return rc;
            if ((rc = stage_probe(sh, first + t, t == 0))) return rc;
        }
        if (t >= 1) {
            const size_t c = t - 1, lo = std::min(n_reads, c * per), hi = std::min(n_reads, (c + 1) * per);
            const uint64_t pc = first + c;
            if ((rc = piece_back(sh, pc))) return rc;
            rc = launch_classify9_from_hits(sh->ix, sh->tx, method, strategy, factor, lower_bound, ranked_only,
                                            hit_source(sh, pc, d_keys45 + lo * kmers_per_read, run_len), hi - lo, kmers_per_read,
                                            fragments_per_read, min_seed_size, max_gap_size, d_out_taxon + lo, d_out_status + lo, sh->s_agg);
            if (rc) return rc;
            if ((rc = piece_done(sh, pc))) return rc;
        }
    }
    return join_streams(sh, caller);
}

int umgap_sharded_check(umgap_sharded* sh) {
    int rc = sharded_args_ok(sh, "umgap_sharded_check");
    if (rc) return rc;
    UMGAP_HIP(hipDeviceSynchronize());
    uint32_t ov = 0, late = 0;
    UMGAP_HIP(hipMemcpy(&late, sh->d_put_timeout, 4, hipMemcpyDeviceToHost));
    if (late) {
        UMGAP_HIP(hipMemset(sh->d_put_timeout, 0, 4));
        return fail(UMGAP_ERR_DEVICE, "sharded lookup: a peer's keys or values did not arrive in time (UMGAP_PUT_TIMEOUT_S) since the last check; the answers of those calls are not valid");
    }
    UMGAP_HIP(hipMemcpy(&ov, sh->d_overflow, 4, hipMemcpyDeviceToHost));
    if (ov) {
        UMGAP_HIP(hipMemset(sh->d_overflow, 0, 4));
        return fail(UMGAP_ERR_BUILD, "sharded lookup: a send region overflowed its capacity since the last check (those keys were answered with a miss); raise `slack`");
    }
    return UMGAP_OK;
}

int umgap_sharded_probe_ms(umgap_sharded* sh, float* ms) {
    if (!sh || !ms) return fail(UMGAP_ERR_INVALID, "umgap_sharded_probe_ms: null argument");
    if (!sh->probe_timed) return fail(UMGAP_ERR_INVALID, "umgap_sharded_probe_ms: no call yet");
    int rc = use_device(sh->device);
    if (rc) return rc;
    UMGAP_HIP(hipEventSynchronize(sh->ev_probe1));
    UMGAP_HIP(hipEventElapsedTime(ms, sh->ev_probe0, sh->ev_probe1));
    return UMGAP_OK;
}

}  // extern "C"
