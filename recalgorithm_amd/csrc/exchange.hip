// Owner bucketing for the row-sharded embedding arenas: global row r lives on rank
// r % world at local row r / world.  One batch's requests are split into `world` fixed-capacity
// buckets (static shapes: the sharded step stays one hipGraph, no host sync); the buckets then
// travel through RCCL all_to_all (host side: recalgorithm_amd/parallel.py).
//
// HBM-bound integer work, one pass: 8 B read + 2-3 x 8 B written per request.  Bucket slots are handed
// out by wave-aggregated atomics: the lanes of a wave that target the same owner are counted with a
// ballot, ONE lane bumps the owner's counter by the group size, and each lane takes base + its rank
// in the group — world <= 64 distinct owners cost at most `world` atomics per 64 requests instead
// of 64 colliding ones.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void exchange_plan_init_kernel(int64_t* __restrict__ send_local,
                                                                 int64_t* __restrict__ send_pos, int64_t slots,
                                                                 int* __restrict__ counters, int world) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < slots) {
        send_local[i] = -1;
        if (send_pos) send_pos[i] = -1;
    }
    if (i < world) counters[i] = 0;
}

__global__ __launch_bounds__(256) void exchange_plan_assign_kernel(const int64_t* __restrict__ rows, int64_t M,
                                                                   unsigned world, int64_t cap,
                                                                   int64_t* __restrict__ send_local,
                                                                   int64_t* __restrict__ send_pos,
                                                                   int64_t* __restrict__ req_slot,
                                                                   int* __restrict__ counters,
                                                                   unsigned char* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    const int64_t row = i < M ? rows[i] : -1;
    const bool valid = row >= 0;
    const unsigned owner = valid ? (unsigned)(row % world) : 0u;
    int64_t slot = -1;
    // peel one owner group per iteration (uniform loop: at most min(world, 64) rounds per wave)
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const unsigned leader = (unsigned)__ffsll((long long)todo) - 1;
        const unsigned o = (unsigned)__shfl((int)owner, (int)leader, 64);
        const unsigned long long grp = __ballot(valid && owner == o) & todo;
        int base = 0;
        if (lane == leader) base = atomicAdd(&counters[o], __popcll(grp));
        base = __shfl(base, (int)leader, 64);
        if ((grp >> lane) & 1ull) slot = base + __popcll(grp & ((1ull << lane) - 1ull));
        todo &= ~grp;
    }
    if (i >= M) return;
    int64_t dest = -1;
    if (valid) {
        if (slot < cap) {
            dest = (int64_t)owner * cap + slot;
            send_local[dest] = row / world;
            if (send_pos) send_pos[dest] = i;
        } else {
            *overflow = 1;          // sticky; the request is dropped (its staged row reads as zeros)
        }
    }
    req_slot[i] = dest;
}

// ---- request de-duplication ---------------------------------------------------------------------------------
// A Zipf batch asks for its hot rows many times; only the FIRST request of each row needs to travel.  An
// open-addressing table (keys = rows, value = the smallest request index that asked for the row) is filled with one
// atomicCAS + one atomicMin per request, then every request reads its row's entry back: deterministic (the
// representative is the smallest index whatever the arrival order), static shapes, no sort.
__global__ __launch_bounds__(256) void dedup_init_kernel(int64_t* __restrict__ keys, int* __restrict__ first, int64_t slots) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < slots) {
        keys[i] = -1;
        first[i] = 0x7fffffff;
    }
}

__global__ __launch_bounds__(256) void dedup_insert_kernel(const int64_t* __restrict__ rows, int64_t M,
                                                           int64_t* __restrict__ keys, int* __restrict__ first,
                                                           unsigned shift, unsigned mask, int* __restrict__ slot_of) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int64_t row = rows[i];
    if (row < 0) return;
    unsigned h = (unsigned)(((unsigned long long)row * 0x9E3779B97F4A7C15ull) >> shift) & mask;
    for (;;) {
        const unsigned long long seen = atomicCAS((unsigned long long*)&keys[h], ~0ull, (unsigned long long)row);
        if (seen == ~0ull || seen == (unsigned long long)row) break;
        h = (h + 1) & mask;                       // load factor <= 1/2: short probes
    }
    atomicMin(&first[h], (int)i);
    slot_of[i] = (int)h;
}

__global__ __launch_bounds__(256) void dedup_resolve_kernel(const int64_t* __restrict__ rows, int64_t M,
                                                            const int* __restrict__ first, const int* __restrict__ slot_of,
                                                            int64_t* __restrict__ unique_rows, int64_t* __restrict__ rep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int64_t row = rows[i];
    const int64_t r = row < 0 ? i : (int64_t)first[slot_of[i]];
    rep[i] = r;
    unique_rows[i] = r == i ? row : -1;
}

struct DedupLayout {
    int64_t slots;
    unsigned shift, mask;
};
DedupLayout dedup_layout(int64_t M) {
    unsigned lg = 6;
    while ((1ll << lg) < 2 * M) ++lg;
    return {1ll << lg, 64u - lg, (unsigned)((1ll << lg) - 1)};
}

}  // namespace

RECALGO_EXPORT int64_t recalgo_dedup_rows_workspace_bytes(int64_t M) {
    if (M < 0 || M >= (1ll << 30)) return -1;
    return dedup_layout(M).slots * 12 + M * 4;
}

RECALGO_EXPORT int recalgo_dedup_rows(const int64_t* rows, int64_t M, int64_t* unique_rows, int64_t* rep, void* workspace,
                                      recalgo_stream_t stream) {
    RECALGO_REQUIRE(M >= 0 && M < (1ll << 30));
    if (M == 0) return 0;
    RECALGO_REQUIRE(rows != nullptr && unique_rows != nullptr && rep != nullptr && workspace != nullptr);
    const DedupLayout L = dedup_layout(M);
    int64_t* keys = (int64_t*)workspace;
    int* first = (int*)(keys + L.slots);
    int* slot_of = first + L.slots;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(dedup_init_kernel, dim3(cdiv(L.slots, 256)), dim3(256), 0, st, keys, first, L.slots);
    hipLaunchKernelGGL(dedup_insert_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, rows, M, keys, first, L.shift, L.mask, slot_of);
    hipLaunchKernelGGL(dedup_resolve_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, rows, M, first, slot_of, unique_rows, rep);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_exchange_plan(const int64_t* rows, int64_t M, int world, int64_t cap,
                                         int64_t* send_local, int64_t* send_pos, int64_t* req_slot,
                                         int* counters, unsigned char* overflow, recalgo_stream_t stream) {
    RECALGO_REQUIRE(M >= 0 && world >= 1 && cap >= 1 && send_local != nullptr);
    RECALGO_REQUIRE(counters != nullptr && overflow != nullptr && (M == 0 || (rows != nullptr && req_slot != nullptr)));
    hipStream_t st = as_stream(stream);
    const int64_t slots = (int64_t)world * cap;
    const int64_t n_init = slots > world ? slots : world;
    hipLaunchKernelGGL(exchange_plan_init_kernel, dim3(cdiv(n_init, 256)), dim3(256), 0, st, send_local, send_pos,
                       slots, counters, world);
    if (M > 0)
        hipLaunchKernelGGL(exchange_plan_assign_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, rows, M,
                           (unsigned)world, cap, send_local, send_pos, req_slot, counters, overflow);
    RECALGO_RETURN_LAST();
}
