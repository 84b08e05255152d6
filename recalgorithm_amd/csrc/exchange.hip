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

}  // namespace

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
