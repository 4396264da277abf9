// Hit names beyond the LDS histogram: the fold of the bulk count kernel's spill list.
//
// An annotation with millions of distinct names (BASELINE config 5: 3 M intervals with unique gene ids)
// does not fit the per-workgroup LDS histogram of count_bulk_rec_kernel (154 KiB = 77 k 16-bit
// counters).  Counting the rest with one memory-side 64-bit atomic per hit -- what
// te_count.py:140-144 / 253-257 amounts to when done naively in parallel -- made 145 M scattered atomics
// per 100 M-unit launch and ran at 3 % of the HBM roofline.  Instead the count kernel appends those
// name ids to a list that is already split by bucket (the top bits of the id; bulk.hip, FastSink), and
// here one workgroup per bucket counts its slices in an LDS histogram and adds the histogram to the
// count vector with plain stores: a name belongs to exactly one bucket, so nothing is atomic in memory,
// nothing is sorted, and the host never waits (the fold is queued right behind the count kernel).
#include <algorithm>
#include <vector>

#include <hip/hip_ext.h>

#include "tecount_internal.h"

namespace {

// Workgroup b: slices (b, 0..n_wg) of `slice_cap` entries, `counts[b * n_wg + wg]` of them valid.
__global__ __launch_bounds__(1024) void spill_bucket_kernel(const uint32_t* __restrict__ spill, const uint32_t* __restrict__ counts,
                                                            uint32_t n_wg, uint32_t slice_cap, uint32_t lo, uint32_t n_names,
                                                            unsigned long long* out, unsigned long long* total) {
    extern __shared__ uint32_t hist[];
    const uint32_t size = 1u << lo, b = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < size; i += 1024u) hist[i] = 0;
    __syncthreads();
    // 16 waves: each takes every 16th slice, its lanes stride over the slice
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t seen = 0;
    for (uint32_t wg = wave; wg < n_wg; wg += 16u) {
        const uint32_t n = counts[(size_t)b * n_wg + wg];
        const uint32_t* __restrict__ src = spill + ((size_t)b * n_wg + wg) * slice_cap;
        for (uint32_t i = lane; i < n; i += 64u) atomicAdd(&hist[src[i] & (size - 1u)], 1u);
        seen += lane == 0 ? n : 0u;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < size; i += 1024u) {
        const uint32_t name = (b << lo) + i;
        if (hist[i] && name < n_names) out[name] += hist[i];  // this workgroup owns the bucket's names
    }
    if (seen) atomicAdd(total, (unsigned long long)seen);
}

template <class T>
int grow(tec_ctx* ctx, T** p, size_t* cap, size_t want) {
    if (want <= *cap) return TEC_OK;
    TEC_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    if (*p) TEC_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    want += want / 8;
    TEC_HIP(ctx, hipMalloc(reinterpret_cast<void**>(p), want * sizeof(T)));
    *cap = want;
    return TEC_OK;
}

}  // namespace

int tec_spill_reserve(tec_ctx* ctx, uint32_t n_wg, uint32_t n_buckets, uint32_t slice_cap) {
    int rc = grow(ctx, &ctx->d_spill, &ctx->spill_cap_total, (size_t)n_wg * n_buckets * slice_cap);
    if (rc) return rc;
    if (!ctx->d_spill_total) {
        TEC_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_spill_total), 8));
        TEC_HIP(ctx, hipMemset(ctx->d_spill_total, 0, 8));
    }
    return grow(ctx, &ctx->d_spill_counts, &ctx->spill_counts_cap, (size_t)n_wg * n_buckets);
}

// Queued behind the count kernel on the same stream.
int tec_spill_fold(tec_ctx* ctx, uint32_t n_wg, uint32_t n_buckets, uint32_t lo, uint32_t slice_cap) {
    if (ctx->spill_ev_used == ctx->spill_ev.size()) {
        tec_event_pair p;
        TEC_HIP(ctx, hipEventCreate(&p.start));
        TEC_HIP(ctx, hipEventCreate(&p.stop));
        p.reduced = nullptr;
        ctx->spill_ev.push_back(p);
    }
    tec_event_pair& ev = ctx->spill_ev[ctx->spill_ev_used++];
    hipExtLaunchKernelGGL(spill_bucket_kernel, dim3(n_buckets), dim3(1024), (uint32_t)(4u << lo), ctx->s_compute, ev.start, ev.stop, 0u,
                          (const uint32_t*)ctx->d_spill, (const uint32_t*)ctx->d_spill_counts, n_wg, slice_cap, lo, ctx->idx.n_names,
                          ctx->d_counts, ctx->d_spill_total);
    TEC_HIP(ctx, hipGetLastError());
    return TEC_OK;
}

// (called with the stream drained) event times and the list length since the last call
int tec_spill_harvest(tec_ctx* ctx) {
    for (size_t i = 0; i < ctx->spill_ev_used; ++i) {
        float ms = 0.f;
        TEC_HIP(ctx, hipEventElapsedTime(&ms, ctx->spill_ev[i].start, ctx->spill_ev[i].stop));
        ctx->spill_ms += ms;
    }
    ctx->spill_ev_used = 0;
    if (ctx->d_spill_total) {
        unsigned long long t = 0;
        TEC_HIP(ctx, hipMemcpy(&t, ctx->d_spill_total, 8, hipMemcpyDeviceToHost));
        ctx->spill_names += t;
        TEC_HIP(ctx, hipMemset(ctx->d_spill_total, 0, 8));
    }
    return TEC_OK;
}

void tec_spill_release(tec_ctx* ctx) {
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    if (ctx->d_spill_counts) (void)hipFree(ctx->d_spill_counts);
    if (ctx->d_spill_total) (void)hipFree(ctx->d_spill_total);
    ctx->d_spill = ctx->d_spill_counts = nullptr;
    ctx->d_spill_total = nullptr;
    ctx->spill_cap_total = ctx->spill_counts_cap = 0;
    for (auto& p : ctx->spill_ev) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    ctx->spill_ev.clear();
    ctx->spill_ev_used = 0;
}
