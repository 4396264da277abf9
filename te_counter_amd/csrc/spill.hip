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

// Workgroup (b, s): slices (b, wg) of `slice_cap` entries (a multiple of four: the slices are read sixteen bytes
// a lane), `counts[b * n_wg + wg]` of them valid, for the wg that fall to part s of gridDim.y.  A slice holds a few
// thousand names, so a wave that read it four bytes a lane and one load at a time spent the launch waiting for
// memory: here a lane has four 16-byte loads in flight before its first LDS add.
__global__ __launch_bounds__(1024) void spill_bucket_kernel(const uint32_t* __restrict__ spill, const uint32_t* __restrict__ counts,
                                                            uint32_t n_wg, uint32_t slice_cap, uint32_t lo, uint32_t n_names,
                                                            unsigned long long* out, unsigned long long* total) {
    extern __shared__ uint32_t hist[];
    const uint32_t size = 1u << lo, mask = size - 1u, b = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < size; i += 1024u) hist[i] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t first = wave + 16u * blockIdx.y, step = 16u * gridDim.y;
    uint32_t seen = 0;
    for (uint32_t wg = first; wg < n_wg; wg += step) {
        const uint32_t n = counts[(size_t)b * n_wg + wg];
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(spill + ((size_t)b * n_wg + wg) * slice_cap);
        const uint32_t nv = (n + 3u) >> 2;
        for (uint32_t i = lane; i < nv; i += 256u) {
            uint4 v[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t at = i + 64u * k;
                v[k] = at < nv ? src[at] : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t at = (i + 64u * k) * 4u;  // (beyond n for the vectors that were not loaded)
                if (at < n) atomicAdd(&hist[v[k].x & mask], 1u);
                if (at + 1u < n) atomicAdd(&hist[v[k].y & mask], 1u);
                if (at + 2u < n) atomicAdd(&hist[v[k].z & mask], 1u);
                if (at + 3u < n) atomicAdd(&hist[v[k].w & mask], 1u);
            }
        }
        seen += lane == 0 ? n : 0u;
    }
    __syncthreads();
    const bool alone = gridDim.y == 1u;  // this workgroup owns the bucket's names: plain adds
    for (uint32_t i = threadIdx.x; i < size; i += 1024u) {
        const uint32_t name = (b << lo) + i, h = hist[i];
        if (h && name < n_names) {
            if (alone)
                out[name] += h;
            else
                atomicAdd(&out[name], (unsigned long long)h);
        }
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
        // (a bucket of 32 768 names is 128 KiB of histogram: more than 64 KiB of dynamic LDS has to be asked for)
        TEC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spill_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
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
    // fewer buckets than CUs: several workgroups share a bucket's slices (and add with atomics)
    const uint32_t parts = std::max(1u, std::min((uint32_t)ctx->n_cu / n_buckets, 8u));
    hipExtLaunchKernelGGL(spill_bucket_kernel, dim3(n_buckets, parts), dim3(1024), (uint32_t)(4u << lo), ctx->s_compute, ev.start, ev.stop, 0u,
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
