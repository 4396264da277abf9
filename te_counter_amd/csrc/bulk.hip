// Bulk SE/PE overlap counting on gfx950: the count kernel and the tec_bulk_* entry points.
//
// Replaces the per-read loops of measureTE.parse_bamse (te_count/te_count.py:201-263) and
// measureTE.parse_bampe (te_count/te_count.py:76-152).  One unit (SE record / PE pair) per lane.
//
// Per unit, exactly as the reference:
//   filters      flag & (0x4|0x200|0x400) -> qcfail ; mapq < q -> lowq ; chrom not in index -> badchrom
//   buckets      b1 = floor((p1-1)/10000), b2 = floor((p2+1)/10000)           (te_count.py:106-107, 222-223)
//   candidate    feature f is looked at iff b1 or b2 lies in [floor(Gl/bs), floor(Gr/bs)]
//                (te_count.py:113-116, 228-231 over genelist.py:371-378) -- evaluated in closed form
//   edge tests   T1: p1 >= Gl and p1+1 <= Gr ; T2: p2-1 >= Gl and p2 <= Gr   (te_count.py:122,125 / 237,240)
//   counting     if any hit has a countable class: +1 for every DISTINCT name among the hits
//                (te_count.py:132-144, 245-257); only-'enhancer' hits crash the reference (:145-147)
// Distinct names need no per-lane buffer: entries are sorted by name id inside a fine bin, so
// duplicates inside one bin scan are adjacent, and a hit from the second edge's bin is checked
// against the first bin by re-scanning it (both lists are short and L1-resident by then).
#include <algorithm>

#include "tecount_internal.h"

namespace {

constexpr uint32_t kNameMask = (1u << TEC_META_NAME_BITS) - 1;
constexpr uint32_t kCountable = (1u << TEC_CLASS_GENE) | (1u << TEC_CLASS_TE) | (1u << TEC_CLASS_SNRNA);
constexpr uint32_t kNoName = 0xFFFFFFFFu;
constexpr int kBlock = 1024;
constexpr uint32_t kChunk = 4096;           // units per workgroup per grid-stride step
constexpr uint32_t kLdsBytes = 79 * 1024;   // per workgroup: two workgroups fit a CU's 160 KiB

__device__ __forceinline__ int floordiv_bs(long long x) {
    long long q = x / TEC_BUCKET_SIZE;
    if (x - q * TEC_BUCKET_SIZE < 0) --q;
    return (int)q;
}

struct Unit {
    int p1, p2;  // edge positions
    int b1, b2;  // the two 10 kb buckets the reference would open
};

__device__ __forceinline__ bool is_candidate(const tec_entry& e, const Unit& u) {
    const int lo = e.left / TEC_BUCKET_SIZE;  // left >= 0 is enforced at build time
    const int hi = floordiv_bs(e.right);
    return (lo <= u.b1 && u.b1 <= hi) || (lo <= u.b2 && u.b2 <= hi);
}
__device__ __forceinline__ bool test1(const tec_entry& e, const Unit& u) { return u.p1 >= e.left && u.p1 < e.right; }
__device__ __forceinline__ bool test2(const tec_entry& e, const Unit& u) { return u.p2 > e.left && u.p2 <= e.right; }

__device__ __forceinline__ void bin_range(const tec_index_dev& idx, int chrom, int pos, uint32_t* a, uint32_t* z) {
    *a = *z = 0;
    if (pos < 0) return;
    const uint32_t b = (uint32_t)pos >> idx.shift;
    if (b >= idx.chrom_nbins[chrom]) return;
    const uint32_t g = idx.chrom_base[chrom] + b;
    *a = idx.bin_off[g];
    *z = idx.bin_off[g + 1];
}

// Where a hit is counted.  The k_lds hottest names (internal ids are ordered by expected hit
// rate, see index.hip) live in a per-workgroup LDS histogram of packed 16-bit counters
// (ds_add_u32 of 1 << 16*(id&1)); a workgroup flushes it to the 64-bit global vector before any
// counter can reach 65536.  Colder names go straight to global 64-bit atomics.
struct Sink {
    uint32_t* lds;
    uint32_t k_lds;
    unsigned long long* counts;
    __device__ __forceinline__ void add(uint32_t name) const {
        if (name < k_lds)
            atomicAdd(&lds[name >> 1], 1u << ((name & 1u) * 16u));
        else
            atomicAdd(&counts[name], 1ull);
    }
};

// Walks the hits of one unit.  COUNT: add 1 to every distinct hit name.  Returns the OR of
// (1 << class) over all hits (0 = no hit).
template <bool COUNT>
__device__ __forceinline__ uint32_t walk_hits(const tec_index_dev& idx, const Unit& u, uint32_t a0, uint32_t a1,
                                              uint32_t b0, uint32_t b1, const Sink& sink) {
    uint32_t classes = 0;
    const tec_entry* __restrict__ ent = idx.entries;
    if (a0 == b0) {  // both edges fall into one fine bin
        uint32_t last = kNoName;
        for (uint32_t j = a0; j < a1; ++j) {
            const tec_entry e = ent[j];
            if ((test1(e, u) || test2(e, u)) && is_candidate(e, u)) {
                classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
                const uint32_t name = e.meta & kNameMask;
                if (COUNT && name != last) sink.add(name);
                last = name;
            }
        }
        return classes;
    }
    uint32_t last = kNoName;
    for (uint32_t j = a0; j < a1; ++j) {
        const tec_entry e = ent[j];
        if (test1(e, u) && is_candidate(e, u)) {
            classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
            const uint32_t name = e.meta & kNameMask;
            if (COUNT && name != last) sink.add(name);
            last = name;
        }
    }
    last = kNoName;
    for (uint32_t j = b0; j < b1; ++j) {
        const tec_entry e = ent[j];
        if (test2(e, u) && is_candidate(e, u)) {
            classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
            const uint32_t name = e.meta & kNameMask;
            if (COUNT && name != last) {
                bool seen = false;  // already counted through the first edge?
                for (uint32_t k = a0; k < a1 && !seen; ++k) {
                    const tec_entry f = ent[k];
                    seen = (f.meta & kNameMask) == name && test1(f, u) && is_candidate(f, u);
                }
                if (!seen) sink.add(name);
            }
            last = name;
        }
    }
    return classes;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void flush_hist(uint32_t* hist, uint32_t words, unsigned long long* counts) {
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) {
        const uint32_t v = hist[w];
        if (v) {
            hist[w] = 0;
            if (v & 0xFFFFu) atomicAdd(&counts[2 * w], (unsigned long long)(v & 0xFFFFu));
            if (v >> 16) atomicAdd(&counts[2 * w + 1], (unsigned long long)(v >> 16));
        }
    }
    __syncthreads();
}

// MIXED = the index holds classes that are not countable in bulk (miRNA..., enhancer): the class
// decision then needs a first pass over the hits before anything may be counted.
// Persistent workgroups walk the batch in chunks of `chunk` units (grid-stride over chunks).
template <bool MIXED>
__global__ __launch_bounds__(kBlock) void count_bulk_kernel(tec_index_dev idx, const int32_t* __restrict__ chrom,
                                                           const int32_t* __restrict__ p1,
                                                           const int32_t* __restrict__ p2,
                                                           const uint16_t* __restrict__ flag,
                                                           const uint8_t* __restrict__ mapq, uint64_t n, int mapq_min,
                                                           unsigned long long* counts, unsigned long long* stats,
                                                           uint32_t k_lds, uint32_t chunk) {
    extern __shared__ uint32_t hist[];
    const uint32_t words = (k_lds + 1u) >> 1;
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) hist[w] = 0;
    __syncthreads();
    const Sink sink{hist, k_lds, counts};
    unsigned n_assigned = 0, n_lowq = 0, n_bad = 0, n_qc = 0, n_enh = 0;
    uint32_t since_flush = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * chunk; base < n; base += (uint64_t)gridDim.x * chunk) {
        if (since_flush + chunk > 65535u) {  // a 16-bit counter gains at most 1 per unit
            flush_hist(hist, words, counts);
            since_flush = 0;
        }
        since_flush += chunk;
        const uint64_t end = base + chunk < n ? base + chunk : n;
        for (uint64_t i = base + threadIdx.x; i < end; i += kBlock) {
            const unsigned fl = flag[i];
            if (fl & (TEC_FLAG_UNMAPPED | TEC_FLAG_QCFAIL | TEC_FLAG_DUP)) {
                ++n_qc;
                continue;
            }
            if ((int)mapq[i] < mapq_min) {
                ++n_lowq;
                continue;
            }
            const int c = chrom[i];
            if (c < 0 || (uint32_t)c >= idx.n_chrom) {
                ++n_bad;
                continue;
            }
            Unit u;
            u.p1 = p1[i];
            u.p2 = p2[i];
            u.b1 = floordiv_bs((long long)u.p1 - 1);
            u.b2 = floordiv_bs((long long)u.p2 + 1);
            uint32_t a0, a1, b0, b1;
            bin_range(idx, c, u.p1, &a0, &a1);
            bin_range(idx, c, u.p2, &b0, &b1);
            if (a0 == a1 && b0 == b1) continue;
            if (a0 == a1) {  // only the second edge has a list: let it run as the "first" one
                a0 = b0;
                a1 = b1;
            } else if (b0 == b1) {
                b0 = a0;
                b1 = a1;
            }
            if (MIXED) {
                const uint32_t classes = walk_hits<false>(idx, u, a0, a1, b0, b1, sink);
                if (classes == 0) continue;
                ++n_assigned;
                if (classes & kCountable)
                    walk_hits<true>(idx, u, a0, a1, b0, b1, sink);
                else if (classes & (1u << TEC_CLASS_ENHANCER))
                    ++n_enh;
            } else {
                if (walk_hits<true>(idx, u, a0, a1, b0, b1, sink)) ++n_assigned;
            }
        }
    }
    flush_hist(hist, words, counts);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)n);
    n_assigned = wave_sum(n_assigned);
    n_lowq = wave_sum(n_lowq);
    n_bad = wave_sum(n_bad);
    n_qc = wave_sum(n_qc);
    n_enh = wave_sum(n_enh);
    if ((threadIdx.x & 63) == 0) {
        if (n_assigned) atomicAdd(&stats[1], (unsigned long long)n_assigned);
        if (n_lowq) atomicAdd(&stats[2], (unsigned long long)n_lowq);
        if (n_bad) atomicAdd(&stats[3], (unsigned long long)n_bad);
        if (n_qc) atomicAdd(&stats[4], (unsigned long long)n_qc);
        if (n_enh) atomicAdd(&stats[5], (unsigned long long)n_enh);
    }
}

struct UnitPtrs {
    const int32_t *chrom, *p1, *p2;
    const uint16_t* flag;
    const uint8_t* mapq;
};

inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }
inline size_t slot_bytes(uint64_t n) { return 3 * al256(n * 4) + al256(n * 2) + al256(n); }
inline UnitPtrs carve(void* base, uint64_t n) {
    char* p = static_cast<char*>(base);
    UnitPtrs u;
    u.chrom = reinterpret_cast<const int32_t*>(p);
    p += al256(n * 4);
    u.p1 = reinterpret_cast<const int32_t*>(p);
    p += al256(n * 4);
    u.p2 = reinterpret_cast<const int32_t*>(p);
    p += al256(n * 4);
    u.flag = reinterpret_cast<const uint16_t*>(p);
    p += al256(n * 2);
    u.mapq = reinterpret_cast<const uint8_t*>(p);
    return u;
}

int harvest_timers(tec_ctx* ctx) {
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.f;
        TEC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i].start, ctx->ev_pool[i].stop));
        ctx->kernel_ms += ms;
    }
    ctx->ev_used = 0;
    return TEC_OK;
}

int launch_count(tec_ctx* ctx, const UnitPtrs& u, uint64_t n) {
    if (n == 0) return TEC_OK;
    if (ctx->ev_used == ctx->ev_pool.size()) {
        if (ctx->ev_pool.size() >= 4096) {  // bound the pool: drain what is in flight
            TEC_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
            int rc = harvest_timers(ctx);
            if (rc) return rc;
        } else {
            tec_event_pair p;
            TEC_HIP(ctx, hipEventCreate(&p.start));
            TEC_HIP(ctx, hipEventCreate(&p.stop));
            ctx->ev_pool.push_back(p);
        }
    }
    tec_event_pair& ev = ctx->ev_pool[ctx->ev_used++];
    const uint64_t want = (n + kChunk - 1) / kChunk;
    const unsigned grid = (unsigned)std::min<uint64_t>(want, (uint64_t)ctx->n_cu * 2);
    const bool mixed = (ctx->idx.class_mask & ~kCountable) != 0;
    const uint32_t k_lds = std::min<uint32_t>(ctx->idx.n_names, kLdsBytes / 2);
    const size_t lds = (size_t)((k_lds + 1) / 2) * 4;
    TEC_HIP(ctx, hipEventRecord(ev.start, ctx->s_compute));
    if (mixed)
        hipLaunchKernelGGL(count_bulk_kernel<true>, dim3(grid), dim3(kBlock), lds, ctx->s_compute, ctx->idx, u.chrom,
                           u.p1, u.p2, u.flag, u.mapq, n, ctx->mapq_min, ctx->d_counts, ctx->d_stats, k_lds, kChunk);
    else
        hipLaunchKernelGGL(count_bulk_kernel<false>, dim3(grid), dim3(kBlock), lds, ctx->s_compute, ctx->idx, u.chrom,
                           u.p1, u.p2, u.flag, u.mapq, n, ctx->mapq_min, ctx->d_counts, ctx->d_stats, k_lds, kChunk);
    TEC_HIP(ctx, hipGetLastError());
    TEC_HIP(ctx, hipEventRecord(ev.stop, ctx->s_compute));
    ++ctx->kernel_launches;
    return TEC_OK;
}

}  // namespace

int tec_bulk_init(tec_ctx* ctx) {
    // 79 KiB of dynamic LDS per workgroup needs the opt-in above the 64 KiB default
    TEC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&count_bulk_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    TEC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&count_bulk_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    return TEC_OK;
}

int tec_bulk_release(tec_ctx* ctx) {
    for (auto& s : ctx->slots) {
        if (s.d_buf) (void)hipFree(s.d_buf);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
        s = tec_slot();
    }
    for (auto& p : ctx->ev_pool) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    ctx->ev_pool.clear();
    ctx->ev_used = 0;
    return TEC_OK;
}

extern "C" int tec_bulk_begin(tec_ctx* ctx, int mapq_min) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_begin: NULL context");
    if (!ctx->have_index) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_begin: no index resident (call tec_index_build)");
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    TEC_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    TEC_HIP(ctx, hipStreamSynchronize(ctx->s_copy));
    int rc = harvest_timers(ctx);
    if (rc) return rc;
    ctx->mapq_min = mapq_min;
    ctx->kernel_ms = 0.0;
    ctx->kernel_launches = 0;
    TEC_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, (size_t)ctx->idx.n_names * sizeof(unsigned long long), ctx->s_compute));
    TEC_HIP(ctx, hipMemsetAsync(ctx->d_stats, 0, 8 * sizeof(unsigned long long), ctx->s_compute));
    ctx->bulk_open = true;
    return TEC_OK;
}

extern "C" int tec_bulk_submit(tec_ctx* ctx, uint64_t n, const int32_t* chrom, const int32_t* p1, const int32_t* p2,
                               const uint16_t* flag, const uint8_t* mapq, uint64_t* ticket) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_submit: NULL context");
    if (!ctx->bulk_open) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_submit: call tec_bulk_begin first");
    if (n && (!chrom || !p1 || !p2 || !flag || !mapq)) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_submit: NULL array");
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t t = ctx->next_ticket++;
    if (ticket) *ticket = t;
    tec_slot& s = ctx->slots[t % tec_ctx::kSlots];
    if (!s.copied) {
        TEC_HIP(ctx, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        TEC_HIP(ctx, hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    } else {
        // the slot's previous batch must have been read by its kernel before it is overwritten
        TEC_HIP(ctx, hipStreamWaitEvent(ctx->s_copy, s.consumed, 0));
    }
    s.ticket = t;
    if (n == 0) {
        TEC_HIP(ctx, hipEventRecord(s.copied, ctx->s_copy));
        TEC_HIP(ctx, hipEventRecord(s.consumed, ctx->s_compute));
        return TEC_OK;
    }
    const size_t need = slot_bytes(n);
    if (need > s.cap) {
        if (s.d_buf) {
            TEC_HIP(ctx, hipEventSynchronize(s.consumed));
            TEC_HIP(ctx, hipFree(s.d_buf));
            s.d_buf = nullptr;
            s.cap = 0;
        }
        TEC_HIP(ctx, hipMalloc(&s.d_buf, need));
        s.cap = need;
    }
    const UnitPtrs d = carve(s.d_buf, n);
    TEC_HIP(ctx, hipMemcpyAsync((void*)d.chrom, chrom, n * 4, hipMemcpyHostToDevice, ctx->s_copy));
    TEC_HIP(ctx, hipMemcpyAsync((void*)d.p1, p1, n * 4, hipMemcpyHostToDevice, ctx->s_copy));
    TEC_HIP(ctx, hipMemcpyAsync((void*)d.p2, p2, n * 4, hipMemcpyHostToDevice, ctx->s_copy));
    TEC_HIP(ctx, hipMemcpyAsync((void*)d.flag, flag, n * 2, hipMemcpyHostToDevice, ctx->s_copy));
    TEC_HIP(ctx, hipMemcpyAsync((void*)d.mapq, mapq, n, hipMemcpyHostToDevice, ctx->s_copy));
    TEC_HIP(ctx, hipEventRecord(s.copied, ctx->s_copy));
    TEC_HIP(ctx, hipStreamWaitEvent(ctx->s_compute, s.copied, 0));
    int rc = launch_count(ctx, d, n);
    if (rc) return rc;
    TEC_HIP(ctx, hipEventRecord(s.consumed, ctx->s_compute));
    return TEC_OK;
}

extern "C" int tec_bulk_wait(tec_ctx* ctx, uint64_t ticket) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_wait: NULL context");
    if (ticket == 0 || ticket >= ctx->next_ticket) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_wait: unknown ticket");
    // If the slot has been recycled since, its newer copy event is later in the same stream.
    tec_slot& s = ctx->slots[ticket % tec_ctx::kSlots];
    TEC_HIP(ctx, hipEventSynchronize(s.copied));
    return TEC_OK;
}

extern "C" int tec_bulk_submit_device(tec_ctx* ctx, uint64_t n, const int32_t* chrom, const int32_t* p1,
                                      const int32_t* p2, const uint16_t* flag, const uint8_t* mapq) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_submit_device: NULL context");
    if (!ctx->bulk_open) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_submit_device: call tec_bulk_begin first");
    if (n && (!chrom || !p1 || !p2 || !flag || !mapq)) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_submit_device: NULL array");
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    UnitPtrs d{chrom, p1, p2, flag, mapq};
    return launch_count(ctx, d, n);
}

extern "C" int tec_bulk_sync(tec_ctx* ctx) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_sync: NULL context");
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    TEC_HIP(ctx, hipStreamSynchronize(ctx->s_copy));
    TEC_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
    return harvest_timers(ctx);
}

extern "C" int tec_bulk_device_buffers(tec_ctx* ctx, void** d_counts, void** d_stats) {
    if (!ctx || !ctx->have_index) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_device_buffers: no index resident");
    if (d_counts) *d_counts = ctx->d_counts;
    if (d_stats) *d_stats = ctx->d_stats;
    return TEC_OK;
}

extern "C" int tec_bulk_finish(tec_ctx* ctx, uint64_t* counts, tec_stats* stats) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_finish: NULL context");
    if (!ctx->bulk_open) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_finish: no bulk run is open");
    int rc = tec_bulk_sync(ctx);
    if (rc) return rc;
    unsigned long long h[8];
    TEC_HIP(ctx, hipMemcpy(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost));
    if (counts) {  // device vector is in internal (hotness) order
        std::vector<uint64_t> tmp(ctx->idx.n_names);
        TEC_HIP(ctx, hipMemcpy(tmp.data(), ctx->d_counts, tmp.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < ctx->idx.n_names; ++i) counts[ctx->int2ext[i]] = tmp[i];
    }
    if (stats) {
        stats->units = h[0];
        stats->assigned = h[1];
        stats->lowq = h[2];
        stats->badchrom = h[3];
        stats->qcfail = h[4];
    }
    ctx->bulk_open = false;
    if (h[5])
        return tec_fail(ctx, TEC_EENHANCER,
                        "%llu unit(s) hit only 'enhancer' features: the reference raises NameError('barcode') "
                        "there (te_count.py:145-147, 258-260)",
                        h[5]);
    return TEC_OK;
}

extern "C" int tec_bulk_kernel_time(tec_ctx* ctx, double* total_ms, uint64_t* launches) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_kernel_time: NULL context");
    int rc = tec_bulk_sync(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->kernel_ms;
    if (launches) *launches = ctx->kernel_launches;
    return TEC_OK;
}
