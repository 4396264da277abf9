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
//
// Kernels in this file:
//   count_bulk_rec_kernel  the product kernel: one compact 32-byte bin record per edge lookup,
//                          continuation pieces for longer lists, six-slot register hit set, packed
//                          16-bit LDS histogram with carry bookkeeping, histogram rows + reduce
//   reduce_slabs_kernel    folds the per-workgroup histogram rows into the u64 count vector
//   count_bulk_kernel      the CSR walk (indexes without records: bins wider than 32 kb, more than
//                          512 chromosomes); also the exact path's walk_hits lives here
#include <algorithm>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "tecount_internal.h"

namespace {

constexpr uint32_t kNameMask = (1u << TEC_META_NAME_BITS) - 1;
constexpr uint32_t kCountable = (1u << TEC_CLASS_GENE) | (1u << TEC_CLASS_TE) | (1u << TEC_CLASS_SNRNA);
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kNoName = 0xFFFFFFFFu;
constexpr int kBlock = 1024;
#ifndef TEC_REC_WAVES
#define TEC_REC_WAVES 4  // min waves per SIMD the record kernel is compiled for (one 1024-thread workgroup per CU)
#endif
constexpr uint32_t kLdsMaxBytes = 154 * 1024;  // one workgroup per CU (room for the static chromosome table)
constexpr uint32_t kChromLds = 512;            // the record kernel keeps every chromosome's {base, bins} in LDS

// Diagnostic builds (make EXTRA=-DTEC_DIAGNOSTICS): wave 0 of every workgroup accumulates the
// shader-clock cycles it spends per stage of the record kernel (with a full s_waitcnt at every
// stage end, so the numbers are latencies, and the kernel is slower than the product build).
#ifdef TEC_DIAGNOSTICS
__device__ unsigned long long g_stage_clk[8];
#define TEC_CLK_DECL unsigned long long clk_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long clk_t = __builtin_amdgcn_s_memtime()
#define TEC_CLK(i)                                                     \
    do {                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
        clk_acc[i] += now__ - clk_t;                                   \
        clk_t = now__;                                                 \
    } while (0)
#define TEC_CLK_STORE                                                                   \
    do {                                                                                \
        if (threadIdx.x == 0)                                                           \
            for (int i__ = 0; i__ < 6; ++i__) atomicAdd(&g_stage_clk[i__], clk_acc[i__]); \
    } while (0)
#else
#define TEC_CLK_DECL
#define TEC_CLK(i)
#define TEC_CLK_STORE
#endif

// floor(x / 10000) for x >= -1 (one 32-bit multiply-high; negative -> -1).
__device__ __forceinline__ int bucket_of(int x) { return x < 0 ? -1 : (int)((unsigned)x / (unsigned)TEC_BUCKET_SIZE); }

struct Unit {
    int p1, p2;     // edge positions
    bool boundary;  // p1 or p2+1 is a multiple of 10000: the only case in which the reference's
                    // bucket-candidate rule can reject a feature that passes an edge test
};

// Does the reference look at this feature at all?  It opens buckets b1 = floor((p1-1)/bs) and
// b2 = floor((p2+1)/bs) only (te_count.py:106-116, 222-231); the feature is listed in buckets
// floor(Gl/bs)..floor(Gr/bs) (genelist.py:371-378).  A feature that passes T1 contains p1 and
// p1+1, so it is listed under floor(p1/bs), which equals b1 unless p1 % bs == 0; likewise for T2
// and b2 unless (p2+1) % bs == 0.  Only then is the full test needed.
__device__ __forceinline__ bool is_candidate_slow(const tec_entry& e, const Unit& u) {
    const int lo = bucket_of(e.left), hi = bucket_of(e.right);
    const int b1 = bucket_of(u.p1 < 0 ? -1 : u.p1 - 1);
    const int b2 = bucket_of(u.p2 < -1 ? -1 : (u.p2 > 0x7FFFFFFE ? 0x7FFFFFFF : u.p2 + 1));
    return (lo <= b1 && b1 <= hi) || (lo <= b2 && b2 <= hi);
}
__device__ __forceinline__ bool is_candidate(const tec_entry& e, const Unit& u) {
    return !u.boundary || is_candidate_slow(e, u);
}
__device__ __forceinline__ bool test1(const tec_entry& e, const Unit& u) { return u.p1 >= e.left && u.p1 < e.right; }
__device__ __forceinline__ bool test2(const tec_entry& e, const Unit& u) { return u.p2 > e.left && u.p2 <= e.right; }

// Where a hit is counted.  The k_lds hottest names (internal ids are ordered by expected hit
// rate, see index.hip) live in a per-workgroup LDS histogram of packed 16-bit counters
// (ds_add_rtn_u32 of n << 16*(id&1)); colder names go straight to global 64-bit atomics.
//
// The 16-bit counters are never flushed before the end of the launch.  Instead every add looks at
// the value it replaced: the one add that carries out of its half sees it (old half + n > 0xFFFF)
// and books 65536 on the name's global counter.  A carry out of the LOW half also lands in the
// neighbour's HIGH half; the same adder books -1 for the neighbour (64-bit two's complement) and,
// if that pushed the high half over as well, the neighbour's 65536.  Atomics on one LDS word are
// totally ordered, so each carry is seen by exactly one adder under any interleaving, and
// global + (final LDS halves) is exact -- with no barrier and no flush inside the loop.
__device__ __noinline__ void wrap_account(unsigned long long* counts, uint32_t name, uint32_t old) {
    atomicAdd(&counts[name], 65536ull);
    if (!(name & 1u)) {
        atomicAdd(&counts[name + 1u], ~0ull);  // the carry this add pushed into the neighbour's half
        if ((old >> 16) == 0xFFFFu) atomicAdd(&counts[name + 1u], 65536ull);
    }
}
__device__ __forceinline__ bool wrapped(uint32_t name, uint32_t old, uint32_t n) {
    return ((old >> ((name & 1u) * 16u)) & 0xFFFFu) + n > 0xFFFFu;
}
__device__ __forceinline__ void lds_add(uint32_t* lds, unsigned long long* counts, uint32_t name, uint32_t n) {
    const uint32_t old = atomicAdd(&lds[name >> 1], n << ((name & 1u) * 16u));
    if (wrapped(name, old, n)) wrap_account(counts, name, old);
}

struct Sink {
    uint32_t* lds;
    uint32_t k_lds;
    unsigned long long* counts;
    bool aggregate;  // coordinate-sorted batches: neighbouring lanes hit the same name
    __device__ __forceinline__ void add_n(uint32_t name, uint32_t n) const {
        if (name < k_lds)
            lds_add(lds, counts, name, n);
        else
            atomicAdd(&counts[name], (unsigned long long)n);
    }
    // Runs of equal names over the lanes that are adding right now are folded into one add by the
    // run's first lane (64 lanes on one LDS word would serialise otherwise).  n <= 64 per add.
    __device__ __forceinline__ void add_runs(uint32_t name) const {
        const unsigned long long active = __ballot(1);
        const unsigned lane = threadIdx.x & 63u;
        const unsigned long long below = active & ((1ull << lane) - 1ull);
        const int prev = below ? 63 - __builtin_clzll(below) : (int)lane;
        const uint32_t pn = (uint32_t)__builtin_amdgcn_ds_bpermute(prev << 2, (int)name);
        const bool head = !below || pn != name;
        const unsigned long long heads = __ballot(head);
        if (head) {
            const unsigned long long above = lane == 63u ? 0ull : (heads >> (lane + 1u)) << (lane + 1u);
            const unsigned next = above ? (unsigned)__builtin_ctzll(above) : 64u;
            const unsigned long long upto = next == 64u ? ~0ull : ((1ull << next) - 1ull);
            add_n(name, (uint32_t)__builtin_popcountll(active & upto & ~((1ull << lane) - 1ull)));
        }
    }
    __device__ __forceinline__ void add(uint32_t name) const {
        if (aggregate)
            add_runs(name);
        else
            add_n(name, 1u);
    }
};

// The record kernel's add: nothing but the LDS (or, for names that do not fit, global) atomic.
// It hands back the replaced word; the caller checks all of a unit's adds for carries together
// (one wait for the LDS returns instead of one per add).
template <bool ALL_LDS>
struct FastSink {
    uint32_t* lds;
    uint32_t k_lds;
    unsigned long long* counts;
    // Names beyond the LDS histogram (an annotation with millions of names: BASELINE config 5) are not
    // counted with one memory-side 64-bit atomic per hit -- 145 M scattered atomics per 100 M-unit launch
    // ran at 3 % of the HBM roofline -- but appended to a spill list that is already split by BUCKET (the
    // top bits of the name id): this workgroup's slice of bucket b is spill[(b * n_wg + wg) * cap ...],
    // filled through cursor b in LDS.  After the launch one workgroup per bucket counts its slices in an
    // LDS histogram and adds it to the count vector with plain stores (spill.hip): no sort, no atomics
    // on memory, nothing for the host to wait for.  A full slice falls back to the global atomic.
    uint32_t* spill;         // nullptr: no spill list
    uint32_t spill_cap;      // entries per (bucket, workgroup) slice
    uint32_t spill_lo;       // bucket = name >> spill_lo
    uint32_t* spill_cursor;  // [n_buckets] in LDS
    __device__ __forceinline__ uint32_t add(uint32_t name) const {
        if (ALL_LDS || name < k_lds) return atomicAdd(&lds[name >> 1], 1u << ((name & 1u) * 16u));
        if (spill) {
            const uint32_t b = name >> spill_lo;
            const uint32_t pos = atomicAdd(&spill_cursor[b], 1u);
            if (pos < spill_cap) {
                spill[((size_t)b * gridDim.x + blockIdx.x) * spill_cap + pos] = name;
                return 0u;
            }
        }
        atomicAdd(&counts[name], 1ull);
        return 0u;  // never reads as a carry
    }
};

// Walks the hits of one unit.  Lists [a0,a1) / [b0,b1) are non-empty; ea / eb are their
// prefetched first entries.  COUNT: add 1 to every distinct hit name.  Returns the OR of
// (1 << class) over all hits (0 = no hit).
template <bool COUNT>
__device__ __forceinline__ uint32_t walk_hits(const tec_entry* __restrict__ ent, const Unit& u, uint32_t a0,
                                              uint32_t a1, uint32_t b0, uint32_t b1, const tec_entry ea,
                                              const tec_entry eb, const Sink& sink) {
    uint32_t classes = 0;
    uint32_t last = kNoName;
    const bool one_bin = a0 == b0;  // both edges fall into one fine bin
    auto first_edge = [&](const tec_entry e) {
        if ((test1(e, u) || (one_bin && test2(e, u))) && is_candidate(e, u)) {
            classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
            const uint32_t name = e.meta & kNameMask;
            if (COUNT && name != last) sink.add(name);
            last = name;
        }
    };
    first_edge(ea);
    for (uint32_t j = a0 + 1; j < a1; j += 2) {  // two gathers in flight per trip
        const bool two = j + 1 < a1;
        const tec_entry e0 = ent[j];
        const tec_entry e1 = ent[two ? j + 1 : j];
        first_edge(e0);
        if (two) first_edge(e1);
    }
    if (one_bin) return classes;

    last = kNoName;
    auto second_edge = [&](const tec_entry e) {
        if (test2(e, u) && is_candidate(e, u)) {
            classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
            const uint32_t name = e.meta & kNameMask;
            if (COUNT && name != last) {
                // already counted through the first edge?  (both lists are short and cached)
                auto same = [&](const tec_entry f) {
                    return (f.meta & kNameMask) == name && test1(f, u) && is_candidate(f, u);
                };
                bool seen = same(ea);
                for (uint32_t k = a0 + 1; k < a1 && !seen; ++k) seen = same(ent[k]);
                if (!seen) sink.add(name);
            }
            last = name;
        }
    };
    second_edge(eb);
    for (uint32_t j = b0 + 1; j < b1; j += 2) {
        const bool two = j + 1 < b1;
        const tec_entry e0 = ent[j];
        const tec_entry e1 = ent[two ? j + 1 : j];
        second_edge(e0);
        if (two) second_edge(e1);
    }
    return classes;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// End of a launch: the workgroup's histogram goes to its own row of the slab with plain 16-byte
// stores, and reduce_slabs_kernel sums the rows into the 64-bit count vector.  (256 workgroups
// flushing 41 k counters each with global atomics -- 10 M memory-side atomics on the same 41 k
// addresses -- took ~20 % of the launch.)  Every wave drains its own outstanding LDS atomics
// before the barrier with an explicit s_waitcnt: hipcc 7.2 was seen to omit it on one loop path of
// the U = 2 record kernel, and a ds_add still in flight when another lane reads the word is a
// lost update (caught by tools/stress_determinism.py as 30 missing counts in ~2 % of launches).
__device__ __forceinline__ void store_hist(const uint32_t* hist, uint32_t words4, uint32_t* slab_row) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(hist);
    uint4* dst = reinterpret_cast<uint4*>(slab_row);
    for (uint32_t q = threadIdx.x; q < words4 / 4u; q += kBlock) dst[q] = src[q];
}

// counts[2w] += sum over rows of the low halves of word w, counts[2w+1] += the high halves.
// The rows of up to kSlabSlots launches are folded at once (one reduce per kSlabSlots launches, or at
// the next sync): slot s holds used.n[s] rows of `stride` words at slabs + s * slot_rows * stride.
// blockIdx.y strides over the rows so that enough loads are in flight.
constexpr int kSlabSlots = 8;
// kernel flag: the index holds every chromosome twice, 2c = its '+' features and 2c + 1 = its '-' features, and a unit
// looks at the copy of its own strand (tec_bulk_set_stranded)
constexpr uint32_t kStrandedRun = 64u;
struct SlabRows {
    uint32_t n[kSlabSlots];
};
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const uint32_t* __restrict__ slabs, uint32_t stride,
                                                           uint32_t words, SlabRows used, uint32_t n_slots,
                                                           uint32_t slot_rows, unsigned long long* counts) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= words) return;
    unsigned long long lo = 0, hi = 0;  // <= 65 535 per row
    for (uint32_t s = 0; s < n_slots; ++s) {
        const uint32_t* __restrict__ base = slabs + (size_t)s * slot_rows * stride;
        const uint32_t rows = used.n[s];
#pragma unroll 8
        for (uint32_t g = blockIdx.y; g < rows; g += gridDim.y) {
            const uint32_t v = base[(size_t)g * stride + w];
            lo += v & 0xFFFFu;
            hi += v >> 16;
        }
    }
    if (lo) atomicAdd(&counts[2u * w], lo);
    if (hi) atomicAdd(&counts[2u * w + 1u], hi);
}

// MIXED = the index holds classes that are not countable in bulk (miRNA..., enhancer): the class
// decision then needs a first pass over the hits before anything may be counted.
// Persistent workgroups walk the batch in chunks of U * kBlock units (grid-stride over chunks).
// Each lane carries U units through four stages so that the three dependent memory levels
// (unit fields -> fine-bin offsets -> entries) of U units are in flight together:
//   stage 1  coalesced loads of the unit SoA                       (5 x U loads per lane)
//   stage 2  filters, fine-bin lookup: CSR offsets of both edges   (up to 4 x U gathers)
//   stage 3  prefetch the first entry of each list                 (up to 2 x U 16-byte gathers)
//   stage 4  walk the lists, count
template <bool MIXED, int U>
__global__ __launch_bounds__(kBlock) void count_bulk_kernel(tec_index_dev idx, const int32_t* __restrict__ chrom,
                                                           const int32_t* __restrict__ p1,
                                                           const int32_t* __restrict__ p2,
                                                           const uint16_t* __restrict__ flag,
                                                           const uint8_t* __restrict__ mapq, uint64_t n, int mapq_min,
                                                           unsigned long long* counts, unsigned long long* stats,
                                                           uint32_t k_lds, uint64_t span, uint32_t flags,
                                                           uint32_t* slabs, uint32_t* spill, uint32_t spill_cap,
                                                           uint32_t* spill_counts) {
    extern __shared__ uint32_t hist[];
    constexpr uint32_t chunk = U * kBlock;
    const uint32_t words = (((k_lds + 1u) >> 1) + 3u) & ~3u;  // whole 16-byte pieces
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) hist[w] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const Sink sink{hist, k_lds, counts, (flags & 1u) != 0};
    const tec_entry* __restrict__ ent = idx.entries;
    const uint32_t sbit = (flags & kStrandedRun) ? 1u : 0u;  // see tec_bulk_set_stranded
    unsigned n_assigned = 0, n_lowq = 0, n_bad = 0, n_qc = 0, n_enh = 0;
    // A workgroup owns spans of consecutive chunks, `gridDim.x` spans apart: at any moment the
    // workgroups are spread over the batch (for coordinate-sorted input: over the genome, so
    // that they do not all pull the same index lines through one L2 channel).
    for (uint64_t s0 = (uint64_t)blockIdx.x * span; s0 < n; s0 += (uint64_t)gridDim.x * span)
    for (uint64_t base = s0; base < n && base < s0 + span; base += chunk) {
        // ---- stage 1
        unsigned fl[U], mq[U];
        int ch[U];
        Unit un[U];
        bool live[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint64_t i = base + (uint64_t)k * kBlock + threadIdx.x;
            live[k] = i < n;
            const uint64_t ii = live[k] ? i : 0;
            // streamed once: non-temporal, so that the batch does not evict the index from L2 / MALL
            fl[k] = __builtin_nontemporal_load(&flag[ii]);
            mq[k] = __builtin_nontemporal_load(&mapq[ii]);
            ch[k] = __builtin_nontemporal_load(&chrom[ii]);
            un[k].p1 = __builtin_nontemporal_load(&p1[ii]);
            un[k].p2 = __builtin_nontemporal_load(&p2[ii]);
        }
        // ---- stage 2
        uint32_t a0[U], a1[U], b0[U], b1[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            a0[k] = a1[k] = b0[k] = b1[k] = 0;
            if (!live[k]) continue;
            if (fl[k] & (TEC_FLAG_UNMAPPED | TEC_FLAG_QCFAIL | TEC_FLAG_DUP)) {
                ++n_qc;
                live[k] = false;
            } else if ((int)mq[k] < mapq_min) {
                ++n_lowq;
                live[k] = false;
            } else if (ch[k] < 0 || (uint32_t)ch[k] >= (idx.n_chrom >> sbit)) {
                ++n_bad;
                live[k] = false;
            } else {
                ch[k] = (int)(((uint32_t)ch[k] << sbit) + (sbit & (fl[k] >> 4)));  // stranded index: chromosome 2c + reverse
                const uint32_t nb = idx.chrom_nbins[ch[k]], cb = idx.chrom_base[ch[k]];
                const uint32_t ba = (uint32_t)un[k].p1 >> idx.shift, bb = (uint32_t)un[k].p2 >> idx.shift;
                if (un[k].p1 >= 0 && ba < nb) {
                    a0[k] = idx.bin_off[cb + ba];
                    a1[k] = idx.bin_off[cb + ba + 1];
                }
                if (un[k].p2 >= 0 && bb < nb) {
                    b0[k] = idx.bin_off[cb + bb];
                    b1[k] = idx.bin_off[cb + bb + 1];
                }
            }
        }
        // ---- stage 3
        tec_entry ea[U], eb[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (!live[k]) continue;
            if (a0[k] == a1[k] && b0[k] == b1[k]) {
                live[k] = false;
                continue;
            }
            if (a0[k] == a1[k]) {  // only the second edge has a list: let it run as the "first" one
                a0[k] = b0[k];
                a1[k] = b1[k];
            } else if (b0[k] == b1[k]) {
                b0[k] = a0[k];
                b1[k] = a1[k];
            }
            ea[k] = ent[a0[k]];
            eb[k] = ea[k];
            if (b0[k] != a0[k]) eb[k] = ent[b0[k]];
            const unsigned q1 = (unsigned)un[k].p1 / (unsigned)TEC_BUCKET_SIZE;
            const unsigned q2 = ((unsigned)un[k].p2 + 1u) / (unsigned)TEC_BUCKET_SIZE;
            un[k].boundary = un[k].p1 < 0 || un[k].p2 < 0 || un[k].p2 == 0x7FFFFFFF ||
                             (unsigned)un[k].p1 == q1 * TEC_BUCKET_SIZE ||
                             (unsigned)un[k].p2 + 1u == q2 * TEC_BUCKET_SIZE;
        }
        // ---- stage 4
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (!live[k]) continue;
            if (MIXED) {
                const uint32_t classes = walk_hits<false>(ent, un[k], a0[k], a1[k], b0[k], b1[k], ea[k], eb[k], sink);
                if (classes == 0) continue;
                ++n_assigned;
                if (classes & kCountable)
                    walk_hits<true>(ent, un[k], a0[k], a1[k], b0[k], b1[k], ea[k], eb[k], sink);
                else if (classes & (1u << TEC_CLASS_ENHANCER))
                    ++n_enh;
            } else {
                if (walk_hits<true>(ent, un[k], a0[k], a1[k], b0[k], b1[k], ea[k], eb[k], sink)) ++n_assigned;
            }
        }
    }
    store_hist(hist, words, slabs + (size_t)blockIdx.x * words);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)n);
    // tallies: waves -> workgroup (LDS) -> five global adds per workgroup (see count_bulk_rec_kernel)
    __shared__ uint32_t tally[5];
    if (threadIdx.x < 5) tally[threadIdx.x] = 0;
    __syncthreads();
    n_assigned = wave_sum(n_assigned);
    n_lowq = wave_sum(n_lowq);
    n_bad = wave_sum(n_bad);
    n_qc = wave_sum(n_qc);
    n_enh = wave_sum(n_enh);
    if ((threadIdx.x & 63) == 0) {
        if (n_assigned) atomicAdd(&tally[0], n_assigned);
        if (n_lowq) atomicAdd(&tally[1], n_lowq);
        if (n_bad) atomicAdd(&tally[2], n_bad);
        if (n_qc) atomicAdd(&tally[3], n_qc);
        if (n_enh) atomicAdd(&tally[4], n_enh);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 5 && tally[threadIdx.x]) atomicAdd(&stats[1 + threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// Record kernel: one compact bin record per edge lookup (index.hip), hits collected in registers.

// Compact bin record (index.hip): {n, off, NE x {left16 | width16 << 16, meta}}; left and right = left + width are
// relative to (bin start - 1) and clamped to the bin, which is all the two edge tests need for a
// position inside the bin.  NE = 3 -> 32 bytes (two 16-byte gathers), NE = 7 -> 64 bytes.
template <int NE>
struct BinRec {
    uint32_t n, off;
    uint32_t lr[NE], meta[NE];
};

// The table is below 4 GiB (index.hip refuses to build records otherwise), so the address is the
// scalar base plus a 32-bit byte offset: no 64-bit vector address arithmetic.
template <int NE>
__device__ __forceinline__ BinRec<NE> load_rec(const uint4* __restrict__ recs, uint32_t g) {
    constexpr int Q = (NE + 1) / 2;  // 16-byte pieces per record
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(recs) + (uint32_t)(g * (uint32_t)(Q * 16)));
    uint4 q[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) q[i] = p[i];
    BinRec<NE> r;
    r.n = q[0].x;
    r.off = q[0].y;
    r.lr[0] = q[0].z;
    r.meta[0] = q[0].w;
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        r.lr[2 * i - 1] = q[i].x;
        r.meta[2 * i - 1] = q[i].y;
        r.lr[2 * i] = q[i].z;
        r.meta[2 * i] = q[i].w;
    }
    return r;
}

// The distinct hit names of one unit in six registers.  0.5 % of the hg38-like units hit more
// than four names -- a third of all wave trips would then take the exact walk with its chain of
// dependent gathers -- but only 0.004 % hit more than six; those are redone by the exact CSR walk
// (walk_hits).  Written without branches (selects only): on CDNA every divergent `if` costs
// scalar exec-mask bookkeeping, and the scalar unit is shared by the CU's four SIMDs --
// profiles/r01b showed more SALU than VALU instructions in this stage.
constexpr uint32_t kHitSlots = 6;
struct HitSet {
    uint32_t h0, h1, h2, h3, h4, h5, n, classes;  // n may exceed kHitSlots: then the unit goes to the exact walk
    // FULL: compare with all kept names (second record); else only with the most recent one:
    // entries are name-sorted inside a record, so while the first record is being scanned an
    // equal name can only be the previous accepted one.  New names are shifted in at h0.
    template <bool MIXED, bool FULL>
    __device__ __forceinline__ void add_if(bool hit, uint32_t meta) {
        const uint32_t name = meta & kNameMask;
        if (MIXED) classes |= hit ? 1u << ((meta >> TEC_META_CLASS_SHIFT) & 7u) : 0u;
        bool fresh = hit & (h0 != name);
        if (FULL) fresh = fresh & (h1 != name) & (h2 != name) & (h3 != name) & (h4 != name) & (h5 != name);
        h5 = fresh ? h4 : h5;
        h4 = fresh ? h3 : h4;
        h3 = fresh ? h2 : h3;
        h2 = fresh ? h1 : h2;
        h1 = fresh ? h0 : h1;
        h0 = fresh ? name : h0;
        n += fresh ? 1u : 0u;
    }
};

// continuation piece i (the table is below 4 GiB as well: 32-bit byte offset)
__device__ __forceinline__ uint4 load_piece(const uint4* __restrict__ cont, uint32_t i) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(cont) + (uint32_t)(i << 4));
}

// Edge tests of one list for a unit that is NOT on a 10 kb boundary (every hit is then a bucket
// candidate, see is_candidate).  x1 / x2m: p1 and p2 - 1 relative to (bin start - 1), or -1 when
// that edge does not lie in this record's bin.  Inline slots beyond the bin's count are zero and
// fail both tests by themselves.  Entries NE..NE+3 arrive as two prefetched continuation pieces
// (c0, c1: zero when the list is shorter); the branches around them are wave-uniform.  Lists
// longer than NE + 4 (0.1 % of the hg38-like lookups) walk the rest of their continuation run.
template <bool MIXED, bool FULL>
__device__ __forceinline__ void test_compact(uint32_t lw, uint32_t meta, int x1, int x2m, HitSet& hs) {
    // the entry as {left, width = right - left} (16 bits each): p lies in [left, right) iff (unsigned)(p - left) < width.
    // T1 is x1 in [left, right), T2 is x2 in (left, right] = x2 - 1 in [left, right); a test that is switched off comes
    // with the position -1, which no entry contains.  Two instructions a test and no scalar mask arithmetic (the form
    // with four compares needed five scalar operations a slot on the unit that the CU's four SIMDs share).
    const uint32_t l = lw & 0xFFFFu, w = lw >> 16;
    const bool t1 = (uint32_t)x1 - l < w;
    const bool t2 = (uint32_t)x2m - l < w;
    hs.add_if<MIXED, FULL>(t1 | t2, meta);
}

template <int NE, bool MIXED, bool FULL>
__device__ __forceinline__ void scan_record(const BinRec<NE>& r, const uint4 c0, const uint4 c1,
                                            const uint4* __restrict__ cont, int x1, int x2m, HitSet& hs) {
#pragma unroll
    for (int k = 0; k < NE; ++k) test_compact<MIXED, FULL>(r.lr[k], r.meta[k], x1, x2m, hs);
    if (__builtin_amdgcn_ballot_w64(r.n > (uint32_t)NE) == 0) return;
    test_compact<MIXED, FULL>(c0.x, c0.y, x1, x2m, hs);
    test_compact<MIXED, FULL>(c0.z, c0.w, x1, x2m, hs);
    if (__builtin_amdgcn_ballot_w64(r.n > (uint32_t)NE + 2u) == 0) return;
    test_compact<MIXED, FULL>(c1.x, c1.y, x1, x2m, hs);
    test_compact<MIXED, FULL>(c1.z, c1.w, x1, x2m, hs);
    if (r.n > (uint32_t)NE + 4u) {
        const uint32_t pieces = (r.n - (uint32_t)NE + 1u) >> 1;
        for (uint32_t j = 2; j < pieces; ++j) {
            const uint4 c = load_piece(cont, r.off + j);
            test_compact<MIXED, FULL>(c.x, c.y, x1, x2m, hs);
            test_compact<MIXED, FULL>(c.z, c.w, x1, x2m, hs);
        }
    }
}

template <bool MIXED, int U, int NE, bool ALL_LDS>
__global__ __launch_bounds__(kBlock, TEC_REC_WAVES) void count_bulk_rec_kernel(tec_index_dev idx, const int32_t* __restrict__ chrom,
                                                               const int32_t* __restrict__ p1,
                                                               const int32_t* __restrict__ p2,
                                                               const uint16_t* __restrict__ flag,
                                                               const uint8_t* __restrict__ mapq, uint64_t n,
                                                               int mapq_min, unsigned long long* counts,
                                                               unsigned long long* stats, uint32_t k_lds, uint64_t span,
                                                               uint32_t flags, uint32_t* slabs, uint32_t* spill,
                                                               uint32_t spill_cap, uint32_t* spill_counts) {
    // (spill_cap: low 24 bits = entries per slice, high 8 bits = log2 of the names per bucket)
    extern __shared__ uint32_t hist[];
    // per-chromosome {first fine bin, bin count} in LDS: one dependent global gather less per unit
    __shared__ uint2 chrom_tab[kChromLds];
    __shared__ uint32_t ticket;
    if (threadIdx.x == 0) ticket = 0;
    for (uint32_t c = threadIdx.x; c < kChromLds; c += kBlock)
        chrom_tab[c] = c < idx.n_chrom ? make_uint2(idx.chrom_base[c], idx.chrom_nbins[c]) : make_uint2(0u, 0u);
    const uint32_t words = (((k_lds + 1u) >> 1) + 3u) & ~3u;  // whole 16-byte pieces
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) hist[w] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const Sink sink{hist, k_lds, counts, (flags & 1u) != 0};  // exact path only
    // bucket cursors of the spill list sit behind the histogram in dynamic LDS
    const uint32_t spill_lo = spill_cap >> 24, slice_cap = spill_cap & 0xFFFFFFu;
    const uint32_t n_buckets = (!ALL_LDS && spill) ? ((idx.n_names + (1u << spill_lo) - 1u) >> spill_lo) : 0u;
    uint32_t* const cursors = hist + words;
    for (uint32_t b = threadIdx.x; b < n_buckets; b += kBlock) cursors[b] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const FastSink<ALL_LDS> fast{hist, k_lds, counts, (!ALL_LDS && spill) ? spill : nullptr, slice_cap, spill_lo, cursors};
#ifdef TEC_DIAGNOSTICS
    const bool mute = (flags & 2u) != 0;  // (diagnostic build: count nothing)
    unsigned sinkhole = 0;                // keeps ablated loads alive
#endif
    const tec_entry* __restrict__ ent = idx.entries;
    const uint32_t sbit = (flags & kStrandedRun) ? 1u : 0u;  // see tec_bulk_set_stranded
    unsigned n_assigned = 0, n_enh = 0;        // per lane (incremented in divergent code)
    unsigned w_lowq = 0, w_bad = 0, w_qc = 0;  // per wave (scalar)
    TEC_CLK_DECL;
    // A workgroup owns spans of consecutive units, `gridDim.x` spans apart (see count_bulk_kernel).
    // Inside the workgroup the waves DRAW their pieces of U x 64 consecutive units from a ticket
    // counter in LDS instead of owning a fixed slice of every chunk: with fixed slices the
    // first-launched (oldest, issue-favoured) waves finished ~18 % ahead and then idled at the
    // final barrier while the rest ran on with too few waves left to hide the gather latency.
    constexpr uint32_t piece = U * 64u;
    const uint32_t per_span = (uint32_t)(span / piece);  // span is a whole number of chunks
    const uint64_t n_spans = (n + span - 1) / span;
    const uint64_t my_spans = blockIdx.x < n_spans ? (n_spans - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t tickets = (uint32_t)(my_spans * per_span);  // the host keeps this below 2^32
    const unsigned lane = threadIdx.x & 63u;
    for (;;) {
        uint64_t base;
        uint32_t left;  // units of this piece
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&ticket, 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= tickets) break;
        const uint32_t sj = t / per_span, sr = t - sj * per_span;
        base = ((uint64_t)blockIdx.x + (uint64_t)sj * gridDim.x) * span + (uint64_t)sr * piece;
        if (base >= n) continue;  // the tail of the batch's last span
        const uint64_t left_u = n - base;
        left = left_u > (uint64_t)piece ? piece : (uint32_t)left_u;
        TEC_CLK(4);  // loop overhead
        Unit un[U];
        bool live[U];
        BinRec<NE> ra[U], rb[U];
        bool one[U];
        int xa1[U], xa2[U], xb2[U];
        int ch[U];
        // ---- stage 1: the unit SoA (streamed, non-temporal).  `base` is wave-uniform, so every
        // address is a scalar base plus a 32-bit lane offset (no 64-bit vector address math)
        unsigned fl[U], mq[U];
        const uint16_t* __restrict__ flag_b = flag + base;
        const uint8_t* __restrict__ mapq_b = mapq + base;
        const int32_t* __restrict__ chrom_b = chrom + base;
        const int32_t* __restrict__ p1_b = p1 + base;
        const int32_t* __restrict__ p2_b = p2 + base;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t i = (uint32_t)k * 64u + lane;
            live[k] = i < left;
            const uint32_t ii = live[k] ? i : 0u;
            fl[k] = __builtin_nontemporal_load(&flag_b[ii]);
            mq[k] = __builtin_nontemporal_load(&mapq_b[ii]);
            ch[k] = __builtin_nontemporal_load(&chrom_b[ii]);
            un[k].p1 = __builtin_nontemporal_load(&p1_b[ii]);
            un[k].p2 = __builtin_nontemporal_load(&p2_b[ii]);
        }
        // stage boundaries are scheduling barriers: hipcc otherwise sinks one unit's loads behind
        // the other unit's waits and the U units run one after the other
        __builtin_amdgcn_sched_barrier(0);
#ifdef TEC_DIAGNOSTICS
        if (flags & 8u) {
#pragma unroll
            for (int k = 0; k < U; ++k) sinkhole ^= fl[k] ^ mq[k] ^ (unsigned)ch[k] ^ (unsigned)un[k].p1 ^ (unsigned)un[k].p2;
            continue;
        }
#endif
        TEC_CLK(0);  // stream latency
        // ---- stage 2: filters, one bin record per edge (a single gather level).  Branch-free on
        // purpose: every lane loads -- a filtered unit or an edge outside the chromosome reads the
        // all-zero record behind the table -- so that the records of all U units go out back to
        // back.  (With `if (valid) load` hipcc sank the unpacking into the branches and waited
        // for each record before issuing the next: four serialized HBM round trips per trip.)
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const bool qc = live[k] && (fl[k] & (TEC_FLAG_UNMAPPED | TEC_FLAG_QCFAIL | TEC_FLAG_DUP)) != 0;
            const bool lowq = live[k] && !qc && (int)mq[k] < mapq_min;
            const bool chrom_ok = (uint32_t)ch[k] < (idx.n_chrom >> sbit);  // negative ids wrap to huge values
            // (stranded index: chromosome 2c + the unit's reverse bit; the exact walk below looks ch up again)
            ch[k] = chrom_ok ? (int)(((uint32_t)ch[k] << sbit) + (sbit & (fl[k] >> 4))) : ch[k];
            const bool bad = live[k] && !qc && !lowq && !chrom_ok;
            // wave-uniform tallies on the scalar unit (one s_bcnt1 each), lane 0 carries them out
            w_qc += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(qc));
            w_lowq += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lowq));
            w_bad += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(bad));
            live[k] = live[k] && !qc && !lowq && chrom_ok;
            const uint32_t chk = chrom_ok ? (uint32_t)ch[k] : 0u;
            // {first bin, bin count} of the chromosome from LDS (indexes with more than kChromLds
            // chromosomes run the CSR kernel: a global or FLAT lookup here would make the compiler
            // drain the outstanding record gathers of the other unit before this one's address)
            const uint2 t = chrom_tab[chk];
            const uint32_t cb = t.x, nb = t.y;
            const uint32_t ba = (uint32_t)un[k].p1 >> idx.shift, bb = (uint32_t)un[k].p2 >> idx.shift;
            const bool va = live[k] && un[k].p1 >= 0 && ba < nb, vb = live[k] && un[k].p2 >= 0 && bb < nb;
            one[k] = va && vb && ba == bb;  // both edges in one bin: its record answers both tests
            ra[k] = load_rec<NE>(idx.recs, va ? cb + ba : idx.zero_rec);
            rb[k] = load_rec<NE>(idx.recs, vb && !one[k] ? cb + bb : idx.zero_rec);
            xa1[k] = un[k].p1 - (int)(ba << idx.shift) + 1;
            xa2[k] = un[k].p2 - (int)(ba << idx.shift) + 1;
            xb2[k] = un[k].p2 - (int)(bb << idx.shift) + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef TEC_DIAGNOSTICS
        if (flags & 4u) {
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (live[k]) sinkhole ^= ra[k].n ^ rb[k].n ^ ra[k].meta[NE - 1] ^ rb[k].lr[NE - 1];
            continue;
        }
#endif
        TEC_CLK(1);  // record gathers
        // ---- stage 2b: entries NE..NE+3 of the longer lists, all in flight together (the
        // continuation table is a few MB: L2 / Infinity Cache hits)
        uint4 ca0[U], ca1[U], cb0[U], cb1[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {  // branch-free as above: shorter lists read the table's zero piece
            ca0[k] = load_piece(idx.cont, ra[k].n > (uint32_t)NE ? ra[k].off : idx.zero_piece);
            ca1[k] = load_piece(idx.cont, ra[k].n > (uint32_t)NE + 2u ? ra[k].off + 1u : idx.zero_piece);
            cb0[k] = load_piece(idx.cont, rb[k].n > (uint32_t)NE ? rb[k].off : idx.zero_piece);
            cb1[k] = load_piece(idx.cont, rb[k].n > (uint32_t)NE + 2u ? rb[k].off + 1u : idx.zero_piece);
        }
        __builtin_amdgcn_sched_barrier(0);
        TEC_CLK(2);  // continuation gathers
        // ---- stage 3: edge tests over the records, distinct names, count
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (!live[k] || (ra[k].n == 0 && rb[k].n == 0)) continue;
            Unit& u = un[k];
            const unsigned q1 = (unsigned)u.p1 / (unsigned)TEC_BUCKET_SIZE;
            const unsigned q2 = ((unsigned)u.p2 + 1u) / (unsigned)TEC_BUCKET_SIZE;
            u.boundary = u.p1 < 0 || u.p2 < 0 || u.p2 == 0x7FFFFFFF || (unsigned)u.p1 == q1 * TEC_BUCKET_SIZE ||
                         (unsigned)u.p2 + 1u == q2 * TEC_BUCKET_SIZE;
            HitSet hs{kNoName, kNoName, kNoName, kNoName, kNoName, kNoName, 0u, 0u};
            if (!u.boundary) {
                // (positions relative to bin start - 1; the second test of a record sees x2 - 1, a test that does not
                // apply to the record sees -1)
                scan_record<NE, MIXED, false>(ra[k], ca0[k], ca1[k], idx.cont, xa1[k], one[k] ? xa2[k] - 1 : -1, hs);
                scan_record<NE, MIXED, true>(rb[k], cb0[k], cb1[k], idx.cont, -1, xb2[k] - 1, hs);
            }
            // exact CSR walk for the rare rest: p1 or p2+1 on a 10 kb boundary (the reference's
            // bucket-candidate rule can bite there), or more than six distinct names
            if (u.boundary || hs.n > kHitSlots) {
                // (the records carry no CSR offsets: look the lists up again; ch is in range, and a
                // record with n > 0 means its edge lies inside the chromosome's bins)
                const uint32_t cbx = idx.chrom_base[ch[k]];
                const uint32_t ga = cbx + ((uint32_t)u.p1 >> idx.shift);
                const uint32_t gb = one[k] ? ga : cbx + ((uint32_t)u.p2 >> idx.shift);
                uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
                if (ra[k].n) { a0 = idx.bin_off[ga]; a1 = idx.bin_off[ga + 1]; }
                if (one[k]) { b0 = a0; b1 = a1; }
                else if (rb[k].n) { b0 = idx.bin_off[gb]; b1 = idx.bin_off[gb + 1]; }
                if (a0 == a1) { a0 = b0; a1 = b1; }
                if (b0 == b1) { b0 = a0; b1 = a1; }
                const tec_entry ea = ent[a0], eb = ent[b0];
                if (MIXED) {
                    const uint32_t classes = walk_hits<false>(ent, u, a0, a1, b0, b1, ea, eb, sink);
                    if (classes) ++n_assigned;
                    if (classes & kCountable)
                        walk_hits<true>(ent, u, a0, a1, b0, b1, ea, eb, sink);
                    else if (classes & (1u << TEC_CLASS_ENHANCER))
                        ++n_enh;
                } else {
                    if (walk_hits<true>(ent, u, a0, a1, b0, b1, ea, eb, sink)) ++n_assigned;
                }
                continue;
            }
            if (hs.n == 0 && (!MIXED || hs.classes == 0)) continue;
            ++n_assigned;
            if (MIXED && !(hs.classes & kCountable)) {
                if (hs.classes & (1u << TEC_CLASS_ENHANCER)) ++n_enh;
                continue;
            }
#ifdef TEC_DIAGNOSTICS
            if (mute) continue;
#endif
            // all adds of the unit first, then one look at the words they replaced
            uint32_t o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0;
            const uint32_t o0 = fast.add(hs.h0);
            if (hs.n > 1) {
                o1 = fast.add(hs.h1);
                if (hs.n > 2) {
                    o2 = fast.add(hs.h2);
                    if (hs.n > 3) {
                        o3 = fast.add(hs.h3);
                        if (hs.n > 4) {
                            o4 = fast.add(hs.h4);
                            if (hs.n > 5) o5 = fast.add(hs.h5);
                        }
                    }
                }
            }
            // absent adds left o = 0: never a carry.  A half that is about to carry (0xFFFF) has its
            // top bit set: one cheap test over all six words, the exact ones only behind it.
            if ((o0 | o1 | o2 | o3 | o4 | o5) & 0x80008000u) {
                if (wrapped(hs.h0, o0, 1u)) wrap_account(counts, hs.h0, o0);
                if (wrapped(hs.h1, o1, 1u)) wrap_account(counts, hs.h1, o1);
                if (wrapped(hs.h2, o2, 1u)) wrap_account(counts, hs.h2, o2);
                if (wrapped(hs.h3, o3, 1u)) wrap_account(counts, hs.h3, o3);
                if (wrapped(hs.h4, o4, 1u)) wrap_account(counts, hs.h4, o4);
                if (wrapped(hs.h5, o5, 1u)) wrap_account(counts, hs.h5, o5);
            }
        }
        TEC_CLK(3);  // tests + counting
    }
    store_hist(hist, words, slabs + (size_t)blockIdx.x * words);
    for (uint32_t b = threadIdx.x; b < n_buckets; b += kBlock)  // (store_hist ended with a barrier: every append is done)
        spill_counts[(size_t)b * gridDim.x + blockIdx.x] = cursors[b] < slice_cap ? cursors[b] : slice_cap;
    TEC_CLK(5);  // final flush
    TEC_CLK_STORE;
#ifdef TEC_DIAGNOSTICS
    if (sinkhole == 0xDEADBEEFu && n == 0) stats[7] = sinkhole;
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)n);
    // tallies: waves -> workgroup (LDS) -> five global adds per workgroup (4 096 waves adding to one
    // line of memory-side atomics would queue behind each other at the end of the launch)
    __shared__ uint32_t tally[5];
    if (threadIdx.x < 5) tally[threadIdx.x] = 0;
    __syncthreads();
    n_assigned = wave_sum(n_assigned);
    n_enh = wave_sum(n_enh);
    if ((threadIdx.x & 63) == 0) {
        if (n_assigned) atomicAdd(&tally[0], n_assigned);
        if (w_lowq) atomicAdd(&tally[1], w_lowq);
        if (w_bad) atomicAdd(&tally[2], w_bad);
        if (w_qc) atomicAdd(&tally[3], w_qc);
        if (n_enh) atomicAdd(&tally[4], n_enh);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 5 && tally[threadIdx.x]) atomicAdd(&stats[1 + threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// tec_bulk_count_noalloc: the one-shot variant for callers that cannot allocate.  Everything the launch needs
// beyond the caller's inputs and outputs -- the per-workgroup histogram rows, the count vector in internal name
// order (the carry bookkeeping of the LDS histogram adds to it) and the tallies -- lives in one file-scope device
// array.  Each GPU holds its own copy for the life of the process; its address is looked up once per context
// (hipGetSymbolAddress in tec_bulk_init) and only that pointer is ever handed to kernels.
constexpr uint32_t kNoAllocMaxNames = 49152;                       // every name must have an LDS counter
constexpr uint32_t kNoAllocMaxRows = 256;                          // workgroups of one launch
constexpr uint64_t kNoAllocMaxUnits = 1ull << 31;
constexpr size_t kNoAllocRowWords = kNoAllocMaxNames / 2;          // packed 16-bit counters
constexpr size_t kNoAllocRowsBytes = (size_t)kNoAllocMaxRows * kNoAllocRowWords * sizeof(uint32_t);  // 24 MiB
constexpr size_t kNoAllocCountsBytes = ((size_t)kNoAllocMaxNames + 2) * sizeof(unsigned long long);
constexpr size_t kNoAllocStatsOff = kNoAllocRowsBytes + ((kNoAllocCountsBytes + 255) & ~size_t(255));
constexpr size_t kNoAllocBytes = kNoAllocStatsOff + 256;
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

__global__ __launch_bounds__(256) void noalloc_clear_kernel(unsigned long long* counts, uint32_t n_counts, unsigned long long* stats) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_counts) counts[i] = 0ull;
    if (i < 8u) stats[i] = 0ull;
}

// internal (hotness) name order -> the caller's name ids; the five log counters and the enhancer-only tally
__global__ __launch_bounds__(256) void noalloc_export_kernel(const unsigned long long* __restrict__ counts, const uint32_t* __restrict__ int2ext,
                                                             uint32_t n_names, const unsigned long long* __restrict__ stats,
                                                             unsigned long long* out_counts, unsigned long long* out_stats) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_names) out_counts[int2ext[i]] = counts[i];
    if (i < 6u && out_stats) out_stats[i] = stats[i];
}

typedef void (*count_kernel_t)(tec_index_dev, const int32_t*, const int32_t*, const int32_t*, const uint16_t*,
                               const uint8_t*, uint64_t, int, unsigned long long*, unsigned long long*, uint32_t,
                               uint64_t, uint32_t, uint32_t*, uint32_t*, uint32_t, uint32_t*);

template <int U>
count_kernel_t pick_rec(bool mixed, bool all_lds) {
    if (all_lds) return mixed ? count_bulk_rec_kernel<true, U, 3, true> : count_bulk_rec_kernel<false, U, 3, true>;
    return mixed ? count_bulk_rec_kernel<true, U, 3, false> : count_bulk_rec_kernel<false, U, 3, false>;
}
template <int U>
count_kernel_t pick_csr(bool mixed) {
    return mixed ? count_bulk_kernel<true, U> : count_bulk_kernel<false, U>;
}
// rec_ne: inline entries per bin record (3), 0 = the CSR walk kernel; all_lds: every name has an
// LDS counter
count_kernel_t pick_kernel(bool mixed, int unroll, int rec_ne, bool all_lds) {
    if (rec_ne == 3) return unroll == 1 ? pick_rec<1>(mixed, all_lds) : pick_rec<2>(mixed, all_lds);
    switch (unroll) {
        case 1: return pick_csr<1>(mixed);
        case 4: return pick_csr<4>(mixed);
        default: return pick_csr<2>(mixed);
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
        if (ctx->launch_ms.size() < 65536) ctx->launch_ms.push_back(ms);
    }
    ctx->ev_used = 0;
    for (size_t i = 0; i < ctx->red_used; ++i) {
        float ms = 0.f;
        TEC_HIP(ctx, hipEventElapsedTime(&ms, ctx->red_pool[i].start, ctx->red_pool[i].stop));
        ctx->reduce_ms += ms;
    }
    ctx->red_used = 0;
    return tec_spill_harvest(ctx);
}

// Folds the histogram rows of the launches since the last reduce into the count vector.
int flush_slabs(tec_ctx* ctx) {
    if (ctx->slab_slot == 0) return TEC_OK;
    if (ctx->red_used == ctx->red_pool.size()) {
        tec_event_pair p;
        TEC_HIP(ctx, hipEventCreate(&p.start));
        TEC_HIP(ctx, hipEventCreate(&p.stop));
        p.reduced = nullptr;
        ctx->red_pool.push_back(p);
    }
    tec_event_pair& ev = ctx->red_pool[ctx->red_used++];
    SlabRows used{};
    uint32_t total_rows = 0;
    for (int k = 0; k < ctx->slab_slot; ++k) {
        used.n[k] = ctx->slab_used[k];
        total_rows += ctx->slab_used[k];
    }
    const uint32_t words = ctx->slab_words;
    const unsigned split = std::max(1u, std::min(total_rows / 64u, 32u));
    hipExtLaunchKernelGGL(reduce_slabs_kernel, dim3((words + 255u) / 256u, split), dim3(256), 0u, ctx->s_compute, ev.start,
                          ev.stop, 0u, (const uint32_t*)ctx->d_slabs, words, words, used, (uint32_t)ctx->slab_slot,
                          (uint32_t)ctx->n_cu, ctx->d_counts);
    TEC_HIP(ctx, hipGetLastError());
    ctx->slab_slot = 0;
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
            p.reduced = nullptr;
            ctx->ev_pool.push_back(p);
        }
    }
    const bool timed = ctx->time_every == 1 || (ctx->time_every > 1 && ctx->kernel_launches % (uint64_t)ctx->time_every == 0);
    tec_event_pair& ev = ctx->ev_pool[timed ? ctx->ev_used++ : ctx->ev_used];
    if (timed) ++ctx->timed_launches;
    const int unroll = ctx->unroll;
    const uint64_t chunk = (uint64_t)unroll * kBlock;
    const uint64_t want = (n + chunk - 1) / chunk;
    const bool mixed = (ctx->idx.class_mask & ~kCountable) != 0 || ctx->force_mixed;
    // LDS histogram: every name if that fits (up to 154 KiB); beyond that the coldest names spill
    // to global atomics.  One 1024-thread workgroup per CU: the kernels are compiled for four
    // waves per SIMD (~100 VGPRs), so a second workgroup would not be co-resident anyway.
    uint32_t budget = ctx->lds_bytes ? ctx->lds_bytes : kLdsMaxBytes;
    // names that do not fit go to a bucketed spill list (FastSink): its bucket cursors live behind the histogram
    unsigned name_bits = 1;
    while (name_bits < 32 && (ctx->idx.n_names >> name_bits)) ++name_bits;
    // (at most 128 buckets of at most 32 768 names: every bucket is a write stream of its own in the count kernel, whose
    // partly filled lines compete for the L2, and 128 KiB is the fold's LDS histogram)
    const uint32_t spill_lo = name_bits > 7 ? std::min(name_bits - 7, 15u) : 0u;
    const uint32_t n_buckets = (ctx->idx.n_names + (1u << spill_lo) - 1u) >> spill_lo;
    const bool may_spill = ctx->idx.n_names > budget / 2 && ctx->spill_enabled && n_buckets <= 1024 && ctx->kernel_variant == 2 &&
                           ctx->idx.n_chrom <= kChromLds && ctx->rec_ne == 3;
    if (may_spill) budget -= (n_buckets * 4u + 15u) & ~15u;
    const uint32_t k_lds = std::min<uint32_t>(ctx->idx.n_names, budget / 2);
    const uint32_t words = (((k_lds + 1u) >> 1) + 3u) & ~3u;  // as the kernels compute it
    const size_t lds = (size_t)words * 4 + (may_spill ? (size_t)n_buckets * 4 : 0);
    const unsigned grid = (unsigned)std::min<uint64_t>(want, (uint64_t)ctx->n_cu);
    // one histogram row per workgroup; reduce_slabs_kernel folds them into the count vector
    if (ctx->slab_slot && ctx->slab_words != words) {  // the histogram width changed: fold what is pending first
        int rc = flush_slabs(ctx);
        if (rc) return rc;
    }
    ctx->slab_words = words;
    const size_t slab_bytes = (size_t)kSlabSlots * ctx->n_cu * words * sizeof(uint32_t);
    if (slab_bytes > ctx->slab_cap) {
        int rc = flush_slabs(ctx);
        if (rc) return rc;
        TEC_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
        if (ctx->d_slabs) TEC_HIP(ctx, hipFree(ctx->d_slabs));
        ctx->d_slabs = nullptr;
        ctx->slab_cap = 0;
        TEC_HIP(ctx, hipMalloc(&ctx->d_slabs, slab_bytes));
        ctx->slab_cap = slab_bytes;
    }
    // spans: ~span_div per workgroup, each a whole number of chunks
    uint64_t span = (n / ((uint64_t)grid * ctx->span_div) + chunk - 1) / chunk * chunk;
    if (span < chunk) span = chunk;
    const bool rec_ok = ctx->kernel_variant == 2 && ctx->idx.n_chrom <= kChromLds && ctx->rec_ne == 3;
    unsigned rows = grid;  // histogram rows of this launch
    uint32_t* const slab = ctx->d_slabs + (size_t)ctx->slab_slot * ctx->n_cu * words;
    // start / stop timestamps ride on the dispatch itself: no barrier packets between the launches
    // names beyond the LDS histogram go to a bucketed spill list (see FastSink), folded right after the launch
    const bool spilling = may_spill && rec_ok && k_lds < ctx->idx.n_names;
    uint32_t spill_cap = 0;
    if (spilling) {
        // room for three names per unit, spread over the buckets (a unit adds at most six on the fast path,
        // config 5 averages 1.45); a slice that fills up falls back to global atomics
        const uint64_t per_wg = (n + grid - 1) / grid;
        spill_cap = (uint32_t)std::min<uint64_t>((per_wg * 3 + n_buckets - 1) / n_buckets + 64, 0xFFFFFFull) & ~3u;  // (the fold reads 16 bytes a lane)
        int rc = tec_spill_reserve(ctx, grid, n_buckets, spill_cap);
        if (rc) return rc;
    }
    hipExtLaunchKernelGGL(pick_kernel(mixed, unroll, rec_ok ? ctx->rec_ne : 0, k_lds >= ctx->idx.n_names), dim3(grid),
                          dim3(kBlock), (uint32_t)lds, ctx->s_compute, timed ? ev.start : (hipEvent_t) nullptr,
                          timed ? ev.stop : (hipEvent_t) nullptr, 0u, ctx->idx, u.chrom, u.p1, u.p2,
                          u.flag, u.mapq, n, ctx->mapq_min, ctx->d_counts, ctx->d_stats, k_lds, span,
                          (uint32_t)ctx->aggregate | (ctx->stranded ? kStrandedRun : 0u), slab, spilling ? ctx->d_spill : (uint32_t*)nullptr,
                          spill_cap | (spill_lo << 24), ctx->d_spill_counts);
    TEC_HIP(ctx, hipGetLastError());
    if (spilling) {
        int rc = tec_spill_fold(ctx, grid, n_buckets, spill_lo, spill_cap);
        if (rc) return rc;
    }
    // the rows wait in their slot; one reduce folds kSlabSlots launches (or what is pending at the next sync)
    ctx->slab_used[ctx->slab_slot++] = rows;
    if (ctx->slab_slot == kSlabSlots) {
        int rc = flush_slabs(ctx);
        if (rc) return rc;
    }
    ++ctx->kernel_launches;
    return TEC_OK;
}

}  // namespace

int tec_bulk_init(tec_ctx* ctx) {
    // more than 64 KiB of dynamic LDS per workgroup needs the opt-in
    for (int mixed = 0; mixed < 2; ++mixed)
        for (int u : {1, 2, 4})
            for (int rec : {0, 3})
                for (int all = 0; all < 2; ++all)
                    TEC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(pick_kernel(mixed != 0, u, rec, all != 0)),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)(kChromLds * sizeof(uint2)) - 64));
    {   // this GPU's copy of the no-allocation scratch array: the device address, looked up with the device current
        void* p = nullptr;
        TEC_HIP(ctx, hipSetDevice(ctx->device));
        TEC_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        ctx->noalloc_scratch = static_cast<unsigned char*>(p);
    }
    const char* e = getenv("TEC_UNROLL");  // tuning knob: units carried per lane (1, 2, 4 or 8)
    const int u = e ? atoi(e) : 2;
    ctx->unroll = (u == 1 || u == 2 || u == 4) ? u : 2;
    e = getenv("TEC_FORCE_MIXED");  // diagnostics: run the mixed-class kernel on any index
    ctx->force_mixed = e && atoi(e) != 0;
    e = getenv("TEC_KERNEL");  // 1: CSR walk kernel, 2: bin-record kernel
    ctx->kernel_variant = e && atoi(e) == 1 ? 1 : 2;
    ctx->rec_ne_wanted = 3;  // inline entries per bin record (32-byte records)
    e = getenv("TEC_AGG");  // tuning knob: fold runs of equal names across lanes (sorted input)
    ctx->aggregate = e ? atoi(e) : 0;
    ctx->lds_bytes = 0u;  // LDS histogram bytes per workgroup: automatic
    ctx->time_every = 1;
    e = getenv("TEC_SPILL");  // 0: names beyond the LDS histogram are counted with global atomics (the round-1 path)
    ctx->spill_enabled = !(e && atoi(e) == 0);
    ctx->span_div = 8;  // spans per workgroup
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
    for (auto& p : ctx->red_pool) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    ctx->red_pool.clear();
    ctx->red_used = 0;
    ctx->slab_slot = 0;
    if (ctx->d_slabs) (void)hipFree(ctx->d_slabs);
    ctx->d_slabs = nullptr;
    ctx->slab_cap = 0;
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
    ctx->stranded = false;
    ctx->slab_slot = 0;
    ctx->kernel_ms = 0.0;
    ctx->reduce_ms = 0.0;
    ctx->kernel_launches = 0;
    ctx->timed_launches = 0;
    ctx->launch_ms.clear();
    ctx->spill_ms = 0.0;
    ctx->spill_names = 0;
    TEC_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, ((size_t)ctx->idx.n_names + 2) * sizeof(unsigned long long), ctx->s_compute));
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

// One batch, counted from start to finish, without a single allocation: see include/tecount.h.
extern "C" int tec_bulk_count_noalloc(tec_ctx* ctx, uint64_t n, const int32_t* d_chrom, const int32_t* d_p1, const int32_t* d_p2,
                                      const uint16_t* d_flag, const uint8_t* d_mapq, int mapq_min, uint64_t* d_counts_out,
                                      uint64_t* d_stats_out, void* stream) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_count_noalloc: NULL context");
    if (!ctx->have_index) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_count_noalloc: no index resident (call tec_index_build)");
    if (!ctx->noalloc_scratch) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_count_noalloc: the scratch array was not resolved");
    if (!d_counts_out) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_count_noalloc: NULL output");
    if (n && (!d_chrom || !d_p1 || !d_p2 || !d_flag || !d_mapq)) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_count_noalloc: NULL array");
    if (ctx->idx.n_names > kNoAllocMaxNames)
        return tec_fail(ctx, TEC_EOVERFLOW, "tec_bulk_count_noalloc: the index has %u names, the fixed scratch array holds counters for %u",
                        ctx->idx.n_names, kNoAllocMaxNames);
    if (n > kNoAllocMaxUnits)
        return tec_fail(ctx, TEC_EOVERFLOW, "tec_bulk_count_noalloc: %llu units in one call, at most %llu", (unsigned long long)n,
                        (unsigned long long)kNoAllocMaxUnits);
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->s_compute;
    uint32_t* const rows = reinterpret_cast<uint32_t*>(ctx->noalloc_scratch);
    unsigned long long* const counts = reinterpret_cast<unsigned long long*>(ctx->noalloc_scratch + kNoAllocRowsBytes);
    unsigned long long* const stats = reinterpret_cast<unsigned long long*>(ctx->noalloc_scratch + kNoAllocStatsOff);
    const uint32_t n_names = ctx->idx.n_names;
    // the array keeps the last call's contents: what must start at zero is cleared by a small kernel (no memset node, so
    // that a caller may capture the call)
    hipLaunchKernelGGL(noalloc_clear_kernel, dim3((n_names + 2u + 255u) / 256u), dim3(256), 0, st, counts, n_names + 2u, stats);
    if (n) {
        const int unroll = ctx->unroll;
        const uint64_t chunk = (uint64_t)unroll * kBlock;
        const uint64_t want = (n + chunk - 1) / chunk;
        const bool mixed = (ctx->idx.class_mask & ~kCountable) != 0 || ctx->force_mixed;
        const uint32_t k_lds = n_names;  // (<= kNoAllocMaxNames: every name has its LDS counter, nothing spills)
        const uint32_t words = (((k_lds + 1u) >> 1) + 3u) & ~3u;
        const unsigned grid = (unsigned)std::min<uint64_t>(want, std::min<uint64_t>((uint64_t)ctx->n_cu, kNoAllocMaxRows));
        uint64_t span = (n / ((uint64_t)grid * ctx->span_div) + chunk - 1) / chunk * chunk;
        if (span < chunk) span = chunk;
        const bool rec_ok = ctx->kernel_variant == 2 && ctx->idx.n_chrom <= kChromLds && ctx->rec_ne == 3;
        hipLaunchKernelGGL(pick_kernel(mixed, unroll, rec_ok ? ctx->rec_ne : 0, true), dim3(grid), dim3(kBlock), (uint32_t)(words * 4u), st,
                           ctx->idx, d_chrom, d_p1, d_p2, d_flag, d_mapq, n, mapq_min, counts, stats, k_lds, span,
                           (uint32_t)ctx->aggregate, rows, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr);
        TEC_HIP(ctx, hipGetLastError());
        SlabRows used{};
        used.n[0] = grid;
        const unsigned split = std::max(1u, std::min(grid / 64u, 32u));
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((words + 255u) / 256u, split), dim3(256), 0, st, (const uint32_t*)rows, words, words,
                           used, 1u, (uint32_t)kNoAllocMaxRows, counts);
    }
    hipLaunchKernelGGL(noalloc_export_kernel, dim3((std::max(n_names, 6u) + 255u) / 256u), dim3(256), 0, st, (const unsigned long long*)counts,
                       (const uint32_t*)ctx->d_int2ext, n_names, (const unsigned long long*)stats,
                       reinterpret_cast<unsigned long long*>(d_counts_out), reinterpret_cast<unsigned long long*>(d_stats_out));
    TEC_HIP(ctx, hipGetLastError());
    return TEC_OK;
}

extern "C" int tec_bulk_set_stranded(tec_ctx* ctx, int on) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_set_stranded: NULL context");
    if (!ctx->bulk_open) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_set_stranded: no bulk run is open");
    if (ctx->kernel_launches) return tec_fail(ctx, TEC_ESTATE, "tec_bulk_set_stranded: units were already submitted to this run");
    if (on && (ctx->idx.n_chrom & 1u))
        return tec_fail(ctx, TEC_EINVAL, "tec_bulk_set_stranded: the index has %u chromosomes: a strand-split index has two per chromosome",
                        ctx->idx.n_chrom);
    ctx->stranded = on != 0;
    return TEC_OK;
}

extern "C" int tec_bulk_sync(tec_ctx* ctx) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_sync: NULL context");
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    {
        // the last launches' histogram rows may still wait for their reduce (it must follow the
        // copy stream's batches too: they were launched on s_compute behind their copy events)
        int rc = flush_slabs(ctx);
        if (rc) return rc;
    }
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
    if (counts) {  // the device vector is in internal (hotness) order: permuted on the device, into the allocation's second half
        const uint32_t n_names = ctx->idx.n_names;
        unsigned long long* const ordered = ctx->d_counts + n_names + 8;
        hipLaunchKernelGGL(noalloc_export_kernel, dim3((n_names + 255u) / 256u), dim3(256), 0, ctx->s_compute,
                           (const unsigned long long*)ctx->d_counts, (const uint32_t*)ctx->d_int2ext, n_names,
                           (const unsigned long long*)ctx->d_stats, ordered, (unsigned long long*)nullptr);
        TEC_HIP(ctx, hipGetLastError());
        TEC_HIP(ctx, hipMemcpyAsync(counts, ordered, (size_t)n_names * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->s_compute));
        TEC_HIP(ctx, hipStreamSynchronize(ctx->s_compute));
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

#ifdef TEC_DIAGNOSTICS
// diagnostics: read and clear the per-stage cycle sums (stream, records, continuation, tests+count,
// loop+flushes, final flush) accumulated by wave 0 of every workgroup
extern "C" int tec_diag_stage_clocks(tec_ctx* ctx, unsigned long long* out) {
    TEC_HIP(ctx, hipDeviceSynchronize());
    TEC_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stage_clk), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {0};
    TEC_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_stage_clk), zero, sizeof zero));
    return TEC_OK;
}
#endif

extern "C" int tec_bulk_kernel_time(tec_ctx* ctx, double* total_ms, uint64_t* launches) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_kernel_time: NULL context");
    int rc = tec_bulk_sync(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->kernel_ms;
    if (launches) *launches = ctx->timed_launches;  // the launches that carried timestamps (all of them by default)
    return TEC_OK;
}

extern "C" int tec_bulk_launch_times(tec_ctx* ctx, float* ms, uint64_t capacity, uint64_t* n) {
    if (!ctx || !n) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_launch_times: NULL argument");
    int rc = tec_bulk_sync(ctx);
    if (rc) return rc;
    *n = ctx->launch_ms.size();
    if (ms)
        for (uint64_t i = 0; i < capacity && i < *n; ++i) ms[i] = ctx->launch_ms[i];
    return TEC_OK;
}

extern "C" int tec_bulk_spill_time(tec_ctx* ctx, double* total_ms, uint64_t* names) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_spill_time: NULL context");
    int rc = tec_bulk_sync(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->spill_ms;
    if (names) *names = ctx->spill_names;
    return TEC_OK;
}

extern "C" int tec_bulk_reduce_time(tec_ctx* ctx, double* total_ms) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_bulk_reduce_time: NULL context");
    int rc = tec_bulk_sync(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->reduce_ms;
    return TEC_OK;
}
