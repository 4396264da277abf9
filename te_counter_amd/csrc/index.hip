// Host-side build of the flat HBM interval index and its upload.
//
// Replaces genelist._optimiseData (te_count/miniglbase/genelist.py:332-396), which files every
// feature under each 10 kb bucket it touches, and the `loc_lookups` table the reference loops
// build per run (te_count/te_count.py:70-73, 195-198, 588-590).
//
// Layout (all chromosomes concatenated, one blob in HBM):
//   bin_off[n_bins + 1]   u32 CSR offsets; a chromosome's fine bins are contiguous
//   entries[n_entries]    {left, right, meta, pad} 16 B, sorted by (name id, strand) inside a fine bin
//   chrom_base[n_chrom], chrom_nbins[n_chrom]
//   recs[n_bins]          compact bin records for the bulk kernel: {n = entries in the bin,
//                         coff, the bin's first NE entries as {left16 | width16<<16, meta}} with
//                         coordinates relative to (bin start - 1) and clamped to the bin -- 32
//                         bytes for NE = 3.  One record answers an edge lookup for every bin with
//                         at most NE entries (one dependent gather level instead of two).
//   cont[]                continuation of the longer lists in the same compact 8-byte form, two
//                         entries per 16-byte piece, a bin's run starting at piece `coff` (zero
//                         padded to a whole piece).  Only ~8 % of the hg38-like bins have one, so
//                         the table is a few MB and stays in L2 / Infinity Cache: a lane fetches
//                         entries NE..NE+3 of both edges in ONE further round trip, all loads in
//                         flight together, instead of walking the 16-byte CSR entries one dependent
//                         HBM gather at a time.
// A feature [left, right] is filed under every fine bin (2^shift bp) that intersects
// [left-1, right+1]: any read edge that can pass one of the reference's 2-bp edge tests
// (te_count.py:122,125 / 237,240 / 645,648) then finds the feature in the ONE fine bin of that
// edge position.  Whether the reference would have looked at the feature at all (its 10 kb
// bucket-candidate rule) is decided on the device from left/right, see bulk.hip.
#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tecount_internal.h"

namespace {

constexpr int kMinShift = 7;
constexpr int kMaxShift = 20;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Coverage of a feature in bp, or false if no edge test can ever pass for it.
inline bool coverage(int32_t gl, int32_t gr, int64_t* lo, int64_t* hi) {
    if ((int64_t)gr < (int64_t)gl - 1) return false;
    *lo = gl > 0 ? (int64_t)gl - 1 : 0;
    *hi = (int64_t)gr + 1;
    return *hi >= 0;
}

}  // namespace

// A buffer whose pages are first touched by the threads that fill it (std::vector would zero it serially).
template <class T>
struct RawBuf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    explicit RawBuf(size_t n_) : p(new T[n_ ? n_ : 1]), n(n_) {}
    T* data() { return p.get(); }
    T* begin() { return p.get(); }
    T& operator[](size_t i) { return p[i]; }
    size_t size() const { return n; }
    void swap(RawBuf& o) {
        p.swap(o.p);
        std::swap(n, o.n);
    }
};

// Host threads for the build (the per-bin work is independent): TEC_INDEX_THREADS, default min(16, cores).
static int build_threads() {
    if (const char* e = getenv("TEC_INDEX_THREADS")) {
        const int t = atoi(e);
        if (t >= 1 && t <= 256) return t;
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc ? hc : 1u));
}
// fn(begin, end, t) over T contiguous slices of [0, n)
template <class F>
static void parallel_slices(int T, uint64_t n, F fn) {
    if (T <= 1 || n < 4096) {
        fn((uint64_t)0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(n * t / T, n * (t + 1) / T, t); });
    fn((uint64_t)0, n / T, 0);
    for (auto& x : th) x.join();
}

extern "C" int tec_index_build(tec_ctx* ctx, uint64_t n, const int32_t* chrom_id, const int32_t* left,
                               const int32_t* right, const uint32_t* meta, uint32_t n_chrom, uint32_t n_names,
                               int bin_shift) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_index_build: NULL context");
    if (ctx->bulk_open) return tec_fail(ctx, TEC_ESTATE, "tec_index_build: a bulk run is open");
    if (n && (!chrom_id || !left || !right || !meta)) return tec_fail(ctx, TEC_EINVAL, "tec_index_build: NULL array");
    if (n_chrom == 0 || n_names == 0) return tec_fail(ctx, TEC_EINVAL, "tec_index_build: empty chromosome or name table");
    if (n_names > (1u << TEC_META_NAME_BITS))
        return tec_fail(ctx, TEC_EINVAL, "tec_index_build: %u names exceed the %d-bit name id", n_names, TEC_META_NAME_BITS);
    if (bin_shift != 0 && (bin_shift < kMinShift || bin_shift > kMaxShift))
        return tec_fail(ctx, TEC_EINVAL, "tec_index_build: bin_shift %d outside [%d, %d]", bin_shift, kMinShift, kMaxShift);

    auto mark = [&](const char*) {};  // (phase boundaries of the build; a timing build prints them)
    // pass 1: validate, per-chromosome extent, entry totals for every candidate shift
    std::vector<int64_t> chrom_hi(n_chrom, -1);
    uint64_t entries_at[kMaxShift + 1] = {0};
    uint64_t hittable = 0;
    uint32_t class_mask = 0;
    std::vector<uint64_t> weight(n_names, 0);  // bp covered per name ~ expected hit rate
    for (uint64_t i = 0; i < n; ++i) {
        if (chrom_id[i] < 0 || (uint32_t)chrom_id[i] >= n_chrom)
            return tec_fail(ctx, TEC_EINVAL, "tec_index_build: feature %llu has chrom id %d", (unsigned long long)i, chrom_id[i]);
        if (left[i] < 0) return tec_fail(ctx, TEC_EINVAL, "tec_index_build: feature %llu has a negative left", (unsigned long long)i);
        if ((meta[i] & ((1u << TEC_META_NAME_BITS) - 1)) >= n_names)
            return tec_fail(ctx, TEC_EINVAL, "tec_index_build: feature %llu has name id >= n_names", (unsigned long long)i);
        class_mask |= 1u << ((meta[i] >> TEC_META_CLASS_SHIFT) & 7u);
        int64_t lo, hi;
        if (!coverage(left[i], right[i], &lo, &hi)) continue;
        ++hittable;
        weight[meta[i] & ((1u << TEC_META_NAME_BITS) - 1)] += (uint64_t)(hi - lo);
        chrom_hi[chrom_id[i]] = std::max(chrom_hi[chrom_id[i]], hi);
        for (int s = kMinShift; s <= kMaxShift; ++s) entries_at[s] += (uint64_t)((hi >> s) - (lo >> s) + 1);
    }
    mark("pass 1 (validate, extents)");
    auto bins_at = [&](int s) {
        uint64_t b = 0;
        for (uint32_t c = 0; c < n_chrom; ++c)
            if (chrom_hi[c] >= 0) b += (uint64_t)(chrom_hi[c] >> s) + 1;
        return b;
    };
    int shift = bin_shift;
    if (shift == 0) {
        // smallest bins whose replication stays moderate (an interval sits in 2.25 bins on average at most) and whose
        // offset table stays small.  Measured on 3 M intervals of ~1 kb: 1 024-bp bins (2.1 copies) 2.59 ms per 100 M
        // units, 512 bp (3.2 copies) 2.69 ms, 2 048 bp (1.5 copies, more entries per record) 2.83 ms
        shift = kMaxShift;
        for (int s = 9; s <= kMaxShift; ++s) {
            if (entries_at[s] <= hittable * 2 + hittable / 4 + 1024 && bins_at(s) <= (48ull << 20)) {
                shift = s;
                break;
            }
        }
    }
    const uint64_t n_entries = entries_at[shift];
    const uint64_t n_bins = bins_at(shift);
    if (n_entries >= 0xFFFFFFF0ull || n_bins >= 0xFFFFFFF0ull)
        return tec_fail(ctx, TEC_EOVERFLOW, "tec_index_build: %llu entries / %llu bins exceed 32-bit offsets",
                        (unsigned long long)n_entries, (unsigned long long)n_bins);

    std::vector<uint32_t> chrom_base(n_chrom, 0), chrom_nbins(n_chrom, 0);
    {
        uint64_t b = 0;
        for (uint32_t c = 0; c < n_chrom; ++c) {
            chrom_base[c] = (uint32_t)b;
            chrom_nbins[c] = chrom_hi[c] >= 0 ? (uint32_t)((chrom_hi[c] >> shift) + 1) : 0;
            b += chrom_nbins[c];
        }
    }

    // Internal name ids: hottest first, so that the names the kernel keeps in its LDS histogram
    // (ids < k_lds, bulk.hip) are the ones that would otherwise serialise on global atomics.
    std::vector<uint32_t> int2ext(n_names), ext2int(n_names);
    for (uint32_t i = 0; i < n_names; ++i) int2ext[i] = i;
    std::stable_sort(int2ext.begin(), int2ext.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
    for (uint32_t i = 0; i < n_names; ++i) ext2int[int2ext[i]] = i;
    const uint32_t name_mask = (1u << TEC_META_NAME_BITS) - 1;

    mark("name order");
    // pass 2: counting sort of (feature, fine bin) pairs
    const int T = build_threads();
    std::vector<uint32_t> bin_off(n_bins + 1, 0);
    parallel_slices(T, n, [&](uint64_t i0, uint64_t i1, int) {
        for (uint64_t i = i0; i < i1; ++i) {
            int64_t lo, hi;
            if (!coverage(left[i], right[i], &lo, &hi)) continue;
            const uint32_t base = chrom_base[chrom_id[i]];
            for (int64_t b = lo >> shift; b <= (hi >> shift); ++b) __atomic_fetch_add(&bin_off[base + b + 1], 1u, __ATOMIC_RELAXED);
        }
    });
    for (uint64_t b = 0; b < n_bins; ++b) bin_off[b + 1] += bin_off[b];
    RawBuf<tec_entry> entries(n_entries);
    {
        // (the order in which threads fill a bin does not matter: the bin is sorted below by a total order)
        std::vector<uint32_t> cursor(bin_off.begin(), bin_off.end() - 1);
        parallel_slices(T, n, [&](uint64_t i0, uint64_t i1, int) {
            for (uint64_t i = i0; i < i1; ++i) {
                int64_t lo, hi;
                if (!coverage(left[i], right[i], &lo, &hi)) continue;
                const uint32_t base = chrom_base[chrom_id[i]];
                const tec_entry e{left[i], right[i], (meta[i] & ~name_mask) | ext2int[meta[i] & name_mask], 0u};
                for (int64_t b = lo >> shift; b <= (hi >> shift); ++b)
                    entries[__atomic_fetch_add(&cursor[base + b], 1u, __ATOMIC_RELAXED)] = e;
            }
        });
    }
    parallel_slices(T, n_bins, [&](uint64_t b0, uint64_t b1, int) {
        for (uint64_t b = b0; b < b1; ++b) {
            const uint32_t a = bin_off[b], z = bin_off[b + 1];
            if (z - a > 1)
                std::sort(entries.begin() + a, entries.begin() + z, [&](const tec_entry& x, const tec_entry& y) {
                    const uint32_t nx = x.meta & name_mask, ny = y.meta & name_mask;
                    if (nx != ny) return nx < ny;
                    const uint32_t sx = x.meta >> TEC_META_STRAND_SHIFT, sy = y.meta >> TEC_META_STRAND_SHIFT;
                    if (sx != sy) return sx < sy;  // (name, strand) pairs adjacent: single-cell dedupe
                    if (x.left != y.left) return x.left < y.left;
                    if (x.right != y.right) return x.right < y.right;
                    return x.meta < y.meta;
                });
        }
    });

    mark("counting sort + bin sorts");
    // Exact duplicates inside a bin (the same exon listed once per transcript) change no hit
    // set -- hits are sets of names / (name, strand) pairs and class bits are OR-ed -- drop them.
    {
        auto same = [](const tec_entry& q, const tec_entry& r) { return q.left == r.left && q.right == r.right && q.meta == r.meta; };
        std::vector<uint32_t> new_off(n_bins + 1, 0);
        parallel_slices(T, n_bins, [&](uint64_t b0, uint64_t b1, int) {  // survivors per bin
            for (uint64_t b = b0; b < b1; ++b) {
                uint32_t k = 0;
                for (uint32_t i = bin_off[b]; i < bin_off[b + 1]; ++i) k += (i == bin_off[b] || !same(entries[i - 1], entries[i])) ? 1u : 0u;
                new_off[b + 1] = k;
            }
        });
        for (uint64_t b = 0; b < n_bins; ++b) new_off[b + 1] += new_off[b];
        RawBuf<tec_entry> kept(new_off[n_bins]);
        parallel_slices(T, n_bins, [&](uint64_t b0, uint64_t b1, int) {
            for (uint64_t b = b0; b < b1; ++b) {
                uint32_t w = new_off[b];
                for (uint32_t i = bin_off[b]; i < bin_off[b + 1]; ++i)
                    if (i == bin_off[b] || !same(entries[i - 1], entries[i])) kept[w++] = entries[i];
            }
        });
        bin_off.swap(new_off);
        entries.swap(kept);
    }
    const uint64_t n_entries_kept = entries.size();

    mark("duplicate removal");
    // upload as one blob
    TEC_HIP(ctx, hipSetDevice(ctx->device));
    TEC_HIP(ctx, hipDeviceSynchronize());
    const size_t o_off = 0;
    const size_t o_ent = align_up(o_off + (n_bins + 1) * sizeof(uint32_t), 256);
    const size_t o_cb = align_up(o_ent + std::max<uint64_t>(n_entries_kept, 1) * sizeof(tec_entry), 256);
    const size_t o_cn = align_up(o_cb + n_chrom * sizeof(uint32_t), 256);
    // compact bin records for the bulk kernel (coordinates relative to bin start - 1, 16 bits)
    // records only while their table stays below 4 GiB (the kernel addresses it with 32-bit byte offsets)
    const int rec_ne = shift <= 15 && n_bins < (1ull << 26) ? ctx->rec_ne_wanted : 0;
    const size_t rec_words = rec_ne ? 2 + 2 * (size_t)rec_ne : 0;
    const size_t o_rec = align_up(o_cn + n_chrom * sizeof(uint32_t), 256);
    RawBuf<uint32_t> recs((size_t)(n_bins + 1) * rec_words + 4);  // + one all-zero record (idx.zero_rec)
    for (size_t k = (size_t)n_bins * rec_words; k < recs.size(); ++k) recs[k] = 0u;
    uint64_t n_pieces = 0;
    std::vector<uint32_t> cont;  // 4 words per piece = two {left16 | width16<<16, meta} entries
    if (rec_ne) {
        uint64_t pieces = 0;
        for (uint64_t b = 0; b < n_bins; ++b) {
            const uint32_t cnt = bin_off[b + 1] - bin_off[b];
            if (cnt > (uint32_t)rec_ne) pieces += (cnt - rec_ne + 1) / 2;
        }
        if (pieces >= (1ull << 28) - 16)  // 16 bytes each, 32-bit byte offsets
            return tec_fail(ctx, TEC_EOVERFLOW, "tec_index_build: %llu continuation pieces exceed 32-bit byte offsets",
                            (unsigned long long)pieces);
        cont.assign((size_t)(pieces + 2) * 4, 0u);  // two zero pieces at the end (idx.zero_piece; a lane may read one piece too far)
        n_pieces = pieces;
    }
    const size_t o_cont = align_up(o_rec + (n_bins + 1) * rec_words * 4 + 16, 256);
    const size_t total = align_up(o_cont + std::max<size_t>(cont.size(), 8) * 4, 256);
    if (rec_ne) {
        const int64_t width = (int64_t)1 << shift;
        // continuation pieces are numbered in bin order: each slice first learns how many lie before it
        std::vector<uint64_t> before(T + 1, 0);
        auto pieces_of = [&](uint64_t b) {
            const uint32_t cnt = bin_off[b + 1] - bin_off[b];
            return cnt > (uint32_t)rec_ne ? (uint64_t)(cnt - rec_ne + 1) / 2 : (uint64_t)0;
        };
        const int Tr = n_bins < 4096 ? 1 : T;
        parallel_slices(Tr, n_bins, [&](uint64_t b0, uint64_t b1, int t) {
            uint64_t k = 0;
            for (uint64_t b = b0; b < b1; ++b) k += pieces_of(b);
            before[t + 1] = k;
        });
        for (int t = 0; t < Tr; ++t) before[t + 1] += before[t];
        parallel_slices(Tr, n_bins, [&](uint64_t b0, uint64_t b1, int t) {
            uint64_t piece = before[t];
            // chromosome of the slice's first bin
            uint32_t c = (uint32_t)(std::upper_bound(chrom_base.begin(), chrom_base.end(), (uint32_t)b0) - chrom_base.begin());
            c = c ? c - 1 : 0;
            for (uint64_t b = b0; b < b1; ++b) {
                while (b >= (uint64_t)chrom_base[c] + chrom_nbins[c]) ++c;  // (chromosomes without bins are skipped)
                const uint32_t lb = (uint32_t)(b - chrom_base[c]);
                uint32_t* r = &recs[b * rec_words];
                for (size_t k = 0; k < rec_words; ++k) r[k] = 0u;
                const uint32_t a = bin_off[b], cnt = bin_off[b + 1] - a;
                const int64_t start = (int64_t)lb << shift;
                r[0] = cnt;
                r[1] = cnt > (uint32_t)rec_ne ? (uint32_t)piece : 0u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const int64_t l = std::min<int64_t>(std::max<int64_t>((int64_t)entries[a + k].left - start + 1, 0), width + 2);
                    const int64_t g = std::min<int64_t>(std::max<int64_t>((int64_t)entries[a + k].right - start + 1, 0), width + 2);
                    uint32_t* w = k < (uint32_t)rec_ne ? &r[2 + 2 * k] : &cont[(size_t)piece * 4 + 2 * (k - rec_ne)];
                    w[0] = (uint32_t)l | ((uint32_t)std::max<int64_t>(g - l, 0) << 16);  // {left, width}: see test_compact (bulk.hip)
                    w[1] = entries[a + k].meta;
                }
                piece += pieces_of(b);
            }
        });
    }
    void* blob = nullptr;
    {
        TEC_HIP(ctx, hipMalloc(&blob, total));
    }
    char* p = static_cast<char*>(blob);
    mark("bin records + device malloc");
    hipError_t e = hipMemcpy(p + o_off, bin_off.data(), (n_bins + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_entries_kept) e = hipMemcpy(p + o_ent, entries.data(), n_entries_kept * sizeof(tec_entry), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p + o_cb, chrom_base.data(), n_chrom * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p + o_cn, chrom_nbins.data(), n_chrom * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p + o_rec, recs.data(), recs.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(p + o_cont, 0, total - o_cont);
    if (e == hipSuccess && !cont.empty()) e = hipMemcpy(p + o_cont, cont.data(), cont.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    unsigned long long* counts = nullptr;
    uint32_t* d_i2e = nullptr;
    mark("uploads");
    if (e == hipSuccess) e = hipMalloc(&d_i2e, (size_t)n_names * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_i2e, int2ext.data(), (size_t)n_names * sizeof(uint32_t), hipMemcpyHostToDevice);
    // + 8: the carry bookkeeping of the packed LDS counters may touch the slot after the last name (bulk.hip)
    // (twice: the second half takes the counts in the caller's name order on their way out, see tec_bulk_finish)
    if (e == hipSuccess) e = hipMalloc(&counts, 2 * ((size_t)n_names + 8) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(counts, 0, ((size_t)n_names + 8) * sizeof(unsigned long long));
    if (e != hipSuccess) {
        (void)hipFree(blob);
        if (counts) (void)hipFree(counts);
        if (d_i2e) (void)hipFree(d_i2e);
        return tec_fail(ctx, TEC_EHIP, "tec_index_build: upload failed: %s", hipGetErrorString(e));
    }
    if (ctx->d_index_blob) (void)hipFree(ctx->d_index_blob);
    if (ctx->d_counts) (void)hipFree(ctx->d_counts);
    if (ctx->d_int2ext) (void)hipFree(ctx->d_int2ext);
    ctx->d_int2ext = d_i2e;
    ctx->d_index_blob = blob;
    ctx->d_counts = counts;
    ctx->idx.bin_off = reinterpret_cast<const uint32_t*>(p + o_off);
    ctx->idx.entries = reinterpret_cast<const tec_entry*>(p + o_ent);
    ctx->idx.chrom_base = reinterpret_cast<const uint32_t*>(p + o_cb);
    ctx->idx.chrom_nbins = reinterpret_cast<const uint32_t*>(p + o_cn);
    ctx->idx.recs = reinterpret_cast<const uint4*>(p + o_rec);
    ctx->idx.cont = reinterpret_cast<const uint4*>(p + o_cont);
    ctx->cont_bytes = cont.size() * sizeof(uint32_t);
    ctx->idx.zero_rec = (uint32_t)n_bins;
    ctx->idx.zero_piece = (uint32_t)n_pieces;
    ctx->rec_ne = rec_ne;
    ctx->idx.n_chrom = n_chrom;
    ctx->idx.n_names = n_names;
    ctx->idx.shift = shift;
    ctx->idx.class_mask = class_mask;
    ctx->int2ext.swap(int2ext);
    ctx->n_entries = n_entries_kept;
    ctx->n_bins = n_bins;
    ctx->index_bytes = total;
    ctx->have_index = true;
    return TEC_OK;
}

extern "C" int tec_index_info(const tec_ctx* ctx, uint64_t* n_entries, uint64_t* n_bins, int* bin_shift,
                              uint64_t* device_bytes) {
    if (!ctx || !ctx->have_index) return TEC_ESTATE;
    if (n_entries) *n_entries = ctx->n_entries;
    if (n_bins) *n_bins = ctx->n_bins;
    if (bin_shift) *bin_shift = ctx->idx.shift;
    if (device_bytes) *device_bytes = ctx->index_bytes;
    return TEC_OK;
}
