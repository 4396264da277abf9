// Single-cell (--sc --se) counting on gfx950: tec_sc_* entry points.
//
// Replaces measureTE.sc_parse_bamse (te_count/te_count.py:298-707).  The reference streams the
// BAM once, keeps a dict of (CB,UB) -> set of "chrom:strand:left:rite" strings, spills it to a
// sorted text "bundle" every 1e7 keys, k-way merges the bundles for the best cells and finally
// maps what is left to features.  Here every record that survives the filters stays resident in
// HBM (32-48 B per record: 288 GB holds billions), and the three parts become whole-array
// passes:
//
//   submit   filters (te_count.py:394-433): flags, mapq, CB|CR present, whitelist lookup (hash
//            table over the sorted, byte-packed whitelist -> id = rank, te_count.py:338,418),
//            UB|UR present, '_'/'alt' chromosomes; order-preserving compaction of survivors
//   part 1   stable radix sort by (barcode id, UMI bytes) -- through a key string without the bits
//            that are equal in every survivor, and one gather of packed rows (sc_part1_sort) ->
//            records of one key are adjacent and in file order.  Epochs ("bundles", te_count.py:377-390) are recovered exactly:
//            the epoch ends at the record that creates the bundle_keys-th distinct key since the
//            last flush.  Inside an (epoch, key) instance the first record supplies the stored
//            string; later records of the same class are duplicates (te_count.py:452-454), a
//            record of another class adds a string (te_count.py:457-462).  Per-barcode counts.
//   part 2   (host picks the maxcells+1000 best barcodes, te_count.py:502)  bundle line order =
//            instance heads sorted by (epoch, barcode, UMI); the reference's reader drops a line
//            iff it is the first of its bundle or a selected barcode lies in [BC(prev), BC(line))
//            (te_count.py:509-543); across bundles the first kept line of a key wins (:552-555)
//   part 3   per surviving (key, class): all-bucket candidate rule + the two inclusive 2-bp
//            window tests (te_count.py:619-649), distinct (name, strand) pairs, type branch with
//            the strand filter (te_count.py:660-684) -> (cell, name) keys, sorted + run-length
//            encoded into COO triplets.
#include <cstring>  // before rocprim: its texture iterator calls the host memset

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "tecount_internal.h"

namespace {

constexpr uint32_t kNameMask = (1u << TEC_META_NAME_BITS) - 1;
constexpr uint32_t kPairMask = ~(7u << TEC_META_CLASS_SHIFT);  // name | strand
constexpr int kB = 256;

template <class T>
struct dvec {
    T* p = nullptr;
    size_t cap = 0;
    // floor: a capacity hint for buffers that will keep growing (the survivor arrays): every
    // regrowth is a hipMalloc (cleared by the driver), a device copy and a hipFree
    hipError_t reserve(size_t n, bool keep = false, size_t used = 0, size_t floor = 0) {
        if (n <= cap) return hipSuccess;
        size_t want = std::max(std::max(n, floor), keep ? cap * 2 : cap + cap / 2);
        T* q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(want, 1) * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && used) e = hipMemcpy(q, p, used * sizeof(T), hipMemcpyDeviceToDevice);
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

enum { ST_QCFAIL = 0, ST_LOWQ, ST_BADBC, ST_SKIPCHROM, ST_NOCB, ST_NOUB, ST_LONGUMI, ST_OR0, ST_OR1, ST_OR2, ST_NLIVE, ST_NOSTRAND, ST_N };  // ST_NLIVE: survivors so far (the scatter kernel's output cursor)

}  // namespace

// Device buffers of a single-cell run.  They outlive the run: tec_sc_end parks them on the context (allocating and
// freeing two dozen arrays of HBM costs more than the kernels of a small run), the next tec_sc_begin on that context
// takes them over, tec_destroy frees them.  Nothing in them is read before it is written.
struct ScBuffers {
    dvec<uint64_t> wl_table;    // open-addressing hash of the whitelist, four words per slot: the barcode's three words
                                // (big-endian packed) and its rank + 1, 0 = empty
    dvec<int32_t> cls_chrom;    // read-side chromosome class -> index chrom id (-1: not in index)
    dvec<uint8_t> cls_skip;     // '_' or 'alt' in the read-side chromosome string
    // survivors, in file order
    dvec<uint32_t> bcid;
    dvec<uint64_t> u0, u1, u2, rec;
    dvec<int32_t> klass, left, rite;
    dvec<unsigned long long> stat;
    // batch temporaries
    dvec<uint32_t> b_live, b_scan, b_bcid;
    dvec<char> b_in, b_in2, tmp;  // two device staging buffers: batch k + 1 is copied while batch k is filtered
    // host batches travel through a ring of three pinned buffers (worker threads copy the caller's arrays in, the copy
    // stream takes them to HBM): pin_done[i] = slot i's copy to HBM has finished, in_free[j] = the ingest kernels have
    // finished reading staging buffer j
    char* pin[3] = {nullptr, nullptr, nullptr};
    size_t pin_cap[3] = {0, 0, 0};
    hipEvent_t pin_done[3] = {nullptr, nullptr, nullptr};
    hipEvent_t in_free[2] = {nullptr, nullptr}, in_copied[2] = {nullptr, nullptr};
    uint64_t n_submits = 0;
    // sorted views (part 1)
    dvec<uint32_t> perm, perm2, s_bcid, gid, s_epoch, headidx, cnt;
    dvec<unsigned long long> first_seen;
    dvec<uint64_t> k64a, k64b, s_u0, s_u1, s_u2, s_rec, marks;
    dvec<uint32_t> k32a, k32b;
    dvec<int32_t> s_klass, s_left, s_rite;
    dvec<uint8_t> ghead, ihead, counted;
    dvec<uint4> rows;  // the survivors as 64-byte rows (four pieces each), built by part 1 for the one gather behind its sort
    // temporaries of parts 2 and 3, and their result: the sorted (cell << 32 | name) keys with their counts stay in HBM
    // until tec_sc_fetch takes them
    struct {
        dvec<uint32_t> sel, selprefix, lines, lkeys, lkeys2, lines2, firstkept, qflag, qscan, qrec, nemit, emoff, cellhits, present, pred;
        dvec<uint8_t> kept_rec;
        dvec<uint64_t> keys, keys2, ukeys, side;
        dvec<unsigned long long> scal;
    } p23;
    // the whitelist the tables in `wl` / `wl_table` were built from: a run with the same barcodes (the usual case: one
    // 10x whitelist for every sample) skips packing, checking, hashing and uploading 0.7 M barcodes
    std::shared_ptr<std::vector<uint8_t>> wl_src;
    uint32_t wl_src_mask = 0;
    // multi-GPU exchange on the device: the survivors grouped by destination GPU (deal) and the rows that arrive (land)
    struct Rows {
        dvec<uint32_t> bcid;
        dvec<uint64_t> u0, u1, u2, rec;
        dvec<int32_t> klass, left, rite;
        void release() { bcid.release(); u0.release(); u1.release(); u2.release(); rec.release(); klass.release(); left.release(); rite.release(); }
        hipError_t reserve(size_t n) {
            hipError_t e;
            if ((e = bcid.reserve(n)) || (e = u0.reserve(n)) || (e = u1.reserve(n)) || (e = u2.reserve(n)) || (e = rec.reserve(n)) ||
                (e = klass.reserve(n)) || (e = left.reserve(n)) || (e = rite.reserve(n)))
                return e;
            return hipSuccess;
        }
        void* column(int c, uint32_t* elem) {
            switch (c) {
                case 0: *elem = 4; return bcid.p;
                case 1: *elem = 8; return u0.p;
                case 2: *elem = 8; return u1.p;
                case 3: *elem = 8; return u2.p;
                case 4: *elem = 8; return rec.p;
                case 5: *elem = 4; return klass.p;
                case 6: *elem = 4; return left.p;
                case 7: *elem = 4; return rite.p;
            }
            *elem = 0;
            return nullptr;
        }
    } deal, land;
    char* pin_out = nullptr;  // pinned landing area of tec_sc_fetch when the submit ring is absent or too small
    size_t pin_out_cap = 0;
    // GPU stream time of the phases (events around each entry point's work; includes the short host gaps
    // between its kernels, so it is an upper bound of the summed kernel time)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void free_all() {
        p23.sel.release(); p23.selprefix.release(); p23.lines.release(); p23.lkeys.release(); p23.lkeys2.release(); p23.lines2.release();
        p23.firstkept.release(); p23.qflag.release(); p23.qscan.release(); p23.qrec.release(); p23.nemit.release(); p23.emoff.release();
        p23.cellhits.release(); p23.kept_rec.release(); p23.keys.release(); p23.keys2.release(); p23.ukeys.release(); p23.side.release(); p23.scal.release();
        p23.present.release(); p23.pred.release();
        deal.release();
        land.release();
        if (pin_out) (void)hipHostFree(pin_out);
        pin_out = nullptr;
        pin_out_cap = 0;
        wl_table.release(); cls_chrom.release(); cls_skip.release();
        bcid.release(); u0.release(); u1.release(); u2.release(); rec.release();
        klass.release(); left.release(); rite.release(); stat.release();
        b_live.release(); b_scan.release(); b_bcid.release(); b_in.release(); b_in2.release(); tmp.release();
        for (int i = 0; i < 3; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (pin_done[i]) (void)hipEventDestroy(pin_done[i]);
            pin[i] = nullptr;
            pin_cap[i] = 0;
            pin_done[i] = nullptr;
        }
        for (int i = 0; i < 2; ++i) {
            if (in_free[i]) (void)hipEventDestroy(in_free[i]);
            if (in_copied[i]) (void)hipEventDestroy(in_copied[i]);
            in_free[i] = in_copied[i] = nullptr;
        }
        perm.release(); perm2.release(); s_bcid.release(); gid.release(); s_epoch.release();
        headidx.release(); cnt.release(); first_seen.release(); k64a.release(); k64b.release();
        s_u0.release(); s_u1.release(); s_u2.release(); k32a.release(); k32b.release();
        s_klass.release(); s_left.release(); s_rite.release(); ghead.release(); ihead.release();
        s_rec.release(); marks.release();
        counted.release(); rows.release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        ev0 = ev1 = nullptr;
    }
};

struct tec_sc_state : ScBuffers {
    int stranded = 0, mapq_min = 20;
    uint64_t bundle_keys = 10000000;
    uint64_t expected_records = 0;  // of the whole input, if the caller can tell (the BAM decoder can): sizes the survivor arrays
    uint32_t n_wl = 0, n_class = 0;
    uint32_t wl_mask = 0;
    uint64_t n_live = 0, n_seen = 0;  // n_live is behind the device's cursor while batches are in flight (sc_settle)
    bool in_flight = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ingest_spans;  // stream time of the batches not yet added to gpu_ms
    uint32_t n_groups = 0, n_epochs = 1;
    bool part1_done = false, sorted = false;
    uint64_t rec_base = 0, max_rec = 0, n_marks = 0;  // global ordinal of this context's first record; largest ordinal anywhere
    std::vector<uint64_t> bounds;                      // flush boundaries (global record ordinals)
    std::vector<uint32_t> cells_seen;
    std::vector<uint64_t> cells_count, cells_first;
    // results
    std::vector<uint32_t> out_cell;
    std::vector<uint64_t> out_hits;
    uint64_t n_coo = 0;  // (cell, name, count) triplets waiting in p23.ukeys / p23.nemit
    uint64_t n_land = 0;         // rows the landing columns were sized for (tec_sc_adopt_begin)
    uint64_t host_row_bytes = 0; // bytes of survivor columns that went to host memory in this run (tec_sc_survivors_fetch)
    uint64_t kept_lines = 0, n_queries = 0, n_assigned = 0;
    uint32_t sort_key_bits = 0;  // of part 1's key string; 0 = sorted column by column
    unsigned long long h_stat[ST_N] = {0};
    double gpu_ms = 0.0;
};

namespace {

#ifdef TEC_SC_TRACE
#define SC_TRACE_BEGIN const double t_a__ = sc_trace_now()
#define SC_TRACE_END(what)                                                                                             \
    do {                                                                                                               \
        const double t_b__ = sc_trace_now();                                                                           \
        if (t_b__ - t_a__ > 0.2) fprintf(stderr, "[sc %9.2f] %6.2f ms  line %d  %.60s\n", t_b__, t_b__ - t_a__, __LINE__, what); \
    } while (0)
#else
#define SC_TRACE_BEGIN
#define SC_TRACE_END(what)
#endif
#ifdef TEC_SC_TRACE
// trace build (make EXTRA=-DTEC_SC_TRACE): every HIP call that takes more than 0.2 ms, with the wall clock since the first one
inline double sc_trace_now() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define SC_HIP(call)                                                                                    \
    do {                                                                                                \
        const double t_a = sc_trace_now();                                                              \
        TEC_HIP(ctx, call);                                                                             \
        const double t_b = sc_trace_now();                                                              \
        if (t_b - t_a > 0.2) fprintf(stderr, "[sc %9.2f] %6.2f ms  line %d  %.60s\n", t_b, t_b - t_a, __LINE__, #call); \
    } while (0)
#else
#define SC_HIP(call) TEC_HIP(ctx, call)
#endif

__device__ __forceinline__ uint64_t load_be64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

__host__ __device__ __forceinline__ uint64_t wl_hash(uint64_t a, uint64_t b, uint64_t c) {
    uint64_t h = a * 0x9E3779B97F4A7C15ull;
    h ^= (h >> 32) ^ (b * 0xC2B2AE3D27D4EB4Full);
    h ^= (h >> 29) ^ (c * 0x165667B19E3779F9ull);
    h *= 0xD6E8FEB86659FD93ull;
    return h ^ (h >> 32);
}

// Whitelist membership and rank (te_count.py:412,418): open-addressing table of 32-byte slots {barcode words a, b, c,
// rank + 1} (rank + 1 == 0: empty), so that a probe is ONE 32-byte access: with the barcodes in an array of their own the
// probe was a table word and then, behind it, the barcode it pointed at -- two dependent random lines per record.
__device__ __forceinline__ int wl_find(const uint4* __restrict__ table, uint32_t mask, uint64_t a, uint64_t b, uint64_t c) {
    uint32_t slot = (uint32_t)wl_hash(a, b, c) & mask;
    for (;;) {
        const uint4 lo = table[2 * (size_t)slot], hi = table[2 * (size_t)slot + 1];
        const uint64_t rank1 = (uint64_t)hi.z | ((uint64_t)hi.w << 32);
        if (rank1 == 0) return -1;
        if ((((uint64_t)lo.x | ((uint64_t)lo.y << 32)) == a) & (((uint64_t)lo.z | ((uint64_t)lo.w << 32)) == b) &
            (((uint64_t)hi.x | ((uint64_t)hi.y << 32)) == c))
            return (int)(rank1 - 1);
        slot = (slot + 1) & mask;
    }
}

// Tallies of a whole block: waves add into LDS, one thread per counter adds to global memory.
// (One global atomic per WAVE on the few adjacent stat counters -- millions of memory-side atomics
// on one 64-byte line, ~7 ns each -- was a third of the ingest time of a 50 M-record run.)
template <int N>
struct BlockTally {
    uint32_t* lds;  // N counters in LDS, zeroed by begin()
    __device__ __forceinline__ void begin() {
        if (threadIdx.x < N) lds[threadIdx.x] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void add(int k, bool pred) {
        const unsigned long long m = __ballot(pred);
        if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(&lds[k], (uint32_t)__builtin_popcountll(m));
    }
    __device__ __forceinline__ void add_n(int k, uint32_t v) {  // per-lane counts: summed over the wave first
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&lds[k], v);
    }
    __device__ __forceinline__ void end(unsigned long long* const* dst) {
        __syncthreads();
        if (threadIdx.x < N && lds[threadIdx.x]) atomicAdd(dst[threadIdx.x], (unsigned long long)lds[threadIdx.x]);
    }
};

// te_count.py:394-433, in the reference's order.  A fixed grid strides over the batch and every workgroup adds its four
// tallies to the run's counters once, at its end: with a workgroup per 256 records a 4 M-record batch ended in 60 k
// memory-side atomics on one line, which took longer than the batch's loads.
__global__ __launch_bounds__(256) void sc_filter_kernel(uint64_t n, uint64_t rec_base, const int32_t* __restrict__ cls,
                                 const uint16_t* __restrict__ flag, const uint8_t* __restrict__ mapq,
                                 const uint8_t* __restrict__ cb, const uint8_t* __restrict__ tagflags, int mapq_min,
                                 const uint4* __restrict__ wl_table, uint32_t wl_mask,
                                 const uint8_t* __restrict__ cls_skip,
                                 uint32_t n_class, uint32_t* live, uint32_t* bcid_out, unsigned long long* stat) {
    __shared__ uint32_t tally_lds[4];
    BlockTally<4> tally{tally_lds};
    tally.begin();
    uint32_t n_qc = 0, n_lowq = 0, n_badbc = 0, n_skipc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        bool ok = false;
        const unsigned fl = flag[i];
        const unsigned tf = tagflags[i];
        if (fl & (TEC_FLAG_UNMAPPED | TEC_FLAG_QCFAIL | TEC_FLAG_DUP)) {
            ++n_qc;
        } else if ((int)mapq[i] < mapq_min) {
            ++n_lowq;
        } else if (!(tf & 1u)) {
            atomicMin(&stat[ST_NOCB], (unsigned long long)(rec_base + i));  // AssertionError('CB or CR tag not found!')
        } else {
            int id = -1;
            if (!(tf & 4u)) {  // bit 2: barcode longer than 24 bytes, cannot be on the whitelist
                const uint8_t* p = cb + 24 * i;
                id = wl_find(wl_table, wl_mask, load_be64(p), load_be64(p + 8), load_be64(p + 16));
            }
            if (id < 0) {
                ++n_badbc;
            } else if (!(tf & 2u)) {
                atomicMin(&stat[ST_NOUB], (unsigned long long)(rec_base + i));  // AssertionError('UB or UR tag not found!')
            } else {
                const int c = cls[i];
                if (c < 0 || (uint32_t)c >= n_class || cls_skip[c]) {
                    ++n_skipc;
                } else {
                    ok = true;
                    bcid_out[i] = (uint32_t)id;
                    // bit 3: a UMI longer than the 24-byte field.  Only a record that reaches the (CB,UB) collapse makes
                    // the run fail: the reference never looks at the UMI of a record it has dropped before (te_count.py:394-433)
                    if (tf & 8u) atomicMin(&stat[ST_LONGUMI], (unsigned long long)(rec_base + i));
                }
            }
        }
        live[i] = ok ? 1u : 0u;
    }
    tally.add_n(0, n_qc);
    tally.add_n(1, n_lowq);
    tally.add_n(2, n_badbc);
    tally.add_n(3, n_skipc);
    unsigned long long* const dst[4] = {&stat[ST_QCFAIL], &stat[ST_LOWQ], &stat[ST_BADBC], &stat[ST_SKIPCHROM]};
    tally.end(dst);
}

__global__ void sc_scatter_kernel(uint64_t n, uint64_t rec_base, const unsigned long long* __restrict__ cursor, const uint32_t* __restrict__ live,
                                  const uint32_t* __restrict__ scan, const uint32_t* __restrict__ bcid_in,
                                  const int32_t* __restrict__ cls, const int32_t* __restrict__ pos,
                                  const int32_t* __restrict__ end, const uint16_t* __restrict__ flag,
                                  const uint8_t* __restrict__ ub, int stranded, uint32_t* bcid, uint64_t* u0,
                                  uint64_t* u1, uint64_t* u2, uint64_t* rec, int32_t* klass, int32_t* left, int32_t* rite,
                                  unsigned long long* stat) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a = 0, b = 0, c = 0;
    if (i < n && live[i]) {
        const uint64_t o = *cursor + scan[i];
        const uint8_t* p = ub + 24 * i;
        a = load_be64(p);
        b = load_be64(p + 8);
        c = load_be64(p + 16);
        bcid[o] = bcid_in[i];
        u0[o] = a;
        u1[o] = b;
        u2[o] = c;
        rec[o] = rec_base + i;
        klass[o] = (cls[i] << 1) | ((stranded && (flag[i] & TEC_FLAG_REVERSE)) ? 1 : 0);
        left[o] = pos[i];
        rite[o] = end[i];
    }
    // which UMI bits are ever set: bounds the radix passes of part 1
    for (int off = 32; off > 0; off >>= 1) {
        a |= __shfl_down(a, off, 64);
        b |= __shfl_down(b, off, 64);
        c |= __shfl_down(c, off, 64);
    }
    // block-level OR in LDS, then one global atomic per word and block
    __shared__ unsigned long long or_lds[3];
    if (threadIdx.x < 3) or_lds[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicOr(&or_lds[0], (unsigned long long)a);
        if (b) atomicOr(&or_lds[1], (unsigned long long)b);
        if (c) atomicOr(&or_lds[2], (unsigned long long)c);
    }
    __syncthreads();
    if (threadIdx.x < 3 && or_lds[threadIdx.x]) {
        // skip the global atomic when it would not change anything (the common case after the first blocks)
        unsigned long long* g = &stat[ST_OR0 + threadIdx.x];
        if ((or_lds[threadIdx.x] & ~__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) atomicOr(g, or_lds[threadIdx.x]);
    }
}

// Behind the scatter of a batch: the output cursor moves on by the batch's survivors.
__global__ void sc_advance_kernel(uint64_t n, const uint32_t* __restrict__ live, const uint32_t* __restrict__ scan, unsigned long long* cursor) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *cursor += (unsigned long long)scan[n - 1] + live[n - 1];
}

// destination GPU of every survivor (cells are dealt by barcode id) and how many go where
__global__ void deal_key_kernel(uint64_t n, const uint32_t* __restrict__ bcid, uint32_t world, uint32_t* key, unsigned long long* counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ unsigned int tally[64];
    if (threadIdx.x < 64) tally[threadIdx.x] = 0;
    __syncthreads();
    if (i < n) {
        const uint32_t d = bcid[i] % world;
        key[i] = d;
        atomicAdd(&tally[d & 63u], 1u);  // (world <= 64)
    }
    __syncthreads();
    if (threadIdx.x < 64 && threadIdx.x < world && tally[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

// which UMI bits are set anywhere, and the largest record ordinal, of rows that arrived from other GPUs
__global__ void land_summary_kernel(uint64_t n, const uint64_t* __restrict__ u0, const uint64_t* __restrict__ u1,
                                    const uint64_t* __restrict__ u2, const uint64_t* __restrict__ rec, unsigned long long* out4) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long a = 0, b = 0, c = 0, m = 0;
    if (i < n) {
        a = u0[i];
        b = u1[i];
        c = u2[i];
        m = rec[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
        a |= __shfl_down(a, off, 64);
        b |= __shfl_down(b, off, 64);
        c |= __shfl_down(c, off, 64);
        const unsigned long long o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicOr(&out4[0], a);
        if (b) atomicOr(&out4[1], b);
        if (c) atomicOr(&out4[2], c);
        atomicMax(&out4[3], m);
    }
}

template <class T>
__global__ void gather_kernel(uint64_t n, const uint32_t* __restrict__ perm, const T* __restrict__ src, T* dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

__global__ void iota_kernel(uint64_t n, uint32_t* p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

// Which UMI bits differ anywhere among the survivors: out8 = {OR of word 0, 1, 2, AND of word 0, 1, 2, "every UMI byte
// is NUL or one of A C G N T" (all ones while true), which of those six occur (bit = the byte's low five bits)}.  A bit
// that is the same in every UMI (the high bits of ASCII letters, the zero padding behind a 12-mer) cannot decide a
// comparison and is left out of the sort key; a byte position of the six-symbol alphabet takes its symbol's rank.
constexpr uint32_t kUmiLetters = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 14) | (1u << 20);  // 'A' 'C' 'G' 'N' 'T' & 31
__device__ __forceinline__ void umi_alphabet(unsigned long long w, bool& ok, uint32_t& seen) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t b = (uint32_t)(w >> (8 * k)) & 0xFFu, lo = b & 31u;
        ok = ok & ((b == 0u) | (((b >> 5) == 2u) & (((kUmiLetters >> lo) & 1u) != 0u)));
        seen |= 1u << lo;
    }
}
__global__ __launch_bounds__(256) void umi_masks_kernel(uint64_t n, const uint64_t* __restrict__ u0, const uint64_t* __restrict__ u1,
                                                        const uint64_t* __restrict__ u2, unsigned long long* out8) {
    unsigned long long o0 = 0, o1 = 0, o2 = 0, a0 = ~0ull, a1 = ~0ull, a2 = ~0ull;
    bool ok = true;
    uint32_t seen = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {  // twelve loads in flight per lane
        unsigned long long x[4], y[4], z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = __builtin_nontemporal_load(&u0[i + k * stride]);
            y[k] = __builtin_nontemporal_load(&u1[i + k * stride]);
            z[k] = __builtin_nontemporal_load(&u2[i + k * stride]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o0 |= x[k]; o1 |= y[k]; o2 |= z[k];
            a0 &= x[k]; a1 &= y[k]; a2 &= z[k];
            umi_alphabet(x[k], ok, seen);
            umi_alphabet(y[k], ok, seen);
            if (z[k]) umi_alphabet(z[k], ok, seen);  // (UMIs of at most 16 bytes: all NUL)
        }
    }
    for (; i < n; i += stride) {
        const unsigned long long x = u0[i], y = u1[i], z = u2[i];
        o0 |= x; o1 |= y; o2 |= z;
        a0 &= x; a1 &= y; a2 &= z;
        umi_alphabet(x, ok, seen);
        umi_alphabet(y, ok, seen);
        umi_alphabet(z, ok, seen);
    }
    unsigned long long conform = ok ? ~0ull : 0ull, pres = seen;
    for (int off = 32; off > 0; off >>= 1) {
        o0 |= __shfl_down(o0, off, 64); o1 |= __shfl_down(o1, off, 64); o2 |= __shfl_down(o2, off, 64);
        a0 &= __shfl_down(a0, off, 64); a1 &= __shfl_down(a1, off, 64); a2 &= __shfl_down(a2, off, 64);
        conform &= __shfl_down(conform, off, 64);
        pres |= __shfl_down(pres, off, 64);
    }
    // waves -> workgroup in LDS -> eight global atomics per workgroup (thousands of waves on one line of memory-side
    // atomics would queue for longer than the pass itself takes)
    __shared__ unsigned long long acc[8];
    if (threadIdx.x < 8) acc[threadIdx.x] = (threadIdx.x < 3 || threadIdx.x == 7) ? 0ull : ~0ull;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicOr(&acc[0], o0); atomicOr(&acc[1], o1); atomicOr(&acc[2], o2);
        atomicAnd(&acc[3], a0); atomicAnd(&acc[4], a1); atomicAnd(&acc[5], a2);
        atomicAnd(&acc[6], conform);
        atomicOr(&acc[7], pres);
    }
    __syncthreads();
    if (threadIdx.x < 3 || threadIdx.x == 7)
        atomicOr(&out8[threadIdx.x], acc[threadIdx.x]);
    else if (threadIdx.x < 7)
        atomicAnd(&out8[threadIdx.x], acc[threadIdx.x]);
}

// The sort key of part 1 as one bit string: barcode id, then what decides the order of two UMIs, most significant
// first -- per UMI byte either the rank of its symbol (alphabet NUL A C G N T: as many bits as the symbols that occur
// need) or, for any other UMI, the bits of the three words that differ anywhere.  Dropping what is equal in every row
// and replacing symbols by their ranks keeps the order of any two rows.  The host cuts the string into runs; no run
// crosses bit 64 of the string.
constexpr int kMaxKeyRuns = 56;
struct KeyRuns {
    uint32_t n;
    unsigned long long ranks;  // symbol ranks, three bits each, at bit 3 * (byte & 31)
    struct Run {
        unsigned long long mask;  // of the run's width in the string
        uint8_t word, src, dst;   // source word (0..2 = UMI words, 3 = barcode id), first source bit, first bit in the string
        uint8_t symbol;           // 1: the source is one byte, the run holds its rank
    } r[kMaxKeyRuns];
};

// One pass over the survivor columns (file order): the packed row of every survivor and its key string (low 64 bits;
// the bits above them, at most 32, when the string is longer).  A string of at most 64 bits says by itself whether two
// neighbours of the sorted order share their (barcode, UMI), so the row then leaves the UMI out: 32 bytes {record
// ordinal, barcode id, class, left, rite}; otherwise 64 bytes with the three UMI words in front.
template <bool WITH_UMI>
__global__ __launch_bounds__(256) void pack_rows_kernel(uint64_t n, const uint32_t* __restrict__ bcid, const uint64_t* __restrict__ u0,
                                                        const uint64_t* __restrict__ u1, const uint64_t* __restrict__ u2,
                                                        const uint64_t* __restrict__ rec, const int32_t* __restrict__ klass,
                                                        const int32_t* __restrict__ left, const int32_t* __restrict__ rite,
                                                        const KeyRuns kr, uint4* __restrict__ rows, uint64_t* __restrict__ key_lo,
                                                        uint32_t* __restrict__ key_hi) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = u0[i], b = u1[i], c = u2[i], o = rec[i];
    const uint32_t bc = bcid[i];
    if (WITH_UMI) {
        uint4* row = rows + 4 * i;
        row[0] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
        row[1] = make_uint4((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)o, (uint32_t)(o >> 32));
        row[2] = make_uint4(bc, (uint32_t)klass[i], (uint32_t)left[i], (uint32_t)rite[i]);
        row[3] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        uint4* row = rows + 2 * i;
        row[0] = make_uint4((uint32_t)o, (uint32_t)(o >> 32), bc, (uint32_t)klass[i]);
        row[1] = make_uint4((uint32_t)left[i], (uint32_t)rite[i], 0u, 0u);
    }
    uint64_t lo = 0, hi = 0;
    for (uint32_t k = 0; k < kr.n; ++k) {  // (wave-uniform: the runs come from the kernel arguments)
        const uint32_t w = kr.r[k].word, src = kr.r[k].src, dst = kr.r[k].dst;
        const uint64_t v = w == 0 ? a : w == 1 ? b : w == 2 ? c : (uint64_t)bc;
        uint64_t f = v >> src;
        if (kr.r[k].symbol) f = kr.ranks >> (3u * ((uint32_t)f & 31u));
        f &= kr.r[k].mask;
        if (dst < 64u)
            lo |= f << dst;
        else
            hi |= f << (dst - 64u);
    }
    key_lo[i] = lo;
    if (WITH_UMI && key_hi) key_hi[i] = (uint32_t)hi;
}

// The sorted views of part 1: row perm[i] of the packed survivors (one piece of one line) spread over the sorted
// columns.  (Eight column gathers pulled eight lines per survivor.)  Rows without the UMI: the first record of every
// (barcode, UMI) key is marked from the sorted key strings here (ghead_kernel's output); the sorted UMI columns, which
// nothing else reads, are not written.
template <bool WITH_UMI>
__global__ __launch_bounds__(256) void gather_rows_kernel(uint64_t n, const uint32_t* __restrict__ perm, const uint4* __restrict__ rows,
                                                          uint32_t* __restrict__ s_bcid, uint64_t* __restrict__ s_u0,
                                                          uint64_t* __restrict__ s_u1, uint64_t* __restrict__ s_u2,
                                                          uint64_t* __restrict__ s_rec, int32_t* __restrict__ s_klass,
                                                          int32_t* __restrict__ s_left, int32_t* __restrict__ s_rite,
                                                          const uint64_t* __restrict__ sorted_key, uint8_t* __restrict__ ghead,
                                                          uint32_t* __restrict__ ghead32) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (WITH_UMI) {
        const uint4* row = rows + 4 * (uint64_t)perm[i];
        const uint4 q0 = row[0], q1 = row[1], q2 = row[2];
        s_u0[i] = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
        s_u1[i] = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
        s_u2[i] = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
        s_rec[i] = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
        s_bcid[i] = q2.x;
        s_klass[i] = (int32_t)q2.y;
        s_left[i] = (int32_t)q2.z;
        s_rite[i] = (int32_t)q2.w;
    } else {
        const uint4* row = rows + 2 * (uint64_t)perm[i];
        const uint4 q0 = row[0], q1 = row[1];
        s_rec[i] = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
        s_bcid[i] = q0.z;
        s_klass[i] = (int32_t)q0.w;
        s_left[i] = (int32_t)q1.x;
        s_rite[i] = (int32_t)q1.y;
        const bool h = i == 0 || sorted_key[i] != sorted_key[i - 1];
        ghead[i] = h;
        ghead32[i] = h;
    }
}

__global__ void ghead_kernel(uint64_t n, const uint32_t* __restrict__ bc, const uint64_t* __restrict__ u0,
                             const uint64_t* __restrict__ u1, const uint64_t* __restrict__ u2, uint8_t* ghead,
                             uint32_t* ghead32) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool h = i == 0 || bc[i] != bc[i - 1] || u0[i] != u0[i - 1] || u1[i] != u1[i - 1] || u2[i] != u2[i - 1];
    ghead[i] = h;
    ghead32[i] = h;
}

// Marks (in key-sorted space; `rec` = global record ordinals, ascending inside a key) the first record
// of every key after record `lower`: the records that create a NEW dict entry once the reference has
// flushed at `lower` (te_count.py:377-390, 464-473).
__global__ void epoch_mark_kernel(uint64_t n, const uint64_t* __restrict__ rec, const uint8_t* __restrict__ ghead,
                                  long long lower, uint32_t* mark) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long o = (long long)rec[i];
    const bool m = o > lower && (ghead[i] || (long long)rec[i - 1] <= lower);
    mark[i] = m ? 1u : 0u;
}

__global__ void compact_rec_kernel(uint64_t n, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ scan,
                                   const uint64_t* __restrict__ rec, uint64_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[scan[i]] = rec[i];
}

// out[0] = number of sorted values <= v (one thread: a 40-step binary search)
__global__ void upper_bound_kernel(const uint64_t* __restrict__ sorted, uint64_t n, uint64_t v, unsigned long long* out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (sorted[mid] <= v)
            lo = mid + 1;
        else
            hi = mid;
    }
    out[0] = lo;
}

__global__ void add_base_kernel(uint64_t n, uint64_t* rec, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i] += base;
}

__global__ void epoch_assign_kernel(uint64_t n, const uint64_t* __restrict__ ord, const uint64_t* __restrict__ bounds,
                                    uint32_t n_bounds, uint32_t* epoch) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = n_bounds;  // number of boundaries strictly below this ordinal
    const uint64_t o = ord[i];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bounds[mid] < o)
            lo = mid + 1;
        else
            hi = mid;
    }
    epoch[i] = lo;
}

__global__ void ihead_kernel(uint64_t n, const uint8_t* __restrict__ ghead, const uint32_t* __restrict__ epoch,
                             uint8_t* ihead, uint32_t* headpos) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool h = ghead[i] || epoch[i] != epoch[i - 1];
    ihead[i] = h;
    headpos[i] = h ? (uint32_t)i : 0u;
}

// counted = a string is stored / the barcode counter bumps (te_count.py:457-473)
// The records are sorted by barcode, so a wave mostly sees one or two barcodes: per distinct barcode of
// the wave ONE add and ONE min go to memory (37 M atomics on 10 k hot words took 3.5 ms of the 50 M-record run).
__global__ __launch_bounds__(256) void part1_count_kernel(uint64_t n, const uint8_t* __restrict__ ihead,
                                                           const uint32_t* __restrict__ headidx, const int32_t* __restrict__ klass,
                                                           const uint32_t* __restrict__ bc, const uint64_t* __restrict__ ord,
                                                           uint8_t* counted, uint32_t* cnt, unsigned long long* first_seen) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const bool c = in && (ihead[i] || klass[i] != klass[headidx[i]]);
    if (in) counted[i] = c;
    const uint32_t b = c ? bc[i] : 0u;
    const unsigned long long o = c ? (unsigned long long)ord[i] : ~0ull;
    const unsigned lane = threadIdx.x & 63u;
    uint64_t todo = __builtin_amdgcn_ballot_w64(c);
    while (todo) {  // wave-uniform
        const int lead = __builtin_ctzll(todo);
        const uint32_t bl = (uint32_t)__shfl((int)b, lead, 64);
        const bool mine = c && b == bl;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(mine);
        unsigned long long m = mine ? o : ~0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long other = (unsigned long long)__shfl_xor((long long)m, d, 64);
            m = other < m ? other : m;
        }
        if ((int)lane == lead) {
            atomicAdd(&cnt[bl], (uint32_t)__builtin_popcountll(mask));
            atomicMin(&first_seen[bl], m);
        }
        todo &= ~mask;
    }
}

struct U8ToU32 {
    __host__ __device__ uint32_t operator()(uint8_t v) const { return v; }
};

template <class F>
__global__ void compact_idx_kernel(uint64_t n, const F* __restrict__ flag, const uint32_t* __restrict__ scan,
                                   uint32_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[scan[i]] = (uint32_t)i;
}

__global__ void scatter_sel_kernel(uint32_t n, const uint32_t* __restrict__ sel_ids, uint32_t* sel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sel[sel_ids[i]] = 1u;
}

// Which (bundle, barcode) pairs have lines at all: present[e * n_wl + b] = 1.
__global__ void presence_kernel(uint64_t n_lines, const uint32_t* __restrict__ line_rec, const uint32_t* __restrict__ bc,
                                const uint32_t* __restrict__ epoch, uint32_t n_wl, uint32_t* present) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    const uint32_t i = line_rec[j];
    present[(size_t)epoch[i] * n_wl + bc[i]] = 1u;
}

// val[e * n_wl + b] = e * (n_wl + 1) + b + 1 where (e, b) has lines, else 0.  An inclusive max-scan over
// the whole table then holds, at slot (e, b), the mark of the largest barcode <= b with lines in bundle
// e -- or a mark of an EARLIER bundle (recognisable by its quotient) if bundle e has none up to b.
__global__ void presence_mark_kernel(uint64_t n, const uint32_t* __restrict__ present, uint32_t n_wl, uint32_t* val) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint32_t e = (uint32_t)(x / n_wl), b = (uint32_t)(x % n_wl);
    val[x] = present[x] ? e * (n_wl + 1u) + b + 1u : 0u;
}

// The bundle reader's drop rule (te_count.py:509-543), one thread per bundle line.  A bundle's lines
// are sorted by (barcode, UMI); with the cells dealt over several GPUs a line's predecessor may live on
// another GPU, so the rule is evaluated against the GLOBAL set of (bundle, barcode) pairs that have
// lines (pred, from the all-reduced presence table): a line is kept iff its barcode is selected and
//   - the previous line of the bundle has the same barcode, or
//   - there is an earlier barcode p with lines in this bundle (else this is the bundle's first line,
//     which the reader never emits) and no selected barcode lies in [p, barcode).
__global__ __launch_bounds__(256) void kept_kernel(uint64_t n_lines, const uint32_t* __restrict__ line_rec, const uint32_t* __restrict__ bc,
                            const uint32_t* __restrict__ epoch, const uint32_t* __restrict__ sel,
                            const uint32_t* __restrict__ selprefix, const uint32_t* __restrict__ gid,
                            const uint32_t* __restrict__ pred, uint32_t n_wl, uint8_t* kept_rec, uint32_t* firstkept,
                            unsigned long long* n_kept) {
    // (a fixed grid strides over the lines: one add of the kept count per workgroup, see sc_filter_kernel)
    __shared__ uint32_t tally_lds[1];
    BlockTally<1> tally{tally_lds};
    tally.begin();
    uint32_t n_k = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lines; j += (uint64_t)gridDim.x * blockDim.x) {
        bool k = false;
        const uint32_t i = line_rec[j];
        const uint32_t b = bc[i], e = epoch[i];
        if (sel[b]) {
            bool same = false;
            if (j > 0) {
                const uint32_t p = line_rec[j - 1];
                same = epoch[p] == e && bc[p] == b;
            }
            if (same) {
                k = true;
            } else {
                // the previous barcode with lines in this bundle (see presence_mark_kernel)
                const uint32_t v = b ? pred[(size_t)e * n_wl + b - 1u] : 0u;
                const bool has = v != 0u && v / (n_wl + 1u) == e;
                k = has && selprefix[b] == selprefix[v - e * (n_wl + 1u) - 1u];
            }
        }
        kept_rec[i] = k;
        if (k) {
            atomicMin(&firstkept[gid[i]], i);
            ++n_k;
        }
    }
    tally.add_n(0, n_k);
    unsigned long long* const dst[1] = {n_kept};
    tally.end(dst);
}

// A record feeds part 3 iff its instance survived part 2 and it is the head, or an extra string
// that no later extra of the same class overwrites (reads[(chrom,strand)] = ..., te_count.py:606).
__global__ void query_flag_kernel(uint64_t n, const uint8_t* __restrict__ ihead, const uint8_t* __restrict__ counted,
                                  const uint32_t* __restrict__ headidx, const uint8_t* __restrict__ kept_rec,
                                  const uint32_t* __restrict__ firstkept, const uint32_t* __restrict__ gid,
                                  const int32_t* __restrict__ klass, uint32_t* qflag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = headidx[i];
    bool q = counted[i] && kept_rec[h] && firstkept[gid[h]] == h;
    if (q && !ihead[i]) {
        for (uint64_t x = i + 1; x < n && !ihead[x]; ++x)
            if (counted[x] && klass[x] == klass[i]) {
                q = false;
                break;
            }
    }
    qflag[i] = q ? 1u : 0u;
}

__device__ __forceinline__ int bucket_of(int x) { return x < 0 ? -1 : (int)((unsigned)x / (unsigned)TEC_BUCKET_SIZE); }

struct ScQuery {
    int left, rite, blo, bhi;
};
// te_count.py:645 / :648 (inclusive windows) and :619-634 (every bucket from (left-1)//bs to rite//bs)
__device__ __forceinline__ bool sc_t1(const tec_entry& e, const ScQuery& q) { return (long long)q.left + 1 >= e.left && q.left <= e.right; }
__device__ __forceinline__ bool sc_t2(const tec_entry& e, const ScQuery& q) { return q.rite >= e.left && (long long)q.rite - 1 <= e.right; }
__device__ __forceinline__ bool sc_cand(const tec_entry& e, const ScQuery& q) {
    // a degenerate feature with floor(left/bs) > floor(right/bs) (right == left - 1 across a 10 kb
    // boundary) is filed under no bucket at all (genelist.py:371-373: empty range)
    const int lo = bucket_of(e.left), hi = bucket_of(e.right);
    return q.bhi >= q.blo && lo <= hi && lo <= q.bhi && hi >= q.blo;
}
__device__ __forceinline__ int floor_bucket(long long x) {
    long long b = x / TEC_BUCKET_SIZE;
    if (x - b * TEC_BUCKET_SIZE < 0) --b;
    return (int)b;
}

__device__ __forceinline__ void sc_range(const tec_index_dev& idx, int chrom, int pos, uint32_t* a, uint32_t* z) {
    *a = *z = 0;
    if (pos < -1) return;
    const uint32_t b = (uint32_t)(pos < 0 ? 0 : pos) >> idx.shift;
    if (b >= idx.chrom_nbins[chrom]) return;
    const uint32_t g = idx.chrom_base[chrom] + b;
    *a = idx.bin_off[g];
    *z = idx.bin_off[g + 1];
}

// Part 3 for one (cell, class, left, rite).  EMIT=false: returns the number of (cell, name)
// increments; EMIT=true: writes them.  Also bumps the per-cell "mapped to something" counter.
// The counting pass keeps a query's first kSide keys in `side` (key j of query t at side[j * nq + t]): the writing
// pass copies them to their place and walks the index again only for the queries that emitted more.
constexpr uint32_t kSide = 4;
template <bool EMIT>
__global__ __launch_bounds__(256) void sc_query_kernel(tec_index_dev idx, uint64_t nq, const uint32_t* __restrict__ qrec,
                                const uint32_t* __restrict__ bc, const int32_t* __restrict__ klass,
                                const int32_t* __restrict__ left, const int32_t* __restrict__ rite,
                                const int32_t* __restrict__ cls_chrom, int stranded, uint32_t* n_emit,
                                const uint32_t* __restrict__ emit_off, uint64_t* keys, uint64_t* side, uint32_t* cellhits,
                                unsigned long long* stat, unsigned long long* n_assigned) {
    // tallies per block (see BlockTally): [0] assigned, [1] hits on a feature without a strand.  A fixed grid strides
    // over the queries in whole-wave steps (every lane of a wave makes the same number of trips: the per-cell adds
    // below are folded over the wave), and the tallies go to memory once per workgroup.
    __shared__ uint32_t tally_lds[2];
    BlockTally<2> tally{tally_lds};
    if (!EMIT) tally.begin();
    uint32_t n_assign = 0, n_nostrand = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t trips = (nq + stride - 1) / stride;
    for (uint64_t trip = 0; trip < trips; ++trip) {
    const uint64_t t = trip * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool did_hit = false;
    uint32_t hit_bc = 0;
    if (t < nq) {
    const uint32_t i = qrec[t];
    const int kl = klass[i];
    const int chrom = cls_chrom[kl >> 1];
    uint32_t emitted = 0;
    uint64_t* out = EMIT ? keys + emit_off[t] : nullptr;
    const uint32_t counted = EMIT ? n_emit[t] : 0u;
    if (EMIT && counted <= kSide) {
        for (uint32_t j = 0; j < counted; ++j) out[j] = side[(uint64_t)j * nq + t];
    } else if (chrom >= 0) {
        ScQuery q;
        q.left = left[i];
        q.rite = rite[i];
        q.blo = floor_bucket((long long)q.left - 1);
        q.bhi = floor_bucket((long long)q.rite);
        uint32_t a0, a1, b0, b1;
        sc_range(idx, chrom, q.left, &a0, &a1);
        sc_range(idx, chrom, q.rite, &b0, &b1);
        const tec_entry* __restrict__ ent = idx.entries;
        const bool one = a0 == b0 && a1 == b1;
        // pass 1: which classes are hit
        uint32_t classes = 0;
        bool nostrand = false;
        for (uint32_t j = a0; j < a1; ++j) {
            const tec_entry e = ent[j];
            if ((sc_t1(e, q) || (one && sc_t2(e, q))) && sc_cand(e, q)) {
                classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
                nostrand |= (e.meta >> TEC_META_STRAND_SHIFT) == 7u;
            }
        }
        if (!one)
            for (uint32_t j = b0; j < b1; ++j) {
                const tec_entry e = ent[j];
                if (sc_t2(e, q) && sc_cand(e, q)) {
                    classes |= 1u << ((e.meta >> TEC_META_CLASS_SHIFT) & 7u);
                    nostrand |= (e.meta >> TEC_META_STRAND_SHIFT) == 7u;
                }
            }
        if (classes) {
            if (!EMIT) {
                did_hit = true;  // self.barcodes[barcode] += 1 (te_count.py:653-655), added below
                hit_bc = bc[i];
                n_nostrand += nostrand ? 1u : 0u;  // KeyError: 'strand' (te_count.py:661)
            }
            const bool gene = classes & (1u << TEC_CLASS_GENE);
            const bool other = classes & ((1u << TEC_CLASS_TE) | (1u << TEC_CLASS_ENHANCER));
            if (gene || other) {
                ++n_assign;
                const uint32_t rs = (uint32_t)(kl & 1);
                const uint64_t cell = (uint64_t)bc[i] << 32;
                uint32_t last = 0xFFFFFFFFu;
                auto emit = [&](uint32_t pair) {
                    // gene branch: antisense pairs are skipped when --strand (te_count.py:665)
                    if (gene && stranded && (pair >> TEC_META_STRAND_SHIFT) != rs) return;
                    if (EMIT)
                        out[emitted] = cell | (pair & kNameMask);
                    else if (emitted < kSide)
                        side[(uint64_t)emitted * nq + t] = cell | (pair & kNameMask);
                    ++emitted;
                };
                for (uint32_t j = a0; j < a1; ++j) {
                    const tec_entry e = ent[j];
                    if ((sc_t1(e, q) || (one && sc_t2(e, q))) && sc_cand(e, q)) {
                        const uint32_t pair = e.meta & kPairMask;
                        if (pair != last) emit(pair);
                        last = pair;
                    }
                }
                if (!one) {
                    last = 0xFFFFFFFFu;
                    for (uint32_t j = b0; j < b1; ++j) {
                        const tec_entry e = ent[j];
                        if (sc_t2(e, q) && sc_cand(e, q)) {
                            const uint32_t pair = e.meta & kPairMask;
                            if (pair != last) {
                                bool seen = false;
                                for (uint32_t k = a0; k < a1 && !seen; ++k) {
                                    const tec_entry f = ent[k];
                                    seen = (f.meta & kPairMask) == pair && sc_t1(f, q) && sc_cand(f, q);
                                }
                                if (!seen) emit(pair);
                            }
                            last = pair;
                        }
                    }
                }
            }
        }
    }
    if (!EMIT) n_emit[t] = emitted;
    }
    if (!EMIT) {
        // the queries are in barcode order, so a wave mostly sees one or two cells: one add per distinct cell of the
        // wave instead of one memory-side atomic per query on a few thousand hot words
        uint64_t todo = __builtin_amdgcn_ballot_w64(did_hit);
        while (todo) {  // wave-uniform
            const int lead = __builtin_ctzll(todo);
            const uint32_t bl = (uint32_t)__shfl((int)hit_bc, lead, 64);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(did_hit && hit_bc == bl);
            if ((int)(threadIdx.x & 63u) == lead) atomicAdd(&cellhits[bl], (uint32_t)__builtin_popcountll(mask));
            todo &= ~mask;
        }
    }
    }
    if (!EMIT) {
        tally.add_n(0, n_assign);
        tally.add_n(1, n_nostrand);
        unsigned long long* const dst[2] = {n_assigned, &stat[ST_NOSTRAND]};
        tally.end(dst);
    }
}

__global__ void map_names_kernel(uint64_t n, uint64_t* keys, const uint32_t* __restrict__ int2ext) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (keys[i] & 0xFFFFFFFF00000000ull) | int2ext[keys[i] & 0xFFFFFFFFull];
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + kB - 1) / kB); }

template <class In, class T>
int scan_u32(tec_ctx* ctx, tec_sc_state* s, In in, uint32_t* out, size_t n, bool inclusive, T op) {
    size_t bytes = 0;
    if (inclusive)
        SC_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, n, op, ctx->s_compute));
    else
        SC_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, op, ctx->s_compute));
    SC_HIP(s->tmp.reserve(bytes));
    if (inclusive)
        SC_HIP(rocprim::inclusive_scan(s->tmp.p, bytes, in, out, n, op, ctx->s_compute));
    else
        SC_HIP(rocprim::exclusive_scan(s->tmp.p, bytes, in, out, 0u, n, op, ctx->s_compute));
    return TEC_OK;
}

template <class K, class VIn>
int sort_pairs(tec_ctx* ctx, tec_sc_state* s, const K* kin, K* kout, VIn vin, uint32_t* vout, size_t n,
               unsigned b0, unsigned b1) {
    size_t bytes = 0;
    SC_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, b0, b1, ctx->s_compute));
    SC_HIP(s->tmp.reserve(bytes));
    SC_HIP(rocprim::radix_sort_pairs(s->tmp.p, bytes, kin, kout, vin, vout, n, b0, b1, ctx->s_compute));
    return TEC_OK;
}

inline unsigned bits_for(uint64_t max_value) {
    unsigned b = 1;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

}  // namespace

// Ends the run on this context, if any: its buffers are parked for the next one.
static void sc_park(tec_ctx* ctx) {
    tec_sc_state* s = ctx->sc;
    if (!s) return;
    ctx->sc = nullptr;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->s_copy);  // a batch may still be on its way
    (void)hipStreamSynchronize(ctx->s_compute);
    for (auto& span : s->ingest_spans) {
        (void)hipEventDestroy(span.first);
        (void)hipEventDestroy(span.second);
    }
    s->ingest_spans.clear();
    if (ctx->sc_parked) {
        s->free_all();
    } else {
        ctx->sc_parked = new ScBuffers(static_cast<ScBuffers&>(*s));
    }
    delete s;
}

// tec_destroy: the run and the parked buffers go.
void tec_sc_release(tec_ctx* ctx) {
    sc_park(ctx);
    if (ctx->sc_parked) {
        ScBuffers* b = static_cast<ScBuffers*>(ctx->sc_parked);
        b->free_all();
        delete b;
        ctx->sc_parked = nullptr;
    }
}

extern "C" int tec_sc_begin(tec_ctx* ctx, int stranded, int mapq_min, uint32_t n_whitelist, const uint8_t* whitelist24,
                            uint64_t bundle_keys, uint32_t n_class, const int32_t* class_index_chrom,
                            const uint8_t* class_skip) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_sc_begin: NULL context");
    if (!ctx->have_index) return tec_fail(ctx, TEC_ESTATE, "tec_sc_begin: no index resident");
    if (n_whitelist == 0 || !whitelist24) return tec_fail(ctx, TEC_EINVAL, "tec_sc_begin: empty whitelist");
    if (n_class && (!class_index_chrom || !class_skip)) return tec_fail(ctx, TEC_EINVAL, "tec_sc_begin: NULL class table");
    if (bundle_keys == 0) return tec_fail(ctx, TEC_EINVAL, "tec_sc_begin: bundle_keys must be positive");
    SC_HIP(hipSetDevice(ctx->device));
    sc_park(ctx);
    tec_sc_state* s = new tec_sc_state();
    if (ctx->sc_parked) {  // buffers parked by the previous run on this context
        ScBuffers* b = static_cast<ScBuffers*>(ctx->sc_parked);
        static_cast<ScBuffers&>(*s) = *b;
        delete b;
        ctx->sc_parked = nullptr;
    }
    ctx->sc = s;
    s->stranded = stranded ? 1 : 0;
    s->mapq_min = mapq_min;
    s->bundle_keys = bundle_keys;
    s->n_wl = n_whitelist;
    s->n_class = n_class;
    const bool same_wl = s->wl_src && s->wl_src->size() == 24 * (size_t)n_whitelist && s->wl_table.p &&
                         memcmp(s->wl_src->data(), whitelist24, 24 * (size_t)n_whitelist) == 0;
    if (same_wl) {
        s->wl_mask = s->wl_src_mask;
    } else {
        s->wl_src.reset();
        std::vector<uint64_t> packed(3 * (size_t)n_whitelist);
        for (size_t i = 0; i < n_whitelist; ++i)
            for (int w = 0; w < 3; ++w) {
                uint64_t v = 0;
                for (int b = 0; b < 8; ++b) v = (v << 8) | whitelist24[24 * i + 8 * w + b];
                packed[3 * i + w] = v;
            }
        for (size_t i = 1; i < n_whitelist; ++i) {
            const uint64_t* a = &packed[3 * (i - 1)];
            const uint64_t* b = &packed[3 * i];
            if (!(a[0] < b[0] || (a[0] == b[0] && (a[1] < b[1] || (a[1] == b[1] && a[2] < b[2])))))
                return tec_fail(ctx, TEC_EINVAL, "tec_sc_begin: whitelist must be strictly ascending (entry %zu)", i);
        }
        {
            uint64_t cap = 16;
            while (cap < 2ull * n_whitelist) cap <<= 1;
            std::vector<uint64_t> table(4 * cap, 0ull);
            const uint32_t mask = (uint32_t)(cap - 1);
            for (uint32_t i = 0; i < n_whitelist; ++i) {
                const uint64_t* k = &packed[3 * (size_t)i];
                uint32_t slot = (uint32_t)wl_hash(k[0], k[1], k[2]) & mask;
                while (table[4 * (size_t)slot + 3]) slot = (slot + 1) & mask;
                table[4 * (size_t)slot] = k[0];
                table[4 * (size_t)slot + 1] = k[1];
                table[4 * (size_t)slot + 2] = k[2];
                table[4 * (size_t)slot + 3] = (uint64_t)i + 1;
            }
            SC_HIP(s->wl_table.reserve(table.size()));
            SC_HIP(hipMemcpy(s->wl_table.p, table.data(), table.size() * 8, hipMemcpyHostToDevice));
            s->wl_mask = mask;
        }
        s->wl_src = std::make_shared<std::vector<uint8_t>>(whitelist24, whitelist24 + 24 * (size_t)n_whitelist);
        s->wl_src_mask = s->wl_mask;
    }
    SC_HIP(s->cls_chrom.reserve(std::max<uint32_t>(n_class, 1)));
    SC_HIP(s->cls_skip.reserve(std::max<uint32_t>(n_class, 1)));
    if (n_class) {
        SC_HIP(hipMemcpy(s->cls_chrom.p, class_index_chrom, n_class * 4, hipMemcpyHostToDevice));
        SC_HIP(hipMemcpy(s->cls_skip.p, class_skip, n_class, hipMemcpyHostToDevice));
    }
    SC_HIP(s->stat.reserve(ST_N));
    unsigned long long init[ST_N] = {0};
    init[ST_NOCB] = init[ST_NOUB] = init[ST_LONGUMI] = ~0ull;
    SC_HIP(hipMemcpy(s->stat.p, init, sizeof init, hipMemcpyHostToDevice));
    return TEC_OK;
}

namespace {
struct PhaseTimer {  // accumulates the stream time between construction and stop()
    tec_ctx* ctx;
    tec_sc_state* s;
    bool on = false;
    PhaseTimer(tec_ctx* c, tec_sc_state* st) : ctx(c), s(st) {
        if (!s->ev0 && hipEventCreate(&s->ev0) != hipSuccess) return;
        if (!s->ev1 && hipEventCreate(&s->ev1) != hipSuccess) return;
        on = hipEventRecord(s->ev0, ctx->s_compute) == hipSuccess;
    }
    void stop() {
        if (!on) return;
        on = false;
        float ms = 0.f;
        if (hipEventRecord(s->ev1, ctx->s_compute) == hipSuccess && hipEventSynchronize(s->ev1) == hipSuccess &&
            hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess)
            s->gpu_ms += ms;
    }
    ~PhaseTimer() { stop(); }
};

// Waits for the batches in flight and brings the host's view up to date: the survivor count from the device's cursor,
// the stream time of the batches into gpu_ms.
int sc_settle(tec_ctx* ctx, tec_sc_state* s) {
    if (!s->in_flight && s->ingest_spans.empty()) return TEC_OK;
    SC_HIP(hipStreamSynchronize(ctx->s_compute));
    unsigned long long live = 0;
    SC_HIP(hipMemcpy(&live, s->stat.p + ST_NLIVE, sizeof live, hipMemcpyDeviceToHost));
    for (auto& span : s->ingest_spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, span.first, span.second) == hipSuccess) s->gpu_ms += ms;
        (void)hipEventDestroy(span.first);
        (void)hipEventDestroy(span.second);
    }
    s->ingest_spans.clear();
    s->in_flight = false;
    if (live >= 0xFFFFFFF0ull) return tec_fail(ctx, TEC_EOVERFLOW, "tec_sc_submit: more than 2^32 surviving records");
    s->n_live = live;
    return TEC_OK;
}

// Filters + order-preserving compaction of one batch that is resident in HBM (te_count.py:394-433), queued on the
// compute stream: the caller's buffers are free once that stream has passed this point (`done`, if given, is recorded
// there).  Nothing here waits for the GPU: where the survivors go is a cursor in device memory, and the survivor arrays
// are sized for every record seen so far to survive.
int sc_ingest(tec_ctx* ctx, tec_sc_state* s, uint64_t n, const int32_t* d_cls, const int32_t* d_pos, const int32_t* d_end,
              const uint16_t* d_flag, const uint8_t* d_mapq, const uint8_t* d_cb, const uint8_t* d_ub, const uint8_t* d_tf,
              hipEvent_t done) {
    hipStream_t st = ctx->s_compute;
    const uint64_t need = s->n_seen + n;  // upper bound of the survivors after this batch
    if (need >= 0xFFFFFFF0ull) return tec_fail(ctx, TEC_EOVERFLOW, "tec_sc_submit: more than 2^32 records");
    // growing a buffer frees memory that queued kernels may still use: settle first (rare: see the floor below)
    const bool grow = n > s->b_live.cap || n > s->b_scan.cap || n > s->b_bcid.cap || need > s->bcid.cap;
    if (grow) {
        int rc = sc_settle(ctx, s);
        if (rc) return rc;
    }
    SC_HIP(s->b_live.reserve(n));
    SC_HIP(s->b_scan.reserve(n));
    SC_HIP(s->b_bcid.reserve(n));
    // survivor arrays: room for sixteen batches like this one (<= 128 M records) at the first allocation, doubling after
    // that -- or, when the caller has said how many records the whole input holds, for all of them (+ 2 %): fresh device
    // memory is scrubbed by the driver at ~9 ms per GB, and regrowth is a copy
    size_t floor = (size_t)std::min<uint64_t>(16 * n, 128ull << 20);
    if (s->expected_records > s->n_seen) floor = (size_t)std::min<uint64_t>(s->expected_records + s->expected_records / 50 + 4096, 0xFFFFFFF0ull);
    const size_t used = (size_t)s->n_seen;  // (an upper bound of what is in use)
    SC_HIP(s->bcid.reserve(need, true, std::min(used, s->bcid.cap), floor));
    SC_HIP(s->u0.reserve(need, true, std::min(used, s->u0.cap), floor));
    SC_HIP(s->u1.reserve(need, true, std::min(used, s->u1.cap), floor));
    SC_HIP(s->u2.reserve(need, true, std::min(used, s->u2.cap), floor));
    SC_HIP(s->rec.reserve(need, true, std::min(used, s->rec.cap), floor));
    SC_HIP(s->klass.reserve(need, true, std::min(used, s->klass.cap), floor));
    SC_HIP(s->left.reserve(need, true, std::min(used, s->left.cap), floor));
    SC_HIP(s->rite.reserve(need, true, std::min(used, s->rite.cap), floor));
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess) (void)hipEventRecord(t0, st);
    hipLaunchKernelGGL(sc_filter_kernel, dim3(std::min(grid_for(n), 2048u)), dim3(kB), 0, st, n, s->n_seen, d_cls, d_flag, d_mapq, d_cb, d_tf,
                       s->mapq_min, reinterpret_cast<const uint4*>(s->wl_table.p), s->wl_mask, s->cls_skip.p, s->n_class, s->b_live.p, s->b_bcid.p,
                       s->stat.p);
    int rc = scan_u32(ctx, s, s->b_live.p, s->b_scan.p, n, false, rocprim::plus<uint32_t>());
    if (rc) return rc;
    hipLaunchKernelGGL(sc_scatter_kernel, dim3(grid_for(n)), dim3(kB), 0, st, n, s->n_seen, s->stat.p + ST_NLIVE, s->b_live.p,
                       s->b_scan.p, s->b_bcid.p, d_cls, d_pos, d_end, d_flag, d_ub, s->stranded, s->bcid.p, s->u0.p, s->u1.p,
                       s->u2.p, s->rec.p, s->klass.p, s->left.p, s->rite.p, s->stat.p);
    hipLaunchKernelGGL(sc_advance_kernel, dim3(1), dim3(64), 0, st, n, s->b_live.p, s->b_scan.p, s->stat.p + ST_NLIVE);
    SC_HIP(hipGetLastError());
    if (t0 && t1 && hipEventRecord(t1, st) == hipSuccess) {
        s->ingest_spans.emplace_back(t0, t1);
    } else {
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
    }
    if (done) SC_HIP(hipEventRecord(done, st));
    s->in_flight = true;
    s->n_seen += n;
    return TEC_OK;
}

// One column of a host batch on its way into the pinned slot.
struct CopyJob {
    char* dst;
    const void* src;
    size_t bytes;
};
void parallel_fill(const CopyJob& job, int threads) {
    const size_t per = ((job.bytes + (size_t)threads - 1) / (size_t)threads + 4095) & ~size_t(4095);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const size_t o = per * (size_t)t;
        if (o < job.bytes) pool.emplace_back([=] { memset(job.dst + o, 0, std::min(per, job.bytes - o)); });
    }
    for (auto& t : pool) t.join();
}
// `threads` workers, each copying its slice of every column (page-cache speed needs more than one core; a thread per
// column and slice would spend more time starting threads than copying)
void parallel_copy(const CopyJob* jobs, int n_jobs, int threads) {
    auto work = [=](int t) {
        for (int j = 0; j < n_jobs; ++j) {
            const size_t per = ((jobs[j].bytes + (size_t)threads - 1) / (size_t)threads + 63) & ~size_t(63);
            const size_t o = per * (size_t)t;
            if (o < jobs[j].bytes) memcpy(jobs[j].dst + o, static_cast<const char*>(jobs[j].src) + o, std::min(per, jobs[j].bytes - o));
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads);
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& t : pool) t.join();
}
}  // namespace

extern "C" int tec_sc_submit(tec_ctx* ctx, uint64_t n, const int32_t* cls, const int32_t* pos, const int32_t* end,
                             const uint16_t* flag, const uint8_t* mapq, const uint8_t* cb24, const uint8_t* ub24,
                             const uint8_t* tagflags) {
    if (!ctx || !ctx->sc) return tec_fail(ctx, TEC_ESTATE, "tec_sc_submit: call tec_sc_begin first");
    tec_sc_state* s = ctx->sc;
    if (s->part1_done) return tec_fail(ctx, TEC_ESTATE, "tec_sc_submit: after tec_sc_part1");
    if (n == 0) return TEC_OK;
    if (!cls || !pos || !end || !flag || !mapq || !cb24 || !ub24 || !tagflags)
        return tec_fail(ctx, TEC_EINVAL, "tec_sc_submit: NULL array");
    if (n >= 0x7FFFFFFFull) return tec_fail(ctx, TEC_EINVAL, "tec_sc_submit: batch too large");
    SC_HIP(hipSetDevice(ctx->device));
#ifdef TEC_SC_TRACE
    const auto te0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        fprintf(stderr, "[sc_submit]   %-22s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te0).count());
    };
#else
    auto lap = [](const char*) {};
#endif
    // the batch as one block: cls | pos | end | flag | mapq | tagflags | cb | ub
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t o_cls = 0, o_pos = o_cls + al(n * 4), o_end = o_pos + al(n * 4), o_flag = o_end + al(n * 4),
                 o_mq = o_flag + al(n * 2), o_tf = o_mq + al(n), o_cb = o_tf + al(n), o_ub = o_cb + al(n * 24),
                 total = o_ub + al(n * 24);
    // ---- 1. the caller's arrays -> a pinned slot (worker threads).  The slot is free once its previous copy to HBM is done
    const int ps = (int)(s->n_submits % 3), ds = (int)(s->n_submits % 2);
    if (!s->pin_done[ps]) SC_HIP(hipEventCreateWithFlags(&s->pin_done[ps], hipEventDisableTiming));
    else SC_HIP(hipEventSynchronize(s->pin_done[ps]));
    if (total > s->pin_cap[ps]) {
        // (pinning costs ~0.1 s per GB: slots are sized for batches of 4 M records once a caller sends large ones, and the
        // whole ring -- three pinned slots, two staging buffers in HBM -- is set up by the first batch that needs it, not
        // one piece per batch over the first three)
        const size_t want = n > 65536 ? std::max(total, (size_t)64 * (4u << 20) + 8 * 256) : total;
        SC_HIP(hipStreamSynchronize(ctx->s_copy));
        SC_HIP(hipStreamSynchronize(ctx->s_compute));
        for (int i = 0; i < 3; ++i) {
            if (s->pin_cap[i] >= want) continue;
            if (s->pin[i]) (void)hipHostFree(s->pin[i]);
            s->pin[i] = nullptr;
            s->pin_cap[i] = 0;
            void* p = nullptr;
            SC_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
            s->pin[i] = static_cast<char*>(p);
            s->pin_cap[i] = want;
            // touch every page now: the first copy into a fresh slot otherwise runs at a third of the speed (10-14 ms
            // instead of 4 ms per 256 MB measured), and the first batches of a run are then what the GPU waits for
            const CopyJob zero = {s->pin[i], nullptr, want};
            parallel_fill(zero, want > (8u << 20) ? 16 : 1);
        }
        SC_HIP(s->b_in.reserve(want));
        SC_HIP(s->b_in2.reserve(want));
    }
    char* h = s->pin[ps];
    lap("slot ready");
    {
        const CopyJob jobs[8] = {{h + o_cls, cls, n * 4}, {h + o_pos, pos, n * 4}, {h + o_end, end, n * 4}, {h + o_flag, flag, n * 2},
                                 {h + o_mq, mapq, n}, {h + o_tf, tagflags, n}, {h + o_cb, cb24, n * 24}, {h + o_ub, ub24, n * 24}};
#ifdef TEC_SC_TRACE
        const auto tc0 = std::chrono::steady_clock::now();
#endif
        parallel_copy(jobs, 8, total > (8u << 20) ? 16 : 1);
#ifdef TEC_SC_TRACE
        fprintf(stderr, "[sc_submit] %zu MB copied into the pinned slot in %.2f ms\n", total >> 20,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
#endif
    }
    // ---- 2. pinned slot -> staging buffer `ds` in HBM on the copy stream, once the ingest that last read that buffer is done
    dvec<char>& din = ds ? s->b_in2 : s->b_in;
    if (!s->in_free[ds]) {
        SC_HIP(hipEventCreateWithFlags(&s->in_free[ds], hipEventDisableTiming));
        SC_HIP(hipEventCreateWithFlags(&s->in_copied[ds], hipEventDisableTiming));
    } else {
        SC_HIP(hipStreamWaitEvent(ctx->s_copy, s->in_free[ds], 0));
    }
    char* b = din.p;
    SC_HIP(hipMemcpyAsync(b, h, total, hipMemcpyHostToDevice, ctx->s_copy));
    SC_HIP(hipEventRecord(s->pin_done[ps], ctx->s_copy));
    SC_HIP(hipEventRecord(s->in_copied[ds], ctx->s_copy));
    // ---- 3. the ingest kernels on the compute stream behind that copy
    SC_HIP(hipStreamWaitEvent(ctx->s_compute, s->in_copied[ds], 0));
    ++s->n_submits;
    lap("copy queued");
    const int rc_ingest = sc_ingest(ctx, s, n, (const int32_t*)(b + o_cls), (const int32_t*)(b + o_pos), (const int32_t*)(b + o_end),
                     (const uint16_t*)(b + o_flag), (const uint8_t*)(b + o_mq), (const uint8_t*)(b + o_cb),
                     (const uint8_t*)(b + o_ub), (const uint8_t*)(b + o_tf), s->in_free[ds]);
    lap("ingest queued");
    return rc_ingest;
}

extern "C" int tec_sc_submit_device(tec_ctx* ctx, uint64_t n, const int32_t* d_cls, const int32_t* d_pos, const int32_t* d_end,
                                    const uint16_t* d_flag, const uint8_t* d_mapq, const uint8_t* d_cb24, const uint8_t* d_ub24,
                                    const uint8_t* d_tagflags) {
    if (!ctx || !ctx->sc) return tec_fail(ctx, TEC_ESTATE, "tec_sc_submit_device: call tec_sc_begin first");
    tec_sc_state* s = ctx->sc;
    if (s->part1_done) return tec_fail(ctx, TEC_ESTATE, "tec_sc_submit_device: after tec_sc_part1");
    if (n == 0) return TEC_OK;
    if (!d_cls || !d_pos || !d_end || !d_flag || !d_mapq || !d_cb24 || !d_ub24 || !d_tagflags)
        return tec_fail(ctx, TEC_EINVAL, "tec_sc_submit_device: NULL array");
    if (n >= 0x7FFFFFFFull) return tec_fail(ctx, TEC_EINVAL, "tec_sc_submit_device: batch too large");
    SC_HIP(hipSetDevice(ctx->device));
    return sc_ingest(ctx, s, n, d_cls, d_pos, d_end, d_flag, d_mapq, d_cb24, d_ub24, d_tagflags, nullptr);
}

int tec_sc_expect_records(tec_ctx* ctx, uint64_t n_records) {
    if (!ctx || !ctx->sc) return TEC_ESTATE;
    ctx->sc->expected_records = n_records;
    return TEC_OK;
}

namespace {

// Part 1, step 1: stable LSD radix sort of a permutation by (bcid, u0, u1, u2); sorted views; key heads.
// The key string of part 1 (see KeyRuns): UMI word 2 lowest, then word 1, word 0, the barcode id on top.  h8 = what
// umi_masks_kernel found.  False when the string is longer than 96 bits or needs more runs than KeyRuns holds (long
// UMIs of arbitrary bytes): the caller then sorts column by column.
bool build_key_runs(const unsigned long long h8[8], unsigned bc_bits, KeyRuns* kr, unsigned* total_bits) {
    const unsigned long long vary[3] = {h8[0] & ~h8[3], h8[1] & ~h8[4], h8[2] & ~h8[5]};
    unsigned dst = 0;
    kr->n = 0;
    kr->ranks = 0;
    auto push = [&](unsigned word, unsigned src, unsigned width, bool symbol) {
        while (width) {
            unsigned take = width;
            if (dst < 64u && dst + take > 64u) {  // no run crosses bit 64 of the string
                if (symbol)
                    dst = 64u;
                else
                    take = 64u - dst;
            }
            if (kr->n == (uint32_t)kMaxKeyRuns || dst + take > 96u) return false;
            KeyRuns::Run& r = kr->r[kr->n++];
            r.mask = take == 64u ? ~0ull : (1ull << take) - 1ull;
            r.word = (uint8_t)word;
            r.src = (uint8_t)src;
            r.dst = (uint8_t)dst;
            r.symbol = symbol ? 1 : 0;
            src += take;
            dst += take;
            width -= take;
        }
        return true;
    };
    if (h8[6]) {
        // every byte is NUL or a letter of ACGNT: a byte position that varies takes the rank of its symbol among the
        // symbols that occur at such positions.  (A letter has bit 6 set, so NUL occurs at a position iff the AND of
        // its bytes has bit 6 clear.)
        static const unsigned lo5[6] = {0u, 1u, 3u, 7u, 14u, 20u};  // NUL A C G N T, ascending byte values
        bool nul = false;
        for (int w = 0; w < 3; ++w)
            for (int k = 0; k < 8; ++k)
                if (((vary[w] >> (8 * k)) & 0xFFu) && !((h8[3 + w] >> (8 * k)) & 0x40u)) nul = true;
        unsigned n_sym = 0;
        for (int y = 0; y < 6; ++y) {
            const bool occurs = y == 0 ? nul : ((h8[7] >> lo5[y]) & 1u) != 0;
            if (occurs) kr->ranks |= (unsigned long long)n_sym++ << (3u * lo5[y]);
        }
        const unsigned sym_bits = n_sym <= 2 ? 1u : n_sym <= 4 ? 2u : 3u;
        for (int w = 2; w >= 0; --w)
            for (int k = 0; k < 8; ++k)
                if ((vary[w] >> (8 * k)) & 0xFFu)
                    if (!push((unsigned)w, 8u * (unsigned)k, sym_bits, true)) return false;
    } else {
        for (int w = 2; w >= 0; --w) {
            unsigned long long m = vary[w];
            while (m) {
                const unsigned first = (unsigned)__builtin_ctzll(m);
                const unsigned long long t = m >> first;
                const unsigned len = t == ~0ull ? 64u : (unsigned)__builtin_ctzll(~t);
                if (!push((unsigned)w, first, len, false)) return false;
                m = len == 64u ? 0ull : m & ~(((1ull << len) - 1ull) << first);
            }
        }
    }
    if (!push(3u, 0u, bc_bits, false)) return false;
    *total_bits = dst;
    return true;
}

// The column-by-column form of part 1's sort: one stable radix round per UMI word that has bits set (least significant
// word first) and one for the barcode id, each behind a gather of its keys, then eight gathers for the sorted views.
// Any UMI sorts this way; sc_part1_sort uses it when the key string does not fit (or under TEC_SC_SORT=columns).
int sc_sort_by_columns(tec_ctx* ctx, tec_sc_state* s) {
    hipStream_t st = ctx->s_compute;
    const uint64_t N = s->n_live;
    const unsigned g = grid_for(N);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kB), 0, st, N, s->perm.p);
    const uint64_t* words[3] = {s->u2.p, s->u1.p, s->u0.p};
    const unsigned long long masks[3] = {s->h_stat[ST_OR2], s->h_stat[ST_OR1], s->h_stat[ST_OR0]};
    for (int w = 0; w < 3; ++w) {
        if (!masks[w]) continue;
        const unsigned b0 = (unsigned)__builtin_ctzll(masks[w]), b1 = 64u - (unsigned)__builtin_clzll(masks[w]);
        hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, words[w], s->k64a.p);
        int rc = sort_pairs(ctx, s, s->k64a.p, s->k64b.p, s->perm.p, s->perm2.p, N, b0, b1);
        if (rc) return rc;
        std::swap(s->perm.p, s->perm2.p);
        std::swap(s->perm.cap, s->perm2.cap);
    }
    {
        hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->bcid.p, s->k32a.p);
        int rc = sort_pairs(ctx, s, s->k32a.p, s->k32b.p, s->perm.p, s->perm2.p, N, 0, bits_for(s->n_wl));
        if (rc) return rc;
        std::swap(s->perm.p, s->perm2.p);
        std::swap(s->perm.cap, s->perm2.cap);
    }
    hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->bcid.p, s->s_bcid.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->u0.p, s->s_u0.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->u1.p, s->s_u1.p);
    if (masks[0])  // (UMIs of at most 16 bytes leave the third word zero everywhere: nothing to gather)
        hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->u2.p, s->s_u2.p);
    else
        SC_HIP(hipMemsetAsync(s->s_u2.p, 0, N * sizeof(uint64_t), st));
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->rec.p, s->s_rec.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->klass.p, s->s_klass.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->left.p, s->s_left.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->rite.p, s->s_rite.p);
    return TEC_OK;
}

// Part 1, step 1: the survivors ordered by (barcode id, UMI), records of one key in file order (a stable sort of rows
// that are in file order).  One streaming pass packs every survivor into a 64-byte row and its key into a bit string
// without the bits that are equal in all rows (10x UMIs: 12 letters of ACGT differ in 4 bits each; with a 20-bit
// barcode id the string has 68 bits against the 212 of the plain columns); a radix sort of the low 64 bits with the row
// number as its value, one more round over the bits above them, and ONE gather of the rows into the sorted columns.
int sc_part1_sort(tec_ctx* ctx, tec_sc_state* s) {
    hipStream_t st = ctx->s_compute;
    SC_HIP(hipStreamSynchronize(st));
    SC_HIP(hipMemcpy(s->h_stat, s->stat.p, sizeof s->h_stat, hipMemcpyDeviceToHost));
    s->sorted = true;
    s->bounds.clear();
    const uint64_t N = s->n_live;
    if (N == 0) return TEC_OK;
    const unsigned g = grid_for(N);
    SC_HIP(s->perm.reserve(N));
    SC_HIP(s->perm2.reserve(N));
    SC_HIP(s->k64a.reserve(N));
    SC_HIP(s->k64b.reserve(N));
    SC_HIP(s->k32a.reserve(N));
    SC_HIP(s->k32b.reserve(N));
    SC_HIP(s->s_bcid.reserve(N));
    SC_HIP(s->s_u0.reserve(N));
    SC_HIP(s->s_u1.reserve(N));
    SC_HIP(s->s_u2.reserve(N));
    SC_HIP(s->s_rec.reserve(N));
    SC_HIP(s->s_klass.reserve(N));
    SC_HIP(s->s_left.reserve(N));
    SC_HIP(s->s_rite.reserve(N));
    const char* how = getenv("TEC_SC_SORT");  // "columns": the column-by-column rounds for every input
    bool by_rows = !(how && !strcmp(how, "columns"));
    KeyRuns kr;
    unsigned key_bits = 0;
    bool heads_done = false;
    if (by_rows) {
        // the bits that differ among the survivors' UMIs
        SC_HIP(s->marks.reserve(64));
        const unsigned long long init[8] = {0ull, 0ull, 0ull, ~0ull, ~0ull, ~0ull, ~0ull, 0ull};
        unsigned long long h[8];
        SC_HIP(hipMemcpyAsync(s->marks.p, init, sizeof init, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(umi_masks_kernel, dim3(std::min(g, 1024u)), dim3(256), 0, st, N, s->u0.p, s->u1.p, s->u2.p,
                           reinterpret_cast<unsigned long long*>(s->marks.p));
        SC_HIP(hipMemcpyAsync(h, s->marks.p, sizeof h, hipMemcpyDeviceToHost, st));
        SC_HIP(hipStreamSynchronize(st));
        by_rows = build_key_runs(h, bits_for(s->n_wl), &kr, &key_bits);
    }
    s->sort_key_bits = by_rows ? key_bits : 0u;
    if (!by_rows) {
        int rc = sc_sort_by_columns(ctx, s);
        if (rc) return rc;
    } else {
        const bool with_umi = key_bits > 64u;
        SC_HIP(s->rows.reserve((with_umi ? 4 : 2) * N));
        SC_HIP(s->ghead.reserve(N));
        if (with_umi)
            hipLaunchKernelGGL(pack_rows_kernel<true>, dim3(g), dim3(256), 0, st, N, s->bcid.p, s->u0.p, s->u1.p, s->u2.p, s->rec.p,
                               s->klass.p, s->left.p, s->rite.p, kr, s->rows.p, s->k64a.p, s->k32a.p);
        else
            hipLaunchKernelGGL(pack_rows_kernel<false>, dim3(g), dim3(256), 0, st, N, s->bcid.p, s->u0.p, s->u1.p, s->u2.p, s->rec.p,
                               s->klass.p, s->left.p, s->rite.p, kr, s->rows.p, s->k64a.p, (uint32_t*)nullptr);
        // (the values are the row numbers 0, 1, 2 ...: an iterator, not an array)
        int rc = sort_pairs(ctx, s, s->k64a.p, s->k64b.p, rocprim::counting_iterator<uint32_t>(0u), s->perm2.p, N, 0, std::min(key_bits, 64u));
        if (rc) return rc;
        std::swap(s->perm.p, s->perm2.p);
        std::swap(s->perm.cap, s->perm2.cap);
        if (with_umi) {
            hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, N, s->perm.p, s->k32a.p, s->k32b.p);
            rc = sort_pairs(ctx, s, s->k32b.p, s->k32a.p, s->perm.p, s->perm2.p, N, 0, key_bits - 64u);
            if (rc) return rc;
            std::swap(s->perm.p, s->perm2.p);
            std::swap(s->perm.cap, s->perm2.cap);
            hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(g), dim3(256), 0, st, N, s->perm.p, s->rows.p, s->s_bcid.p, s->s_u0.p,
                               s->s_u1.p, s->s_u2.p, s->s_rec.p, s->s_klass.p, s->s_left.p, s->s_rite.p, (const uint64_t*)nullptr,
                               (uint8_t*)nullptr, (uint32_t*)nullptr);
        } else {
            hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(g), dim3(256), 0, st, N, s->perm.p, s->rows.p, s->s_bcid.p, s->s_u0.p,
                               s->s_u1.p, s->s_u2.p, s->s_rec.p, s->s_klass.p, s->s_left.p, s->s_rite.p, s->k64b.p, s->ghead.p,
                               s->k32a.p);
            heads_done = true;
        }
    }
    SC_HIP(hipGetLastError());
    // ---- keys (groups)
    SC_HIP(s->ghead.reserve(N));
    SC_HIP(s->gid.reserve(N));
    if (!heads_done)
        hipLaunchKernelGGL(ghead_kernel, dim3(g), dim3(kB), 0, st, N, s->s_bcid.p, s->s_u0.p, s->s_u1.p, s->s_u2.p, s->ghead.p,
                           s->k32a.p);
    int rc = scan_u32(ctx, s, s->k32a.p, s->gid.p, N, true, rocprim::plus<uint32_t>());
    if (rc) return rc;
    uint32_t last_gid = 0;
    SC_HIP(hipMemcpyAsync(&last_gid, s->gid.p + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    s->n_groups = last_gid;  // inclusive scan of heads: the last value is the number of keys
    return TEC_OK;
}

// Part 1, step 2 (repeated): the records that open a NEW (CB,UB) dict entry after record `lower` -- their
// ordinals, ascending, stay in s->marks.  The reference flushes once such entries number bundle_keys
// (te_count.py:377): with the cells dealt over several GPUs the bundle_keys-th one is found by the
// caller over all GPUs' marks (tec_sc_epoch_rank); on one GPU it is marks[bundle_keys - 1].
int sc_epoch_marks(tec_ctx* ctx, tec_sc_state* s, long long lower, uint64_t* n_marks) {
    hipStream_t st = ctx->s_compute;
    const uint64_t N = s->n_live;
    *n_marks = 0;
    s->n_marks = 0;
    if (N == 0) return TEC_OK;
    const unsigned g = grid_for(N);
    hipLaunchKernelGGL(epoch_mark_kernel, dim3(g), dim3(kB), 0, st, N, s->s_rec.p, s->ghead.p, lower, s->k32a.p);
    int rc = scan_u32(ctx, s, s->k32a.p, s->k32b.p, N, false, rocprim::plus<uint32_t>());
    if (rc) return rc;
    uint32_t ls = 0, lf = 0;
    SC_HIP(hipMemcpyAsync(&ls, s->k32b.p + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SC_HIP(hipMemcpyAsync(&lf, s->k32a.p + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    const uint64_t m = (uint64_t)ls + lf;
    s->n_marks = m;
    *n_marks = m;
    if (!m) return TEC_OK;
    SC_HIP(s->k64a.reserve(std::max<uint64_t>(N, m)));
    SC_HIP(s->marks.reserve(m));
    hipLaunchKernelGGL(compact_rec_kernel, dim3(g), dim3(kB), 0, st, N, s->k32a.p, s->k32b.p, s->s_rec.p, s->k64a.p);
    size_t bytes = 0;
    SC_HIP(rocprim::radix_sort_keys(nullptr, bytes, s->k64a.p, s->marks.p, m, 0, bits_for(s->max_rec + 1), st));
    SC_HIP(s->tmp.reserve(bytes));
    SC_HIP(rocprim::radix_sort_keys(s->tmp.p, bytes, s->k64a.p, s->marks.p, m, 0, bits_for(s->max_rec + 1), st));
    SC_HIP(hipStreamSynchronize(st));  // callers read marks with plain hipMemcpy, which does not wait for this (non-blocking) stream
    return TEC_OK;
}

// Part 1, step 3: bundles from the flush boundaries, (bundle, key) instances, duplicates vs extra
// strings, per-barcode counts and first-seen ordinals.
int sc_part1_finish(tec_ctx* ctx, tec_sc_state* s, const std::vector<uint64_t>& bounds) {
    hipStream_t st = ctx->s_compute;
    const uint64_t N = s->n_live;
    s->bounds = bounds;
    s->n_epochs = (uint32_t)bounds.size() + 1;
    s->part1_done = true;
    s->cells_seen.clear();
    s->cells_count.clear();
    s->cells_first.clear();
    if (N == 0) return TEC_OK;
    const unsigned g = grid_for(N);
    SC_HIP(s->s_epoch.reserve(N));
    if (bounds.empty()) {
        SC_HIP(hipMemsetAsync(s->s_epoch.p, 0, N * 4, st));
    } else {
        SC_HIP(s->k64b.reserve(std::max<size_t>(N, bounds.size())));
        SC_HIP(hipMemcpyAsync(s->k64b.p, bounds.data(), bounds.size() * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(epoch_assign_kernel, dim3(g), dim3(kB), 0, st, N, s->s_rec.p, s->k64b.p, (uint32_t)bounds.size(),
                           s->s_epoch.p);
        SC_HIP(hipStreamSynchronize(st));
    }
    SC_HIP(s->ihead.reserve(N));
    SC_HIP(s->headidx.reserve(N));
    SC_HIP(s->counted.reserve(N));
    hipLaunchKernelGGL(ihead_kernel, dim3(g), dim3(kB), 0, st, N, s->ghead.p, s->s_epoch.p, s->ihead.p, s->k32a.p);
    int rc = scan_u32(ctx, s, s->k32a.p, s->headidx.p, N, true, rocprim::maximum<uint32_t>());
    if (rc) return rc;
    SC_HIP(s->cnt.reserve(s->n_wl));
    SC_HIP(s->first_seen.reserve(s->n_wl));
    SC_HIP(hipMemsetAsync(s->cnt.p, 0, (size_t)s->n_wl * 4, st));
    SC_HIP(hipMemsetAsync(s->first_seen.p, 0xFF, (size_t)s->n_wl * 8, st));
    hipLaunchKernelGGL(part1_count_kernel, dim3(g), dim3(kB), 0, st, N, s->ihead.p, s->headidx.p, s->s_klass.p, s->s_bcid.p,
                       s->s_rec.p, s->counted.p, s->cnt.p, s->first_seen.p);
    SC_HIP(hipGetLastError());
    std::vector<uint32_t> hc(s->n_wl);
    std::vector<unsigned long long> hf(s->n_wl);
    SC_HIP(hipMemcpyAsync(hc.data(), s->cnt.p, (size_t)s->n_wl * 4, hipMemcpyDeviceToHost, st));
    SC_HIP(hipMemcpyAsync(hf.data(), s->first_seen.p, (size_t)s->n_wl * 8, hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    for (uint32_t c = 0; c < s->n_wl; ++c)
        if (hc[c]) {
            s->cells_seen.push_back(c);
            s->cells_count.push_back(hc[c]);
            s->cells_first.push_back(hf[c]);
        }
    return TEC_OK;
}

int sc_check_open(tec_ctx* ctx, const char* who, bool need_sorted) {
    if (!ctx || !ctx->sc) return tec_fail(ctx, TEC_ESTATE, "%s: call tec_sc_begin first", who);
    if (ctx->sc->part1_done) return tec_fail(ctx, TEC_ESTATE, "%s: part 1 is already finished", who);
    if (need_sorted && !ctx->sc->sorted) return tec_fail(ctx, TEC_ESTATE, "%s: call tec_sc_part1_sort first", who);
    return sc_settle(ctx, ctx->sc);  // (batches may still be in flight: the survivor count is the device's until then)
}

}  // namespace

// ---- the survivors of this GPU's records, for the multi-GPU exchange (cells are dealt to GPUs by
// barcode id, so every (CB,UB) key and every cell row is then local to one GPU)
extern "C" int tec_sc_survivors(tec_ctx* ctx, uint64_t* n) {
    if (!ctx || !ctx->sc || !n) return tec_fail(ctx, TEC_ESTATE, "tec_sc_survivors: call tec_sc_begin first");
    SC_HIP(hipSetDevice(ctx->device));
    int rc = sc_settle(ctx, ctx->sc);
    if (rc) return rc;
    *n = ctx->sc->n_live;
    return TEC_OK;
}

extern "C" int tec_sc_survivors_fetch(tec_ctx* ctx, uint32_t* bcid, uint64_t* u0, uint64_t* u1, uint64_t* u2, uint64_t* rec,
                                      int32_t* klass, int32_t* left, int32_t* rite) {
    int rc = sc_check_open(ctx, "tec_sc_survivors_fetch", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    const uint64_t n = s->n_live;
    if (!n) return TEC_OK;
    if (!bcid || !u0 || !u1 || !u2 || !rec || !klass || !left || !rite) return tec_fail(ctx, TEC_EINVAL, "tec_sc_survivors_fetch: NULL array");
    SC_HIP(hipSetDevice(ctx->device));
    SC_HIP(hipStreamSynchronize(ctx->s_compute));
    s->host_row_bytes += n * 52;
    SC_HIP(hipMemcpy(bcid, s->bcid.p, n * 4, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(u0, s->u0.p, n * 8, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(u1, s->u1.p, n * 8, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(u2, s->u2.p, n * 8, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(rec, s->rec.p, n * 8, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(klass, s->klass.p, n * 4, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(left, s->left.p, n * 4, hipMemcpyDeviceToHost));
    SC_HIP(hipMemcpy(rite, s->rite.p, n * 4, hipMemcpyDeviceToHost));
    return TEC_OK;
}

// Replaces the survivor set (rows in any order: they are sorted by record ordinal on the device, which
// is the file order every later step relies on).
extern "C" int tec_sc_survivors_replace(tec_ctx* ctx, uint64_t n, const uint32_t* bcid, const uint64_t* u0, const uint64_t* u1,
                                        const uint64_t* u2, const uint64_t* rec, const int32_t* klass, const int32_t* left,
                                        const int32_t* rite) {
    int rc = sc_check_open(ctx, "tec_sc_survivors_replace", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (s->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_survivors_replace: after tec_sc_part1_sort");
    if (n && (!bcid || !u0 || !u1 || !u2 || !rec || !klass || !left || !rite)) return tec_fail(ctx, TEC_EINVAL, "tec_sc_survivors_replace: NULL array");
    if (n >= 0xFFFFFFF0ull) return tec_fail(ctx, TEC_EOVERFLOW, "tec_sc_survivors_replace: more than 2^32 rows");
    SC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->s_compute;
    SC_HIP(hipStreamSynchronize(st));
    unsigned long long ors[3] = {0, 0, 0}, max_rec = 0;
    for (uint64_t i = 0; i < n; ++i) {
        ors[0] |= u0[i];
        ors[1] |= u1[i];
        ors[2] |= u2[i];
        if (rec[i] > max_rec) max_rec = rec[i];
    }
    if (max_rec + 1 > s->max_rec) s->max_rec = max_rec + 1;
    SC_HIP(hipMemcpy(s->stat.p + ST_OR0, ors, sizeof ors, hipMemcpyHostToDevice));
    s->n_live = n;
    if (!n) return TEC_OK;
    SC_HIP(s->bcid.reserve(n)); SC_HIP(s->u0.reserve(n)); SC_HIP(s->u1.reserve(n)); SC_HIP(s->u2.reserve(n));
    SC_HIP(s->rec.reserve(n)); SC_HIP(s->klass.reserve(n)); SC_HIP(s->left.reserve(n)); SC_HIP(s->rite.reserve(n));
    SC_HIP(s->perm.reserve(n)); SC_HIP(s->perm2.reserve(n)); SC_HIP(s->k64a.reserve(n)); SC_HIP(s->k64b.reserve(n));
    SC_HIP(s->k32a.reserve(n));
    const unsigned g = grid_for(n);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kB), 0, st, n, s->perm.p);
    SC_HIP(hipMemcpyAsync(s->k64a.p, rec, n * 8, hipMemcpyHostToDevice, st));
    rc = sort_pairs(ctx, s, s->k64a.p, s->rec.p, s->perm.p, s->perm2.p, n, 0, bits_for(max_rec + 1));
    if (rc) return rc;
    // the other columns follow the permutation (through the two scratch columns)
    auto col64 = [&](const uint64_t* h, uint64_t* d) -> int {
        SC_HIP(hipMemcpyAsync(s->k64a.p, h, n * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->k64a.p, d);
        return TEC_OK;
    };
    auto col32 = [&](const void* h, void* d) -> int {
        SC_HIP(hipMemcpyAsync(s->k32a.p, h, n * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->k32a.p, static_cast<uint32_t*>(d));
        return TEC_OK;
    };
    if ((rc = col64(u0, s->u0.p)) || (rc = col64(u1, s->u1.p)) || (rc = col64(u2, s->u2.p))) return rc;
    if ((rc = col32(bcid, s->bcid.p)) || (rc = col32(klass, s->klass.p)) || (rc = col32(left, s->left.p)) || (rc = col32(rite, s->rite.p))) return rc;
    SC_HIP(hipStreamSynchronize(st));
    return TEC_OK;
}

// ---- the same exchange without the host: the survivors are grouped by destination on the device (tec_sc_deal), the
// caller moves the groups between GPUs (RCCL point-to-point on tensors wrapped over the deal / landing columns, or
// tec_sc_adopt_copy between contexts of one process), tec_sc_adopt_commit orders what arrived by record ordinal.
// Columns are numbered as tec_sc_survivors_fetch lists them: bcid u0 u1 u2 rec klass left rite.
extern "C" int tec_sc_deal(tec_ctx* ctx, uint32_t world, uint64_t* counts) {
    int rc = sc_check_open(ctx, "tec_sc_deal", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (s->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_deal: after tec_sc_part1_sort");
    if (!counts || world == 0 || world > 64) return tec_fail(ctx, TEC_EINVAL, "tec_sc_deal: 1 <= world <= 64");
    SC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->s_compute;
    for (uint32_t d = 0; d < world; ++d) counts[d] = 0;
    const uint64_t n = s->n_live;
    if (!n) return TEC_OK;
    const unsigned g = grid_for(n);
    SC_HIP(s->deal.reserve(n));
    SC_HIP(s->perm.reserve(n)); SC_HIP(s->perm2.reserve(n)); SC_HIP(s->k32a.reserve(n)); SC_HIP(s->k32b.reserve(n));
    SC_HIP(s->marks.reserve(64));
    SC_HIP(hipMemsetAsync(s->marks.p, 0, 64 * sizeof(uint64_t), st));
    hipLaunchKernelGGL(deal_key_kernel, dim3(g), dim3(kB), 0, st, n, s->bcid.p, world, s->k32a.p, reinterpret_cast<unsigned long long*>(s->marks.p));
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kB), 0, st, n, s->perm.p);
    // stable: inside a destination's group the rows keep their file order
    rc = sort_pairs(ctx, s, s->k32a.p, s->k32b.p, s->perm.p, s->perm2.p, n, 0, bits_for(world));
    if (rc) return rc;
    hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->bcid.p, s->deal.bcid.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->u0.p, s->deal.u0.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->u1.p, s->deal.u1.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->u2.p, s->deal.u2.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->rec.p, s->deal.rec.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->klass.p, s->deal.klass.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->left.p, s->deal.left.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->rite.p, s->deal.rite.p);
    SC_HIP(hipGetLastError());
    unsigned long long h[64];
    SC_HIP(hipMemcpyAsync(h, s->marks.p, world * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));  // (the caller's transport reads the deal columns from other streams)
    for (uint32_t d = 0; d < world; ++d) counts[d] = h[d];
    return TEC_OK;
}

extern "C" int tec_sc_deal_column(tec_ctx* ctx, int column, void** d_ptr, uint32_t* elem_bytes) {
    if (!ctx || !ctx->sc || !d_ptr || !elem_bytes) return tec_fail(ctx, TEC_ESTATE, "tec_sc_deal_column: no single-cell run");
    *d_ptr = ctx->sc->deal.column(column, elem_bytes);
    if (!*elem_bytes) return tec_fail(ctx, TEC_EINVAL, "tec_sc_deal_column: columns are 0..7");
    return TEC_OK;
}

extern "C" int tec_sc_adopt_begin(tec_ctx* ctx, uint64_t n_rows) {
    int rc = sc_check_open(ctx, "tec_sc_adopt_begin", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (s->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_adopt_begin: after tec_sc_part1_sort");
    if (n_rows >= 0xFFFFFFF0ull) return tec_fail(ctx, TEC_EOVERFLOW, "tec_sc_adopt_begin: more than 2^32 rows");
    SC_HIP(hipSetDevice(ctx->device));
    SC_HIP(s->land.reserve(std::max<uint64_t>(n_rows, 1)));
    s->n_land = n_rows;
    return TEC_OK;
}

extern "C" int tec_sc_adopt_column(tec_ctx* ctx, int column, void** d_ptr, uint32_t* elem_bytes) {
    if (!ctx || !ctx->sc || !d_ptr || !elem_bytes) return tec_fail(ctx, TEC_ESTATE, "tec_sc_adopt_column: no single-cell run");
    *d_ptr = ctx->sc->land.column(column, elem_bytes);
    if (!*elem_bytes) return tec_fail(ctx, TEC_EINVAL, "tec_sc_adopt_column: columns are 0..7");
    return TEC_OK;
}

// n_rows of a column from device memory of this process (a peer context's deal column, or this context's own) into the
// landing column from row dst_row on.
extern "C" int tec_sc_adopt_copy(tec_ctx* ctx, int column, uint64_t dst_row, const void* d_src, uint64_t n_rows) {
    if (!ctx || !ctx->sc) return tec_fail(ctx, TEC_ESTATE, "tec_sc_adopt_copy: no single-cell run");
    tec_sc_state* s = ctx->sc;
    uint32_t elem = 0;
    char* dst = static_cast<char*>(s->land.column(column, &elem));
    if (!elem) return tec_fail(ctx, TEC_EINVAL, "tec_sc_adopt_copy: columns are 0..7");
    if (dst_row + n_rows > s->n_land) return tec_fail(ctx, TEC_EINVAL, "tec_sc_adopt_copy: beyond the rows announced to tec_sc_adopt_begin");
    if (!n_rows) return TEC_OK;
    if (!d_src) return tec_fail(ctx, TEC_EINVAL, "tec_sc_adopt_copy: NULL source");
    SC_HIP(hipSetDevice(ctx->device));
    SC_HIP(hipMemcpyAsync(dst + dst_row * elem, d_src, n_rows * elem, hipMemcpyDeviceToDevice, ctx->s_compute));
    SC_HIP(hipStreamSynchronize(ctx->s_compute));  // (the source may be another context's buffer: done before that context moves on)
    return TEC_OK;
}

extern "C" int tec_sc_adopt_commit(tec_ctx* ctx) {
    int rc = sc_check_open(ctx, "tec_sc_adopt_commit", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (s->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_adopt_commit: after tec_sc_part1_sort");
    SC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->s_compute;
    const uint64_t n = s->n_land;
    SC_HIP(s->marks.reserve(64));
    SC_HIP(hipMemsetAsync(s->marks.p, 0, 4 * sizeof(uint64_t), st));
    if (n) hipLaunchKernelGGL(land_summary_kernel, dim3(grid_for(n)), dim3(kB), 0, st, n, s->land.u0.p, s->land.u1.p, s->land.u2.p, s->land.rec.p,
                              reinterpret_cast<unsigned long long*>(s->marks.p));
    unsigned long long h[4] = {0, 0, 0, 0};
    SC_HIP(hipMemcpyAsync(h, s->marks.p, sizeof h, hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    if (n && h[3] + 1 > s->max_rec) s->max_rec = h[3] + 1;
    SC_HIP(hipMemcpy(s->stat.p + ST_OR0, h, 3 * sizeof(unsigned long long), hipMemcpyHostToDevice));
    s->n_live = n;
    if (!n) return TEC_OK;
    SC_HIP(s->bcid.reserve(n)); SC_HIP(s->u0.reserve(n)); SC_HIP(s->u1.reserve(n)); SC_HIP(s->u2.reserve(n));
    SC_HIP(s->rec.reserve(n)); SC_HIP(s->klass.reserve(n)); SC_HIP(s->left.reserve(n)); SC_HIP(s->rite.reserve(n));
    SC_HIP(s->perm.reserve(n)); SC_HIP(s->perm2.reserve(n));
    const unsigned g = grid_for(n);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(kB), 0, st, n, s->perm.p);
    // file order = ascending record ordinal (the groups arrive per source GPU, each in its own file order)
    rc = sort_pairs(ctx, s, s->land.rec.p, s->rec.p, s->perm.p, s->perm2.p, n, 0, bits_for(h[3] + 1));
    if (rc) return rc;
    hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.bcid.p, s->bcid.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.u0.p, s->u0.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.u1.p, s->u1.p);
    hipLaunchKernelGGL(gather_kernel<uint64_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.u2.p, s->u2.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.klass.p, s->klass.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.left.p, s->left.p);
    hipLaunchKernelGGL(gather_kernel<int32_t>, dim3(g), dim3(kB), 0, st, n, s->perm2.p, s->land.rite.p, s->rite.p);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(st));
    return TEC_OK;
}

// Bytes of survivor columns that tec_sc_survivors_fetch has copied to host memory in this run (0 when the exchange ran
// on the device).
extern "C" int tec_sc_host_row_bytes(tec_ctx* ctx, uint64_t* bytes) {
    if (!ctx || !ctx->sc || !bytes) return tec_fail(ctx, TEC_ESTATE, "tec_sc_host_row_bytes: no single-cell run");
    *bytes = ctx->sc->host_row_bytes;
    return TEC_OK;
}

// ---- part 1 in steps (multi-GPU callers; tec_sc_part1 below runs them for one GPU)
extern "C" int tec_sc_set_record_base(tec_ctx* ctx, uint64_t base, uint64_t total_records) {
    int rc = sc_check_open(ctx, "tec_sc_set_record_base", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (s->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_set_record_base: after tec_sc_part1_sort");
    SC_HIP(hipSetDevice(ctx->device));
    if (base && s->n_live) hipLaunchKernelGGL(add_base_kernel, dim3(grid_for(s->n_live)), dim3(kB), 0, ctx->s_compute, s->n_live, s->rec.p, base);
    SC_HIP(hipGetLastError());
    s->rec_base += base;
    s->max_rec = std::max<uint64_t>(total_records, s->rec_base + s->n_seen);
    return TEC_OK;
}

extern "C" int tec_sc_part1_sort(tec_ctx* ctx, uint64_t* n_keys) {
    int rc = sc_check_open(ctx, "tec_sc_part1_sort", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    SC_HIP(hipSetDevice(ctx->device));
    if (s->max_rec < s->rec_base + s->n_seen) s->max_rec = s->rec_base + s->n_seen;
    PhaseTimer timer(ctx, s);
    rc = sc_part1_sort(ctx, s);
    if (rc) return rc;
    if (n_keys) *n_keys = s->n_groups;
    return TEC_OK;
}

/* ordinals (ST_NOCB / ST_NOUB) of the first record that lacks a barcode / UMI tag, ~0 if none: the
 * reference dies at the smaller one (te_count.py:409, 426); valid after tec_sc_part1_sort */
extern "C" int tec_sc_sort_key_bits(tec_ctx* ctx, uint32_t* bits) {
    if (!ctx || !ctx->sc || !bits) return tec_fail(ctx, TEC_ESTATE, "tec_sc_sort_key_bits: no single-cell run");
    if (!ctx->sc->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_sort_key_bits: call tec_sc_part1_sort first");
    *bits = ctx->sc->sort_key_bits;
    return TEC_OK;
}

extern "C" int tec_sc_missing_tags(tec_ctx* ctx, uint64_t* first_no_cb, uint64_t* first_no_ub) {
    if (!ctx || !ctx->sc || !ctx->sc->sorted) return tec_fail(ctx, TEC_ESTATE, "tec_sc_missing_tags: call tec_sc_part1_sort first");
    tec_sc_state* s = ctx->sc;
    const unsigned long long none = ~0ull;
    if (first_no_cb) *first_no_cb = s->h_stat[ST_NOCB] == none ? none : s->h_stat[ST_NOCB] + s->rec_base;
    if (first_no_ub) *first_no_ub = s->h_stat[ST_NOUB] == none ? none : s->h_stat[ST_NOUB] + s->rec_base;
    return TEC_OK;
}

/* ordinal of the first record that reaches the (CB,UB) collapse with a UMI longer than the 24-byte field (tagflags bit
 * 3), ~0 if none: such a run is refused (TEC_EOVERFLOW from tec_sc_part1); valid after tec_sc_part1_sort */
extern "C" int tec_sc_long_umi(tec_ctx* ctx, uint64_t* first) {
    if (!ctx || !ctx->sc || !ctx->sc->sorted || !first) return tec_fail(ctx, TEC_ESTATE, "tec_sc_long_umi: call tec_sc_part1_sort first");
    tec_sc_state* s = ctx->sc;
    *first = s->h_stat[ST_LONGUMI] == ~0ull ? ~0ull : s->h_stat[ST_LONGUMI] + s->rec_base;
    return TEC_OK;
}

extern "C" int tec_sc_epoch_marks(tec_ctx* ctx, int64_t lower, uint64_t* n_marks) {
    int rc = sc_check_open(ctx, "tec_sc_epoch_marks", true);
    if (rc) return rc;
    if (!n_marks) return tec_fail(ctx, TEC_EINVAL, "tec_sc_epoch_marks: NULL argument");
    SC_HIP(hipSetDevice(ctx->device));
    PhaseTimer timer(ctx, ctx->sc);
    return sc_epoch_marks(ctx, ctx->sc, lower, n_marks);
}

extern "C" int tec_sc_epoch_rank(tec_ctx* ctx, uint64_t v, uint64_t* count_le) {
    int rc = sc_check_open(ctx, "tec_sc_epoch_rank", true);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (!count_le) return tec_fail(ctx, TEC_EINVAL, "tec_sc_epoch_rank: NULL argument");
    *count_le = 0;
    if (!s->n_marks) return TEC_OK;
    SC_HIP(hipSetDevice(ctx->device));
    unsigned long long* d = s->stat.p + ST_N - 1;  // scratch word (ST_NOSTRAND is reset before part 3)
    hipLaunchKernelGGL(upper_bound_kernel, dim3(1), dim3(1), 0, ctx->s_compute, s->marks.p, s->n_marks, v, d);
    unsigned long long h = 0;
    SC_HIP(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->s_compute));
    SC_HIP(hipStreamSynchronize(ctx->s_compute));
    const unsigned long long zero = 0;
    SC_HIP(hipMemcpyAsync(d, &zero, 8, hipMemcpyHostToDevice, ctx->s_compute));
    *count_le = h;
    return TEC_OK;
}

extern "C" int tec_sc_epoch_kth(tec_ctx* ctx, uint64_t k, uint64_t* ordinal) {
    int rc = sc_check_open(ctx, "tec_sc_epoch_kth", true);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    if (!ordinal || k == 0 || k > s->n_marks) return tec_fail(ctx, TEC_EINVAL, "tec_sc_epoch_kth: k out of range");
    SC_HIP(hipSetDevice(ctx->device));
    SC_HIP(hipMemcpy(ordinal, s->marks.p + (k - 1), 8, hipMemcpyDeviceToHost));
    return TEC_OK;
}

extern "C" int tec_sc_part1_finish(tec_ctx* ctx, uint32_t n_bounds, const uint64_t* bounds, uint64_t* n_cells_seen) {
    int rc = sc_check_open(ctx, "tec_sc_part1_finish", true);
    if (rc) return rc;
    if (n_bounds && !bounds) return tec_fail(ctx, TEC_EINVAL, "tec_sc_part1_finish: NULL bounds");
    SC_HIP(hipSetDevice(ctx->device));
    PhaseTimer timer(ctx, ctx->sc);
    rc = sc_part1_finish(ctx, ctx->sc, std::vector<uint64_t>(bounds, bounds + n_bounds));
    if (rc) return rc;
    if (n_cells_seen) *n_cells_seen = ctx->sc->cells_seen.size();
    return TEC_OK;
}

extern "C" int tec_sc_part1(tec_ctx* ctx, uint64_t* n_cells_seen) {
    int rc = sc_check_open(ctx, "tec_sc_part1", false);
    if (rc) return rc;
    tec_sc_state* s = ctx->sc;
    SC_HIP(hipSetDevice(ctx->device));
    if (n_cells_seen) *n_cells_seen = 0;
    if (!s->sorted) {
        rc = tec_sc_part1_sort(ctx, nullptr);
        if (rc) return rc;
    }
    // the reference dies at the first record (file order) that lacks a tag
    if (s->h_stat[ST_NOCB] != ~0ull || s->h_stat[ST_NOUB] != ~0ull) {
        s->part1_done = true;
        if (s->h_stat[ST_NOCB] < s->h_stat[ST_NOUB])
            return tec_fail(ctx, TEC_ENOCB, "CB or CR tag not found! (record %llu)", s->h_stat[ST_NOCB]);
        return tec_fail(ctx, TEC_ENOUB, "UB or UR tag not found! (record %llu)", s->h_stat[ST_NOUB]);
    }
    if (s->h_stat[ST_LONGUMI] != ~0ull) {
        s->part1_done = true;
        return tec_fail(ctx, TEC_EOVERFLOW, "UMI longer than 24 bytes is not supported (record %llu)", s->h_stat[ST_LONGUMI]);
    }
    PhaseTimer timer(ctx, s);
    // bundles (te_count.py:377-390): the flush fires once len(umis) has reached bundle_keys
    std::vector<uint64_t> bounds;
    if ((uint64_t)s->n_groups >= s->bundle_keys) {
        long long lower = -1;
        for (;;) {
            uint64_t m = 0;
            rc = sc_epoch_marks(ctx, s, lower, &m);
            if (rc) return rc;
            if (m < s->bundle_keys) break;
            uint64_t b = 0;
            SC_HIP(hipMemcpy(&b, s->marks.p + (s->bundle_keys - 1), 8, hipMemcpyDeviceToHost));
            bounds.push_back(b);
            lower = (long long)b;
            if (bounds.size() > s->n_live / s->bundle_keys + 2) return tec_fail(ctx, TEC_EHIP, "tec_sc_part1: epoch loop runaway");
        }
    }
    rc = sc_part1_finish(ctx, s, bounds);
    if (rc) return rc;
    if (n_cells_seen) *n_cells_seen = s->cells_seen.size();
    return TEC_OK;
}

extern "C" int tec_sc_part1_fetch(tec_ctx* ctx, uint32_t* bcid, uint64_t* count, uint64_t* first_seen) {
    if (!ctx || !ctx->sc || !ctx->sc->part1_done) return tec_fail(ctx, TEC_ESTATE, "tec_sc_part1_fetch: run tec_sc_part1 first");
    tec_sc_state* s = ctx->sc;
    const size_t n = s->cells_seen.size();
    if (n && bcid) memcpy(bcid, s->cells_seen.data(), n * 4);
    if (n && count) memcpy(count, s->cells_count.data(), n * 8);
    if (n && first_seen) memcpy(first_seen, s->cells_first.data(), n * 8);
    return TEC_OK;
}

namespace {
// Bundle lines = (bundle, key) instance heads in (bundle, barcode, UMI) order -> lines_out (device), count.
int sc_lines(tec_ctx* ctx, tec_sc_state* s, dvec<uint32_t>& lines, dvec<uint32_t>& lkeys, dvec<uint32_t>& lkeys2,
             dvec<uint32_t>& lines2, uint64_t* n_lines_out) {
    hipStream_t st = ctx->s_compute;
    const uint64_t N = s->n_live;
    const unsigned g = grid_for(N);
    SC_HIP(s->k32b.reserve(N));
    int rc = scan_u32(ctx, s, rocprim::make_transform_iterator(s->ihead.p, U8ToU32()), s->k32b.p, N, false, rocprim::plus<uint32_t>());
    if (rc) return rc;
    uint32_t ls = 0;
    uint8_t lf = 0;
    SC_HIP(hipMemcpyAsync(&ls, s->k32b.p + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SC_HIP(hipMemcpyAsync(&lf, s->ihead.p + (N - 1), 1, hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    const uint64_t n_lines = (uint64_t)ls + (lf ? 1u : 0u);
    *n_lines_out = n_lines;
    SC_HIP(lines.reserve(n_lines));
    hipLaunchKernelGGL(compact_idx_kernel<uint8_t>, dim3(g), dim3(kB), 0, st, N, s->ihead.p, s->k32b.p, lines.p);
    if (s->n_epochs > 1) {
        SC_HIP(lkeys.reserve(n_lines));
        SC_HIP(lkeys2.reserve(n_lines));
        SC_HIP(lines2.reserve(n_lines));
        hipLaunchKernelGGL(gather_kernel<uint32_t>, dim3(grid_for(n_lines)), dim3(kB), 0, st, n_lines, lines.p, s->s_epoch.p, lkeys.p);
        rc = sort_pairs(ctx, s, lkeys.p, lkeys2.p, lines.p, lines2.p, n_lines, 0, bits_for(s->n_epochs));
        if (rc) return rc;
        std::swap(lines.p, lines2.p);
        std::swap(lines.cap, lines2.cap);
    }
    return TEC_OK;
}
}  // namespace

// Which (bundle, barcode) pairs have bundle lines on this GPU: present[e * n_whitelist + b] in {0, 1},
// n_bundles * n_whitelist entries.  Multi-GPU callers OR the tables of all GPUs and hand the result to
// tec_sc_part23_global (the drop rule looks at a line's predecessor in the WHOLE bundle).
extern "C" int tec_sc_presence(tec_ctx* ctx, uint32_t n_bundles, uint8_t* present) {
    if (!ctx || !ctx->sc || !ctx->sc->part1_done) return tec_fail(ctx, TEC_ESTATE, "tec_sc_presence: finish part 1 first");
    tec_sc_state* s = ctx->sc;
    if (!present || n_bundles < s->n_epochs) return tec_fail(ctx, TEC_EINVAL, "tec_sc_presence: table too small");
    SC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->s_compute;
    const size_t cells = (size_t)n_bundles * s->n_wl;
    memset(present, 0, cells);
    if (s->n_live == 0) return TEC_OK;
    dvec<uint32_t> lines, lkeys, lkeys2, lines2, pres;
    auto cleanup = [&]() { lines.release(); lkeys.release(); lkeys2.release(); lines2.release(); pres.release(); };
    uint64_t n_lines = 0;
    int rc = sc_lines(ctx, s, lines, lkeys, lkeys2, lines2, &n_lines);
    hipError_t e = rc ? hipSuccess : pres.reserve(cells);
    if (!rc && e == hipSuccess) e = hipMemsetAsync(pres.p, 0, cells * 4, st);
    if (!rc && e == hipSuccess) {
        hipLaunchKernelGGL(presence_kernel, dim3(grid_for(n_lines)), dim3(kB), 0, st, n_lines, lines.p, s->s_bcid.p, s->s_epoch.p,
                           s->n_wl, pres.p);
        std::vector<uint32_t> h(cells);
        e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(h.data(), pres.p, cells * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < cells; ++i) present[i] = h[i] ? 1 : 0;
    }
    cleanup();
    if (rc) return rc;
    if (e != hipSuccess) return tec_fail(ctx, TEC_EHIP, "tec_sc_presence: %s", hipGetErrorString(e));
    return TEC_OK;
}

static int sc_part23_impl(tec_ctx* ctx, uint32_t n_selected, const uint32_t* selected_bcid, uint32_t n_bundles,
                          const uint8_t* present_global, uint64_t* kept_lines, uint64_t* n_cells_hit, uint64_t* n_coo);

extern "C" int tec_sc_part23(tec_ctx* ctx, uint32_t n_selected, const uint32_t* selected_bcid, uint64_t* kept_lines,
                             uint64_t* n_cells_hit, uint64_t* n_coo) {
    return sc_part23_impl(ctx, n_selected, selected_bcid, 0, nullptr, kept_lines, n_cells_hit, n_coo);
}

extern "C" int tec_sc_part23_global(tec_ctx* ctx, uint32_t n_selected, const uint32_t* selected_bcid, uint32_t n_bundles,
                                    const uint8_t* present, uint64_t* kept_lines, uint64_t* n_cells_hit, uint64_t* n_coo) {
    if (!present) return tec_fail(ctx, TEC_EINVAL, "tec_sc_part23_global: NULL presence table");
    return sc_part23_impl(ctx, n_selected, selected_bcid, n_bundles, present, kept_lines, n_cells_hit, n_coo);
}

static int sc_part23_impl(tec_ctx* ctx, uint32_t n_selected, const uint32_t* selected_bcid, uint32_t n_bundles,
                          const uint8_t* present_global, uint64_t* kept_lines, uint64_t* n_cells_hit, uint64_t* n_coo) {
    if (!ctx || !ctx->sc || !ctx->sc->part1_done) return tec_fail(ctx, TEC_ESTATE, "tec_sc_part23: run tec_sc_part1 first");
    tec_sc_state* s = ctx->sc;
    SC_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->s_compute;
    s->out_cell.clear(); s->out_hits.clear();
    s->n_coo = 0;
    s->kept_lines = s->n_queries = s->n_assigned = 0;
    if (kept_lines) *kept_lines = 0;
    if (n_cells_hit) *n_cells_hit = 0;
    if (n_coo) *n_coo = 0;
    const uint64_t N = s->n_live;
    if (N == 0 || n_selected == 0) return TEC_OK;
    PhaseTimer timer(ctx, s);
    for (uint32_t i = 0; i < n_selected; ++i)
        if (selected_bcid[i] >= s->n_wl || (i && selected_bcid[i] <= selected_bcid[i - 1]))
            return tec_fail(ctx, TEC_EINVAL, "tec_sc_part23: selected barcodes must be ascending whitelist ids");
    const unsigned g = grid_for(N);
    // ---- selection mask and its exclusive prefix (number of selected ids below c)
    // (the temporaries belong to the context: a run of the same size allocates nothing)
    auto& B = s->p23;
    dvec<uint32_t>&sel = B.sel, &selprefix = B.selprefix, &lines = B.lines, &lkeys = B.lkeys, &lkeys2 = B.lkeys2, &lines2 = B.lines2,
                  &firstkept = B.firstkept, &qflag = B.qflag, &qscan = B.qscan, &qrec = B.qrec, &nemit = B.nemit, &emoff = B.emoff,
                  &cellhits = B.cellhits, &present = B.present, &pred = B.pred;
    dvec<uint8_t>& kept_rec = B.kept_rec;
    dvec<uint64_t>&keys = B.keys, &keys2 = B.keys2, &ukeys = B.ukeys, &side = B.side;
    dvec<unsigned long long>& scal = B.scal;
    auto cleanup = [&]() {};
#define SC_TRY(call)                         \
    do {                                     \
        SC_TRACE_BEGIN;                      \
        hipError_t e2__ = (call);            \
        SC_TRACE_END(#call);                 \
        if (e2__ != hipSuccess) {            \
            cleanup();                       \
            return tec_fail(ctx, TEC_EHIP, "%s failed: %s", #call, hipGetErrorString(e2__)); \
        }                                    \
    } while (0)
#define SC_RC(call)          \
    do {                     \
        int rc__ = (call);   \
        if (rc__) {          \
            cleanup();       \
            return rc__;     \
        }                    \
    } while (0)
    SC_TRY(sel.reserve(s->n_wl));
    SC_TRY(selprefix.reserve(s->n_wl));
    SC_TRY(hipMemsetAsync(sel.p, 0, (size_t)s->n_wl * 4, st));
    SC_TRY(s->k32a.reserve(std::max<size_t>(N, n_selected)));
    SC_TRY(hipMemcpyAsync(s->k32a.p, selected_bcid, (size_t)n_selected * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scatter_sel_kernel, dim3((n_selected + kB - 1) / kB), dim3(kB), 0, st, n_selected, s->k32a.p, sel.p);
    SC_RC(scan_u32(ctx, s, sel.p, selprefix.p, s->n_wl, false, rocprim::plus<uint32_t>()));
    // ---- bundle lines = instance heads in (epoch, barcode, UMI) order
    uint64_t n_lines = 0;
    SC_RC(sc_lines(ctx, s, lines, lkeys, lkeys2, lines2, &n_lines));
    // ---- which (bundle, barcode) pairs have lines -- everywhere (given) or here -- and, from that, every
    // line's predecessor barcode in its bundle
    const uint32_t nb = present_global ? n_bundles : s->n_epochs;
    if (nb < s->n_epochs) {
        cleanup();
        return tec_fail(ctx, TEC_EINVAL, "tec_sc_part23_global: the presence table has fewer bundles than this GPU");
    }
    const size_t pcells = (size_t)nb * s->n_wl;
    if ((uint64_t)nb * (s->n_wl + 1ull) >= 0xFFFFFFFFull) {
        cleanup();
        return tec_fail(ctx, TEC_EOVERFLOW, "tec_sc_part23: bundles x whitelist beyond 2^32");
    }
    SC_TRY(present.reserve(pcells));
    SC_TRY(pred.reserve(pcells));
    if (present_global) {
        std::vector<uint32_t> h(pcells);
        for (size_t i = 0; i < pcells; ++i) h[i] = present_global[i] ? 1u : 0u;
        SC_TRY(hipMemcpy(present.p, h.data(), pcells * 4, hipMemcpyHostToDevice));
    } else {
        SC_TRY(hipMemsetAsync(present.p, 0, pcells * 4, st));
        hipLaunchKernelGGL(presence_kernel, dim3(grid_for(n_lines)), dim3(kB), 0, st, n_lines, lines.p, s->s_bcid.p, s->s_epoch.p,
                           s->n_wl, present.p);
    }
    hipLaunchKernelGGL(presence_mark_kernel, dim3(grid_for(pcells)), dim3(kB), 0, st, (uint64_t)pcells, present.p, s->n_wl, pred.p);
    SC_RC(scan_u32(ctx, s, pred.p, pred.p, pcells, true, rocprim::maximum<uint32_t>()));
    // ---- drop rule, first kept line of a key across bundles
    SC_TRY(kept_rec.reserve(N));
    SC_TRY(firstkept.reserve((size_t)s->n_groups + 1));
    SC_TRY(scal.reserve(4));
    SC_TRY(hipMemsetAsync(kept_rec.p, 0, N, st));
    SC_TRY(hipMemsetAsync(firstkept.p, 0xFF, ((size_t)s->n_groups + 1) * 4, st));
    SC_TRY(hipMemsetAsync(scal.p, 0, 4 * 8, st));
    hipLaunchKernelGGL(kept_kernel, dim3(std::min(grid_for(n_lines), 4096u)), dim3(kB), 0, st, n_lines, lines.p, s->s_bcid.p, s->s_epoch.p,
                       sel.p, selprefix.p, s->gid.p, pred.p, s->n_wl, kept_rec.p, firstkept.p, scal.p);
    // ---- part 3 queries
    SC_TRY(qflag.reserve(N));
    SC_TRY(qscan.reserve(N));
    hipLaunchKernelGGL(query_flag_kernel, dim3(g), dim3(kB), 0, st, N, s->ihead.p, s->counted.p, s->headidx.p, kept_rec.p,
                       firstkept.p, s->gid.p, s->s_klass.p, qflag.p);
    SC_RC(scan_u32(ctx, s, qflag.p, qscan.p, N, false, rocprim::plus<uint32_t>()));
    uint32_t qs = 0, qf = 0;
    SC_TRY(hipMemcpyAsync(&qs, qscan.p + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SC_TRY(hipMemcpyAsync(&qf, qflag.p + (N - 1), 4, hipMemcpyDeviceToHost, st));
    unsigned long long h_scal[4] = {0};
    SC_TRY(hipMemcpyAsync(h_scal, scal.p, sizeof h_scal, hipMemcpyDeviceToHost, st));
    SC_TRY(hipStreamSynchronize(st));
    s->kept_lines = h_scal[0];
    if (kept_lines) *kept_lines = s->kept_lines;
    const uint64_t nq = (uint64_t)qs + qf;
    s->n_queries = nq;
    if (nq == 0) {
        cleanup();
        return TEC_OK;
    }
    SC_TRY(qrec.reserve(nq));
    hipLaunchKernelGGL(compact_idx_kernel<uint32_t>, dim3(g), dim3(kB), 0, st, N, qflag.p, qscan.p, qrec.p);
    SC_TRY(nemit.reserve(nq));
    SC_TRY(emoff.reserve(nq));
    SC_TRY(side.reserve((size_t)kSide * nq));
    SC_TRY(cellhits.reserve(s->n_wl));
    SC_TRY(hipMemsetAsync(cellhits.p, 0, (size_t)s->n_wl * 4, st));
    hipLaunchKernelGGL(sc_query_kernel<false>, dim3(std::min(grid_for(nq), 4096u)), dim3(kB), 0, st, ctx->idx, nq, qrec.p, s->s_bcid.p,
                       s->s_klass.p, s->s_left.p, s->s_rite.p, s->cls_chrom.p, s->stranded, nemit.p, (const uint32_t*)nullptr,
                       (uint64_t*)nullptr, side.p, cellhits.p, s->stat.p, scal.p + 1);
    SC_RC(scan_u32(ctx, s, nemit.p, emoff.p, nq, false, rocprim::plus<uint32_t>()));
    uint32_t es = 0, ef = 0;
    SC_TRY(hipMemcpyAsync(&es, emoff.p + (nq - 1), 4, hipMemcpyDeviceToHost, st));
    SC_TRY(hipMemcpyAsync(&ef, nemit.p + (nq - 1), 4, hipMemcpyDeviceToHost, st));
    SC_TRY(hipMemcpyAsync(h_scal, scal.p, sizeof h_scal, hipMemcpyDeviceToHost, st));
    SC_TRY(hipMemcpyAsync(s->h_stat, s->stat.p, sizeof s->h_stat, hipMemcpyDeviceToHost, st));
    SC_TRY(hipStreamSynchronize(st));
    s->n_assigned = h_scal[1];
    if (s->h_stat[ST_NOSTRAND]) {
        cleanup();
        return tec_fail(ctx, TEC_ENOSTRAND, "a hit feature has no 'strand' (the reference raises KeyError, te_count.py:661)");
    }
    const uint64_t n_keys = (uint64_t)es + ef;
    // per-cell hit counters (self.barcodes of part 3), ascending barcode id
    {
        std::vector<uint32_t> h(s->n_wl);
        SC_TRY(hipMemcpy(h.data(), cellhits.p, (size_t)s->n_wl * 4, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < s->n_wl; ++c)
            if (h[c]) {
                s->out_cell.push_back(c);
                s->out_hits.push_back(h[c]);
            }
    }
    if (n_cells_hit) *n_cells_hit = s->out_cell.size();
    if (n_keys) {
        if (n_keys >= 0xFFFFFFF0ull) {
            cleanup();
            return tec_fail(ctx, TEC_EOVERFLOW, "tec_sc_part23: more than 2^32 (cell, feature) increments");
        }
        SC_TRY(keys.reserve(n_keys));
        SC_TRY(keys2.reserve(n_keys));
        hipLaunchKernelGGL(sc_query_kernel<true>, dim3(std::min(grid_for(nq), 4096u)), dim3(kB), 0, st, ctx->idx, nq, qrec.p, s->s_bcid.p,
                           s->s_klass.p, s->s_left.p, s->s_rite.p, s->cls_chrom.p, s->stranded, nemit.p, emoff.p,
                           keys.p, side.p, (uint32_t*)nullptr, s->stat.p, scal.p + 1);
        hipLaunchKernelGGL(map_names_kernel, dim3(grid_for(n_keys)), dim3(kB), 0, st, n_keys, keys.p, ctx->d_int2ext);
        // keys = cell << 32 | name: the name ids use the low bits_for(names) bits only, so the sort runs over the name bits
        // and then (stable) over the cell bits instead of over all 32 + bits_for(cells)
        size_t bytes = 0;
        const unsigned nb_name = std::min(32u, bits_for(ctx->idx.n_names)), nb_cell = bits_for(s->n_wl);
        SC_TRY(rocprim::radix_sort_keys(nullptr, bytes, keys.p, keys2.p, n_keys, 0, nb_name, st));
        SC_TRY(s->tmp.reserve(bytes));
        SC_TRY(rocprim::radix_sort_keys(s->tmp.p, bytes, keys.p, keys2.p, n_keys, 0, nb_name, st));
        bytes = 0;
        SC_TRY(rocprim::radix_sort_keys(nullptr, bytes, keys2.p, keys.p, n_keys, 32, 32 + nb_cell, st));
        SC_TRY(s->tmp.reserve(bytes));
        SC_TRY(rocprim::radix_sort_keys(s->tmp.p, bytes, keys2.p, keys.p, n_keys, 32, 32 + nb_cell, st));
        std::swap(keys.p, keys2.p);
        std::swap(keys.cap, keys2.cap);
        SC_TRY(ukeys.reserve(n_keys));
        SC_TRY(nemit.reserve(n_keys));
        bytes = 0;
        SC_TRY(rocprim::run_length_encode(nullptr, bytes, keys2.p, (unsigned)n_keys, ukeys.p, nemit.p, scal.p + 2, st));
        SC_TRY(s->tmp.reserve(bytes));
        SC_TRY(rocprim::run_length_encode(s->tmp.p, bytes, keys2.p, (unsigned)n_keys, ukeys.p, nemit.p, scal.p + 2, st));
        SC_TRY(hipMemcpyAsync(h_scal, scal.p, sizeof h_scal, hipMemcpyDeviceToHost, st));
        SC_TRY(hipStreamSynchronize(st));
        s->n_coo = (uint64_t)h_scal[2];  // the triplets stay in ukeys / nemit until tec_sc_fetch
    }
    if (n_coo) *n_coo = s->n_coo;
    cleanup();
    return TEC_OK;
#undef SC_TRY
#undef SC_RC
}

extern "C" int tec_sc_fetch(tec_ctx* ctx, uint32_t* cell_bcid, uint64_t* cell_hits, uint32_t* coo_cell, uint32_t* coo_name,
                            uint64_t* coo_count) {
    if (!ctx || !ctx->sc) return tec_fail(ctx, TEC_ESTATE, "tec_sc_fetch: no single-cell run");
    tec_sc_state* s = ctx->sc;
    if (!s->out_cell.empty()) {
        if (cell_bcid) memcpy(cell_bcid, s->out_cell.data(), s->out_cell.size() * 4);
        if (cell_hits) memcpy(cell_hits, s->out_hits.data(), s->out_hits.size() * 8);
    }
    const size_t nu = (size_t)s->n_coo;
    if (nu && (coo_cell || coo_name || coo_count)) {
        // keys and counts -> pinned memory at PCIe speed, then worker threads split them into the caller's three arrays
        SC_HIP(hipSetDevice(ctx->device));
        const size_t need = nu * 12;
        char* stage = s->pin_cap[0] >= need ? s->pin[0] : nullptr;  // (the submit ring is idle by now)
        if (!stage) {
            if (s->pin_out_cap < need) {
                if (s->pin_out) (void)hipHostFree(s->pin_out);
                s->pin_out = nullptr;
                s->pin_out_cap = 0;
                void* p = nullptr;
                SC_HIP(hipHostMalloc(&p, need + need / 4, hipHostMallocDefault));
                s->pin_out = static_cast<char*>(p);
                s->pin_out_cap = need + need / 4;
            }
            stage = s->pin_out;
        }
        hipStream_t st = ctx->s_compute;
        SC_HIP(hipMemcpyAsync(stage, s->p23.ukeys.p, nu * 8, hipMemcpyDeviceToHost, st));
        SC_HIP(hipMemcpyAsync(stage + nu * 8, s->p23.nemit.p, nu * 4, hipMemcpyDeviceToHost, st));
        SC_HIP(hipStreamSynchronize(st));
        const uint64_t* hk = reinterpret_cast<const uint64_t*>(stage);
        const uint32_t* hn = reinterpret_cast<const uint32_t*>(stage + nu * 8);
        const int threads = nu > (1u << 18) ? 16 : 1;
        const size_t per = (nu + (size_t)threads - 1) / (size_t)threads;
        auto work = [=](int t) {
            const size_t a = per * (size_t)t, b = std::min(nu, a + per);
            for (size_t i = a; i < b; ++i) {
                if (coo_cell) coo_cell[i] = (uint32_t)(hk[i] >> 32);
                if (coo_name) coo_name[i] = (uint32_t)(hk[i] & 0xFFFFFFFFull);
                if (coo_count) coo_count[i] = hn[i];
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& t : pool) t.join();
    }
    return TEC_OK;
}

extern "C" int tec_sc_stats(tec_ctx* ctx, tec_sc_stats_t* out) {
    if (!ctx || !ctx->sc || !out) return tec_fail(ctx, TEC_ESTATE, "tec_sc_stats: no single-cell run");
    tec_sc_state* s = ctx->sc;
    out->records = s->n_seen;
    out->qcfail = s->h_stat[ST_QCFAIL];
    out->lowq = s->h_stat[ST_LOWQ];
    out->invalid_barcode = s->h_stat[ST_BADBC];
    out->skipped_chrom = s->h_stat[ST_SKIPCHROM];
    out->survivors = s->n_live;
    out->keys = s->n_groups;
    out->bundles = s->n_live ? s->n_epochs : 0;
    out->kept_lines = s->kept_lines;
    out->assigned = s->n_assigned;
    out->gpu_us = (uint64_t)(s->gpu_ms * 1000.0);
    return TEC_OK;
}

extern "C" int tec_sc_end(tec_ctx* ctx) {
    if (!ctx) return tec_fail(ctx, TEC_EINVAL, "tec_sc_end: NULL context");
    sc_park(ctx);
    return TEC_OK;
}
