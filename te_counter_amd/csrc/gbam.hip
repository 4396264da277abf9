// The BAM front end on the GPU: tec_gbam_* entry points.
//
// Replaces pysam.AlignmentFile(filename) / next(sam) and the per-record attribute reads of the
// reference's loops (te_count/te_count.py:65,78-98 / 190,203-214 / 351,393-438) for BGZF-compressed
// BAM input.  The host only moves COMPRESSED bytes: a staging thread preads the file in 256 MB pieces
// into a ring of pinned buffers, walks the BGZF headers and sends the blocks to HBM on the copy stream
// (stage_chunk).  Everything else runs on the MI355X:
//
//   ginflate_kernel     phase 1 of the inflation: one LANE per BGZF block decodes the Huffman bit stream (ginflate.h:
//                       16-bit tables in LDS, 320 B per lane -> 512 lanes per CU), stores literal bytes at their final
//                       place and turns every match into a 32-bit token, four tokens per 16-byte store into 8 KiB pages
//                       drawn from a pool; the lane never reads the inflated stream
//   gresolve_kernel     phase 2: one WAVE per block replays the matches, 64 at a time: positions by a wave prefix sum,
//                       copies in rounds (a match goes as soon as its source lies below the first match still pending).  A block that phase 1 refuses is redone by the host's
//                       referee decoder (inflate_fast.h, then zlib) and patched in
//   seg_guess_kernel    the inflated stream is cut into 64 KiB segments; one wave per segment looks for
//                       the first offset at which a chain of BAM records plausibly starts (bamscan.h)
//   seg_walk_kernel     one lane per segment walks the record chain from that offset to the segment's
//                       end: record count, landing position, 16-bit record offsets
//   (host)              follows the landing positions from the known start: a segment's records are
//                       taken only if the chain of the previous segment lands exactly on its guess;
//                       otherwise that segment is walked again from the landing position.  The
//                       guesses are therefore a speed-up only: the record boundaries are exact.
//   extract_kernel      one lane per record: tid -> index chromosome / class, pos, reference_end from
//                       the CIGAR, flag, mapq, mate-name key, CB/UB tags -> structure-of-arrays in HBM
//   pair_kernel         PE: records (2k, 2k+1) -> counting units (te_count.py:78-98)
// and the units go straight to the count kernel (tec_bulk_submit_device) or the single-cell ingest
// (tec_sc_submit_device) without ever visiting host memory.
//
// Pipeline (run_decode): a chunk is 16 384 blocks, chunk k lives in slot k % kDepth (its own compressed bytes, stream
// buffer, token pool and stream).  The staging thread launches a chunk's inflate kernels the moment its bytes are on
// their way to HBM, so up to kDepth - 1 launches share the chip while the host steps and row kernels of an earlier
// chunk run on the compute stream.
//
// Multi-GPU: a reader owns a byte range of the file (include/tecount.h, tec_bam_open_range has the
// rules); the first record of a range that does not start at 0 is a guess that the caller verifies
// with the previous range's end token.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "bamscan.h"
#include "ginflate.h"
#include "tecount_internal.h"

namespace {

constexpr uint32_t kSegShift = 16;
constexpr uint32_t kSeg = 1u << kSegShift;
constexpr uint32_t kMaxRecPerSeg = kSeg / 36u + 2u;  // a record is at least 36 bytes with its size field
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr size_t kCarryMax = 64u << 20;  // longest record tail carried from one chunk to the next
constexpr int kInflLanes = 256;          // lanes per workgroup of the phase-1 kernel: 640 bytes of LDS each = all 160 KiB of a CU
constexpr int kStreams = 5;              // inflate streams (slot i launches on stream i % kStreams)
constexpr int kDepth = 8;                // chunks in flight: being staged, inflating or being consumed (chunk k lives in slot k % kDepth)
constexpr size_t kScratchPerWave = 64u * (tec_ginf::kLongSymsPerLane * 2u + tec_ginf::kLenWords * 8u);  // global scratch of the inflate kernel
constexpr uint32_t kFlagFilter = TEC_FLAG_UNMAPPED | TEC_FLAG_QCFAIL | TEC_FLAG_DUP;

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct BlockDesc {
    uint32_t in_off, in_len;  // DEFLATE body inside the chunk's compressed bytes
    uint32_t out_off, isize;  // inflated bytes inside the chunk's inflated stream
};

// ---- per-lane memory of the inflate kernel: tables in LDS, lane-interleaved (entry i of lane L of a workgroup of
// kThreads lanes at u16 index i * kThreads + L: same-i accesses of a wave are consecutive, random-i accesses always
// land in the lane pair's own bank); the symbols of long codes and the code lengths of the block being
// set up in global memory, interleaved per wave
template <int kThreads>
struct DevMem {
    uint16_t* t;
    uint16_t* s;
    uint64_t* w;  // code lengths of the dynamic block being set up, 16 per word, interleaved per wave like s
    __device__ __forceinline__ uint64_t lenw(uint32_t k) const { return w[k * 64u]; }
    __device__ __forceinline__ void set_lenw(uint32_t k, uint64_t v) { w[k * 64u] = v; }
    __device__ __forceinline__ uint32_t tab(uint32_t i) const { return t[i * kThreads]; }
    __device__ __forceinline__ void set_tab(uint32_t i, uint32_t v) { t[i * kThreads] = (uint16_t)v; }
    // sorted symbols of the long codes: the first kLongLds ranks behind the table in LDS, the others in global scratch
    __device__ __forceinline__ uint32_t lsym(uint32_t i) const {
        return i < tec_ginf::kLongLds ? t[(tec_ginf::kLitEntries + i) * kThreads] : s[i * 64u];
    }
    __device__ __forceinline__ uint32_t lsym_near(uint32_t i) const { return t[(tec_ginf::kLitEntries + i) * kThreads]; }
    __device__ __forceinline__ void set_lsym(uint32_t i, uint32_t v) {
        if (i < tec_ginf::kLongLds)
            t[(tec_ginf::kLitEntries + i) * kThreads] = (uint16_t)v;
        else
            s[i * 64u] = (uint16_t)v;
    }
};

// ---- token pages.  A block's tokens go into pages of kPageTokens that its lane draws from a pool with one atomic add
// per page; block b's page numbers are pages[b * kPagesPerBlock ...].
constexpr uint32_t kPageTokens = 2048;     // 8 KiB
constexpr uint32_t kPagesPerBlock = 12;    // 24 576 tokens: a block of 64 KiB has at most 65 536 / 3 + 65 536 / 256 + 2
struct DevSink {
    uint32_t* pool;
    uint32_t n_pages;
    uint32_t* cursor;    // next free page of the pool
    uint32_t* my_pages;  // this block's page table
    uint32_t* cur = nullptr;
    uint32_t k = 0;
    __device__ __forceinline__ bool put(uint32_t t) {
        const uint32_t i = k & (kPageTokens - 1u);
        if (i == 0u) {
            const uint32_t idx = k / kPageTokens;
            if (idx >= kPagesPerBlock) return false;
            const uint32_t pg = atomicAdd(cursor, 1u);
            if (pg >= n_pages) return false;  // the pool is exhausted: the referee takes the block
            my_pages[idx] = pg;
            cur = pool + (size_t)pg * kPageTokens;
        }
        cur[i] = t;
        ++k;
        return true;
    }
};

template <int kThreads>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void ginflate_kernel(const BlockDesc* __restrict__ desc, uint32_t n,
                                                                const uint8_t* __restrict__ comp, uint8_t* infl,
                                                                uint16_t* syms_scr, uint32_t* n_failed, uint32_t* failed,
                                                                uint32_t* pool, uint32_t n_pages, uint32_t* pages, uint32_t* ntok) {
    extern __shared__ uint16_t lds_tab[];
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;
    if (b >= n) return;
    const size_t wave = b >> 6, lane = b & 63u;
    uint8_t* const scr = reinterpret_cast<uint8_t*>(syms_scr) + wave * kScratchPerWave;
    DevMem<kThreads> m{lds_tab + threadIdx.x, reinterpret_cast<uint16_t*>(scr) + lane,
             reinterpret_cast<uint64_t*>(scr + 64u * tec_ginf::kLongSymsPerLane * 2u) + lane};
    const BlockDesc d = desc[b];
    DevSink sink{pool, n_pages, n_failed + 1, pages + (size_t)b * kPagesPerBlock};
    int rc = tec_ginf::kOk;
    if (d.isize) rc = tec_ginf::inflate_stream(comp + d.in_off, d.in_len, infl + d.out_off, d.isize, m, sink);
    ntok[b] = rc == tec_ginf::kOk ? sink.k : 0u;
    if (rc != tec_ginf::kOk) failed[atomicAdd(n_failed, 1u)] = b;
}

// ---- phase 2.  Exact-length copies only: unlike a sequential decoder, a neighbour lane may already have written
// the bytes behind this match.
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void st_tail(uint8_t* q, uint64_t v, uint32_t rem) {  // the low `rem` (< 8) bytes of v
    if (rem & 4u) {
        const uint32_t w = (uint32_t)v;
        memcpy(q, &w, 4);
        q += 4;
        v >>= 32;
    }
    if (rem & 2u) {
        const uint16_t w = (uint16_t)v;
        memcpy(q, &w, 2);
        q += 2;
        v >>= 16;
    }
    if (rem & 1u) *q = (uint8_t)v;
}
__device__ __forceinline__ void copy_match(uint8_t* q, uint32_t len, uint32_t dist) {
    const uint8_t* p = q - dist;
    uint32_t k = 0;
    if (dist >= 8u) {  // eight source bytes are never bytes the same group writes
        for (; k + 8u <= len; k += 8u) {
            const uint64_t v = ld8(p + k);
            memcpy(q + k, &v, 8);
        }
        if (k < len) st_tail(q + k, ld8(p + k), len - k);
    } else {  // a pattern of `dist` bytes: one load, replicated in the register, stored in whole periods
        uint64_t v = ld8(p);  // (bytes from q on are not written yet: masked off)
        const uint32_t bits = 8u * dist;
        v &= (1ull << bits) - 1ull;
        v |= v << bits;                             // 2 periods (dist <= 4: still inside 64 bits)
        v |= dist <= 2u ? v << (2u * bits) : 0ull;  // 4 periods
        v |= dist == 1u ? v << 32 : 0ull;           // 8 periods
        if (dist == 3u) v |= v << 48;               // (3: 6 bytes from the line above, the rest here)
        const uint32_t g = dist == 1u ? 8u : (8u / dist) * dist;  // 8 8 6 8 5 6 7
        for (; k + 8u <= len; k += g) memcpy(q + k, &v, 8);
        if (k < len) st_tail(q + k, v, len - k);
    }
}

// One wave per block.  Tokens [t0, t0 + 64) of a block sit in one page (pages hold 2048 tokens).
__global__ __launch_bounds__(256) void gresolve_kernel(const BlockDesc* __restrict__ desc, uint32_t n, uint8_t* infl,
                                                        const uint32_t* __restrict__ pool, const uint32_t* __restrict__ pages,
                                                        const uint32_t* __restrict__ ntok) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (b >= n) return;
    const uint32_t T = ntok[b];
    if (T == 0u) return;  // empty, or refused by phase 1 (the host redoes it)
    uint8_t* const out = infl + desc[b].out_off;
    const uint32_t* const pg = pages + (size_t)b * kPagesPerBlock;
    uint32_t op = 0;
    for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
        const uint32_t t = t0 + lane;
        const uint32_t* page = pool + (size_t)pg[t0 / kPageTokens] * kPageTokens;
        const uint32_t tok = t < T ? page[t & (kPageTokens - 1u)] : 0u;
        const uint32_t len = tec_ginf::tok_match_len(tok);
        const uint32_t nb = tec_ginf::tok_before(tok) + len;  // bytes in place in front of the match, then the match
        // output positions: inclusive prefix sum over the lanes
        uint32_t incl = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            incl += lane >= (uint32_t)d ? up : 0u;
        }
        const uint32_t dst = op + incl - len;  // where the match starts
        op += (uint32_t)__shfl((int)incl, 63, 64);
        uint8_t* const q = out + dst;
        const uint32_t dist = (tok & 0x7FFFu) + 1u;
        const uint32_t need_end = dst - dist + (len < dist ? len : dist);  // the source bytes that must be final end here
        unsigned long long pending = __ballot(len != 0u);
        while (pending) {  // wave-uniform; the first pending match always qualifies
            const uint32_t first = (uint32_t)__builtin_ctzll(pending);
            const uint32_t hwm = (uint32_t)__shfl((int)dst, (int)first, 64);  // everything below is final
            const bool mine = ((pending >> lane) & 1ull) != 0ull;
            const bool go = mine && need_end <= hwm;
            if (go) copy_match(q, len, dist);
            pending &= ~__ballot(go);
        }
    }
}

// One wave per segment: the smallest offset in the segment at which three consecutive records look
// right.  start0 != kNone: the stream's first record start is known (segments before it hold none).
__global__ __launch_bounds__(256) void seg_guess_kernel(const uint8_t* __restrict__ s, uint32_t total, uint32_t n_seg,
                                                        uint32_t start0, int32_t n_ref, const uint32_t* __restrict__ ref_len,
                                                        uint32_t* guess) {
    const uint32_t seg = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= n_seg) return;
    if (start0 != kNone && seg <= (start0 >> kSegShift)) {
        if (lane == 0) guess[seg] = seg == (start0 >> kSegShift) ? start0 : kNone;
        return;
    }
    const uint32_t base = seg << kSegShift;
    const uint32_t end = total - base < kSeg ? total : base + kSeg;
    for (uint32_t x0 = base; x0 < end; x0 += 64u) {
        const uint32_t x = x0 + lane;
        const bool ok = x < end && tec_bamscan::plausible_record(s, total, x, n_ref, ref_len, 3);
        const unsigned long long mask = __ballot(ok);
        if (mask) {
            if (lane == 0) guess[seg] = x0 + (uint32_t)__builtin_ctzll(mask);
            return;
        }
    }
    if (lane == 0) guess[seg] = kNone;
}

// One lane per segment: follows block_size fields from the segment's start to the first record that
// starts beyond the segment (or is incomplete: the tail that is carried to the next chunk).
// only_seg != kNone: redo just that segment from only_start (the host's repair of a wrong guess).
__global__ __launch_bounds__(64) void seg_walk_kernel(const uint8_t* __restrict__ s, uint32_t total, uint32_t n_seg,
                                                      const uint32_t* __restrict__ guess, uint32_t* count, uint32_t* land,
                                                      uint16_t* offs, uint32_t only_seg, uint32_t only_start) {
    uint32_t seg = blockIdx.x * 64u + threadIdx.x;
    uint32_t p;
    if (only_seg != kNone) {
        if (seg != 0) return;
        seg = only_seg;
        p = only_start;
    } else {
        if (seg >= n_seg) return;
        p = guess[seg];
    }
    if (p == kNone) {
        count[seg] = 0;
        land[seg] = kNone;
        return;
    }
    const uint32_t base = seg << kSegShift;
    const uint32_t seg_end = total - base < kSeg ? total : base + kSeg;
    uint16_t* o = offs + (size_t)seg * kMaxRecPerSeg;
    uint32_t c = 0;
    while (p < seg_end) {
        if (total - p < 4u) break;
        const uint32_t bs = tec_bamscan::rd32(s + p);
        if (bs < 32u || total - p - 4u < bs) break;  // incomplete (or not a record): the chain ends here
        o[c++] = (uint16_t)(p - base);
        p += 4u + bs;
    }
    count[seg] = c;
    land[seg] = p;
}

struct RecArrays {  // structure of arrays in HBM; which members are used depends on the mode
    int32_t *chrom, *pos, *end;
    uint16_t* flag;
    uint8_t* mapq;
    uint64_t* qkey;
    uint8_t *cb, *ub, *tf;
};

enum { kModeSE = 0, kModePE = 1, kModeSC = 2, kModeDump = 3 };  // Dump: every field of every record, kept resident (tests, tec_gbam_decode)

// One wave per segment, one lane per record.  seg_base[seg] == kNone: the segment's records are not
// on the chain.  Row of record j of the segment: slot0 + seg_base[seg] + j.
template <int MODE>
__global__ __launch_bounds__(64) void extract_kernel(const uint8_t* __restrict__ s, const uint32_t* __restrict__ seg_base,
                                                     const uint32_t* __restrict__ count, const uint16_t* __restrict__ offs,
                                                     uint32_t clip, uint64_t slot0, RecArrays out,
                                                     const int32_t* __restrict__ tidmap, uint32_t n_tid, int mapq_min,
                                                     uint64_t ordinal0, unsigned long long* first_noend, uint32_t* bad_layout) {
    const uint32_t seg = blockIdx.x;
    const uint32_t b = seg_base[seg];
    if (b == kNone) return;
    uint32_t c = count[seg];
    if (c > clip) c = clip;
    const uint16_t* o = offs + (size_t)seg * kMaxRecPerSeg;
    for (uint32_t j = threadIdx.x; j < c; j += 64u) {
        const uint8_t* rec = s + ((size_t)seg << kSegShift) + o[j];
        tec_bamscan::Core core;
        if (!tec_bamscan::parse_core(rec, MODE != kModePE, &core)) {
            atomicAdd(bad_layout, 1u);
            continue;
        }
        const uint64_t w = slot0 + b + j;
        const int32_t mapped = (core.tid >= 0 && (uint32_t)core.tid < n_tid) ? tidmap[core.tid] : -1;
        out.chrom[w] = mapped;
        out.pos[w] = core.pos;
        out.flag[w] = (uint16_t)core.flag;
        out.mapq[w] = (uint8_t)core.mapq;
        if (MODE == kModePE || MODE == kModeDump) out.qkey[w] = tec_bamscan::mate_key_hash(core.name, core.l_name ? core.l_name - 1u : 0u);
        if (MODE != kModePE) out.end[w] = core.end;
        if (MODE == kModeSE) {
            // a record that reaches `loc2 + 1` with reference_end None kills the reference (te_count.py:223)
            const bool live = !(core.flag & kFlagFilter) && (int)core.mapq >= mapq_min && mapped >= 0;
            if (live && core.end == tec_bamscan::kNoEnd) atomicMin(first_noend, (unsigned long long)(ordinal0 + b + j));
        }
        if (MODE == kModeSC || MODE == kModeDump) {
            const uint32_t tf = tec_bamscan::scan_tags(core.aux, core.rec_end, out.cb + 24 * w, out.ub + 24 * w);
            out.tf[w] = (uint8_t)tf;  // (bit 3, a UMI longer than the 24-byte field, is judged by the single-cell filter)
        }
    }
}

// PE units (te_count.py:78-98): records first + 2u and first + 2u + 1 of the record arrays.
__global__ __launch_bounds__(256) void pair_kernel(uint64_t n_units, uint64_t first, RecArrays r, int32_t* u_chrom, int32_t* u_p1,
                                                   int32_t* u_p2, uint16_t* u_flag, uint8_t* u_mapq, int mapq_min,
                                                   uint64_t unit0, unsigned long long* first_mismatch) {
    const uint64_t u = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (u >= n_units) return;
    const uint64_t a = first + 2u * u, b = a + 1u;
    // (the flags OR-ed for the filters; the reverse bit is read 1's alone: only the stranded-bulk extension looks at it)
    const uint32_t fa = r.flag[a];
    const uint32_t fl = ((fa | (uint32_t)r.flag[b]) & ~(uint32_t)TEC_FLAG_REVERSE) | (fa & (uint32_t)TEC_FLAG_REVERSE);
    const uint32_t mq = r.mapq[a];
    u_chrom[u] = r.chrom[a];
    u_p1[u] = r.pos[a];
    u_p2[u] = r.pos[b];
    u_flag[u] = (uint16_t)fl;
    u_mapq[u] = (uint8_t)mq;
    // mate names are compared only for pairs that passed the flag and mapq filters (te_count.py:81-94)
    if (!(fl & kFlagFilter) && (int)mq >= mapq_min && r.qkey[a] != r.qkey[b]) atomicMin(first_mismatch, (unsigned long long)(unit0 + u));
}

__global__ void copy_rec_kernel(RecArrays r, uint64_t src, uint64_t dst) {
    r.chrom[dst] = r.chrom[src];
    r.pos[dst] = r.pos[src];
    r.flag[dst] = r.flag[src];
    r.mapq[dst] = r.mapq[src];
    r.qkey[dst] = r.qkey[src];
}

// ---- a grow-only device buffer
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes, bool keep = false, size_t used = 0) {
        if (bytes <= cap) return hipSuccess;
        size_t want = keep ? std::max(bytes, cap * 2) : bytes + bytes / 8;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (keep && p && used) e = hipMemcpy(q, p, used, hipMemcpyDeviceToDevice);
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
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

struct Piece {  // one buffer of the pinned staging ring
    uint8_t* p = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;  // its copy to HBM
    bool busy = false;
};

struct Staged {  // one chunk of compressed blocks: staged through the pinned ring into d_comp
    size_t bytes = 0;           // compressed bytes of the chunk
    uint64_t file_off = 0;      // of byte 0
    std::vector<BlockDesc> desc;
    std::vector<uint64_t> block_off;  // file offset of every block
    uint64_t inflated = 0;
    uint64_t next_off = 0;      // file offset of the first block after this chunk
    bool last = false;          // no owned block follows
    bool at_eof = false;        // next_off is the end of the file
    std::string error;
    // the staging thread also starts the chunk's copy to HBM (copy stream): it overlaps the GPU's work on the chunk before
    DBuf d_comp, d_desc;
    hipEvent_t copied = nullptr;
    // the chunk's inflate launch (its own scratch and failed-block list: two launches are in flight at a time)
    DBuf d_syms, d_ctl, d_pages, d_ntok;
    hipEvent_t inflate_done = nullptr;  // completion of the chunk's phase-2 kernel
    bool launched = false;
    int buf = 0;  // which of the two inflated-stream buffers it writes
};

}  // namespace

// Device buffers and pinned staging of a reader.  They outlive the reader: tec_gbam_close parks them on the
// context (allocating and freeing gigabytes of HBM and pinned memory costs more than decoding a small file),
// the next reader on that context takes them over, tec_destroy frees them.
struct GbamBuffers {
    DBuf d_tokpool[kDepth];  // token pages of the chunk in slot i (phase 1 fills, phase 2 replays)
    DBuf d_infl2[kDepth], d_ctl, d_guess, d_count, d_land, d_offs, d_segbase, d_reflen, d_tidmap, d_carry;
    DBuf r_chrom, r_pos, r_end, r_flag, r_mapq, r_qkey, r_cb, r_ub, r_tf;  // record arrays
    DBuf u_chrom, u_p1, u_p2, u_flag, u_mapq;                                // PE units
    DBuf d_flags;
    uint64_t r_cap = 0, u_cap = 0;
    Staged slots[kDepth];                 // slot i owns stream buffer i and token pool i
    // Chunk k's two inflate kernels go to stream k % kStreams, phase 2 in order behind phase 1: the launches of up to
    // kStreams chunks overlap when the streams sit on different hardware queues.  A process has GPU_MAX_HW_QUEUES of them
    // (HIP's default: four) for all its streams and which one a stream gets is not ours to choose: with four, two of these
    // streams share a queue and their launches run back to back; tec_create asks for sixteen before HIP initialises.
    hipStream_t s_infl[kStreams] = {};
    int cur_buf = 0;                      // buffer of the chunk being consumed (the carry sits in front of its stream)
    Piece pieces[3];  // one being walked / copied, one being read, one whose copy to HBM drains
    int next_piece = 0;
};

struct tec_gbam : GbamBuffers {
    tec_ctx* ctx = nullptr;
    int fd = -1;
    uint64_t file_size = 0;
    std::string err;
    std::vector<std::string> ref_names;
    std::vector<uint32_t> ref_lens;
    bool ranged = false;
    uint64_t range_begin = 0, range_end = ~0ull;
    uint64_t first_data_off = 0;     // file offset of the first BGZF block this reader stages
    uint64_t hdr_block_off = 0;      // range 0: block that holds the first record ...
    uint32_t hdr_inblock = 0;        // ... and its offset inside that block's inflated bytes
    // one chunk = one launch of each inflate kernel.  Phase 1 is one lane per block, so a launch takes a block's serial decode
    // time (~30 ms) whether it holds 16 384 blocks or 65 536: chunks are SMALL and up to kDepth of them are in flight -- the
    // first launch starts one staging piece into the file, and what is left to do once the last byte is staged is one small chunk
    size_t chunk_bytes = 2047u << 20;
    uint32_t chunk_blocks = 16384;  // 32 workgroups of 512 lanes; (stream offsets are 32-bit: <= 4 GiB inflated)
    size_t piece_bytes = 256u << 20;  // pinned staging ring: two buffers of this size
    int read_threads = 8;
    // run state
    uint64_t n_records = 0;   // owned records delivered so far
    bool start_known = false;
    bool have_first = false;
    uint64_t tok_first[2] = {0, 0}, tok_end[2] = {0, 0};
    uint64_t blocks = 0, blocks_host = 0, comp_bytes = 0, infl_bytes = 0, repairs = 0;
    double t_stage_wait = 0, t_inflate = 0, t_segments = 0, t_rows = 0;  // seconds of the decode loop, by phase
    double mean_block_bytes = 0;  // compressed bytes per BGZF block, from the chunks staged so far
    double t_stage_alloc = 0, t_stage_read = 0;                          // staging thread: pinned allocation, pread
    // deferred PE pairing
    bool deferred = false;
    uint64_t n_resident = 0;  // records kept in r_* (deferred mode), starting at row 1
    bool skip_header = false;  // the next chunk is the start of the file: records begin behind the BAM header
    uint32_t n_tid = 0;
};

namespace {

int gfail(tec_gbam* g, int code, const std::string& msg) {
    if (g) g->err = msg;
    if (g && g->ctx) g->ctx->err = msg;
    return code;
}

#define G_HIP(call)                                                                                       \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return gfail(g, TEC_EHIP, std::string(#call " failed: ") + hipGetErrorString(e__) + " (gbam.hip:" + \
                                          std::to_string(__LINE__) + ")");                                \
    } while (0)

// Fills `st` with the owned blocks that start at file offset `off` (at most chunk_bytes / chunk_blocks).
// epilogue: ignore the range end (blocks needed to complete the last owned record).
void stage_chunk(tec_gbam* g, Staged* st, uint64_t off, bool epilogue, uint32_t max_blocks) {
    st->desc.clear();
    st->block_off.clear();
    st->bytes = 0;
    st->inflated = 0;
    st->file_off = off;
    st->error.clear();
    st->launched = false;
    st->last = st->at_eof = false;
    const uint64_t limit = epilogue ? g->file_size : std::min<uint64_t>(g->range_end, g->file_size);
    if (off >= g->file_size) {
        st->next_off = g->file_size;
        st->last = st->at_eof = true;
        return;
    }
    // window of the file to stage: what max_blocks blocks are expected to need (+ 6 %), from the mean block size seen so far
    // -- before the first chunk, from a 4 MiB probe at `off`.  A window that turns out short just ends the chunk early.
    if (g->mean_block_bytes <= 0) {
        std::vector<uint8_t> probe((size_t)std::min<uint64_t>(4u << 20, g->file_size - off) + 32, 0);
        const ssize_t k = pread(g->fd, probe.data(), probe.size() - 32, (off_t)off);
        size_t o = 0, bsize, body_off, body_len;
        uint32_t isize, nb = 0;
        while (k > 0 && tec_bgzf_parse(probe.data(), (size_t)k, o, &bsize, &body_off, &body_len, &isize)) {
            o += bsize;
            ++nb;
        }
        g->mean_block_bytes = nb ? (double)o / nb : 65536.0;
    }
    const uint64_t expect = (uint64_t)(g->mean_block_bytes * 1.06 * max_blocks) + (1u << 20);
    const size_t cap_bytes = (size_t)std::min<uint64_t>(std::min<uint64_t>(g->chunk_bytes, expect) + 65536 + 64, g->file_size - off);
    hipStream_t cp = g->ctx->s_copy;
    hipError_t e = st->copied ? hipSuccess : hipEventCreateWithFlags(&st->copied, hipEventDisableTiming);
    if (e == hipSuccess) e = st->d_comp.reserve(cap_bytes + 64);
    if (e != hipSuccess) {
        st->error = std::string("cannot allocate the chunk's buffer on the GPU: ") + hipGetErrorString(e);
        return;
    }
    // The window travels in PIECES through a ring of three small pinned buffers (pinning costs ~0.1 s per GB, a
    // chunk is 1-2 GB): pread a piece, walk its block headers, send its whole blocks to HBM on the copy stream.
    // While a piece is walked (a pointer chase through cold cache lines: 3 ms per 256 MB) the reader threads already
    // fill the other buffer: a block cut by the end of a piece starts within its last 64 KiB, so the next piece is
    // read from 64 KiB before that end and the walk goes on at whatever offset in it the next block turns out to have.
    constexpr size_t kOverlap = 65536 + 64;
    struct ReadOp {
        Piece* pc = nullptr;
        uint64_t base = 0;  // file offset of byte 0 of the buffer
        size_t win = 0;
        std::vector<std::thread> th;
        std::vector<int> ok;
        bool active = false;
        double t0 = 0;
    };
    auto begin_read = [&](ReadOp& r, uint64_t base) -> bool {
        Piece& pc = g->pieces[g->next_piece];
        g->next_piece = (g->next_piece + 1) % 3;
        double t0 = now_s();
        if (pc.busy) {
            (void)hipEventSynchronize(pc.done);
            pc.busy = false;
        }
        // (no more than the chunk can still take: what is read beyond its last block is read again for the next chunk)
        const uint64_t room = (uint64_t)cap_bytes - std::min<uint64_t>(cap_bytes, base > off ? base - off : 0) + kOverlap;
        const size_t win = (size_t)std::min<uint64_t>(std::min<uint64_t>(g->piece_bytes, room), g->file_size - base);
        if (win + 32 > pc.cap) {
            if (pc.p) (void)hipHostFree(pc.p);
            pc.p = nullptr;
            pc.cap = 0;
            const size_t want = (size_t)std::min<uint64_t>(g->piece_bytes, std::max<uint64_t>(win, g->file_size - g->first_data_off)) + 32;
            void* p = nullptr;
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess ||
                (!pc.done && hipEventCreateWithFlags(&pc.done, hipEventDisableTiming) != hipSuccess)) {
                st->error = "cannot allocate pinned staging memory";
                return false;
            }
            pc.p = static_cast<uint8_t*>(p);
            pc.cap = want;
        }
        g->t_stage_alloc += now_s() - t0;
        r.pc = &pc;
        r.base = base;
        r.win = win;
        r.t0 = now_s();
        const int nt = win > (32u << 20) ? g->read_threads : 1;
        r.ok.assign(nt, 1);
        r.th.clear();
        const size_t per = (win + nt - 1) / nt;
        uint8_t* dst = pc.p;
        const int fd = g->fd;
        for (int t = 0; t < nt; ++t) {
            int* okp = &r.ok[t];
            r.th.emplace_back([=] {  // parallel pread: page cache -> pinned memory
                size_t a = (size_t)t * per, bnd = std::min(win, a + per);
                while (a < bnd) {
                    const ssize_t k = pread(fd, dst + a, bnd - a, (off_t)(base + a));
                    if (k <= 0) {
                        *okp = 0;
                        return;
                    }
                    a += (size_t)k;
                }
            });
        }
        r.active = true;
        return true;
    };
    auto end_read = [&](ReadOp& r, bool wanted) {
        for (auto& x : r.th) x.join();
        r.th.clear();
        r.active = false;
        if (!wanted) return;
        for (int v : r.ok)
            if (!v) st->error = "read error on the alignment file";
        memset(r.pc->p + r.win, 0, 32);
        g->t_stage_read += now_s() - r.t0;
    };
    size_t o = 0;  // bytes of the chunk that are parsed and on their way
    uint64_t infl = 0;
    bool full = false;
    ReadOp cur, nxt;
    if (off >= limit) st->last = true;
    else if (!begin_read(cur, off)) cur.active = false;
    while (cur.active) {
        end_read(cur, true);
        if (!st->error.empty()) break;
        const size_t delta = (size_t)(off + o - cur.base);  // the first unparsed block inside this buffer
        const uint8_t* pb = cur.pc->p + delta;
        const size_t win = cur.win - delta;
        const bool to_eof = cur.base + cur.win >= g->file_size;
        // read ahead while walking -- unless this piece very likely completes the chunk
        if (!to_eof && o + win + kOverlap < cap_bytes && cur.base + cur.win < limit + kOverlap) {
            if (!begin_read(nxt, cur.base + cur.win - kOverlap)) break;
        }
        size_t q = 0;  // whole blocks of this piece
        while (st->desc.size() < max_blocks) {
            if (off + o + q >= limit) {
                st->last = true;
                break;
            }
            size_t bsize, body_off, body_len;
            uint32_t isize;
            if (!tec_bgzf_parse(pb, win, q, &bsize, &body_off, &body_len, &isize)) {
                if (to_eof && win - q < 18 + 8) {  // nothing but a stub at the very end
                    if (win - q != 0) st->error = "truncated BGZF block at the end of the file";
                    st->last = true;
                } else if (to_eof || win - q >= 65536 + 64) {
                    st->error = "not a BGZF block (corrupt or not a BAM file)";
                }
                break;  // else: the block runs past this piece; the next piece holds it whole
            }
            if ((o + q + bsize > cap_bytes || infl + isize > 0xFFFF0000ull - kCarryMax - 65536ull) && !st->desc.empty()) {
                full = true;  // the chunk's compressed buffer or its 32-bit stream offsets are exhausted
                break;
            }
            st->desc.push_back({(uint32_t)(o + body_off), (uint32_t)body_len, (uint32_t)infl, isize});
            st->block_off.push_back(off + o + q);
            infl += isize;
            q += bsize;
        }
        if (!st->error.empty()) break;
        if (q) {
            e = hipMemcpyAsync(st->d_comp.as<uint8_t>() + o, pb, q + tec_ginf::kInPad, hipMemcpyHostToDevice, cp);
            if (e == hipSuccess) e = hipEventRecord(cur.pc->done, cp);
            if (e != hipSuccess) {
                st->error = std::string("copy of a staged piece to the GPU failed: ") + hipGetErrorString(e);
                break;
            }
            cur.pc->busy = true;
            o += q;
        } else if (!st->last && !full && st->desc.size() < max_blocks) {
            st->error = "not a BGZF block (corrupt or not a BAM file)";  // a whole piece without one complete block
            break;
        }
        if (st->desc.size() >= max_blocks || full || st->last || off + o >= g->file_size) break;
        if (off + o >= limit) {
            st->last = true;
            break;
        }
        if (nxt.active && (off + o < nxt.base || off + o > nxt.base + kOverlap)) end_read(nxt, false);  // (cannot happen: read again)
        if (!nxt.active && !begin_read(nxt, off + o)) break;
        cur = std::move(nxt);
        nxt = ReadOp();
    }
    if (cur.active) end_read(cur, false);
    if (nxt.active) end_read(nxt, false);
    st->bytes = o;
    st->inflated = infl;
    if (!st->desc.empty()) g->mean_block_bytes = (double)o / (double)st->desc.size();
    st->next_off = off + o;
    if (st->next_off >= g->file_size) st->last = st->at_eof = true;
    if (st->next_off >= limit) st->last = true;
    if (!st->desc.empty() && st->error.empty()) {  // the block descriptors follow the compressed bytes
        e = st->d_desc.reserve(st->desc.size() * sizeof(BlockDesc));
        if (e == hipSuccess) e = hipMemcpyAsync(st->d_desc.p, st->desc.data(), st->desc.size() * sizeof(BlockDesc), hipMemcpyHostToDevice, cp);
        if (e == hipSuccess) e = hipEventRecord(st->copied, cp);
        if (e != hipSuccess) st->error = std::string("copy of a staged chunk to the GPU failed: ") + hipGetErrorString(e);
    }
}

struct ChunkResult {
    uint64_t n_records = 0;   // on the chain in this chunk
    uint32_t total = 0;       // stream bytes (carry + inflated)
    uint32_t tail = 0;        // stream offset where the incomplete tail starts (== total: none)
    uint32_t first_pos = kNone;  // stream offset of the chunk's first record on the chain
    uint32_t carry_in = 0;
};

struct RunArgs {
    int mode = kModeSE;
    int mapq_min = 0;
    bool deferred = false;
};

struct RunState {
    uint32_t carry_len = 0;
    bool have_odd = false;  // PE streaming: row 0 of the record arrays holds an unpaired record
    unsigned long long* d_first_noend = nullptr;
    unsigned long long* d_first_mismatch = nullptr;
    uint32_t* d_bad_layout = nullptr;
    uint64_t units_done = 0;
};

int ensure_records(tec_gbam* g, uint64_t rows, int mode, bool keep, uint64_t used_rows) {
    // (the buffers outlive one file -- they are cached on the context -- and a later run may be of another mode:
    // every array the mode uses is brought to the common row capacity, the others stay as they are)
    const uint64_t want = rows <= g->r_cap ? g->r_cap : (keep ? std::max<uint64_t>(rows, g->r_cap * 2) : rows + rows / 8 + 1024);
    auto grow = [&](DBuf& b, size_t elt) { return b.reserve((size_t)want * elt, keep, (size_t)used_rows * elt); };
    G_HIP(grow(g->r_chrom, 4));
    G_HIP(grow(g->r_pos, 4));
    G_HIP(grow(g->r_flag, 2));
    G_HIP(grow(g->r_mapq, 1));
    if (mode == kModePE || mode == kModeDump) G_HIP(grow(g->r_qkey, 8));
    if (mode != kModePE) G_HIP(grow(g->r_end, 4));
    if (mode == kModeSC || mode == kModeDump) {
        G_HIP(grow(g->r_cb, 24));
        G_HIP(grow(g->r_ub, 24));
        G_HIP(grow(g->r_tf, 1));
    }
    g->r_cap = want;
    return TEC_OK;
}

RecArrays rec_arrays(tec_gbam* g) {
    return RecArrays{g->r_chrom.as<int32_t>(), g->r_pos.as<int32_t>(), g->r_end.as<int32_t>(), g->r_flag.as<uint16_t>(),
                     g->r_mapq.as<uint8_t>(), g->r_qkey.as<uint64_t>(), g->r_cb.as<uint8_t>(), g->r_ub.as<uint8_t>(),
                     g->r_tf.as<uint8_t>()};
}

int ensure_units(tec_gbam* g, uint64_t n) {
    if (n <= g->u_cap) return TEC_OK;
    const uint64_t want = n + n / 8 + 1024;
    G_HIP(g->u_chrom.reserve(want * 4));
    G_HIP(g->u_p1.reserve(want * 4));
    G_HIP(g->u_p2.reserve(want * 4));
    G_HIP(g->u_flag.reserve(want * 2));
    G_HIP(g->u_mapq.reserve(want));
    g->u_cap = want;
    return TEC_OK;
}

// Pairs rows [first, first + 2 n_units) of the record arrays and hands the units to the count kernel.
int pair_and_count(tec_gbam* g, RunState* rs, const RunArgs& a, uint64_t first, uint64_t n_units) {
    if (!n_units) return TEC_OK;
    int rc = ensure_units(g, n_units);
    if (rc) return rc;
    hipStream_t st = g->ctx->s_compute;
    hipLaunchKernelGGL(pair_kernel, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, st, n_units, first, rec_arrays(g),
                       g->u_chrom.as<int32_t>(), g->u_p1.as<int32_t>(), g->u_p2.as<int32_t>(), g->u_flag.as<uint16_t>(),
                       g->u_mapq.as<uint8_t>(), a.mapq_min, rs->units_done, rs->d_first_mismatch);
    G_HIP(hipGetLastError());
    rc = tec_bulk_submit_device(g->ctx, n_units, g->u_chrom.as<int32_t>(), g->u_p1.as<int32_t>(), g->u_p2.as<int32_t>(),
                                g->u_flag.as<uint16_t>(), g->u_mapq.as<uint8_t>());
    if (rc) return rc;
    // the unit buffers are reused by the next chunk: the count kernel must have read them
    G_HIP(hipStreamSynchronize(st));
    rs->units_done += n_units;
    return TEC_OK;
}

// Both inflate kernels of a staged chunk into stream buffer `buf`, on inflate stream `lane_stream` (staging thread; the epilogue: calling thread).  Stream
// buffer `buf` must be free (the caller knows: its last reader synchronised); the chunk's copy to HBM is waited for on the device.
int launch_inflate(tec_gbam* g, Staged* st, int buf, int lane_stream) {
    tec_ctx* ctx = g->ctx;
    st->buf = buf;
    st->launched = true;
    const uint32_t nb = (uint32_t)st->desc.size();
    hipStream_t si = g->s_infl[lane_stream];
    if (!si) {
        G_HIP(hipStreamCreateWithFlags(&g->s_infl[lane_stream], hipStreamNonBlocking));
        si = g->s_infl[lane_stream];
    }
    if (!st->inflate_done) G_HIP(hipEventCreate(&st->inflate_done));
    DBuf& infl = g->d_infl2[buf];
    if (kCarryMax + st->inflated + 256 > infl.cap) {
        // (growing copies the carry area: nothing may be in flight on either side)
        G_HIP(hipStreamSynchronize(ctx->s_compute));
        G_HIP(hipStreamSynchronize(si));
        G_HIP(infl.reserve(kCarryMax + st->inflated + 256, true, kCarryMax));  // keep: the carry sits below kCarryMax
    }
    const uint32_t waves = (nb + 63u) / 64u;
    G_HIP(st->d_syms.reserve((size_t)std::max(waves, 1u) * kScratchPerWave));
    G_HIP(st->d_ctl.reserve(64 + (size_t)std::max(nb, 1u) * 4));
    G_HIP(st->d_pages.reserve((size_t)std::max(nb, 1u) * kPagesPerBlock * 4));
    G_HIP(st->d_ntok.reserve((size_t)std::max(nb, 1u) * 4));
    if (!nb) return TEC_OK;
    // token pool: four bytes per match.  A pool of 3/4 of the inflated size holds any stream whose matches average
    // 5.4 bytes or more (BAM payloads average 8 to 30); each block leaves at most one page partly filled.  A block
    // that finds the pool empty is refused and redone by the host's referee -- slow, never wrong
    constexpr size_t page_bytes = (size_t)kPageTokens * 4;
    const size_t want_pages = ((size_t)st->inflated / 4 * 3 + page_bytes - 1) / page_bytes + (size_t)nb;
    DBuf& pool = g->d_tokpool[buf];
    if (want_pages * page_bytes > pool.cap) G_HIP(pool.reserve(want_pages * page_bytes));  // (the slot's previous chunk is done)
    const uint32_t n_pages = (uint32_t)std::min<size_t>(pool.cap / page_bytes, 0xFFFFFFFFu);
    uint32_t* d_nfailed = st->d_ctl.as<uint32_t>();  // [0] failed blocks, [1] pages drawn from the pool
    G_HIP(hipMemsetAsync(d_nfailed, 0, 64, si));
    G_HIP(hipStreamWaitEvent(si, st->copied, 0));  // the staging thread's copy of this chunk
    auto p1 = [&](auto kernel, uint32_t lanes) {
        kernel<<<(nb + lanes - 1) / lanes, lanes, (size_t)lanes * tec_ginf::kEntriesPerLane * 2, si>>>(
            st->d_desc.as<BlockDesc>(), nb, st->d_comp.as<uint8_t>(), infl.as<uint8_t>() + kCarryMax, st->d_syms.as<uint16_t>(), d_nfailed,
            d_nfailed + 16, pool.as<uint32_t>(), n_pages, st->d_pages.as<uint32_t>(), st->d_ntok.as<uint32_t>());
    };
    p1(ginflate_kernel<kInflLanes>, (uint32_t)kInflLanes);
    G_HIP(hipGetLastError());
    gresolve_kernel<<<(nb + 3u) / 4u, 256, 0, si>>>(st->d_desc.as<BlockDesc>(), nb, infl.as<uint8_t>() + kCarryMax, pool.as<uint32_t>(),
                                                   st->d_pages.as<uint32_t>(), st->d_ntok.as<uint32_t>());
    G_HIP(hipGetLastError());
    G_HIP(hipEventRecord(st->inflate_done, si));
    return TEC_OK;
}

// Inflates, segments, walks and extracts one staged chunk.  limit_one: epilogue -- only the record
// that starts at stream offset 0 (in the carried bytes) is taken.
int process_chunk(tec_gbam* g, Staged* st, const RunArgs& a, RunState* rs, bool limit_one, ChunkResult* res) {
    tec_ctx* ctx = g->ctx;
    hipStream_t cs = ctx->s_compute;
    const uint32_t nb = (uint32_t)st->desc.size();
    const uint32_t carry = rs->carry_len;
    res->carry_in = carry;
    if ((uint64_t)carry + st->inflated >= 0xFFFF0000ull) return gfail(g, TEC_EOVERFLOW, "chunk larger than 4 GiB of inflated data");
    const uint32_t total = carry + (uint32_t)st->inflated;
    res->total = total;
    res->tail = total;
    if (total == 0) return TEC_OK;
    // ---- the inflated stream (the kernel was started ahead of time for all but the first chunk and the epilogue)
    if (!st->launched) {  // (the epilogue; chunks of the main loop were launched by the staging thread)
        int rc = launch_inflate(g, st, g->cur_buf, 0);
        if (rc) return rc;
    }
    g->cur_buf = st->buf;
    uint32_t* d_nfailed = st->d_ctl.as<uint32_t>();
    uint32_t* d_failed = d_nfailed + 16;
    uint8_t* d_stream_base = g->d_infl2[st->buf].as<uint8_t>() + kCarryMax;  // inflated bytes of this chunk start here
    uint8_t* d_stream = d_stream_base - carry;                               // stream offset 0
    double t0 = now_s();
    if (nb) G_HIP(hipEventSynchronize(st->inflate_done));
    if (nb) {
        uint32_t n_failed = 0;
        G_HIP(hipMemcpy(&n_failed, d_nfailed, 4, hipMemcpyDeviceToHost));
        if (n_failed > nb) return gfail(g, TEC_EHIP, "the inflate kernel reported more failed blocks than the chunk holds");
        if (n_failed) {  // the referee: the host's decoder (and zlib behind it) has the last word
            std::vector<uint32_t> bad(n_failed);
            G_HIP(hipMemcpy(bad.data(), d_failed, (size_t)n_failed * 4, hipMemcpyDeviceToHost));
            std::vector<uint8_t> tmp(65536);
            for (uint32_t b : bad) {
                if (b >= nb) return gfail(g, TEC_EHIP, "the inflate kernel reported a block that is not in the chunk");
                const BlockDesc& d = st->desc[b];
                std::vector<uint8_t> body(d.in_len + 16, 0);  // (the pinned ring has moved on: read the block again)
                if (pread(g->fd, body.data(), d.in_len, (off_t)(st->file_off + d.in_off)) != (ssize_t)d.in_len ||
                    !tec_bgzf_inflate_host(body.data(), d.in_len, tmp.data(), d.isize))
                    return gfail(g, TEC_EINVAL, "BGZF inflate failed (corrupt block at file offset " + std::to_string(st->block_off[b]) + ")");
                G_HIP(hipMemcpy(d_stream_base + d.out_off, tmp.data(), d.isize, hipMemcpyHostToDevice));
            }
            g->blocks_host += n_failed;
        }
        g->blocks += nb;
        g->comp_bytes += st->bytes;
        g->infl_bytes += st->inflated;
    }
    g->t_inflate += now_s() - t0;
    t0 = now_s();
    // ---- segments
    const uint32_t n_seg = (total + kSeg - 1) >> kSegShift;
    G_HIP(g->d_guess.reserve((size_t)n_seg * 4));
    G_HIP(g->d_count.reserve((size_t)n_seg * 4));
    G_HIP(g->d_land.reserve((size_t)n_seg * 4));
    G_HIP(g->d_segbase.reserve((size_t)n_seg * 4));
    G_HIP(g->d_offs.reserve((size_t)n_seg * kMaxRecPerSeg * 2));
    uint32_t start0 = kNone;
    if (g->start_known) {
        start0 = 0;
        if (g->skip_header) {  // the very first chunk of the file: records begin behind the header
            for (size_t i = 0; i < st->block_off.size(); ++i)
                if (st->block_off[i] == g->hdr_block_off) start0 = st->desc[i].out_off + g->hdr_inblock;
            g->skip_header = false;
        }
    }
    hipLaunchKernelGGL(seg_guess_kernel, dim3((n_seg + 3) / 4), dim3(256), 0, cs, d_stream, total, n_seg, start0,
                       (int32_t)g->ref_lens.size(), g->d_reflen.as<uint32_t>(), g->d_guess.as<uint32_t>());
    hipLaunchKernelGGL(seg_walk_kernel, dim3((n_seg + 63) / 64), dim3(64), 0, cs, d_stream, total, n_seg, g->d_guess.as<uint32_t>(),
                       g->d_count.as<uint32_t>(), g->d_land.as<uint32_t>(), g->d_offs.as<uint16_t>(), kNone, 0u);
    G_HIP(hipGetLastError());
    std::vector<uint32_t> guess(n_seg), count(n_seg), land(n_seg), seg_base(n_seg, kNone);
    G_HIP(hipMemcpyAsync(guess.data(), g->d_guess.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost, cs));
    G_HIP(hipMemcpyAsync(count.data(), g->d_count.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost, cs));
    G_HIP(hipMemcpyAsync(land.data(), g->d_land.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost, cs));
    G_HIP(hipStreamSynchronize(cs));
    // ---- the chain: from the known start (or, for a range that begins mid-file, the first guess)
    uint32_t cur = kNone;
    if (start0 != kNone) {
        cur = start0 >> kSegShift;
        if (cur >= n_seg) cur = kNone;  // the header ends beyond this chunk's data (cannot happen with real headers)
    } else {
        for (uint32_t s2 = 0; s2 < n_seg; ++s2)
            if (guess[s2] != kNone) {
                cur = s2;
                break;
            }
    }
    uint64_t n_rec = 0;
    uint32_t tail = (start0 != kNone || cur != kNone) ? total : total;  // no record start found at all: nothing carried
    if (cur != kNone) {
        res->first_pos = guess[cur];
        for (;;) {
            seg_base[cur] = (uint32_t)n_rec;
            n_rec += limit_one ? std::min(count[cur], 1u) : count[cur];
            const uint32_t p = land[cur];
            const uint32_t seg_end = total - (cur << kSegShift) < kSeg ? total : (cur << kSegShift) + kSeg;
            if (limit_one) {
                tail = p;  // (unused by the epilogue)
                break;
            }
            if (p < seg_end || p >= total) {  // an incomplete record: the tail; or the data ends exactly with a record
                tail = p < total ? p : total;
                break;
            }
            const uint32_t t = p >> kSegShift;
            if (guess[t] != p) {  // the guess of segment t is not on the chain: walk it again from p
                hipLaunchKernelGGL(seg_walk_kernel, dim3(1), dim3(64), 0, cs, d_stream, total, n_seg, g->d_guess.as<uint32_t>(),
                                   g->d_count.as<uint32_t>(), g->d_land.as<uint32_t>(), g->d_offs.as<uint16_t>(), t, p);
                G_HIP(hipGetLastError());
                G_HIP(hipMemcpyAsync(&count[t], g->d_count.as<uint32_t>() + t, 4, hipMemcpyDeviceToHost, cs));
                G_HIP(hipMemcpyAsync(&land[t], g->d_land.as<uint32_t>() + t, 4, hipMemcpyDeviceToHost, cs));
                G_HIP(hipStreamSynchronize(cs));
                guess[t] = p;
                ++g->repairs;
            }
            cur = t;
        }
    } else if (start0 == kNone) {
        tail = total;  // a range that begins inside a record longer than this chunk: nothing of it is ours
    }
    res->n_records = n_rec;
    res->tail = tail;
    g->t_segments += now_s() - t0;
    t0 = now_s();
    if (!n_rec) return TEC_OK;
    // ---- rows of the record arrays
    uint64_t slot0;
    if (a.mode == kModePE) {
        slot0 = 1 + (a.deferred ? g->n_resident : 0);
        int rc = ensure_records(g, slot0 + n_rec + 2, a.mode, a.deferred || rs->have_odd, a.deferred ? slot0 : 1);
        if (rc) return rc;
    } else if (a.mode == kModeDump) {
        slot0 = g->n_resident;
        int rc = ensure_records(g, slot0 + n_rec + 2, a.mode, true, slot0);
        if (rc) return rc;
    } else {
        slot0 = 0;
        int rc = ensure_records(g, n_rec + 2, a.mode, false, 0);
        if (rc) return rc;
    }
    G_HIP(hipMemcpyAsync(g->d_segbase.p, seg_base.data(), (size_t)n_seg * 4, hipMemcpyHostToDevice, cs));
    const RecArrays ra = rec_arrays(g);
    const uint32_t clip = limit_one ? 1u : 0xFFFFFFFFu;
    const int32_t* tm = g->d_tidmap.as<int32_t>();
#define LAUNCH_EXTRACT(M)                                                                                                   \
    hipLaunchKernelGGL(extract_kernel<M>, dim3(n_seg), dim3(64), 0, cs, d_stream, g->d_segbase.as<uint32_t>(),               \
                       g->d_count.as<uint32_t>(), g->d_offs.as<uint16_t>(), clip, slot0, ra, tm, g->n_tid, a.mapq_min,      \
                       g->n_records, rs->d_first_noend, rs->d_bad_layout)
    if (a.mode == kModeSE) LAUNCH_EXTRACT(kModeSE);
    else if (a.mode == kModePE) LAUNCH_EXTRACT(kModePE);
    else if (a.mode == kModeSC) LAUNCH_EXTRACT(kModeSC);
    else LAUNCH_EXTRACT(kModeDump);
#undef LAUNCH_EXTRACT
    G_HIP(hipGetLastError());
    // ---- hand the rows on
    if (a.mode == kModeSE) {
        int rc = tec_bulk_submit_device(ctx, n_rec, ra.chrom, ra.pos, ra.end, ra.flag, ra.mapq);
        if (rc) return rc;
        G_HIP(hipStreamSynchronize(cs));  // the rows are overwritten by the next chunk
    } else if (a.mode == kModeSC) {
        if (g->n_records == 0 && st->bytes) {  // the first chunk: records per compressed byte x the bytes of the range
            const uint64_t range_bytes = std::min<uint64_t>(g->range_end, g->file_size) - std::min<uint64_t>(g->first_data_off, g->file_size);
            (void)tec_sc_expect_records(ctx, (uint64_t)((double)n_rec / (double)st->bytes * (double)range_bytes * 1.02) + n_rec / 50);
        }
        int rc = tec_sc_submit_device(ctx, n_rec, ra.chrom, ra.pos, ra.end, ra.flag, ra.mapq, ra.cb, ra.ub, ra.tf);
        if (rc) return rc;
    } else if (a.deferred || a.mode == kModeDump) {
        g->n_resident += n_rec;
    } else {
        const uint64_t have = n_rec + (rs->have_odd ? 1 : 0);
        const uint64_t first = rs->have_odd ? 0 : 1;
        const uint64_t n_units = have / 2;
        int rc = pair_and_count(g, rs, a, first, n_units);
        if (rc) return rc;
        if (have & 1) {
            const uint64_t src = first + 2 * n_units;
            if (src != 0) hipLaunchKernelGGL(copy_rec_kernel, dim3(1), dim3(1), 0, cs, ra, src, (uint64_t)0);
            rs->have_odd = true;
        } else {
            rs->have_odd = false;
        }
    }
    G_HIP(hipStreamSynchronize(cs));
    g->t_rows += now_s() - t0;
    return TEC_OK;
}

// Stream offset -> {file offset of the BGZF block that holds the byte, offset inside it}.  Offsets
// inside the carried bytes cannot be mapped (the caller never asks for them).
bool stream_to_token(const Staged& st, uint32_t carry, uint32_t pos, uint64_t tok[2]) {
    if (pos < carry) return false;
    const uint32_t rel = pos - carry;
    for (size_t i = 0; i < st.desc.size(); ++i)
        if (rel >= st.desc[i].out_off && rel < st.desc[i].out_off + st.desc[i].isize) {
            tok[0] = st.block_off[i];
            tok[1] = rel - st.desc[i].out_off;
            return true;
        }
    return false;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// entry points

extern "C" const char* tec_gbam_last_error(const tec_gbam* g) {
    static thread_local std::string none;
    return g ? g->err.c_str() : none.c_str();
}

extern "C" int tec_gbam_open(tec_ctx* ctx, const char* path, uint64_t range_begin, uint64_t range_end, tec_gbam** out) {
    if (!ctx || !path || !out) return tec_fail(ctx, TEC_EINVAL, "tec_gbam_open: NULL argument");
    *out = nullptr;
    // header and reference names through the host decoder (a few blocks at the start of the file)
    tec_bam* hb = nullptr;
    int rc = tec_bam_open(path, 1, &hb);
    if (rc != TEC_OK) return tec_fail(ctx, rc, "%s", tec_bam_last_error(nullptr));
    std::unique_ptr<tec_gbam> g(new tec_gbam());
    g->ctx = ctx;
    // (the buffers parked on the context by the previous reader are taken over at the very end: an open that fails
    // leaves them where they are)
    const int n_ref = tec_bam_n_refs(hb);
    for (int i = 0; i < n_ref; ++i) g->ref_names.emplace_back(tec_bam_ref_name(hb, i));
    g->ref_lens.resize((size_t)n_ref);
    for (int i = 0; i < n_ref; ++i) g->ref_lens[i] = tec_bam_ref_len(hb, i);
    uint64_t first[2], end[2];
    int hf = 0, fin = 0;
    tec_bam_range_tokens(hb, first, end, &hf, &fin);
    g->file_size = tec_bam_file_size(hb);
    tec_bam_close(hb);
    if (g->file_size == 0) return tec_fail(ctx, TEC_EINVAL, "tec_gbam_open: %s is not a BGZF-compressed BAM file on disk", path);
    g->hdr_block_off = first[0];
    g->hdr_inblock = (uint32_t)first[1];
    g->ranged = !(range_begin == 0 && range_end >= g->file_size);
    g->range_begin = range_begin;
    g->range_end = std::min<uint64_t>(range_end, g->file_size);
    g->fd = open(path, O_RDONLY);
    if (g->fd < 0) return tec_fail(ctx, TEC_EINVAL, "tec_gbam_open: cannot open %s", path);
    if (const char* e = getenv("TEC_GBAM_CHUNK_MB")) {
        const long mb = atol(e);
        if (mb >= 1 && mb <= 3072) g->chunk_bytes = (size_t)mb << 20;
    }
    if (const char* e = getenv("TEC_GBAM_PIECE_KB")) {  // (tests: several pieces per chunk on small files)
        const long kb = atol(e);
        if (kb >= 192 && kb <= (2047l << 10)) g->piece_bytes = (size_t)kb << 10;
    }
    if (const char* e = getenv("TEC_GBAM_CHUNK_BLOCKS")) {
        const long nb = atol(e);
        if (nb >= 1 && nb <= 65536) g->chunk_blocks = (uint32_t)nb;
    }
    if (g->range_begin == 0) {
        g->first_data_off = 0;
        g->start_known = true;
        g->skip_header = true;
        if (hf) {
            g->tok_first[0] = first[0];
            g->tok_first[1] = first[1];
        }
    } else {
        // first BGZF block at or after range_begin: scan a window of the file for the block header's fixed bytes
        const uint64_t from = std::min<uint64_t>(g->range_begin, g->file_size);
        const size_t win = (size_t)std::min<uint64_t>(g->file_size - from, 4u << 20);
        std::vector<uint8_t> w(win + 16);
        const ssize_t got = win ? pread(g->fd, w.data(), win, (off_t)from) : 0;
        g->first_data_off = g->file_size;  // nothing to stage unless a block is found
        if (got > 0) {
            const size_t o = tec_bgzf_find_block(w.data(), (size_t)got, 0);
            if (o < (size_t)got) {
                g->first_data_off = from + o;
            } else if (from + (uint64_t)got < g->file_size) {
                close(g->fd);
                return tec_fail(ctx, TEC_EINVAL, "tec_gbam_open: no BGZF block found within 4 MiB of offset %llu", (unsigned long long)from);
            }
        }
        g->start_known = false;
    }
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = g->d_reflen.reserve(std::max<size_t>(1, (size_t)n_ref) * 4);
    if (e == hipSuccess && n_ref) e = hipMemcpy(g->d_reflen.p, g->ref_lens.data(), (size_t)n_ref * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess)  // (more than 64 KiB of dynamic LDS has to be asked for)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ginflate_kernel<kInflLanes>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                kInflLanes * (int)tec_ginf::kEntriesPerLane * 2);
    if (e != hipSuccess) {
        close(g->fd);
        g->d_reflen.release();
        return tec_fail(ctx, TEC_EHIP, "tec_gbam_open: %s", hipGetErrorString(e));
    }
    if (ctx->gbam_cache) {  // buffers parked by the previous reader on this context
        GbamBuffers* c = static_cast<GbamBuffers*>(ctx->gbam_cache);
        DBuf own_reflen = g->d_reflen;  // (just filled; the parked one is dropped in its favour)
        static_cast<GbamBuffers&>(*g) = *c;
        g->d_reflen.release();
        g->d_reflen = own_reflen;
        delete c;
        ctx->gbam_cache = nullptr;
    }
    *out = g.release();
    return TEC_OK;
}

extern "C" int tec_gbam_n_refs(const tec_gbam* g) { return g ? (int)g->ref_names.size() : -1; }
extern "C" const char* tec_gbam_ref_name(const tec_gbam* g, int i) {
    return (g && i >= 0 && (size_t)i < g->ref_names.size()) ? g->ref_names[i].c_str() : nullptr;
}
extern "C" uint64_t tec_gbam_file_size(const tec_gbam* g) { return g ? g->file_size : 0; }

static void free_buffers(GbamBuffers* g) {
    for (auto& b : g->d_tokpool) b.release();
    for (auto& b : g->d_infl2) b.release();
    for (DBuf* b : {&g->d_ctl, &g->d_guess, &g->d_count, &g->d_land, &g->d_offs,
                    &g->d_segbase, &g->d_flags, &g->d_reflen, &g->d_tidmap, &g->d_carry, &g->r_chrom, &g->r_pos, &g->r_end, &g->r_flag,
                    &g->r_mapq, &g->r_qkey, &g->r_cb, &g->r_ub, &g->r_tf, &g->u_chrom, &g->u_p1, &g->u_p2, &g->u_flag, &g->u_mapq})
        b->release();
    for (auto& pc : g->pieces) {
        if (pc.p) (void)hipHostFree(pc.p);
        if (pc.done) (void)hipEventDestroy(pc.done);
        pc = Piece();
    }
    for (auto& s : g->slots) {
        s.d_comp.release();
        s.d_desc.release();
        s.d_syms.release();
        s.d_ctl.release();
        s.d_pages.release();
        s.d_ntok.release();
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.inflate_done) (void)hipEventDestroy(s.inflate_done);
        s.copied = s.inflate_done = nullptr;
    }
    for (hipStream_t& si : g->s_infl) {
        if (si) (void)hipStreamDestroy(si);
        si = nullptr;
    }
}

void tec_gbam_cache_release(tec_ctx* ctx) {
    if (!ctx || !ctx->gbam_cache) return;
    GbamBuffers* c = static_cast<GbamBuffers*>(ctx->gbam_cache);
    free_buffers(c);
    delete c;
    ctx->gbam_cache = nullptr;
}

extern "C" void tec_gbam_close(tec_gbam* g) {
    if (!g) return;
    if (g->ctx) {
        (void)hipSetDevice(g->ctx->device);
        (void)hipStreamSynchronize(g->ctx->s_compute);
        (void)hipStreamSynchronize(g->ctx->s_copy);  // a staged chunk that was never processed may still be copying
        for (hipStream_t si : g->s_infl)
            if (si) (void)hipStreamSynchronize(si);  // ... or inflating
    }
    if (g->ctx && !g->ctx->gbam_cache) {
        g->ctx->gbam_cache = new GbamBuffers(static_cast<GbamBuffers&>(*g));  // parked for the next reader
    } else {
        free_buffers(g);
    }
    if (g->fd >= 0) close(g->fd);
    delete g;
}

namespace {

// The decode loop: staged chunks -> process_chunk, the carry between chunks, the epilogue record of a
// range, the tokens.  A producer thread stages chunk k+1 (pread into pinned memory, block headers)
// while the GPU works on chunk k.
int run_decode(tec_gbam* g, const RunArgs& a, tec_gbam_report* rep) {
    tec_ctx* ctx = g->ctx;
    hipStream_t cs = ctx->s_compute;
    G_HIP(hipSetDevice(ctx->device));
    RunState rs;
    G_HIP(g->d_ctl.reserve(64 + 4));
    DBuf& flags = g->d_flags;
    G_HIP(flags.reserve(64));
    const unsigned long long init[4] = {~0ull, ~0ull, 0ull, 0ull};
    G_HIP(hipMemcpy(flags.p, init, sizeof init, hipMemcpyHostToDevice));
    rs.d_first_noend = flags.as<unsigned long long>();
    rs.d_first_mismatch = rs.d_first_noend + 1;
    rs.d_bad_layout = reinterpret_cast<uint32_t*>(rs.d_first_noend + 2);
    g->cur_buf = 0;
    g->deferred = a.deferred;
    g->n_resident = 0;
    g->n_records = 0;
    // Stream buffers at their final size BEFORE the pipeline starts: a chunk holds at most chunk_blocks blocks of at
    // most 64 KiB, and the number of chunks follows from the file size and the mean block size (a 4 MiB probe).  The
    // staging thread launches the inflates, this thread copies record tails into the fronts of the same buffers:
    // neither may see a buffer move.  (Should an estimate fall short, the staging thread waits until this thread has
    // caught up and grows the buffer then.)
    {
        const uint64_t range_bytes = std::min<uint64_t>(g->range_end, g->file_size) - std::min<uint64_t>(g->first_data_off, g->file_size);
        if (g->mean_block_bytes <= 0 && range_bytes) {
            std::vector<uint8_t> probe((size_t)std::min<uint64_t>(4u << 20, range_bytes) + 32, 0);
            const ssize_t k = pread(g->fd, probe.data(), probe.size() - 32, (off_t)g->first_data_off);
            size_t o = 0, bsize, body_off, body_len;
            uint32_t isize, nb = 0;
            while (k > 0 && tec_bgzf_parse(probe.data(), (size_t)k, o, &bsize, &body_off, &body_len, &isize)) {
                o += bsize;
                ++nb;
            }
            g->mean_block_bytes = nb ? (double)o / nb : 65536.0;
        }
        const uint64_t est_blocks = (uint64_t)((double)range_bytes / std::max(g->mean_block_bytes, 28.0) * 1.1) + 16;
        const uint64_t n_chunks = (est_blocks + g->chunk_blocks - 1) / g->chunk_blocks;
        const int n_full = (int)std::min<uint64_t>(kDepth, n_chunks + 1);  // (+ 1: the epilogue reads into the buffer after the last chunk's)
        const size_t cap = kCarryMax + (size_t)std::min<uint64_t>(g->chunk_blocks, est_blocks) * 65536 + 256;
        for (int i = 0; i < kDepth; ++i) G_HIP(g->d_infl2[i].reserve(i < n_full ? cap : kCarryMax + 256));
    }

    // ---- the staging thread: chunk k -> slot k % kDepth: pread + header walk + H2D (stage_chunk), then -- at once, from this
    // thread -- the chunk's two inflate kernels.  The GPU starts on the file as soon as the first chunk is staged and never waits
    // for the consumer's host steps.
    std::mutex mu;
    std::condition_variable cv;
    int ready[kDepth] = {};  // 0 free, 1 staged and launched
    uint64_t processed = 0;  // chunks this thread is done with (rows extracted, tail moved on, compute stream drained)
    bool stop = false;
    uint64_t next_off = g->first_data_off;
#ifdef TEC_GBAM_TRACE  // trace build (make EXTRA=-DTEC_GBAM_TRACE): when each chunk is staged / launched / processed
    const double trace_t0 = now_s();
#define GBAM_TRACE(what, k, n) fprintf(stderr, "[gbam %8.2f ms] chunk %llu %s (%llu)\n", (now_s() - trace_t0) * 1e3, (unsigned long long)(k), what, (unsigned long long)(n))
#else
#define GBAM_TRACE(what, k, n) (void)0
#endif
    std::thread producer([&] {
        uint64_t off = next_off;
        for (uint64_t k = 0;; ++k) {
            const int slot = (int)(k % kDepth);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || ready[slot] == 0; });
                if (stop) break;
            }
            (void)hipSetDevice(ctx->device);
            Staged* st = &g->slots[slot];
            GBAM_TRACE("slot free", k, slot);
            stage_chunk(g, st, off, false, g->chunk_blocks);
            GBAM_TRACE("staged", k, st->desc.size());
            if (st->error.empty() && !st->desc.empty()) {
                if (kCarryMax + st->inflated + 256 > g->d_infl2[slot].cap) {  // (an estimate fell short: see above)
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || processed == k; });
                    if (stop) break;
                }
                if (launch_inflate(g, st, slot, (int)(k % kStreams)) != TEC_OK) st->error = "inflate launch failed: " + g->err;
            }
            const bool last = st->last || !st->error.empty();
            off = st->next_off;
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[slot] = 1;
            }
            cv.notify_all();
            if (last) break;
        }
    });
    auto finish_producer = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (producer.joinable()) producer.join();
    };

    int rc = TEC_OK;
    int slot = 0;
    uint64_t end_off = g->first_data_off;  // file offset behind the last staged owned block
    bool at_eof = false;
    bool first_token_pending = !g->start_known;
    for (uint64_t k = 0;; ++k) {
        slot = (int)(k % kDepth);
        {
            const double tw = now_s();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[slot] == 1; });
            g->t_stage_wait += now_s() - tw;
        }
        Staged* st = &g->slots[slot];
        if (!st->error.empty()) {
            rc = gfail(g, TEC_EINVAL, st->error);
            break;
        }
        ChunkResult res;
        GBAM_TRACE("taken", k, st->desc.size());
        rc = process_chunk(g, st, a, &rs, false, &res);
        GBAM_TRACE("processed", k, res.n_records);
        if (rc) break;
        if (first_token_pending && res.first_pos != kNone) {
            uint64_t tok[2];
            if (stream_to_token(*st, res.carry_in, res.first_pos, tok)) {
                g->tok_first[0] = tok[0];
                g->tok_first[1] = tok[1];
                g->have_first = true;
            }
            g->start_known = true;
            first_token_pending = false;
        }
        if (g->start_known && res.n_records && !g->have_first) g->have_first = true;
        g->n_records += res.n_records;
        // the tail of the stream (an incomplete record) moves in front of the next chunk's data
        uint32_t new_carry = g->start_known ? res.total - res.tail : 0;
        if (new_carry > kCarryMax) {
            rc = gfail(g, TEC_EOVERFLOW, "a BAM record larger than 64 MiB is not supported");
            break;
        }
        const int nbuf = (slot + 1) % kDepth;  // where the next chunk (or the epilogue) has its stream
        if (new_carry) {
            const uint8_t* src = g->d_infl2[slot].as<uint8_t>() + kCarryMax - res.carry_in + res.tail;
            uint8_t* dst = g->d_infl2[nbuf].as<uint8_t>() + kCarryMax - new_carry;
            hipError_t e = hipMemcpyAsync(dst, src, new_carry, hipMemcpyDeviceToDevice, cs);
            if (e != hipSuccess) {
                rc = gfail(g, TEC_EHIP, std::string("carry copy failed: ") + hipGetErrorString(e));
                break;
            }
        }
        {
            // rows are extracted, the tail has moved on: once the compute stream has drained, the slot's buffers may be written
            // again (the staging thread launches the next phase 1 into them without any device-side wait)
            hipError_t e = hipStreamSynchronize(cs);
            if (e != hipSuccess) {
                rc = gfail(g, TEC_EHIP, std::string("stream synchronisation failed: ") + hipGetErrorString(e));
                break;
            }
        }
        g->cur_buf = nbuf;
        rs.carry_len = new_carry;
        end_off = st->next_off;
        at_eof = st->at_eof;
        const bool last = st->last;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!last) ready[slot] = 0;
            processed = k + 1;
        }
        cv.notify_all();
        if (last) break;
    }
    finish_producer();
    if (rc) {
        return rc;
    }

    // ---- the epilogue: a record that starts in owned bytes and ends beyond the range; the end token
    Staged* ep = &g->slots[(slot + 1) % kDepth];
    uint64_t off = end_off;
    bool token_done = false;
    if (rs.carry_len && g->start_known) {
        if (at_eof) {
                return gfail(g, TEC_EINVAL, "truncated BAM record");
        }
        uint32_t blocks = 1;
        for (;;) {
            stage_chunk(g, ep, off, true, blocks);
            if (!ep->error.empty()) {
                        return gfail(g, TEC_EINVAL, ep->error);
            }
            if (ep->desc.empty()) {
                        return gfail(g, TEC_EINVAL, "truncated BAM record");
            }
            // is the record at stream offset 0 complete with these blocks?  (its size field may itself straddle)
            ChunkResult res;
            const uint32_t carry = rs.carry_len;
            rc = process_chunk(g, ep, a, &rs, true, &res);
            if (rc) {
                        return rc;
            }
            if (res.n_records == 1) {
                g->n_records += 1;
                // where it ends: 4 + block_size from stream offset 0
                uint32_t bs = 0;
                G_HIP(hipStreamSynchronize(cs));
                G_HIP(hipMemcpy(&bs, g->d_infl2[g->cur_buf].as<uint8_t>() + kCarryMax - carry, 4, hipMemcpyDeviceToHost));
                const uint32_t endpos = 4u + bs;
                uint64_t tok[2];
                if (stream_to_token(*ep, carry, endpos, tok)) {
                    g->tok_end[0] = tok[0];
                    g->tok_end[1] = tok[1];
                    token_done = true;
                } else {
                    off = ep->next_off;  // the record ends exactly with the staged blocks: the next data block starts the next record
                }
                break;
            }
            if (ep->at_eof && ep->desc.size() < blocks) {
                        return gfail(g, TEC_EINVAL, "truncated BAM record");
            }
            blocks *= 2;  // not complete yet: take more blocks (the carry in HBM is unchanged: process_chunk leaves it in place)
        }
        rs.carry_len = 0;
    }
    if (!token_done) {
        // the next record starts with the first non-empty block at or after `off`
        g->tok_end[0] = g->file_size;
        g->tok_end[1] = 0;
        std::vector<uint8_t> w(65536 + 64);
        while (off < g->file_size) {
            const size_t want = (size_t)std::min<uint64_t>(w.size() - 16, g->file_size - off);
            const ssize_t got = pread(g->fd, w.data(), want, (off_t)off);
            size_t bsize, body_off, body_len;
            uint32_t isize;
            if (got <= 0 || !tec_bgzf_parse(w.data(), (size_t)got, 0, &bsize, &body_off, &body_len, &isize)) break;
            if (isize) {
                g->tok_end[0] = off;
                break;
            }
            off += bsize;
        }
    }
    // ---- flags, first / last record of a deferred PE run
    unsigned long long h[4];
    G_HIP(hipStreamSynchronize(cs));
    G_HIP(hipMemcpy(h, flags.p, sizeof h, hipMemcpyDeviceToHost));
    if ((uint32_t)h[2]) return gfail(g, TEC_EINVAL, "corrupt BAM record layout");
    if (rep) {
        memset(rep, 0, sizeof *rep);
        rep->n_records = g->n_records;
        rep->first_noend_live = h[0];
        rep->first_name_mismatch = h[1];
        rep->has_records = g->n_records ? 1 : 0;
        rep->first_token[0] = g->tok_first[0];
        rep->first_token[1] = g->tok_first[1];
        rep->end_token[0] = g->tok_end[0];
        rep->end_token[1] = g->tok_end[1];
        rep->blocks = g->blocks;
        rep->blocks_redone_on_host = g->blocks_host;
        rep->compressed_bytes = g->comp_bytes;
        rep->inflated_bytes = g->infl_bytes;
        rep->segment_repairs = g->repairs;
        rep->seconds_stage_wait = g->t_stage_wait;
        rep->seconds_inflate = g->t_inflate;
        rep->seconds_segments = g->t_segments;
        rep->seconds_rows = g->t_rows;
        rep->seconds_stage_alloc = g->t_stage_alloc;
        rep->seconds_stage_read = g->t_stage_read;
        rep->units_counted = rs.units_done;
        rep->odd_record_left = rs.have_odd ? 1 : 0;
        if (a.mode == kModePE && a.deferred && g->n_resident) {
            const RecArrays ra = rec_arrays(g);
            auto fetch = [&](uint64_t row, tec_gbam_rec* r) -> hipError_t {
                hipError_t e = hipMemcpy(&r->chrom, ra.chrom + row, 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(&r->pos, ra.pos + row, 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(&r->flag, ra.flag + row, 2, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(&r->mapq, ra.mapq + row, 1, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(&r->qkey, ra.qkey + row, 8, hipMemcpyDeviceToHost);
                return e;
            };
            G_HIP(fetch(1, &rep->first_rec));
            G_HIP(fetch(g->n_resident, &rep->last_rec));
        }
    }
    return TEC_OK;
}

int set_tidmap(tec_gbam* g, const int32_t* map, uint32_t n) {
    G_HIP(hipSetDevice(g->ctx->device));
    G_HIP(g->d_tidmap.reserve(std::max<size_t>(1, n) * 4));
    if (n) G_HIP(hipMemcpy(g->d_tidmap.p, map, (size_t)n * 4, hipMemcpyHostToDevice));
    g->n_tid = n;
    return TEC_OK;
}

}  // namespace

extern "C" int tec_gbam_run_bulk(tec_gbam* g, int paired, int mapq_min, const int32_t* tid_to_chrom, uint32_t n_tid,
                                 int defer_pairing, tec_gbam_report* rep) {
    if (!g) return TEC_EINVAL;
    if (!g->ctx->bulk_open) return gfail(g, TEC_ESTATE, "tec_gbam_run_bulk: call tec_bulk_begin first");
    if (n_tid && !tid_to_chrom) return gfail(g, TEC_EINVAL, "tec_gbam_run_bulk: NULL chromosome map");
    int rc = set_tidmap(g, tid_to_chrom, n_tid);
    if (rc) return rc;
    RunArgs a;
    a.mode = paired ? kModePE : kModeSE;
    a.mapq_min = mapq_min;
    a.deferred = paired && defer_pairing;
    return run_decode(g, a, rep);
}

// Deferred PE pairing (multi-GPU): once every rank's record count is known, this rank's pairs are rows
// skip_first + 2k, skip_first + 2k + 1 of its resident records; extra_mate (may be NULL) is the next
// range's first record, the mate of this range's last record when its tail is unpaired.
extern "C" int tec_gbam_finish_pairs(tec_gbam* g, int mapq_min, int skip_first, const tec_gbam_rec* extra_mate,
                                     uint64_t unit_ordinal0, uint64_t* n_units_out, uint64_t* first_name_mismatch) {
    if (!g) return TEC_EINVAL;
    if (!g->deferred) return gfail(g, TEC_ESTATE, "tec_gbam_finish_pairs: the run was not deferred");
    G_HIP(hipSetDevice(g->ctx->device));
    uint64_t n = g->n_resident;
    if (extra_mate && n) {
        int rc = ensure_records(g, 1 + n + 2, kModePE, true, 1 + n);
        if (rc) return rc;
        const RecArrays ra = rec_arrays(g);
        const uint64_t row = 1 + n;
        G_HIP(hipMemcpy(ra.chrom + row, &extra_mate->chrom, 4, hipMemcpyHostToDevice));
        G_HIP(hipMemcpy(ra.pos + row, &extra_mate->pos, 4, hipMemcpyHostToDevice));
        G_HIP(hipMemcpy(ra.flag + row, &extra_mate->flag, 2, hipMemcpyHostToDevice));
        G_HIP(hipMemcpy(ra.mapq + row, &extra_mate->mapq, 1, hipMemcpyHostToDevice));
        G_HIP(hipMemcpy(ra.qkey + row, &extra_mate->qkey, 8, hipMemcpyHostToDevice));
        ++n;
    }
    const uint64_t skip = skip_first ? 1 : 0;
    const uint64_t n_units = n > skip ? (n - skip) / 2 : 0;
    DBuf flag;
    G_HIP(flag.reserve(16));
    const unsigned long long none = ~0ull;
    G_HIP(hipMemcpy(flag.p, &none, 8, hipMemcpyHostToDevice));
    RunState rs;
    rs.d_first_mismatch = flag.as<unsigned long long>();
    rs.units_done = unit_ordinal0;
    RunArgs a;
    a.mode = kModePE;
    a.mapq_min = mapq_min;
    int rc = pair_and_count(g, &rs, a, 1 + skip, n_units);
    unsigned long long h = none;
    if (!rc) {
        hipError_t e = hipMemcpy(&h, flag.p, 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = gfail(g, TEC_EHIP, hipGetErrorString(e));
    }
    flag.release();
    if (rc) return rc;
    if (n_units_out) *n_units_out = n_units;
    if (first_name_mismatch) *first_name_mismatch = h;
    return TEC_OK;
}

extern "C" int tec_gbam_run_sc(tec_gbam* g, int mapq_min, const int32_t* tid_to_class, uint32_t n_tid, tec_gbam_report* rep) {
    if (!g) return TEC_EINVAL;
    if (!g->ctx->sc) return gfail(g, TEC_ESTATE, "tec_gbam_run_sc: call tec_sc_begin first");
    if (n_tid && !tid_to_class) return gfail(g, TEC_EINVAL, "tec_gbam_run_sc: NULL class map");
    int rc = set_tidmap(g, tid_to_class, n_tid);
    if (rc) return rc;
    RunArgs a;
    a.mode = kModeSC;
    a.mapq_min = mapq_min;
    return run_decode(g, a, rep);
}

// Decode only: every record of the reader's range stays in HBM; tec_gbam_fetch copies the rows to the
// host.  tid is the BAM reference id (no mapping).  For tests and for callers that want the records.
extern "C" int tec_gbam_decode(tec_gbam* g, tec_gbam_report* rep) {
    if (!g) return TEC_EINVAL;
    std::vector<int32_t> ident(g->ref_names.size());
    for (size_t i = 0; i < ident.size(); ++i) ident[i] = (int32_t)i;
    int rc = set_tidmap(g, ident.data(), (uint32_t)ident.size());
    if (rc) return rc;
    RunArgs a;
    a.mode = kModeDump;
    return run_decode(g, a, rep);
}

extern "C" int tec_gbam_fetch(tec_gbam* g, uint64_t first, uint64_t n, int32_t* tid, int32_t* pos, int32_t* end, uint16_t* flag,
                              uint8_t* mapq, uint64_t* qkey, uint8_t* cb24, uint8_t* ub24, uint8_t* tagflags) {
    if (!g) return TEC_EINVAL;
    if (first + n > g->n_resident) return gfail(g, TEC_EINVAL, "tec_gbam_fetch: rows out of range");
    if (!n) return TEC_OK;
    G_HIP(hipSetDevice(g->ctx->device));
    const RecArrays ra = rec_arrays(g);
    if (tid) G_HIP(hipMemcpy(tid, ra.chrom + first, n * 4, hipMemcpyDeviceToHost));
    if (pos) G_HIP(hipMemcpy(pos, ra.pos + first, n * 4, hipMemcpyDeviceToHost));
    if (end) G_HIP(hipMemcpy(end, ra.end + first, n * 4, hipMemcpyDeviceToHost));
    if (flag) G_HIP(hipMemcpy(flag, ra.flag + first, n * 2, hipMemcpyDeviceToHost));
    if (mapq) G_HIP(hipMemcpy(mapq, ra.mapq + first, n, hipMemcpyDeviceToHost));
    if (qkey) G_HIP(hipMemcpy(qkey, ra.qkey + first, n * 8, hipMemcpyDeviceToHost));
    if (cb24) G_HIP(hipMemcpy(cb24, ra.cb + 24 * first, n * 24, hipMemcpyDeviceToHost));
    if (ub24) G_HIP(hipMemcpy(ub24, ra.ub + 24 * first, n * 24, hipMemcpyDeviceToHost));
    if (tagflags) G_HIP(hipMemcpy(tagflags, ra.tf + first, n, hipMemcpyDeviceToHost));
    return TEC_OK;
}
