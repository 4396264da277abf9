// Internal declarations shared by the translation units of libtecount.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/tecount.h"

// One fine-bin entry: 16 bytes so that a lane fetches it with a single global_load_dwordx4.
struct tec_entry {
    int32_t left;
    int32_t right;
    uint32_t meta;  // name id | class << 26 | strand << 29
    uint32_t pad;
};
static_assert(sizeof(tec_entry) == 16, "entry must be one dwordx4");

// Device view of the resident index.
struct tec_index_dev {
    const uint32_t* bin_off;      // [n_bins + 1] CSR offsets into entries, all chromosomes concatenated
    const tec_entry* entries;     // [n_entries], sorted by name id inside each fine bin
    const uint32_t* chrom_base;   // [n_chrom] first fine bin of each chromosome
    const uint32_t* chrom_nbins;  // [n_chrom] number of fine bins of each chromosome
    const uint4* recs;            // [n_bins] compact bin records {n, coff, NE x {left16|width16<<16, meta}}: see index.hip
    const uint4* cont;            // continuation pieces (two compact entries each) of bins with more than NE entries
    uint32_t zero_rec;            // index of an all-zero record behind the table (n_bins)
    uint32_t zero_piece;          // index of an all-zero continuation piece behind the table
    uint32_t n_chrom;
    uint32_t n_names;
    int shift;
    uint32_t class_mask;          // OR over features of (1 << class)
};

struct tec_event_pair {
    hipEvent_t start, stop, reduced;  // a kernel between start and stop (`reduced` unused)
};

struct tec_slot {  // one staged batch in HBM
    void* d_buf = nullptr;
    size_t cap = 0;
    hipEvent_t copied = nullptr;    // H2D of this slot finished (host buffers reusable)
    hipEvent_t consumed = nullptr;  // kernel reading this slot finished (slot reusable)
    uint64_t ticket = 0;
};

struct tec_sc_state;

struct tec_ctx {
    int device = 0;
    std::string err;
    hipStream_t s_copy = nullptr;
    hipStream_t s_compute = nullptr;

    // index
    bool have_index = false;
    tec_index_dev idx{};
    void* d_index_blob = nullptr;
    uint64_t n_entries = 0, n_bins = 0, index_bytes = 0, cont_bytes = 0;
    std::vector<uint32_t> int2ext;  // internal (hotness-ordered) name id -> caller's name id
    uint32_t* d_int2ext = nullptr;

    // single cell
    tec_sc_state* sc = nullptr;
    void* sc_parked = nullptr;   // device buffers parked by the last single-cell run (sc.hip)
    void* gbam_cache = nullptr;  // device / pinned buffers parked by the last on-GPU BAM reader (gbam.hip)

    // bulk state
    bool bulk_open = false;
    int mapq_min = 20;
    unsigned long long* d_counts = nullptr;  // [n_names]
    unsigned long long* d_stats = nullptr;   // [8]: units, assigned, lowq, badchrom, qcfail, enhancer_flag
    static constexpr int kSlots = 3;
    tec_slot slots[kSlots];
    uint64_t next_ticket = 1;
    std::vector<tec_event_pair> ev_pool;
    size_t ev_used = 0;
    double kernel_ms = 0.0, reduce_ms = 0.0;
    std::vector<float> launch_ms;  // duration of every timed count-kernel launch since tec_bulk_begin
    uint32_t* d_slabs = nullptr;  // [workgroups x histogram words] per-workgroup histogram rows of a launch
    size_t slab_cap = 0;
    int time_every = 1;
    int slab_slot = 0;              // launches whose rows wait for the next reduce
    uint32_t slab_used[8] = {0};    // rows written by each of them
    uint32_t slab_words = 0;
    std::vector<tec_event_pair> red_pool;  // start / stop of the reduce kernels
    size_t red_used = 0;
    uint64_t kernel_launches = 0, timed_launches = 0;
    int n_cu = 256;
    int unroll = 2;  // units carried per lane by the count kernel
    int aggregate = 0;
    bool stranded = false;  // this bulk run counts against a strand-split index (tec_bulk_set_stranded)
    int kernel_variant = 2;
    bool force_mixed = false;
    int rec_ne_wanted = 3;  // inline entries per bin record asked for (TEC_REC_NE)
    int rec_ne = 0;         // ... of the resident index (0: no records, bin shift too wide)
    uint32_t lds_bytes = 0;  // 0 = automatic
    int span_div = 8;
    unsigned char* noalloc_scratch = nullptr;  // device address of this GPU's copy of bulk.hip's file-scope scratch array
    // spill list of hit names beyond the LDS histogram (bulk.hip FastSink, spill.hip)
    bool spill_enabled = true;
    uint32_t *d_spill = nullptr, *d_spill_counts = nullptr;
    unsigned long long* d_spill_total = nullptr;
    size_t spill_cap_total = 0, spill_counts_cap = 0;
    std::vector<tec_event_pair> spill_ev;  // start / stop of the fold kernels
    size_t spill_ev_used = 0;
    double spill_ms = 0.0;
    uint64_t spill_names = 0;
};

int tec_fail(tec_ctx* ctx, int code, const char* fmt, ...);

#define TEC_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return tec_fail((ctx), TEC_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                \
    } while (0)

// bulk.hip
int tec_bulk_init(tec_ctx* ctx);
int tec_bulk_release(tec_ctx* ctx);
// spill.hip
int tec_spill_reserve(tec_ctx* ctx, uint32_t n_wg, uint32_t n_buckets, uint32_t slice_cap);
int tec_spill_fold(tec_ctx* ctx, uint32_t n_wg, uint32_t n_buckets, uint32_t lo, uint32_t slice_cap);
int tec_spill_harvest(tec_ctx* ctx);
void tec_spill_release(tec_ctx* ctx);
// sc.hip
void tec_sc_release(tec_ctx* ctx);
// gbam.hip
// (sc.hip) how many records the whole input is expected to hold: sizes the survivor arrays at the first batch
int tec_sc_expect_records(tec_ctx* ctx, uint64_t n_records);
void tec_gbam_cache_release(tec_ctx* ctx);
// bamio.hip: BGZF helpers shared with gbam.hip
size_t tec_bgzf_find_block(const uint8_t* m, size_t len, size_t from);
bool tec_bgzf_parse(const uint8_t* m, size_t len, size_t o, size_t* bsize, size_t* body_off, size_t* body_len, uint32_t* isize);
bool tec_bgzf_inflate_host(const uint8_t* body, size_t body_len, uint8_t* out, uint32_t isize);
