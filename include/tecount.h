/*
 * tecount.h -- C ABI of libtecount.so: the MI355X (gfx950) read-vs-annotation overlap counter.
 *
 * The reference (oaxiom/te_counter) has no FFI seam: its hot path is three Python methods of
 * one class, measureTE.parse_bamse / parse_bampe / sc_parse_bamse (te_count/te_count.py:167,
 * 42, 298), called from bin/te_count:98-120.  The entry points below are what a binding of that
 * path would bind; each one names the reference code it replaces.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every call returns 0 on success and a negative TEC_E* code on failure; tec_last_error()
 *     gives the message.  Nothing throws across the boundary.
 *   - "host" pointers are ordinary (ideally pinned, see tec_alloc_pinned) memory; "device"
 *     pointers are HBM addresses on the context's GPU.
 *   - a context is single-producer (one thread submits); calls drop no locks of the caller.
 *   - coordinates are the BAM/pysam ones the reference uses verbatim: 0-based reference_start,
 *     reference_end one past the last aligned base; annotation left/right exactly as stored in
 *     the index (no conversion, te_count/miniglbase/location.py:31).
 */
#ifndef TECOUNT_H
#define TECOUNT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEC_VERSION 100 /* 0.1.0 */

#define TEC_OK 0
#define TEC_EINVAL -1   /* bad argument / call order */
#define TEC_EHIP -2     /* a HIP runtime call failed */
#define TEC_ENOMEM -3
#define TEC_ESTATE -4   /* e.g. submit before begin, no index uploaded */
#define TEC_EENHANCER -5 /* bulk: a read whose only countable type is 'enhancer'; the reference
                            dies with NameError there (te_count.py:145-147, 258-260) */
#define TEC_EOVERFLOW -6
#define TEC_ENOCB -7     /* --sc: a record has neither CB nor CR (reference: AssertionError, te_count.py:409) */
#define TEC_ENOUB -8     /* --sc: a record has neither UB nor UR (reference: AssertionError, te_count.py:426) */
#define TEC_ENOSTRAND -9 /* --sc: a hit feature has no 'strand' key (reference: KeyError, te_count.py:661) */

/* The reference's bucket width: a semantic constant (te_count/miniglbase/config.py:36). */
#define TEC_BUCKET_SIZE 10000

/* Feature classes, derived by the host from the index's `type` strings.  Which classes count
 * where: te_count.py:134-147 / 247-260 (bulk), 662-684 (single cell). */
#define TEC_CLASS_OTHER 0    /* e.g. miRNA: can make a read "assigned" but never counts      */
#define TEC_CLASS_GENE 1     /* protein_coding | lincRNA | lncRNA                            */
#define TEC_CLASS_TE 2
#define TEC_CLASS_SNRNA 3    /* counted in bulk, not in single cell                          */
#define TEC_CLASS_ENHANCER 4 /* crashes bulk, counted in single cell                         */

/* meta word of a feature: name id | class << 26 | strand id << 29 */
#define TEC_META_NAME_BITS 26
#define TEC_META_CLASS_SHIFT 26
#define TEC_META_STRAND_SHIFT 29
#define TEC_STRAND_PLUS 0  /* '+' */
#define TEC_STRAND_MINUS 1 /* '-' */
                           /* 2..6: any other strand string, one id per distinct string */
#define TEC_STRAND_MISSING 7 /* the feature has no 'strand' key at all */

/* Record flag bits the path looks at (SAM spec; pysam is_unmapped / is_qcfail / is_duplicate /
 * is_reverse, te_count.py:81-86, 204, 394, 438). */
#define TEC_FLAG_UNMAPPED 0x4
#define TEC_FLAG_REVERSE 0x10
#define TEC_FLAG_QCFAIL 0x200
#define TEC_FLAG_DUP 0x400

typedef struct tec_ctx tec_ctx;

/* Log-only counters of the bulk loops (te_count.py:158-162, 269-274). */
typedef struct tec_stats {
    uint64_t units;     /* records (SE) or pairs (PE) submitted                                   */
    uint64_t assigned;  /* "Reads were assigned to a gene"                                        */
    uint64_t lowq;      /* "Read quality is too low"                                              */
    uint64_t badchrom;  /* "Reads mapped to an invalid chromosome"                                */
    uint64_t qcfail;    /* "Reads are QC fails"                                                   */
} tec_stats;

int tec_version(void);
/* Message of the last failure on this context (ctx == NULL: of the last failed tec_create). */
const char* tec_last_error(const tec_ctx* ctx);

/* One context per GPU.  device_id is a HIP device ordinal. */
int tec_create(int device_id, tec_ctx** out);
void tec_destroy(tec_ctx* ctx);

/* Page-locked host memory for the SoA batches (so that hipMemcpyAsync really is async). */
int tec_alloc_pinned(tec_ctx* ctx, size_t bytes, void** out);
int tec_free_pinned(tec_ctx* ctx, void* p);

/* ---- Index -------------------------------------------------------------------------------
 * Replaces the unpickled genelist (`self.genome.buckets` / `.linearData`) that the reference
 * loops reach into (te_count.py:31-35, 68-73, 100-126; built by
 * te_count/miniglbase/genelist.py:332-396).
 *
 * Input: one row per annotation feature, in any order.  chrom_id indexes the caller's table of
 * index-side chromosome keys (0 <= chrom_id < n_chrom).  The library builds, on the host, a
 * per-chromosome CSR over fine bins of 2^bin_shift bp (entries = {left, right, meta}) and
 * uploads it to HBM.  The reference's 10 kb "is this feature a candidate" rule is evaluated in
 * closed form per entry on the device, so the result is independent of bin_shift.
 * bin_shift == 0 lets the library choose.
 */
int tec_index_build(tec_ctx* ctx, uint64_t n_features, const int32_t* chrom_id, const int32_t* left,
                    const int32_t* right, const uint32_t* meta, uint32_t n_chrom, uint32_t n_names,
                    int bin_shift);
/* Facts about the resident index (any pointer may be NULL). */
int tec_index_info(const tec_ctx* ctx, uint64_t* n_entries, uint64_t* n_bins, int* bin_shift,
                   uint64_t* device_bytes);

/* ---- Bulk counting -----------------------------------------------------------------------
 * Replaces the `while 1:` loops of parse_bamse (te_count.py:201-263) and parse_bampe
 * (te_count.py:76-152).  A "unit" is one record (SE) or one pair of consecutive records (PE);
 * the host decoder has already
 *   - mapped the read's chromosome to an index chrom id, -1 if
 *     reference_name.replace('chr','') is not an index key (te_count.py:96-102, 212-218);
 *   - set p1 = reference_start, p2 = reference_end (SE, te_count.py:213-214) or
 *     p1 = read1.reference_start, p2 = read2.reference_start (PE, te_count.py:97-98);
 *   - PE: OR-ed both mates' flags into `flag` (except TEC_FLAG_REVERSE, which is read 1's: only
 *     tec_bulk_set_stranded looks at it) and taken read1's mapq (te_count.py:81-88).
 * The unit layout is therefore the same for SE and PE: 15 bytes per unit.
 */
int tec_bulk_begin(tec_ctx* ctx, int mapq_min);

/* Host SoA batch -> HBM (hipMemcpyAsync on the context's copy stream) -> count kernel.
 * Returns as soon as the work is queued.  *ticket (optional) identifies the batch: the host
 * buffers may be overwritten once tec_bulk_wait(ctx, ticket) has returned. */
int tec_bulk_submit(tec_ctx* ctx, uint64_t n_units, const int32_t* chrom_id, const int32_t* p1,
                    const int32_t* p2, const uint16_t* flag, const uint8_t* mapq, uint64_t* ticket);
int tec_bulk_wait(tec_ctx* ctx, uint64_t ticket);

/* Same, for a batch that is already resident in HBM (device pointers); no copy is made and the
 * buffers must stay valid until tec_bulk_sync / tec_bulk_finish. */
int tec_bulk_submit_device(tec_ctx* ctx, uint64_t n_units, const int32_t* d_chrom_id, const int32_t* d_p1,
                           const int32_t* d_p2, const uint16_t* d_flag, const uint8_t* d_mapq);

/* Wait for everything submitted so far. */
int tec_bulk_sync(tec_ctx* ctx);

/* EXTENSION, no counterpart in the reference: bulk counting that respects the read strand.  The reference refuses
 * `--strand` for bulk input before it reads anything (NotImplementedError, te_count.py:58-59, 183-184), so no reference
 * output exists for it and nothing pins this mode; it applies the single-cell path's rule (te_count.py:654, 665: a
 * feature counts only if its strand equals '-' for a reverse read, '+' otherwise) to the bulk loops.  The caller builds
 * the index STRAND-SPLIT -- chromosome 2c holds the '+' features of chromosome c, 2c + 1 its '-' features, features of
 * any other strand are left out -- and calls this between tec_bulk_begin and the first submit; units keep the plain
 * chromosome id c, and TEC_FLAG_REVERSE of a unit's flag selects the copy (PE: the host puts read 1's bit there).
 * tec_bulk_begin switches it off again. */
int tec_bulk_set_stranded(tec_ctx* ctx, int on);

/* One batch counted from start to finish WITHOUT ANY ALLOCATION.  No counterpart in the reference (which allocates
 * Python objects per read): this entry point is for callers that cannot call hipMalloc -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation.  It is tec_bulk_begin + tec_bulk_submit_device +
 * tec_bulk_finish for one device-resident batch, with the results left in HBM:
 *   d_counts_out  n_names uint64, in the CALLER's name ids (as tec_bulk_finish delivers them); overwritten
 *   d_stats_out   6 uint64 {units, assigned, lowq, badchrom, qcfail, units that hit only 'enhancer' features}
 *                 (the last one non-zero is where tec_bulk_finish returns TEC_EENHANCER); may be NULL
 *   stream        a hipStream_t, or NULL for the context's own compute stream; the call only queues work on it
 * The scratch memory of the launch (per-workgroup histogram rows, an internal count vector, tallies) is a fixed array
 * of 24.4 MiB inside the library's code object, one copy per GPU for the life of the process, shared by every call on
 * that GPU: calls that overlap on one GPU race, and the limits are fixed -- at most 49 152 names in the index and 2^31
 * units per call (TEC_EOVERFLOW beyond).  It neither needs nor disturbs an open tec_bulk_begin run. */
int tec_bulk_count_noalloc(tec_ctx* ctx, uint64_t n_units, const int32_t* d_chrom_id, const int32_t* d_p1, const int32_t* d_p2,
                           const uint16_t* d_flag, const uint8_t* d_mapq, int mapq_min, uint64_t* d_counts_out,
                           uint64_t* d_stats_out, void* stream);

/* Device address of the running uint64 count vector (n_names entries, in the library's INTERNAL name order, identical on every GPU that holds the same index; tec_bulk_finish maps it back to name ids) and of
 * the 5-word stats block {units, assigned, lowq, badchrom, qcfail}; for the caller's multi-GPU
 * sum-reduce (RCCL through torch.distributed).  Valid until tec_destroy / the next
 * tec_index_build. */
int tec_bulk_device_buffers(tec_ctx* ctx, void** d_counts, void** d_stats);

/* Drain, then copy counts (n_names uint64) and stats to the host.  Returns TEC_EENHANCER if any
 * read hit only 'enhancer'-class features (counts are still delivered).  The count vector is
 * what final_results holds when the reference loop ends (te_count.py:165, 277), keyed by name
 * id instead of by string. */
int tec_bulk_finish(tec_ctx* ctx, uint64_t* counts, tec_stats* stats);

/* Duration (ms, HIP events on the stream the kernel runs on) and number of count-kernel
 * launches since tec_bulk_begin; for bench.py's roofline figure. */
int tec_bulk_kernel_time(tec_ctx* ctx, double* total_ms, uint64_t* launches);

/* The individual durations (ms) behind tec_bulk_kernel_time: *n = number of timed launches since
 * tec_bulk_begin; the first min(capacity, *n) of them are written to ms (may be NULL). */
int tec_bulk_launch_times(tec_ctx* ctx, float* ms, uint64_t capacity, uint64_t* n);

/* With more names than the per-workgroup LDS histogram holds (77 k), hits on the remaining names are appended to a
 * list that is sorted and run-length encoded after every launch: its GPU time (ms) and length since tec_bulk_begin. */
int tec_bulk_spill_time(tec_ctx* ctx, double* total_ms, uint64_t* names);

/* Duration (ms) of the small kernels that fold the per-workgroup histogram rows into the count
 * vector after each count kernel, since tec_bulk_begin. */
int tec_bulk_reduce_time(tec_ctx* ctx, double* total_ms);


/* ---- Single-cell counting (--sc --se) ------------------------------------------------------
 * Replaces measureTE.sc_parse_bamse (te_count.py:298-707).  The host decoder hands over, per
 * record: cls = id of the read-side chromosome string reference_name.replace('chr','')
 * (te_count.py:431), pos/end = reference_start/reference_end, flag, mapq, the CB (else CR) and
 * UB (else UR) tag values as zero-padded 24-byte fields, and tagflags (bit 0: CB|CR present,
 * bit 1: UB|UR present, bit 2: barcode longer than 24 bytes, bit 3: UMI longer than 24 bytes -- a
 * record that reaches the (CB,UB) collapse with bit 3 set makes tec_sc_part1 fail with TEC_EOVERFLOW).  Everything the reference keeps in
 * dicts, sets and spill files stays in HBM; the spill ("bundle") boundaries, the bundle reader's
 * line-drop rule and the cross-bundle merge are reproduced exactly.
 *
 * Call order: begin, submit* (file order), part1, part1_fetch, [host picks the best barcodes,
 * te_count.py:502], part23, fetch, stats, end.
 */
typedef struct tec_sc_stats_t {
    uint64_t records;         /* submitted                                                     */
    uint64_t qcfail, lowq;    /* te_count.py:394-400                                           */
    uint64_t invalid_barcode; /* not on the whitelist (te_count.py:412-415)                    */
    uint64_t skipped_chrom;   /* '_' or 'alt' chromosomes (te_count.py:432-433)                */
    uint64_t survivors;       /* records that reach the (CB,UB) collapse                       */
    uint64_t keys;            /* distinct (CB,UB)                                              */
    uint64_t bundles;         /* spill epochs                                                  */
    uint64_t kept_lines;      /* "Preserved N" (te_count.py:545,572)                           */
    uint64_t assigned;        /* __read_assinged_to_feature (te_count.py:686)                  */
    uint64_t gpu_us;          /* GPU stream time of submit + part 1 + parts 2/3, microseconds  */
} tec_sc_stats_t;

/* whitelist24: n_whitelist x 24 bytes, zero padded, strictly ascending bytewise == the order of
 * sorted(set(lines)) (te_count.py:338): a barcode's id is its row.  bundle_keys: the reference's
 * literal 1e7 (te_count.py:377).  class_index_chrom[c] = index chrom id of read-side chromosome
 * string c, or -1 (te_count.py:614); class_skip[c] = '_' in c or 'alt' in c (te_count.py:432). */
int tec_sc_begin(tec_ctx* ctx, int stranded, int mapq_min, uint32_t n_whitelist, const uint8_t* whitelist24,
                 uint64_t bundle_keys, uint32_t n_class, const int32_t* class_index_chrom, const uint8_t* class_skip);
/* Host SoA batch, in file order.  The caller's buffers are free on return (they have been copied into a pinned
 * staging ring); the copy to HBM and the filter / compaction kernels are queued behind earlier batches and run
 * while the caller prepares the next one.  Any later tec_sc_* call that needs their result waits for them. */
int tec_sc_submit(tec_ctx* ctx, uint64_t n, const int32_t* cls, const int32_t* pos, const int32_t* end,
                  const uint16_t* flag, const uint8_t* mapq, const uint8_t* cb24, const uint8_t* ub24,
                  const uint8_t* tagflags);
/* Same, for a batch that is already resident in HBM (device pointers).  The kernels are QUEUED on the context's
 * compute stream: the arrays must stay untouched until work queued on that stream BEHIND this call has started
 * (libtecount's own decoder reuses them from that stream only), or until tec_sc_survivors / tec_sc_part1* returns. */
int tec_sc_submit_device(tec_ctx* ctx, uint64_t n, const int32_t* d_cls, const int32_t* d_pos, const int32_t* d_end,
                         const uint16_t* d_flag, const uint8_t* d_mapq, const uint8_t* d_cb24, const uint8_t* d_ub24,
                         const uint8_t* d_tagflags);
/* Part 1 (te_count.py:371-476).  Returns TEC_ENOCB / TEC_ENOUB where the reference asserts.
 * *n_cells_seen = number of barcodes with a non-zero raw count. */
int tec_sc_part1(tec_ctx* ctx, uint64_t* n_cells_seen);
/* ---- part 1 in steps, and the hooks a multi-GPU caller needs --------------------------------------
 * With several GPUs every rank decodes and filters its own slice of the file, then the survivors are
 * dealt to the GPUs by barcode id (tec_sc_survivors_fetch / _replace through the caller's all-to-all), so
 * that every (CB,UB) key and every cell row is local to one GPU.  What stays global is order-dependent:
 *   - record ordinals: tec_sc_set_record_base adds the number of records in the slices before this one
 *     (total_records = records in the whole file);
 *   - the spill ("bundle") boundaries: the reference flushes when its dict holds 1e7 keys
 *     (te_count.py:377), i.e. at the record that opens the bundle_keys-th NEW key since the last flush, in
 *     file order.  tec_sc_epoch_marks collects this GPU's new-key records after `lower` (-1 = start);
 *     the caller sums the counts over the GPUs and, while the sum reaches bundle_keys, finds the ordinal
 *     with global rank bundle_keys by bisection on tec_sc_epoch_rank (one GPU: tec_sc_epoch_kth);
 *   - the drop rule of the bundle reader (te_count.py:509-543) looks at the previous line of the whole
 *     bundle: tec_sc_presence / tec_sc_part23_global carry the OR of all GPUs' (bundle, barcode) tables;
 *   - the barcode selection (te_count.py:502) over the gathered tec_sc_part1_fetch rows.
 * tec_sc_part1 / tec_sc_part23 run the same steps for a single GPU. */
int tec_sc_set_record_base(tec_ctx* ctx, uint64_t base, uint64_t total_records);
int tec_sc_survivors(tec_ctx* ctx, uint64_t* n);
int tec_sc_survivors_fetch(tec_ctx* ctx, uint32_t* bcid, uint64_t* u0, uint64_t* u1, uint64_t* u2, uint64_t* rec, int32_t* klass,
                           int32_t* left, int32_t* rite);
int tec_sc_survivors_replace(tec_ctx* ctx, uint64_t n, const uint32_t* bcid, const uint64_t* u0, const uint64_t* u1,
                             const uint64_t* u2, const uint64_t* rec, const int32_t* klass, const int32_t* left,
                             const int32_t* rite);
/* The same exchange without host memory (extension: the reference is one process, te_count.py:298-707; these stand where
 * a multi-process port of its part 1 would shuffle its `umis` dict by barcode).  tec_sc_deal groups this GPU's survivors
 * by destination GPU = barcode id % world (file order kept inside a group) into the "deal" columns and reports the
 * group sizes; the caller moves group d to GPU d -- RCCL point-to-point on tensors wrapped over tec_sc_deal_column /
 * tec_sc_adopt_column pointers, or tec_sc_adopt_copy between contexts of one process -- into the landing columns that
 * tec_sc_adopt_begin(n_rows) sized; tec_sc_adopt_commit orders what arrived by record ordinal and makes it this GPU's
 * survivors.  Columns 0..7 = bcid u0 u1 u2 rec klass left rite, element sizes 4 8 8 8 8 4 4 4 bytes.
 * tec_sc_host_row_bytes: bytes of survivor columns that tec_sc_survivors_fetch copied to the host in this run. */
int tec_sc_deal(tec_ctx* ctx, uint32_t world, uint64_t* counts);
int tec_sc_deal_column(tec_ctx* ctx, int column, void** d_ptr, uint32_t* elem_bytes);
int tec_sc_adopt_begin(tec_ctx* ctx, uint64_t n_rows);
int tec_sc_adopt_column(tec_ctx* ctx, int column, void** d_ptr, uint32_t* elem_bytes);
int tec_sc_adopt_copy(tec_ctx* ctx, int column, uint64_t dst_row, const void* d_src, uint64_t n_rows);
int tec_sc_adopt_commit(tec_ctx* ctx);
int tec_sc_host_row_bytes(tec_ctx* ctx, uint64_t* bytes);
int tec_sc_part1_sort(tec_ctx* ctx, uint64_t* n_keys);
/* How tec_sc_part1_sort ordered the survivors (the reference's `umis` dict and its sorted bundle lines,
 * te_count.py:371-390, 464-473, need no such choice): *bits = length of the (barcode id, UMI) key string without the bits
 * that are equal in every survivor, sorted in one or two radix rounds; 0 = the string was longer than 96 bits (or
 * TEC_SC_SORT=columns) and the columns were sorted one by one.  Results do not depend on it. */
int tec_sc_sort_key_bits(tec_ctx* ctx, uint32_t* bits);
int tec_sc_missing_tags(tec_ctx* ctx, uint64_t* first_no_cb, uint64_t* first_no_ub);
/* Ordinal of the first surviving record whose UMI does not fit the 24-byte field (~0: none); after tec_sc_part1_sort.
 * (The reference keeps UMIs as Python strings of any length, te_count.py:420-427; this library refuses such a run.) */
int tec_sc_long_umi(tec_ctx* ctx, uint64_t* first);
int tec_sc_epoch_marks(tec_ctx* ctx, int64_t lower, uint64_t* n_marks);
int tec_sc_epoch_rank(tec_ctx* ctx, uint64_t v, uint64_t* count_le);
int tec_sc_epoch_kth(tec_ctx* ctx, uint64_t k, uint64_t* ordinal);
int tec_sc_part1_finish(tec_ctx* ctx, uint32_t n_bounds, const uint64_t* bounds, uint64_t* n_cells_seen);
int tec_sc_presence(tec_ctx* ctx, uint32_t n_bundles, uint8_t* present);
int tec_sc_part23_global(tec_ctx* ctx, uint32_t n_selected, const uint32_t* selected_bcid, uint32_t n_bundles,
                         const uint8_t* present, uint64_t* kept_lines, uint64_t* n_cells_hit, uint64_t* n_coo);

/* Barcodes seen (ascending id), their raw counts (self.barcodes of part 1) and the ordinal of the
 * record that first counted them (dict insertion order, for the stable sort at te_count.py:502). */
int tec_sc_part1_fetch(tec_ctx* ctx, uint32_t* bcid, uint64_t* count, uint64_t* first_seen);
/* Parts 2 and 3 (te_count.py:493-691) for the selected barcodes (ascending ids). */
int tec_sc_part23(tec_ctx* ctx, uint32_t n_selected, const uint32_t* selected_bcid, uint64_t* kept_lines,
                  uint64_t* n_cells_hit, uint64_t* n_coo);
/* cell_bcid/cell_hits: self.barcodes of part 3 in insertion order (ascending id);
 * coo_*: final_results as (barcode id, name id, count) triplets sorted by (barcode, name). */
int tec_sc_fetch(tec_ctx* ctx, uint32_t* cell_bcid, uint64_t* cell_hits, uint32_t* coo_cell, uint32_t* coo_name,
                 uint64_t* coo_count);
int tec_sc_stats(tec_ctx* ctx, tec_sc_stats_t* out);
int tec_sc_end(tec_ctx* ctx);

/* ---- Alignment decoding (the step before the path) ------------------------------------------
 * Replaces pysam.AlignmentFile(filename, 'r') / next(sam) and the AlignedSegment attributes the
 * reference reads (te_count.py:65,78-98 / 190,203-214 / 351,393-438).  pysam (htslib) is a
 * third-party dependency that is not part of the reference tree and is not pinned there; this
 * decoder follows the SAM/BAM specification.  BGZF-compressed BAM and plain-text SAM are both
 * accepted (pysam's 'r' mode auto-detects as well).
 *
 * tec_bam_read fills up to max_records rows: tid (-1 = '*'), pos = reference_start (0-based),
 * end = reference_end (pos + reference-consuming CIGAR length; INT32_MIN when the record has no
 * CIGAR, where pysam returns None), flag, mapq, and optionally
 *   qkey      a 64-bit hash of '_'.join(qname.split('/')[0:-1]), the string the PE path compares
 *             between mates (te_count.py:92)
 *   cb24/ub24 the CB (else CR) / UB (else UR) 'Z' tag values, zero padded to 24 bytes
 *   tagflags  bit 0 CB|CR present, bit 1 UB|UR present, bit 2 barcode > 24 bytes, bit 3 UMI > 24 bytes
 * *n_out < max_records means end of file.
 */
typedef struct tec_bam tec_bam;
int tec_bam_open(const char* path, int n_threads, tec_bam** out);
/* Multi-GPU decoding: one reader per rank over a byte range [range_begin, range_end) of the compressed
 * file.  The reader delivers exactly the records that START inside a BGZF block whose file offset lies
 * in the range (a record that straddles into the next range is completed from the following blocks).
 * A range that does not begin at 0 finds its first BGZF block by the block header's fixed bytes and
 * GUESSES its first record start (field plausibility over three consecutive records); the caller makes
 * the decomposition exact by comparing tokens: range k's end token -- {file offset of the BGZF block,
 * offset inside its inflated bytes} of the first record it does NOT own -- must equal range k+1's first
 * token.  The reference reads the file front to back with one pysam iterator (te_count.py:65,78-79). */
int tec_bam_open_range(const char* path, int n_threads, uint64_t range_begin, uint64_t range_end, tec_bam** out);
/* Valid after the reader has returned its last record (*finished != 0).  *has_first == 0: no record
 * starts in this range. */
int tec_bam_range_tokens(tec_bam* r, uint64_t first[2], uint64_t end[2], int* has_first, int* finished);
uint64_t tec_bam_file_size(const tec_bam* r);
void tec_bam_close(tec_bam* r);
/* Raw DEFLATE (RFC 1951) of one BGZF block body into exactly out_len bytes: the decoder the BAM
 * reader uses in place of pysam / htslib (te_count.py:65, 190, 351); use_zlib == 1 runs zlib's
 * inflate instead (the referee in tests), use_zlib == 2 the host build of the decoder one GPU lane
 * runs in the on-GPU front end (csrc/ginflate.h -- the same source, so the CPU suite pins it against
 * zlib).  TEC_OK, or TEC_EINVAL for a malformed / short / long stream. */
int tec_inflate_raw(const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_len, int use_zlib);

const char* tec_bam_last_error(const tec_bam* r);
int tec_bam_n_refs(const tec_bam* r);
uint32_t tec_bam_ref_len(const tec_bam* r, int i);
const char* tec_bam_ref_name(const tec_bam* r, int i);
int tec_bam_read(tec_bam* r, uint64_t max_records, int32_t* tid, int32_t* pos, int32_t* end, uint16_t* flag,
                 uint8_t* mapq, uint64_t* qkey, uint8_t* cb24, uint8_t* ub24, uint8_t* tagflags, uint64_t* n_out);

/* ---- Alignment decoding ON THE GPU ----------------------------------------------------------
 * The same boundary as tec_bam_* (pysam.AlignmentFile + next(sam), te_count.py:65,78-98 / 190,203-214 /
 * 351,393-438) for BGZF-compressed BAM files, with the decoding moved to the MI355X: the host copies
 * the COMPRESSED file bytes into pinned memory and on to HBM; a kernel inflates the BGZF blocks (one
 * lane per block), kernels find the record boundaries, extract the fields the reference reads and --
 * for PE -- pair consecutive records; the resulting rows feed tec_bulk_submit_device /
 * tec_sc_submit_device directly.  Results are those of the host decoder bit for bit (a block the
 * inflate kernel refuses is redone by the host decoder; record boundaries are verified, not guessed).
 *
 * range_begin / range_end: byte range of the compressed file owned by this reader (0, UINT64_MAX = the
 * whole file); ownership and token rules as for tec_bam_open_range. */
typedef struct tec_gbam tec_gbam;
typedef struct tec_gbam_rec {  /* the fields of one record that PE pairing needs (te_count.py:81-98) */
    int32_t chrom;             /* index chromosome id of the record's reference, -1: not in the index */
    int32_t pos;
    uint16_t flag;
    uint8_t mapq;
    uint64_t qkey;             /* hash of the mate-name key (te_count.py:92) */
} tec_gbam_rec;
typedef struct tec_gbam_report {
    uint64_t n_records;            /* records owned by this reader */
    int has_records;
    uint64_t first_token[2], end_token[2];
    uint64_t first_noend_live;     /* SE: ordinal of the first record that passes every filter but has no CIGAR
                                      (pysam: reference_end None -> the reference dies, te_count.py:223); ~0: none */
    uint64_t first_name_mismatch;  /* PE: unit ordinal of the first live pair whose mate names differ
                                      (the reference dies, te_count.py:92-94); ~0: none */
    uint64_t units_counted;        /* PE, not deferred: pairs handed to the count kernel */
    int odd_record_left;           /* PE, not deferred: a last unpaired record was dropped (te_count.py:78-79) */
    tec_gbam_rec first_rec, last_rec; /* PE, deferred: for the cross-range pairing */
    uint64_t blocks, blocks_redone_on_host, compressed_bytes, inflated_bytes, segment_repairs;
    /* wall seconds of the decode loop by phase: waiting for the staging thread (pread into pinned memory),
     * H2D + inflate kernel, segment guess / walk / chain, row extraction (+ pairing, + count kernel) */
    double seconds_stage_wait, seconds_inflate, seconds_segments, seconds_rows;
    double seconds_stage_alloc, seconds_stage_read; /* inside the staging thread: pinned allocation, pread */
} tec_gbam_report;
int tec_gbam_open(tec_ctx* ctx, const char* path, uint64_t range_begin, uint64_t range_end, tec_gbam** out);
void tec_gbam_close(tec_gbam* g);
const char* tec_gbam_last_error(const tec_gbam* g);
int tec_gbam_n_refs(const tec_gbam* g);
const char* tec_gbam_ref_name(const tec_gbam* g, int i);
uint64_t tec_gbam_file_size(const tec_gbam* g);
/* Bulk (after tec_bulk_begin): decodes the reader's records and counts them.  tid_to_chrom[tid]: index
 * chromosome id or -1 (te_count.py:96-102, 212-218).  paired: units are pairs of consecutive records.
 * defer_pairing (multi-GPU PE): keep the records resident in HBM instead of pairing them, because
 * which records are mates depends on the parity of the number of records in the ranges before this one;
 * finish with tec_gbam_finish_pairs once that is known. */
int tec_gbam_run_bulk(tec_gbam* g, int paired, int mapq_min, const int32_t* tid_to_chrom, uint32_t n_tid, int defer_pairing,
                      tec_gbam_report* rep);
int tec_gbam_finish_pairs(tec_gbam* g, int mapq_min, int skip_first, const tec_gbam_rec* extra_mate, uint64_t unit_ordinal0,
                          uint64_t* n_units, uint64_t* first_name_mismatch);
/* Single cell (after tec_sc_begin): tid_to_class[tid] = read-side chromosome class (tec_sc_begin). */
int tec_gbam_run_sc(tec_gbam* g, int mapq_min, const int32_t* tid_to_class, uint32_t n_tid, tec_gbam_report* rep);

/* Decode only: the reader's records stay resident in HBM (tid = BAM reference id, end = INT32_MIN without a
 * CIGAR, qkey / cb24 / ub24 / tagflags as for tec_bam_read); tec_gbam_fetch copies rows [first, first + n)
 * to host arrays (any pointer may be NULL). */
int tec_gbam_decode(tec_gbam* g, tec_gbam_report* rep);
int tec_gbam_fetch(tec_gbam* g, uint64_t first, uint64_t n, int32_t* tid, int32_t* pos, int32_t* end, uint16_t* flag,
                   uint8_t* mapq, uint64_t* qkey, uint8_t* cb24, uint8_t* ub24, uint8_t* tagflags);

/* ---- Dense single-cell TSV (the step after the path) ----------------------------------------
 * Replaces the densifying loops of measureTE.sc_save_result (te_count.py:744-754).  header: the
 * complete first line ('name\t<features...>\n'); labels: the row barcodes, '\0'-separated, in
 * row order; row_cell[i]: barcode id of row i; coo_*: (barcode id, name id, count) sorted by
 * (barcode, name) as tec_sc_fetch returns them.  Bytes identical to the reference's writer. */
int tec_write_dense_tsv(const char* path, const char* header, uint64_t header_len, uint32_t n_names, uint32_t n_rows,
                        const char* labels, const uint32_t* row_cell, uint64_t nnz, const uint32_t* coo_cell,
                        const uint32_t* coo_name, const uint64_t* coo_count);

#ifdef __cplusplus
}
#endif
#endif /* TECOUNT_H */
