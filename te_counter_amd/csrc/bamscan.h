// BAM record parsing shared by the host decoder (bamio.hip) and the on-GPU decoder (gbam.hip): the
// same source compiles for both, so the CPU test suite exercises what the kernels run.
//
// Stands where the reference reads pysam AlignedSegment attributes (te_count/te_count.py:81-98,
// 204-214, 394-438); pysam / htslib are not part of the reference tree, so the rules are the SAM/BAM
// specification's (SAMv1 section 4.2) plus the two htslib behaviours that change a value the
// reference consumes: bam_endpos() counts a CIGAR without reference-consuming ops as one base, and
// a long CIGAR parked in a CG:B,I tag behind a placeholder CIGAR that starts with <l_seq>S is the record's real CIGAR.
#pragma once

#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define TEC_HD __host__ __device__ __forceinline__
#else
#define TEC_HD inline
#endif

namespace tec_bamscan {

constexpr int32_t kNoEnd = INT32_MIN;  // reference_end of a record without CIGAR (pysam: None)

TEC_HD uint32_t rd32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;  // little-endian on both x86-64 and gfx950
}
TEC_HD uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// Up to eight bytes at p as a little-endian word; bytes at or beyond e read as `fill`.  One 8-byte
// load wherever the record allows it: a lane that walks names and tag strings byte by byte spends
// its time in dependent one-byte round trips.
TEC_HD uint64_t word_upto(const uint8_t* p, const uint8_t* e, uint8_t fill) {
    uint64_t w;
    if (e - p >= 8) {
        memcpy(&w, p, 8);
        return w;
    }
    w = 0x0101010101010101ull * fill;
    for (int k = 0; k < 8 && p + k < e; ++k) w = (w & ~(0xFFull << (8 * k))) | ((uint64_t)p[k] << (8 * k));
    return w;
}
// 0x80 in every byte of w that is zero (exact: no carries between bytes).
TEC_HD uint64_t zero_bytes(uint64_t w) {
    const uint64_t m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((w & m) + m) | w | m);
}

// Size in bytes of the value of an aux field of type `ty` starting at p (p < e), or 0 if it cannot be
// told (unknown type, or the value runs past e).
TEC_HD size_t aux_value_size(char ty, const uint8_t* p, const uint8_t* e) {
    size_t sz;
    switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': {
            const uint8_t* z = p;
            uint64_t hit = 0;
            while (z < e && !(hit = zero_bytes(word_upto(z, e, 1)))) z += 8;
            if (!hit) return 0;  // no terminator inside the record
            sz = (size_t)(z - p) + (size_t)(__builtin_ctzll(hit) >> 3) + 1;
            break;
        }
        case 'B': {
            if (e - p < 5) return 0;
            const char st = (char)p[0];
            size_t es;
            switch (st) {
                case 'c': case 'C': es = 1; break;
                case 's': case 'S': es = 2; break;
                case 'i': case 'I': case 'f': es = 4; break;
                default: return 0;
            }
            sz = 5 + (size_t)rd32(p + 1) * es;
            break;
        }
        default: return 0;
    }
    return (size_t)(e - p) < sz ? 0 : sz;
}

// '_'.join(qname.split('/')[0:-1]) (te_count.py:92) as a 64-bit FNV-1a hash: mate names are only
// compared for equality.  A name without '/' has the empty key.
TEC_HD uint64_t mate_key_hash(const uint8_t* q, size_t n) {
    const uint8_t* e = q + n;
    size_t last = n;
    bool any = false;
    for (size_t i = 0; i < n; i += 8) {  // the last '/'
        const uint64_t z = zero_bytes(word_upto(q + i, e, 0) ^ 0x2F2F2F2F2F2F2F2Full);
        if (z) {
            last = i + (size_t)((63 - __builtin_clzll(z)) >> 3);
            any = true;
        }
    }
    uint64_t h = 1469598103934665603ull;
    if (!any) return h;
    for (size_t i = 0; i < last; i += 8) {
        const uint64_t w = word_upto(q + i, e, 0);
        const size_t m = last - i < 8 ? last - i : 8;
        for (size_t b = 0; b < m; ++b) {
            uint8_t c = (uint8_t)(w >> (8 * b));
            c = c == '/' ? (uint8_t)'_' : c;
            h = (h ^ c) * 1099511628211ull;
        }
    }
    return h;
}

struct Core {
    int32_t tid, pos, end;
    uint32_t flag, mapq;
    uint32_t l_name, n_cigar, l_seq;
    const uint8_t *name, *cigar, *aux, *rec_end;
};

// Reference bases consumed by n CIGAR ops at p (M D N = X).
TEC_HD int64_t cigar_ref_len(const uint8_t* p, uint32_t n) {
    int64_t r = 0;
    for (uint32_t c = 0; c < n; ++c) {
        const uint32_t v = rd32(p + 4 * (size_t)c), op = v & 15u;
        if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) r += v >> 4;
    }
    return r;
}

// The fixed part of the record at rec (rec points at block_size; the whole record is in memory).
// want_end: also compute reference_end (a CIGAR walk).  false = the record's layout is inconsistent.
TEC_HD bool parse_core(const uint8_t* rec, bool want_end, Core* c) {
    const uint32_t bs = rd32(rec);
    const uint8_t* p = rec + 4;
    c->rec_end = p + bs;
    c->tid = (int32_t)rd32(p);
    c->pos = (int32_t)rd32(p + 4);
    c->l_name = p[8];
    c->mapq = p[9];
    c->n_cigar = rd16(p + 12);
    c->flag = rd16(p + 14);
    c->l_seq = rd32(p + 16);
    c->name = p + 32;
    c->cigar = c->name + c->l_name;
    const uint64_t fixed = 32ull + c->l_name + 4ull * c->n_cigar + ((uint64_t)c->l_seq + 1) / 2 + c->l_seq;
    if (bs < 32u || fixed > bs) return false;
    c->aux = p + fixed;
    c->end = kNoEnd;
    if (want_end && c->n_cigar) {
        int64_t reflen;
        const uint8_t* cg = nullptr;
        uint32_t cg_n = 0;
        // htslib (sam.c bam_tag2cigar): a CIGAR whose first op is <l_seq>S on a placed read is a placeholder for the ops in
        // the record's first CG tag, if that is a B array of (u)int32 holding at least n_cigar_op (and fewer than 2^29) ops
        if (c->tid >= 0 && c->pos >= 0) {
            const uint32_t op0 = rd32(c->cigar);
            if ((op0 & 15u) == 4u && (op0 >> 4) == c->l_seq) {
                const uint8_t* a = c->aux;
                while (c->rec_end - a >= 3) {
                    const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
                    a += 3;
                    if (t0 == 'C' && t1 == 'G') {
                        if (ty == 'B' && c->rec_end - a >= 5 && (a[0] == 'I' || a[0] == 'i')) {
                            const uint32_t n = rd32(a + 1);
                            if (n >= c->n_cigar && n < (1u << 29) && (uint64_t)(c->rec_end - a - 5) >= 4ull * n) {
                                cg = a + 5;
                                cg_n = n;
                            }
                        }
                        break;
                    }
                    const size_t sz = aux_value_size(ty, a, c->rec_end);
                    if (!sz) break;
                    a += sz;
                }
            }
        }
        reflen = cg ? cigar_ref_len(cg, cg_n) : cigar_ref_len(c->cigar, c->n_cigar);
        // htslib bam_endpos(): a CIGAR that consumes no reference base still covers one
        c->end = (int32_t)((int64_t)c->pos + (reflen ? reflen : 1));
    }
    return true;
}

// CB (else CR) and UB (else UR) 'Z' values of the aux block [p, e) -> zero-padded 24-byte fields.
// tagflags: bit 0 CB|CR present, bit 1 UB|UR present, bit 2 barcode > 24 bytes, bit 3 UMI > 24 bytes.
// The reference builds dict(read.get_tags()) (te_count.py:403): with a repeated tag the LAST value wins.
TEC_HD uint32_t scan_tags(const uint8_t* p, const uint8_t* e, uint8_t* cb24, uint8_t* ub24) {
    const uint8_t *cb = nullptr, *cr = nullptr, *ub = nullptr, *ur = nullptr;
    uint32_t ncb = 0, ncr = 0, nub = 0, nur = 0;
    while (e - p >= 3) {
        const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
        p += 3;
        const size_t sz = aux_value_size(ty, p, e);
        if (!sz) break;
        if (ty == 'Z') {
            const uint32_t n = (uint32_t)sz - 1u;
            if (t0 == 'C' && t1 == 'B') { cb = p; ncb = n; }
            else if (t0 == 'C' && t1 == 'R') { cr = p; ncr = n; }
            else if (t0 == 'U' && t1 == 'B') { ub = p; nub = n; }
            else if (t0 == 'U' && t1 == 'R') { ur = p; nur = n; }
        }
        p += sz;
    }
    uint32_t tf = 0;
    // the 24-byte fields, zero padded, as three words each (the value's terminating NUL and what follows it are masked off)
    auto field = [](const uint8_t* v, uint32_t n, uint8_t* out24) {
        for (uint32_t k = 0; k < 3u; ++k) {
            uint64_t w = 0;
            if (v && n <= 24u && 8u * k < n) {
                w = word_upto(v + 8u * k, v + n, 0);
            }
            memcpy(out24 + 8u * k, &w, 8);
        }
    };
    const uint8_t* b = cb ? cb : cr;
    const uint32_t nb = cb ? ncb : ncr;
    if (b) tf |= nb > 24u ? 5u : 1u;
    field(b, nb, cb24);
    const uint8_t* u = ub ? ub : ur;
    const uint32_t nu = ub ? nub : nur;
    if (u) tf |= nu > 24u ? 10u : 2u;
    field(u, nu, ub24);
    return tf;
}

// Could a BAM record start at byte x of the inflated stream [s, s + total)?  Used only to GUESS where
// the records of a byte range or of a 64 KiB segment start; every guess is afterwards verified
// against the record chain that reaches it, so a wrong answer costs time, never correctness.
// ref_len may be null.  depth: how many following records must look right as well.
TEC_HD bool plausible_header(const uint8_t* s, uint64_t total, uint64_t x, int32_t n_ref, const uint32_t* ref_len, uint64_t* next) {
    if (total - x < 36) return false;  // shortest record: 32 fixed + 1-byte name + ... (block_size >= 33)
    const uint8_t* r = s + x;
    const uint32_t bs = rd32(r);
    if (bs < 33u || bs > (1u << 28)) return false;
    const int32_t tid = (int32_t)rd32(r + 4), pos = (int32_t)rd32(r + 8);
    const uint32_t l_name = r[12], n_cig = rd16(r + 16), l_seq = rd32(r + 20);
    const int32_t ntid = (int32_t)rd32(r + 24), npos = (int32_t)rd32(r + 28);
    if (tid < -1 || tid >= n_ref || ntid < -1 || ntid >= n_ref || pos < -1 || npos < -1) return false;
    if (ref_len && tid >= 0 && (uint32_t)pos > ref_len[tid]) return false;
    if (l_name == 0u || l_seq > (1u << 28)) return false;
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
    if (fixed > bs) return false;
    // the name: printable characters and a terminating NUL (when those bytes are in view)
    const uint64_t name_end = x + 36 + l_name;
    if (name_end <= total) {
        if (r[36 + l_name - 1] != 0) return false;
        for (uint32_t k = 0; k + 1 < l_name; ++k)
            if (r[36 + k] < 0x21 || r[36 + k] > 0x7E) return false;
        // CIGAR op codes 0..8
        const uint32_t look = n_cig < 4u ? n_cig : 4u;
        if (name_end + 4ull * look <= total)
            for (uint32_t k = 0; k < look; ++k)
                if ((rd32(r + 36 + l_name + 4 * k) & 15u) > 8u) return false;
    }
    *next = x + 4 + bs;
    return true;
}
TEC_HD bool plausible_record(const uint8_t* s, uint64_t total, uint64_t x, int32_t n_ref, const uint32_t* ref_len, int depth) {
    uint64_t nx = 0;
    if (!plausible_header(s, total, x, n_ref, ref_len, &nx)) return false;
    for (int d = 0; d < depth; ++d) {
        if (total - (nx < total ? nx : total) < 36) return nx <= total + (1u << 28);  // the chain leaves the view: accept
        uint64_t nn = 0;
        if (!plausible_header(s, total, nx, n_ref, ref_len, &nn)) return false;
        nx = nn;
    }
    return true;
}

}  // namespace tec_bamscan
