// Raw DEFLATE (RFC 1951) for one lane of a GPU wavefront: the decoder behind the on-GPU BGZF front end
// (gbam.hip), written so that the very same source also compiles for the host, where the CPU test
// suite runs it against zlib on every stream shape zlib can produce.
//
// Why a different decoder from inflate_fast.h: on the GPU one lane inflates one BGZF block, and what
// limits the number of lanes per CU is table memory.  Tables are therefore small and 16-bit:
//   literal/length  9-bit first level (512 x u16), distance 6-bit first level (64 x u16)
//   = 1152 bytes per lane -> 128 lanes per CU in 144 KiB of LDS.
// Codes longer than the first level (rare in BAM payloads) are decoded canonically, bit by bit, from
// per-length counts kept in registers (packed) and a sorted symbol list in the lane's scratch memory.
// Base values and extra-bit counts of length / distance symbols are computed arithmetically instead
// of being looked up.
//
// Replaces, on the device, the BGZF inflation that the reference gets from pysam / htslib
// (te_count/te_count.py:65,190,351); follows RFC 1951 only.
#pragma once

#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define TEC_HD __host__ __device__ __forceinline__
#else
#define TEC_HD inline
#endif

namespace tec_ginf {

constexpr int kLitRoot = 9;
constexpr int kDistRoot = 6;
constexpr uint32_t kLitEntries = 1u << kLitRoot;
constexpr uint32_t kDistEntries = 1u << kDistRoot;
constexpr uint32_t kEntriesPerLane = kLitEntries + kDistEntries;  // u16 each: 1152 bytes
constexpr uint32_t kLensPerLane = 320;                            // code-length scratch (bytes)
constexpr uint32_t kSymsPerLane = 320;                            // sorted symbols: 288 literal/length + 32 distance (u16)

// Reasons a block is handed back to the host's referee decoder (zlib / inflate_fast.h).
enum : int { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadCodeLengths = 3, kBadCode = 4, kBadSymbol = 5, kBadDistance = 6,
             kOutputOverrun = 7, kInputOverrun = 8, kShortOutput = 9 };

// Counts of codes per length, 12 bits each, five lengths per 64-bit word (lengths 1..15).
struct Counts {
    uint64_t w[3];
    TEC_HD void clear() { w[0] = w[1] = w[2] = 0; }
    TEC_HD void add(uint32_t len) {  // len 1..15
        const uint32_t k = len - 1u;
        const uint64_t one = 1ull << (12u * (k % 5u));
        const uint32_t q = k / 5u;
        w[0] += q == 0 ? one : 0ull;
        w[1] += q == 1 ? one : 0ull;
        w[2] += q == 2 ? one : 0ull;
    }
    TEC_HD uint32_t get(uint32_t len) const {
        const uint32_t k = len - 1u;
        const uint32_t q = k / 5u;
        const uint64_t v = q == 0 ? w[0] : (q == 1 ? w[1] : w[2]);
        return (uint32_t)(v >> (12u * (k % 5u))) & 0xFFFu;
    }
};

// What the canonical (bit by bit) decoder of codes longer than the first level needs.
struct LongCodes {
    Counts counts;
    uint32_t first;  // first code of length root+1, i.e. puff's `first` after the iteration len = root
    uint32_t index;  // number of symbols with length <= root
};

// Per-lane memory.  The device instantiation keeps `tab` in LDS and the two scratch arrays in global
// memory, lane-interleaved; the host instantiation uses plain arrays.
//   uint32_t tab(uint32_t i) / void set_tab(uint32_t i, uint32_t v)     i < kEntriesPerLane
//   uint32_t len(uint32_t i) / void set_len(uint32_t i, uint32_t v)     i < kLensPerLane
//   uint32_t sym(uint32_t i) / void set_sym(uint32_t i, uint32_t v)     i < kSymsPerLane
struct HostMem {
    uint16_t t[kEntriesPerLane];
    uint8_t l[kLensPerLane];
    uint16_t s[kSymsPerLane];
    uint32_t tab(uint32_t i) const { return t[i]; }
    void set_tab(uint32_t i, uint32_t v) { t[i] = (uint16_t)v; }
    uint32_t len(uint32_t i) const { return l[i]; }
    void set_len(uint32_t i, uint32_t v) { l[i] = (uint8_t)v; }
    uint32_t sym(uint32_t i) const { return s[i]; }
    void set_sym(uint32_t i, uint32_t v) { s[i] = (uint16_t)v; }
};

struct BitReader {
    const uint8_t* ip;
    const uint8_t* in;
    const uint8_t* in_end;
    uint64_t buf;
    uint32_t cnt;
    // At least 32 valid bits afterwards.  Never reads beyond in_end + 4 (the caller pads its buffers
    // by four bytes): past the end the last word is fed again, ip keeps counting, and overrun() /
    // the ip checks of the caller reject the stream.
    TEC_HD void refill() {
        if (cnt <= 32u) {
            uint32_t v;
            memcpy(&v, ip < in_end ? ip : in_end, 4);
            buf |= (uint64_t)v << cnt;
            ip += 4;
            cnt += 32u;
        }
    }
    TEC_HD uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    TEC_HD void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
    }
    TEC_HD uint32_t get(uint32_t n) {  // n <= 16, after a refill
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    // bits consumed so far beyond the end of the input (> 0: the stream is truncated)
    TEC_HD bool overrun() const { return (int64_t)(ip - in) * 8 - (int64_t)cnt > (int64_t)(in_end - in) * 8; }
};

TEC_HD uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32u - len); }

// Canonical Huffman tables of one code (RFC 1951 3.2.2).  lens(i), i < n: code lengths.  Entries
// [tab_base, tab_base + 2^root): (symbol << 4) | length for codes of at most `root` bits, 0 for
// prefixes of longer codes and for unused patterns.  sym_base: where the symbols sorted by (length,
// value) go (needed by the long-code decoder).  Returns false for an over-subscribed code or an
// incomplete one that zlib would refuse (inftrees.c: incomplete is allowed only for no code at all,
// or for a literal/length or distance code that consists of a single one-bit code).
template <class Mem, class LenFn>
TEC_HD bool build_tables(Mem& m, LenFn lens, uint32_t n, uint32_t root, uint32_t tab_base, uint32_t sym_base, bool is_cl,
                         LongCodes* lc) {
    Counts cnt;
    cnt.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t l = lens(i);
        if (l) cnt.add(l);
    }
    // completeness, offsets of every length in the sorted symbol list
    int32_t left = 1;
    uint32_t max_len = 0, total = 0;
    Counts offs;  // running offsets (<= 320: fits 12 bits)
    offs.clear();
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = cnt.get(l);
        left = (left << 1) - (int32_t)c;
        if (left < 0) return false;
        if (c) max_len = l;
        // offs[l] = total: add `total` into slot l
        const uint32_t k = l - 1u, q = k / 5u;
        const uint64_t add = (uint64_t)total << (12u * (k % 5u));
        offs.w[0] += q == 0 ? add : 0ull;
        offs.w[1] += q == 1 ? add : 0ull;
        offs.w[2] += q == 2 ? add : 0ull;
        total += c;
    }
    if (left > 0 && max_len > 0 && (is_cl || max_len != 1u)) return false;
    // symbols sorted by (length, value)
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t l = lens(i);
        if (!l) continue;
        m.set_sym(sym_base + offs.get(l), i);
        offs.add(l);
    }
    for (uint32_t i = 0; i < (1u << root); ++i) m.set_tab(tab_base + i, 0u);
    // codes in canonical order: within a length consecutive, (code + count) << 1 between lengths
    uint32_t code = 0, idx = 0;
    lc->counts = cnt;
    lc->first = 0;
    lc->index = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = cnt.get(l);
        if (l <= root) {
            for (uint32_t j = 0; j < c; ++j) {
                const uint32_t s = m.sym(sym_base + idx + j);
                const uint32_t e = (s << 4) | l;
                for (uint32_t x = bitrev(code + j, l); x < (1u << root); x += 1u << l) m.set_tab(tab_base + x, e);
            }
        }
        // (codes longer than root leave their first-level entries 0: the long-code path)
        code = (code + c) << 1;
        idx += c;
        if (l == root) {
            lc->first = code;  // first code of length root + 1
            lc->index = idx;
        }
    }
    return true;
}

// One symbol of a code whose first-level lookup missed: canonical decoding from bit root+1 on.
// Returns the symbol, or 0xFFFF if the bits are no code.
template <class Mem>
TEC_HD uint32_t decode_long(const Mem& m, BitReader& br, uint32_t root, uint32_t sym_base, const LongCodes& lc) {
    uint32_t code = bitrev(br.peek(root), root) << 1;
    br.drop(root);
    uint32_t first = lc.first, index = lc.index;
    for (uint32_t l = root + 1u; l <= 15u; ++l) {
        code |= br.peek(1);
        br.drop(1);
        const uint32_t c = lc.counts.get(l);
        if (code - first < c) return m.sym(sym_base + index + (code - first));
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return 0xFFFFu;
}

// Inflates the raw DEFLATE stream in[0, in_len) into exactly out_len bytes.  kOk, or the reason the
// block must be redone by the referee.  `in` must be readable up to in + in_len + 4.
template <class Mem>
TEC_HD int inflate_stream(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, Mem& m) {
    BitReader br{in, in, in + in_len, 0ull, 0u};
    uint32_t op = 0;
    bool last = false;
    while (!last) {
        br.refill();
        last = br.get(1) != 0;
        const uint32_t type = br.get(2);
        if (type == 3u) return kBadBlockType;
        if (type == 0u) {  // stored: to the byte boundary, LEN, NLEN, bytes
            br.drop(br.cnt & 7u);
            br.refill();
            const uint32_t len = br.get(16);
            br.refill();
            const uint32_t nlen = br.get(16);
            if ((len ^ nlen) != 0xFFFFu) return kBadStored;
            // whole bytes still in the bit buffer belong to the payload: step back to them
            br.ip -= br.cnt >> 3;
            br.buf = 0;
            br.cnt = 0;
            if ((uint32_t)(br.in_end - br.ip) < len) return kInputOverrun;
            if (out_len - op < len) return kOutputOverrun;
            for (uint32_t k = 0; k < len; ++k) out[op + k] = br.ip[k];
            br.ip += len;
            op += len;
            continue;
        }
        uint32_t hlit = 288, hdist = 32;
        if (type == 1u) {  // fixed codes (RFC 1951 3.2.6)
            for (uint32_t i = 0; i < 288u; ++i) m.set_len(i, i < 144u ? 8u : (i < 256u ? 9u : (i < 280u ? 7u : 8u)));
            for (uint32_t i = 0; i < 32u; ++i) m.set_len(288u + i, 5u);
        } else {  // dynamic codes: the code-length code first (3.2.7)
            br.refill();
            hlit = br.get(5) + 257u;
            hdist = br.get(5) + 1u;
            const uint32_t hclen = br.get(4) + 4u;
            if (hlit > 286u || hdist > 30u) return kBadCodeLengths;
            // the 19 code-length-code lengths, 3 bits each, packed in one register in symbol order
            uint64_t cl = 0;
            for (uint32_t i = 0; i < hclen; ++i) {
                br.refill();
                // order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, five bits per entry
                const uint64_t ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) |
                                        (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
                const uint64_t ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
                const uint32_t s = i < 12u ? (uint32_t)(ord_lo >> (5u * i)) & 31u : (uint32_t)(ord_hi >> (5u * (i - 12u))) & 31u;
                cl |= (uint64_t)br.get(3) << (3u * s);
            }
            LongCodes unused;
            // 7-bit table in the (not yet built) literal/length area; its symbol list goes to the tail of the scratch
            if (!build_tables(m, [&](uint32_t i) { return (uint32_t)(cl >> (3u * i)) & 7u; }, 19u, 7u, 0u, 288u, true, &unused))
                return kBadCodeLengths;
            uint32_t i = 0, prev = 0;
            const uint32_t want = hlit + hdist;
            while (i < want) {
                br.refill();
                const uint32_t e = m.tab(br.peek(7));
                if ((e & 15u) == 0u) return kBadCodeLengths;
                br.drop(e & 15u);
                const uint32_t s = e >> 4;
                if (s < 16u) {
                    m.set_len(i++, s);
                    prev = s;
                } else {
                    uint32_t rep, v = 0;
                    if (s == 16u) {
                        if (i == 0) return kBadCodeLengths;
                        v = prev;
                        rep = 3u + br.get(2);
                    } else if (s == 17u) {
                        rep = 3u + br.get(3);
                    } else {
                        rep = 11u + br.get(7);
                    }
                    if (i + rep > want) return kBadCodeLengths;
                    for (uint32_t k = 0; k < rep; ++k) m.set_len(i + k, v);
                    i += rep;
                    prev = v;
                }
            }
            if (br.overrun()) return kInputOverrun;
            if (m.len(256) == 0u) return kBadCodeLengths;  // no end-of-block code
        }
        LongCodes ll, ld;
        if (!build_tables(m, [&](uint32_t i) { return m.len(i); }, hlit, (uint32_t)kLitRoot, 0u, 0u, false, &ll)) return kBadCode;
        if (!build_tables(m, [&](uint32_t i) { return m.len(hlit + i); }, hdist, (uint32_t)kDistRoot, kLitEntries, 288u, false, &ld))
            return kBadCode;
        // ---- the block's symbols
        for (;;) {
            br.refill();
            uint32_t e = m.tab(br.peek(kLitRoot));
            uint32_t s;
            if (e & 15u) {
                br.drop(e & 15u);
                s = e >> 4;
            } else {
                s = decode_long(m, br, (uint32_t)kLitRoot, 0u, ll);
                if (s == 0xFFFFu) return kBadSymbol;
            }
            if (s < 256u) {
                if (op >= out_len) return kOutputOverrun;
                out[op++] = (uint8_t)s;
                continue;
            }
            if (s == 256u) break;
            s -= 257u;
            if (s >= 29u) return kBadSymbol;
            // length: symbols 0..7 -> 3..10, then groups of four per extra bit, 28 -> 258
            const uint32_t lx = s < 8u ? 0u : (s == 28u ? 0u : (s - 4u) >> 2);
            uint32_t length = s < 8u ? 3u + s : (s == 28u ? 258u : 3u + ((4u + (s & 3u)) << lx));
            br.refill();
            length += br.get(lx);
            br.refill();
            uint32_t d;
            e = m.tab(kLitEntries + br.peek(kDistRoot));
            if (e & 15u) {
                br.drop(e & 15u);
                d = e >> 4;
            } else {
                d = decode_long(m, br, (uint32_t)kDistRoot, 288u, ld);
                if (d == 0xFFFFu) return kBadDistance;
            }
            if (d >= 30u) return kBadDistance;
            const uint32_t dx = d < 4u ? 0u : (d - 2u) >> 1;
            uint32_t dist = d < 4u ? 1u + d : 1u + ((2u + (d & 1u)) << dx);
            br.refill();
            dist += br.get(dx);
            if (dist > op) return kBadDistance;
            if (out_len - op < length) return kOutputOverrun;
            if (br.ip > br.in_end + 8) return kInputOverrun;
            uint8_t* q = out + op;
            const uint8_t* p = q - dist;
            op += length;
            if (dist == 1u) {  // a run of one byte (quality strings)
                const uint8_t v = *p;
                for (uint32_t k = 0; k < length; ++k) q[k] = v;
            } else if (dist >= 4u) {  // four source bytes are never bytes this match writes within the same group
                uint32_t k = 0;
                for (; k + 4u <= length; k += 4u) {
                    const uint8_t a = p[k], b = p[k + 1], c = p[k + 2], e4 = p[k + 3];
                    q[k] = a;
                    q[k + 1] = b;
                    q[k + 2] = c;
                    q[k + 3] = e4;
                }
                for (; k < length; ++k) q[k] = p[k];
            } else {
                for (uint32_t k = 0; k < length; ++k) q[k] = p[k];
            }
        }
        if (br.ip > br.in_end + 8) return kInputOverrun;
    }
    if (br.overrun()) return kInputOverrun;
    return op == out_len ? kOk : kShortOutput;
}

}  // namespace tec_ginf
