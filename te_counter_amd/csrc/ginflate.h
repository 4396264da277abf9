// Raw DEFLATE (RFC 1951) for one lane of a GPU wavefront: the decoder behind the on-GPU BGZF front end
// (gbam.hip), written so that the very same source also compiles for the host, where the CPU test
// suite runs it against zlib on every stream shape zlib can produce.
//
// Inflation is split in two (gbam.hip has the kernels):
//   phase 1  (this file, inflate_stream)  one LANE per BGZF block walks the bit stream -- the part of DEFLATE that
//            is serial by nature.  Literal bytes (and the bytes of stored blocks) go straight to their final place in
//            the inflated stream; every MATCH becomes a 32-bit token {literals in front of it, length, distance}.
//            The lane never reads the inflated stream back: its loads are the 8-byte input refills only.  (When the
//            lane also performed the copies, every match was a load that had to wait for the lane's own earlier
//            stores -- gfx9 retires loads and stores in order through one counter -- and a wave spent ten times its
//            instruction time per symbol.)
//   phase 2  (resolve_tokens below for the host; gresolve_kernel on the device)  one WAVE per block replays the
//            matches: 64 tokens at a time, output positions by a wave prefix sum, copies in rounds -- a match goes
//            as soon as its source bytes lie below the first match still pending.
//
// Why a different decoder from inflate_fast.h: what limits the number of lanes per CU is table memory, hence
//   * tables are small and 16-bit, in LDS: literal/length 7-bit first level (128 entries), distance
//     5-bit (32) = 320 bytes per lane -> 512 lanes (eight waves, two per SIMD) per CU in exactly the
//     160 KiB of LDS; the second wave of a SIMD issues while the first waits for memory;
//   * the code lengths of a dynamic block go to the lane's global scratch as they are decoded (sixteen
//     nibbles per 64-bit store) and come back in symbol order for the two passes of the table builder; the
//     distance lengths stay in two registers;
//   * codes longer than the first level (rare in BAM payloads) are decoded canonically, bit by bit,
//     from per-length counts packed in registers; only their symbols live in global scratch;
//   * base values and extra-bit counts of length / distance symbols are computed arithmetically;
//   * input arrives as 8-byte loads issued one step ahead of their use.
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

#ifndef TEC_GINF_LITROOT
#define TEC_GINF_LITROOT 7
#endif
constexpr int kLitRoot = TEC_GINF_LITROOT;
#ifndef TEC_GINF_DISTROOT
#define TEC_GINF_DISTROOT 5
#endif
constexpr int kDistRoot = TEC_GINF_DISTROOT;
constexpr uint32_t kLitEntries = 1u << kLitRoot;
constexpr uint32_t kDistEntries = 1u << kDistRoot;
constexpr uint32_t kEntriesPerLane = kLitEntries + kDistEntries;  // u16 each: 640 bytes
constexpr uint32_t kLenWords = 20;                                 // code lengths of a dynamic block: 320 nibbles in 64-bit words (global)
constexpr uint32_t kLongSymsPerLane = 320;                            // symbols of codes longer than the first level (u16, global)
constexpr uint32_t kInPad = 16;                                       // readable bytes the caller guarantees behind the input

// Reasons a block is handed back to the host's referee decoder (zlib / inflate_fast.h).
enum : int { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadCodeLengths = 3, kBadCode = 4, kBadSymbol = 5, kBadDistance = 6,
             kOutputOverrun = 7, kInputOverrun = 8, kShortOutput = 9, kTokenOverrun = 10 };

// ---- tokens: what phase 1 hands to phase 2, one 32-bit word per match
//   bits 31:30  kind   0 nothing (padding), 2 match, 3 skip
//   match       bits 29:23 bytes already in place in front of the match (literals, stored bytes: 0..127),
//               bits 22:15 length - 3 (0..255), bits 14:0 distance - 1
//   skip        bits 19:0 bytes already in place (a run longer than 127, or the tail behind the last match)
// A block of n bytes has at most n / 3 + n / 128 + 2 tokens.
constexpr uint32_t kTokMatch = 2u << 30, kTokSkip = 3u << 30;
constexpr uint32_t kTokMaxBefore = 127u, kTokMaxSkip = 0xFFFFFu;
TEC_HD uint32_t tok_before(uint32_t t) { return (t >> 30) == 2u ? (t >> 23) & 127u : ((t >> 30) == 3u ? t & kTokMaxSkip : 0u); }
TEC_HD uint32_t tok_match_len(uint32_t t) { return (t >> 30) == 2u ? ((t >> 15) & 0xFFu) + 3u : 0u; }

// Host sink: a growing array.  (The device sink, gbam.hip, packs four tokens per 16-byte store into pages
// drawn from a pool.)  put() returns false when the sink is full: the block is then handed to the referee.
struct VectorSink {
    uint32_t* tok;
    uint32_t cap, n = 0;
    bool put(uint32_t t) {
        if (n >= cap) return false;
        tok[n++] = t;
        return true;
    }
};

// Phase 2 on the host: replays the matches into out[0, out_len), whose literal bytes are in place.  Returns false if
// the tokens do not describe exactly out_len bytes (phase 1 has checked distances and lengths; this is a second line
// of defence).
inline bool resolve_tokens(const uint32_t* tok, uint32_t n, uint8_t* out, uint32_t out_len) {
    uint32_t op = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t t = tok[i], before = tok_before(t), len = tok_match_len(t);
        if (out_len - op < before || out_len - op - before < len) return false;
        op += before;
        if (len) {
            const uint32_t dist = (t & 0x7FFFu) + 1u;
            if (dist > op) return false;
            for (uint32_t j = 0; j < len; ++j) out[op + j] = out[op + j - dist];
            op += len;
        }
    }
    return op == out_len;
}

// Per-lane memory.  The device instantiation keeps `tab` in LDS and the long-code symbols in global
// memory; the host instantiation uses plain arrays.
//   uint32_t tab(uint32_t i) / void set_tab(uint32_t i, uint32_t v)     i < kEntriesPerLane
//   uint32_t sym(uint32_t i) / void set_sym(uint32_t i, uint32_t v)     i < kLongSymsPerLane
//   uint64_t lenw(uint32_t k) / void set_lenw(uint32_t k, uint64_t v)   k < kLenWords
struct HostMem {
    uint16_t t[kEntriesPerLane];
    uint16_t s[kLongSymsPerLane];
    uint64_t lw[kLenWords];
    uint64_t lenw(uint32_t k) const { return lw[k]; }
    void set_lenw(uint32_t k, uint64_t v) { lw[k] = v; }
    uint32_t tab(uint32_t i) const { return t[i]; }
    void set_tab(uint32_t i, uint32_t v) { t[i] = (uint16_t)v; }
    uint32_t sym(uint32_t i) const { return s[i]; }
    void set_sym(uint32_t i, uint32_t v) { s[i] = (uint16_t)v; }
};

// The code lengths of a dynamic block, four bits each, are written once in symbol order (sixteen per
// 64-bit word, to the lane's global scratch: stores do not stall the lane) and read back in symbol
// order by the two passes of build_tables (twenty dependent loads per pass).
template <class Mem>
struct LenWriter {
    Mem& m;
    uint64_t acc = 0;
    uint32_t i = 0;
    TEC_HD void put(uint32_t v) {
        acc |= (uint64_t)v << (4u * (i & 15u));
        if ((i & 15u) == 15u) {
            m.set_lenw(i >> 4, acc);
            acc = 0;
        }
        ++i;
    }
    TEC_HD void finish() {
        if (i & 15u) m.set_lenw(i >> 4, acc);
    }
};

// Counts of codes per length, 12 bits each, five lengths per 64-bit word (lengths 1..15).
struct Counts {
    uint64_t w[3];
    TEC_HD void clear() { w[0] = w[1] = w[2] = 0; }
    TEC_HD void add_n(uint32_t len, uint32_t n) {  // len 1..15
        const uint32_t k = len - 1u;
        const uint64_t v = (uint64_t)n << (12u * (k % 5u));
        const uint32_t q = k / 5u;
        w[0] += q == 0 ? v : 0ull;
        w[1] += q == 1 ? v : 0ull;
        w[2] += q == 2 ? v : 0ull;
    }
    TEC_HD void add(uint32_t len) { add_n(len, 1u); }
    TEC_HD uint32_t get(uint32_t len) const {
        const uint32_t k = len - 1u;
        const uint32_t q = k / 5u;
        const uint64_t v = q == 0 ? w[0] : (q == 1 ? w[1] : w[2]);
        return (uint32_t)(v >> (12u * (k % 5u))) & 0xFFFu;
    }
};

// Next canonical code per length, 16 bits each, four lengths per 64-bit word.
struct NextCodes {
    uint64_t w[4];
    TEC_HD void clear() { w[0] = w[1] = w[2] = w[3] = 0; }
    TEC_HD void set(uint32_t len, uint32_t v) {
        const uint32_t q = len >> 2, sh = 16u * (len & 3u);
        const uint64_t mask = ~(0xFFFFull << sh), val = (uint64_t)(v & 0xFFFFu) << sh;
        w[0] = q == 0 ? (w[0] & mask) | val : w[0];
        w[1] = q == 1 ? (w[1] & mask) | val : w[1];
        w[2] = q == 2 ? (w[2] & mask) | val : w[2];
        w[3] = q == 3 ? (w[3] & mask) | val : w[3];
    }
    TEC_HD uint32_t take(uint32_t len) {  // returns the code and advances it
        const uint32_t q = len >> 2, sh = 16u * (len & 3u);
        const uint64_t v = q == 0 ? w[0] : (q == 1 ? w[1] : (q == 2 ? w[2] : w[3]));
        const uint64_t one = 1ull << sh;
        w[0] += q == 0 ? one : 0ull;
        w[1] += q == 1 ? one : 0ull;
        w[2] += q == 2 ? one : 0ull;
        w[3] += q == 3 ? one : 0ull;
        return (uint32_t)(v >> sh) & 0xFFFFu;
    }
};

// What the canonical (bit by bit) decoder of codes longer than the first level needs.
struct LongCodes {
    Counts counts;
    uint32_t first;  // first code of length root+1, i.e. puff's `first` after the iteration len = root
};

// Input: 64-bit bit buffer fed 32 bits at a time from a two-deep queue of 8-byte loads (the load of
// the word after next is in flight while the current one is consumed).
struct BitReader {
    const uint8_t* in;
    const uint8_t* in_end;
    const uint8_t* lp;  // address of the next 8-byte load
    uint64_t buf, cur, nxt;
    uint32_t cnt;       // valid bits in buf
    uint32_t cur_left;  // bytes of `cur` not yet moved into buf (8, 4 or 0)
    uint32_t taken;     // bytes of the stream moved into buf so far
    TEC_HD uint64_t load8() {
        uint64_t v;
        memcpy(&v, lp < in_end ? lp : in_end, 8);  // never beyond in_end + 8
        lp += 8;
        return v;
    }
    TEC_HD void start(const uint8_t* from) {
        lp = from;
        buf = 0;
        cnt = 0;
        cur = load8();
        nxt = load8();
        cur_left = 8;
        taken = (uint32_t)(from - in);
    }
    TEC_HD void refill() {  // at least 32 valid bits afterwards
        if (cnt <= 32u) {
            if (cur_left == 0u) {
                cur = nxt;
                nxt = load8();
                cur_left = 8;
            }
            buf |= (cur & 0xFFFFFFFFull) << cnt;
            cur >>= 32;
            cur_left -= 4u;
            cnt += 32u;
            taken += 4u;
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
    // stream position (bytes) of the first bit that has not been consumed, rounded down
    TEC_HD uint32_t byte_pos() const { return taken - (cnt >> 3); }
    TEC_HD bool overrun() const { return (uint64_t)taken * 8u - cnt > (uint64_t)(in_end - in) * 8u; }
    TEC_HD bool far_overrun() const { return taken > (uint32_t)(in_end - in) + 16u; }
};

TEC_HD uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32u - len); }

// Canonical Huffman tables of one code (RFC 1951 3.2.2).  lens(i), i < n: code lengths.  Entries
// [tab_base, tab_base + 2^root): (symbol << 4) | length for codes of at most `root` bits, 0 for
// prefixes of longer codes and for unused patterns.  Symbols of longer codes go, sorted by (length,
// value), to the lane's global list from sym_base on.  Returns false for an over-subscribed code or
// an incomplete one that zlib would refuse (inftrees.c: incomplete is allowed only for no code at
// all, or for a literal/length or distance code that consists of a single one-bit code).
template <class Mem, class LenFn>
TEC_HD bool build_tables(Mem& m, LenFn lens, uint32_t n, uint32_t root, uint32_t tab_base, uint32_t sym_base, bool is_cl,
                         LongCodes* lc) {
    Counts cnt;
    cnt.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t l = lens(i);
        if (l) cnt.add(l);
    }
    int32_t left = 1;
    uint32_t max_len = 0, code = 0, n_long = 0;
    NextCodes next;
    next.clear();
    Counts long_offs;  // where the symbols of each length > root start in the lane's long-symbol list
    long_offs.clear();
    lc->counts = cnt;
    lc->first = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = cnt.get(l);
        left = (left << 1) - (int32_t)c;
        if (left < 0) return false;
        if (c) max_len = l;
        next.set(l, code);
        code = (code + c) << 1;
        if (l == root) lc->first = code;  // first code of length root + 1
        if (l > root) {
            long_offs.add_n(l, n_long);
            n_long += c;
        }
    }
    if (left > 0 && max_len > 0 && (is_cl || max_len != 1u)) return false;
    for (uint32_t i = 0; i < (1u << root); ++i) m.set_tab(tab_base + i, 0u);
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens(s);
        if (!l) continue;
        if (l <= root) {
            const uint32_t e = (s << 4) | l;
            for (uint32_t x = bitrev(next.take(l), l); x < (1u << root); x += 1u << l) m.set_tab(tab_base + x, e);
        } else {  // first-level entries of its prefix stay 0: the long-code path
            m.set_sym(sym_base + long_offs.get(l), s);
            long_offs.add(l);
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
    uint32_t first = lc.first, index = 0;
#pragma unroll 1
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

// Phase 1: walks the raw DEFLATE stream in[0, in_len), which inflates to exactly out_len bytes: literal and stored
// bytes are written to `out` at their final place, matches become tokens (sink.put).  kOk, or the reason the block
// must be redone by the referee.  `in` must be readable up to in + in_len + kInPad.
template <class Mem, class Sink>
TEC_HD int inflate_stream(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, Mem& m, Sink& sink) {
    BitReader br;
    br.in = in;
    br.in_end = in + in_len;
    br.start(in);
    uint32_t op = 0;
    uint32_t run = 0;  // bytes in place since the last token
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
            const uint32_t pos = br.byte_pos();  // whole bytes still in the bit buffer belong to the payload
            if (pos > in_len || in_len - pos < len) return kInputOverrun;
            if (out_len - op < len) return kOutputOverrun;
            const uint8_t* src = in + pos;
            uint8_t* q = out + op;
            uint32_t k = 0;
            for (; k + 8u <= len; k += 8u) {
                uint64_t v;
                memcpy(&v, src + k, 8);
                memcpy(q + k, &v, 8);
            }
            for (; k < len; ++k) q[k] = src[k];
            op += len;
            run += len;
            br.start(src + len);
            continue;
        }
        uint32_t hlit = 288, hdist = 32;
        bool fixed = false;
        uint64_t dl_lo = 0x5555555555555555ull, dl_hi = 0x5555555555555555ull;  // distance code lengths, 4 bits each (fixed: all 5)
        if (type == 1u) {  // fixed codes (RFC 1951 3.2.6)
            fixed = true;
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
            // its 7-bit table sits in the (not yet built) literal/length area; no code is longer than 7 bits
            if (!build_tables(m, [&](uint32_t i) { return (uint32_t)(cl >> (3u * i)) & 7u; }, 19u, 7u, 0u, 0u, true, &unused))
                return kBadCodeLengths;
            dl_lo = dl_hi = 0;
            LenWriter<Mem> lw{m};
            uint32_t prev = 0;
            bool has_eob = false;
            const uint32_t want = hlit + hdist;
            auto put = [&](uint32_t v) {
                if (lw.i >= hlit) {  // the distance code lengths stay in registers (<= 32 x 4 bits)
                    const uint32_t j = lw.i - hlit;
                    const uint64_t x = (uint64_t)v << (4u * (j & 15u));
                    dl_lo |= j < 16u ? x : 0ull;
                    dl_hi |= j < 16u ? 0ull : x;
                } else if (lw.i == 256u) {
                    has_eob = v != 0u;
                }
                lw.put(v);
            };
            while (lw.i < want) {
                br.refill();
                const uint32_t e = m.tab(br.peek(7));
                if ((e & 15u) == 0u) return kBadCodeLengths;
                br.drop(e & 15u);
                const uint32_t s = e >> 4;
                if (s < 16u) {
                    put(s);
                    prev = s;
                } else {
                    uint32_t rep, v = 0;
                    if (s == 16u) {
                        if (lw.i == 0) return kBadCodeLengths;
                        v = prev;
                        rep = 3u + br.get(2);
                    } else if (s == 17u) {
                        rep = 3u + br.get(3);
                    } else {
                        rep = 11u + br.get(7);
                    }
                    if (lw.i + rep > want) return kBadCodeLengths;
                    for (uint32_t k = 0; k < rep; ++k) put(v);
                    prev = v;
                }
            }
            lw.finish();
            if (br.overrun()) return kInputOverrun;
            if (!has_eob) return kBadCodeLengths;  // no end-of-block code
        }
        LongCodes ll, ld;
        uint64_t lcur = 0;  // both passes of build_tables ask for the lengths in symbol order
        auto lit_len = [&](uint32_t i) -> uint32_t {
            if (fixed) return i < 144u ? 8u : (i < 256u ? 9u : (i < 280u ? 7u : 8u));
            if ((i & 15u) == 0u) lcur = m.lenw(i >> 4);
            return (uint32_t)(lcur >> (4u * (i & 15u))) & 15u;
        };
        if (!build_tables(m, lit_len, hlit, (uint32_t)kLitRoot, 0u, 0u, false, &ll)) return kBadCode;
        if (!build_tables(m, [&](uint32_t i) { return (uint32_t)((i < 16u ? dl_lo : dl_hi) >> (4u * (i & 15u))) & 15u; }, hdist,
                          (uint32_t)kDistRoot, kLitEntries, 288u, false, &ld))
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
                ++run;
                // up to two more literals from the bits already in the buffer (>= 18 after any first code, and a
                // first-level code has at most kLitRoot bits): lanes in a literal run advance three symbols in the
                // time a neighbour lane spends on one match
#pragma unroll
                for (int extra = 0; extra < 2; ++extra) {
                    const uint32_t e2 = m.tab(br.peek(kLitRoot));
                    if ((e2 & 15u) == 0u || (e2 >> 4) >= 256u || op >= out_len) break;
                    br.drop(e2 & 15u);
                    out[op++] = (uint8_t)(e2 >> 4);
                    ++run;
                }
                continue;
            }
            if (s == 256u) break;
            s -= 257u;
            if (s >= 29u) return kBadSymbol;
            // length: symbols 0..7 -> 3..10, then groups of four per extra bit, 28 -> 258
            const uint32_t lx = s < 8u ? 0u : (s == 28u ? 0u : (s - 4u) >> 2);
            uint32_t length = s < 8u ? 3u + s : (s == 28u ? 258u : 3u + ((4u + (s & 3u)) << lx));
            length += br.get(lx);  // (the refill at the top left >= 33 bits: a 15-bit code and 5 extra bits fit)
            br.refill();           // >= 33 bits again: a 15-bit distance code and 13 extra bits fit
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
            dist += br.get(dx);
            if (dist > op) return kBadDistance;
            if (out_len - op < length) return kOutputOverrun;
            if (br.far_overrun()) return kInputOverrun;
            op += length;
            if (run > kTokMaxBefore) {  // a long run of bytes in place gets a token of its own
                if (!sink.put(kTokSkip | run)) return kTokenOverrun;
                run = 0;
            }
            if (!sink.put(kTokMatch | (run << 23) | ((length - 3u) << 15) | (dist - 1u))) return kTokenOverrun;
            run = 0;
        }
        if (br.far_overrun()) return kInputOverrun;
    }
    if (br.overrun()) return kInputOverrun;
    if (run && !sink.put(kTokSkip | run)) return kTokenOverrun;  // (run <= out_len <= 65 536)
    return op == out_len ? kOk : kShortOutput;
}

}  // namespace tec_ginf
