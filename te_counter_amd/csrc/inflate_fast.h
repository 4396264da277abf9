// A raw-DEFLATE (RFC 1951) decoder for BGZF blocks: whole input and whole output in memory, output
// size known in advance (the block's ISIZE).  Written for throughput on the host side of the BAM
// decoder (bamio.hip): 64-bit bit buffer refilled with one unaligned load, 11-bit literal/length and
// 8-bit distance lookup tables with second-level tables for longer codes, word-wise match copies.
// zlib stays the referee: a block this decoder rejects is handed to zlib's inflate, so a valid
// stream is never refused and a corrupt one gets zlib's verdict (bamio.hip).
//
// The tables follow the canonical-Huffman construction of RFC 1951 section 3.2.2; nothing here is
// taken from the reference (which relies on pysam / htslib for this step, te_count.py:65,190,351).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace tec_inflate {

constexpr int kLitBits = 11;   // first-level bits of the literal/length table
constexpr int kDistBits = 8;   // first-level bits of the distance table
constexpr int kMaxCodeLen = 15;
constexpr int kLitTabSize = (1 << kLitBits) + 1024;   // + room for all second-level tables (<= 2 * 288)
constexpr int kDistTabSize = (1 << kDistBits) + 512;

// Table entry (32 bits):
//   bits  0..3   code bits to consume (for a second-level entry: the bits beyond the first level)
//   bits  4..7   kind: 0 invalid, 1 literal, 2 length/distance base, 3 end of block, 4 link to a second-level table
//   bits  8..12  number of extra bits (kind 2) / index bits of the second-level table (kind 4)
//   bits 16..31  literal byte | base value | offset of the second-level table
constexpr uint32_t kKindLiteral = 1u << 4, kKindBase = 2u << 4, kKindEnd = 3u << 4, kKindLink = 4u << 4;

inline uint32_t entry(uint32_t kind, uint32_t nbits, uint32_t extra, uint32_t value) {
    return nbits | kind | (extra << 8) | (value << 16);
}

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) {
        r = (r << 1) | (code & 1u);
        code >>= 1;
    }
    return r;
}

// Builds the lookup table of one Huffman code.  lens[0..n): code lengths (0 = unused symbol).
// is_dist: symbols are distance codes, else literal/length codes.  Returns false for an
// over-subscribed code, an incomplete one that zlib would refuse, or one whose second-level tables
// do not fit.
inline bool build_table(const uint8_t* lens, int n, bool is_dist, int first_bits, uint32_t* tab, int tab_size,
                        bool is_code_lengths = false) {
    int count[kMaxCodeLen + 1] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    uint32_t next[kMaxCodeLen + 2];
    uint32_t code = 0;
    int64_t left = 1;
    for (int l = 1; l <= kMaxCodeLen; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;  // over-subscribed
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    // zlib's rule (inftrees.c): an incomplete code is an error, except no code at all or -- for the
    // literal/length and distance codes -- a single code of one bit
    if (left > 0) {
        int max_len = 0;
        for (int l = kMaxCodeLen; l >= 1 && !max_len; --l)
            if (count[l]) max_len = l;
        if (max_len > 0 && (is_code_lengths || max_len != 1)) return false;
    }
    const int first_size = 1 << first_bits;
    memset(tab, 0, sizeof(uint32_t) * (size_t)tab_size);
    // second-level tables: one per first-level prefix that has longer codes, sized by its longest code
    uint8_t sub_bits[1 << kLitBits];
    memset(sub_bits, 0, (size_t)first_size);
    {
        uint32_t nx[kMaxCodeLen + 2];
        memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (l == 0) continue;
            const uint32_t c = nx[l]++;
            if (l > first_bits) {
                const uint32_t rev = reverse_bits(c, l);
                const uint32_t prefix = rev & (uint32_t)(first_size - 1);
                if (l - first_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - first_bits);
            }
        }
    }
    int used = first_size;
    for (int p = 0; p < first_size; ++p) {
        if (!sub_bits[p]) continue;
        const int sz = 1 << sub_bits[p];
        if (used + sz > tab_size) return false;
        tab[p] = entry(kKindLink, (uint32_t)first_bits, sub_bits[p], (uint32_t)used);
        used += sz;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l == 0) continue;
        const uint32_t c = next[l]++;
        const uint32_t rev = reverse_bits(c, l);
        uint32_t kind, extra = 0, value;
        if (is_dist) {
            if (s >= 30) continue;  // codes 30, 31 never occur in valid data: leave them invalid
            kind = kKindBase;
            extra = kDistExtra[s];
            value = kDistBase[s];
        } else if (s < 256) {
            kind = kKindLiteral;
            value = (uint32_t)s;
        } else if (s == 256) {
            kind = kKindEnd;
            value = 0;
        } else {
            if (s - 257 >= 29) continue;  // 286, 287: invalid
            kind = kKindBase;
            extra = kLenExtra[s - 257];
            value = kLenBase[s - 257];
        }
        if (l <= first_bits) {
            const uint32_t e = entry(kind, (uint32_t)l, extra, value);
            for (uint32_t i = rev; i < (uint32_t)first_size; i += 1u << l) tab[i] = e;
        } else {
            const uint32_t prefix = rev & (uint32_t)(first_size - 1);
            const uint32_t link = tab[prefix];
            const uint32_t base = link >> 16, bits = (link >> 8) & 31u;
            const uint32_t e = entry(kind, (uint32_t)(l - first_bits), extra, value);
            for (uint32_t i = rev >> first_bits; i < (1u << bits); i += 1u << (l - first_bits)) tab[base + i] = e;
        }
    }
    return true;
}

struct Tables {
    uint32_t lit[kLitTabSize];
    uint32_t dist[kDistTabSize];
};

inline uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // little-endian host (x86-64)
}

// Inflates exactly out_len bytes from in[0..in_len).  Returns true iff the stream is well formed,
// ends with a final block and produces exactly out_len bytes.  Never reads or writes outside the
// two buffers.
inline bool inflate_block(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, Tables* T) {
    const uint8_t* ip = in;
    const uint8_t* const in_end = in + in_len;
    uint8_t* op = out;
    uint8_t* const out_end = out + out_len;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    // careful refill: byte by byte, zeros past the end (reading past the end is detected by `over`)
    size_t over = 0;
    auto refill = [&]() {
        while (bitcnt <= 56) {
            if (ip < in_end)
                bitbuf |= (uint64_t)*ip++ << bitcnt;
            else
                ++over;
            bitcnt += 8;
        }
    };
    auto getbits = [&](int nb) -> uint32_t {
        const uint32_t v = (uint32_t)(bitbuf & ((1ull << nb) - 1ull));
        bitbuf >>= nb;
        bitcnt -= nb;
        return v;
    };
    bool last = false;
    while (!last) {
        refill();
        last = getbits(1) != 0;
        const uint32_t type = getbits(2);
        if (type == 0) {  // stored
            const int drop = bitcnt & 7;
            getbits(drop);
            // whole bytes still in the buffer go back: first the virtual ones past the end, then real ones
            while (bitcnt >= 8 && over) {
                --over;
                bitcnt -= 8;
            }
            if (over) return false;  // the header itself ran past the end
            while (bitcnt >= 8 && ip > in) {
                --ip;
                bitcnt -= 8;
            }
            bitbuf = 0;
            bitcnt = 0;
            if (in_end - ip < 4) return false;
            const uint32_t len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
            ip += 4;
            if ((len ^ nlen) != 0xFFFFu) return false;
            if ((size_t)(in_end - ip) < len || (size_t)(out_end - op) < len) return false;
            memcpy(op, ip, len);
            ip += len;
            op += len;
            continue;
        }
        if (type == 3) return false;
        if (type == 1) {  // fixed codes
            uint8_t lens[288 + 32];
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
            if (!build_table(lens, 288, false, kLitBits, T->lit, kLitTabSize)) return false;
            if (!build_table(lens + 288, 32, true, kDistBits, T->dist, kDistTabSize)) return false;
        } else {  // dynamic codes
            refill();
            const int hlit = (int)getbits(5) + 257, hdist = (int)getbits(5) + 1, hclen = (int)getbits(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                refill();
                cl[order[i]] = (uint8_t)getbits(3);
            }
            uint32_t cltab[1 << 7];
            if (!build_table(cl, 19, false, 7, cltab, 1 << 7, true)) return false;  // lengths <= 7: no second level
            uint8_t lens[286 + 30 + 138];
            int i = 0;
            while (i < hlit + hdist) {
                refill();
                const uint32_t e = cltab[bitbuf & 127u];
                if ((e & 0xF0u) != kKindLiteral) return false;  // symbols 0..18 are built as "literals"
                getbits((int)(e & 15u));
                const uint32_t sym = e >> 16;
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                } else {
                    int rep;
                    uint8_t v = 0;
                    if (sym == 16) {
                        if (i == 0) return false;
                        v = lens[i - 1];
                        rep = 3 + (int)getbits(2);
                    } else if (sym == 17) {
                        rep = 3 + (int)getbits(3);
                    } else {
                        rep = 11 + (int)getbits(7);
                    }
                    if (i + rep > hlit + hdist) return false;
                    while (rep--) lens[i++] = v;
                }
            }
            if ((int64_t)over * 8 > (int64_t)bitcnt) return false;  // the code lengths ran past the end
            if (lens[256] == 0) return false;  // no end-of-block code
            if (!build_table(lens, hlit, false, kLitBits, T->lit, kLitTabSize)) return false;
            if (!build_table(lens + hlit, hdist, true, kDistBits, T->dist, kDistTabSize)) return false;
        }
        // ---- the block's symbols
        const uint32_t* const lit = T->lit;
        const uint32_t* const dst = T->dist;
        for (;;) {
            // fast path: far from both buffer ends, so no per-symbol bounds checks are needed
            // (a symbol consumes at most 15 + 5 + 15 + 13 = 48 bits and writes at most 258 + 7 bytes)
            while (in_end - ip >= 16 && out_end - op >= 272) {
                bitbuf |= load64(ip) << bitcnt;
                ip += (63 - bitcnt) >> 3;
                bitcnt |= 56;
                uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1u)];
                // up to three first-level literals from one refill (<= 33 of the >= 56 bits)
                if ((e & 0xF0u) == kKindLiteral) {
                    bitbuf >>= (e & 15u);
                    bitcnt -= (int)(e & 15u);
                    *op++ = (uint8_t)(e >> 16);
                    e = lit[bitbuf & ((1u << kLitBits) - 1u)];
                    if ((e & 0xF0u) == kKindLiteral) {
                        bitbuf >>= (e & 15u);
                        bitcnt -= (int)(e & 15u);
                        *op++ = (uint8_t)(e >> 16);
                        e = lit[bitbuf & ((1u << kLitBits) - 1u)];
                        if ((e & 0xF0u) == kKindLiteral) {
                            bitbuf >>= (e & 15u);
                            bitcnt -= (int)(e & 15u);
                            *op++ = (uint8_t)(e >> 16);
                            continue;
                        }
                    }
                    // e is the next symbol's first-level entry, not consumed yet: top the buffer up
                    // (a length/distance pair needs up to 48 bits)
                    bitbuf |= load64(ip) << bitcnt;
                    ip += (63 - bitcnt) >> 3;
                    bitcnt |= 56;
                }
                if ((e & 0xF0u) == kKindLink) {
                    e = lit[(e >> 16) + ((bitbuf >> kLitBits) & ((1u << ((e >> 8) & 31u)) - 1u))];
                    bitbuf >>= kLitBits;
                    bitcnt -= kLitBits;
                }
                bitbuf >>= (e & 15u);
                bitcnt -= (int)(e & 15u);
                if ((e & 0xF0u) == kKindLiteral) {  // a long-code literal
                    *op++ = (uint8_t)(e >> 16);
                    continue;
                }
                if ((e & 0xF0u) != kKindBase) {
                    if ((e & 0xF0u) == kKindEnd) goto block_done;
                    return false;
                }
                {
                    const uint32_t xb = (e >> 8) & 31u;
                    const uint32_t length = (e >> 16) + (uint32_t)(bitbuf & ((1ull << xb) - 1ull));
                    bitbuf >>= xb;
                    bitcnt -= (int)xb;
                    uint32_t d = dst[bitbuf & ((1u << kDistBits) - 1u)];
                    if ((d & 0xF0u) == kKindLink) {
                        d = dst[(d >> 16) + ((bitbuf >> kDistBits) & ((1u << ((d >> 8) & 31u)) - 1u))];
                        bitbuf >>= kDistBits;
                        bitcnt -= kDistBits;
                    }
                    if ((d & 0xF0u) != kKindBase) return false;
                    bitbuf >>= (d & 15u);
                    bitcnt -= (int)(d & 15u);
                    const uint32_t db = (d >> 8) & 31u;
                    const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf & ((1ull << db) - 1ull));
                    bitbuf >>= db;
                    bitcnt -= (int)db;
                    if (dist > (size_t)(op - out)) return false;
                    const uint8_t* src = op - dist;
                    uint8_t* const end = op + length;
                    if (dist >= 8) {
                        do {
                            memcpy(op, src, 8);
                            op += 8;
                            src += 8;
                        } while (op < end);
                    } else if (dist == 1) {
                        memset(op, *src, length);
                    } else {
                        do {
                            *op++ = *src++;
                        } while (op < end);
                    }
                    op = end;
                }
            }
            // careful path: one symbol with every check
            {
                refill();
                uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1u)];
                if ((e & 0xF0u) == kKindLink) {
                    e = lit[(e >> 16) + ((bitbuf >> kLitBits) & ((1u << ((e >> 8) & 31u)) - 1u))];
                    getbits(kLitBits);
                }
                getbits((int)(e & 15u));
                if ((e & 0xF0u) == kKindLiteral) {
                    if (op >= out_end) return false;
                    *op++ = (uint8_t)(e >> 16);
                } else if ((e & 0xF0u) == kKindEnd) {
                    goto block_done;
                } else if ((e & 0xF0u) == kKindBase) {
                    const uint32_t length = (e >> 16) + getbits((int)((e >> 8) & 31u));
                    refill();
                    uint32_t d = dst[bitbuf & ((1u << kDistBits) - 1u)];
                    if ((d & 0xF0u) == kKindLink) {
                        d = dst[(d >> 16) + ((bitbuf >> kDistBits) & ((1u << ((d >> 8) & 31u)) - 1u))];
                        getbits(kDistBits);
                    }
                    if ((d & 0xF0u) != kKindBase) return false;
                    getbits((int)(d & 15u));
                    const uint32_t dist = (d >> 16) + getbits((int)((d >> 8) & 31u));
                    if (dist > (size_t)(op - out) || length > (size_t)(out_end - op)) return false;
                    const uint8_t* src = op - dist;
                    for (uint32_t k = 0; k < length; ++k) op[k] = src[k];
                    op += length;
                } else {
                    return false;
                }
                if (over > 8) return false;  // ran off the end of the input
            }
        }
    block_done:;
    }
    // bits consumed beyond the input's end mean a truncated stream
    if ((int64_t)over * 8 > (int64_t)bitcnt) return false;
    return op == out_end;
}

}  // namespace tec_inflate
