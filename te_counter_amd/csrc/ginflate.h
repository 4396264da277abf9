// Raw DEFLATE (RFC 1951) for one lane of a GPU wavefront: the decoder behind the on-GPU BGZF front end
// (gbam.hip), written so that the very same source also compiles for the host, where the CPU test
// suite runs it against zlib on every stream shape zlib can produce.
//
// Inflation is split in two (gbam.hip has the kernels):
//   phase 1  (this file, inflate_stream)  one LANE per BGZF block walks the bit stream -- the part of DEFLATE that
//            is serial by nature.  Literal bytes (and the bytes of stored blocks) go straight to their final place in
//            the inflated stream; every MATCH becomes a 32-bit token {bytes in place in front of it, length, distance}.
//            The lane never reads the inflated stream back: its loads are the 8-byte input refills only.
//   phase 2  (resolve_tokens below for the host; gresolve_kernel on the device)  one WAVE per block replays the
//            matches: 64 tokens at a time, output positions by a wave prefix sum, copies in rounds -- a match goes
//            as soon as its source bytes lie below the first match still pending.
//
// A wave executes the union of what its 64 lanes do, one instruction at a time, and a lone wave issues an
// instruction every eight cycles or so: the time of phase 1 is (instructions per trip of the symbol loop, summed
// over every path ANY lane takes) x (trips of the slowest lane).  The symbol loop is therefore written as one
// straight line that every lane walks the same way, with selects instead of branches and nothing that loops:
//   * a trip = up to two literals that the first-level table yields directly, then one full literal/length
//     symbol and -- computed by all lanes, consumed by those that saw a length -- one distance symbol;
//   * literal/length codes: a 7-bit first-level table in LDS; longer codes are decoded canonically WITHOUT a
//     loop -- the 15 bits ahead, bit-reversed, are compared with the first code of every length (one register per
//     length, eight compare-and-select pairs), which yields length and rank at once; the symbol of that rank sits
//     in a sorted list beside the table (LDS; ranks beyond 192: global scratch);
//   * distance codes: no table at all -- the same compare chain over lengths 1..15, the (at most 32) symbols
//     sorted by code in six registers;
//   * literals are gathered in a register and stored eight bytes at a time (or when a match follows);
//   * errors are sticky flags looked at when the lane leaves the loop; every address is clamped.
// Tables cost 640 bytes of LDS per lane: 256 lanes (one wave per SIMD) per CU.
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
// Hides a value from the optimiser (device code): a chain of selects over registers is otherwise turned into a table in
// scratch memory that is read with a computed index -- a memory access in the middle of the symbol loop.
#ifdef __HIP_DEVICE_COMPILE__
#define TEC_OPAQUE(x) asm volatile("" : "+v"(x))
#define TEC_OPAQUE_ORDERED(x) asm volatile("" : "+v"(x) : : "memory")  // ... and memory accesses stay on their side of it
#else
#define TEC_OPAQUE(x) (void)(x)
#define TEC_OPAQUE_ORDERED(x) (void)(x)
#endif

namespace tec_ginf {

constexpr int kLitRoot = 7;
constexpr uint32_t kLitEntries = 1u << kLitRoot;
constexpr uint32_t kLongLds = 192;                                  // long-code symbols kept beside the table
constexpr uint32_t kEntriesPerLane = kLitEntries + kLongLds;        // u16 each: 640 bytes
constexpr uint32_t kLenWords = 20;                                  // code lengths of a dynamic block: 320 nibbles in 64-bit words (global)
constexpr uint32_t kLongSymsPerLane = 320;                          // sorted long-code symbols by rank; ranks >= kLongLds live in global scratch
constexpr uint32_t kInPad = 16;                                     // readable bytes the caller guarantees behind the input

// Reasons a block is handed back to the host's referee decoder (zlib / inflate_fast.h).
enum : int { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadCodeLengths = 3, kBadCode = 4, kBadSymbol = 5, kBadDistance = 6,
             kOutputOverrun = 7, kInputOverrun = 8, kShortOutput = 9, kTokenOverrun = 10 };

// ---- tokens: what phase 1 hands to phase 2, one 32-bit word per match
//   match   bit 31 set, bits 30:23 bytes already in place in front of the match (literals, stored bytes: 0..255),
//           bits 22:15 length - 3 (0..255), bits 14:0 distance - 1
//   skip    bit 31 clear, bits 30:0 bytes already in place (a run longer than 255, or the tail behind the last match;
//           0 = nothing)
// A block of n bytes has at most n / 3 + n / 256 + 2 tokens.
constexpr uint32_t kTokMatch = 1u << 31;
constexpr uint32_t kTokMaxBefore = 255u;
TEC_HD uint32_t tok_before(uint32_t t) { return (t >> 31) ? (t >> 23) & 255u : t; }
TEC_HD uint32_t tok_match_len(uint32_t t) { return (t >> 31) ? ((t >> 15) & 255u) + 3u : 0u; }

// Host sink: a growing array.  (The device sink, gbam.hip, writes into pages drawn from a pool.)  put() returns
// false when the sink is full: the block is then handed to the referee.
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

// Per-lane memory.  The device instantiation keeps `tab` and the head of the long-symbol list in LDS, the rest in
// global memory; the host instantiation uses plain arrays.
//   uint32_t tab(uint32_t i) / void set_tab(uint32_t i, uint32_t v)     i < kLitEntries
//   uint32_t lsym(uint32_t i) / void set_lsym(uint32_t i, uint32_t v)   i < kLongSymsPerLane
//   uint32_t lsym_near(uint32_t i)                                       i < kLongLds: the part of the list that is always at hand
//   uint64_t lenw(uint32_t k) / void set_lenw(uint32_t k, uint64_t v)   k < kLenWords
struct HostMem {
    uint16_t t[kLitEntries];
    uint16_t s[kLongSymsPerLane];
    uint64_t lw[kLenWords];
    uint64_t lenw(uint32_t k) const { return lw[k]; }
    void set_lenw(uint32_t k, uint64_t v) { lw[k] = v; }
    uint32_t tab(uint32_t i) const { return t[i]; }
    void set_tab(uint32_t i, uint32_t v) { t[i] = (uint16_t)v; }
    uint32_t lsym(uint32_t i) const { return s[i]; }
    uint32_t lsym_near(uint32_t i) const { return s[i]; }
    void set_lsym(uint32_t i, uint32_t v) { s[i] = (uint16_t)v; }
};

// The code lengths of a dynamic block, four bits each, are written once in symbol order (sixteen per
// 64-bit word, to the lane's global scratch: stores do not stall the lane) and read back in symbol
// order by the two passes of the table builder (twenty dependent loads per pass).
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

// Canonical decoding without a loop.  The codes of a canonical Huffman code, padded with zeros to 15 bits, are
// ordered by length: length l owns [E[l-1], E[l]) with E[l] = (first code of length l + number of such codes)
// << (15 - l), E non-decreasing.  For lengths l = kLo .. 14 one register q[l - kLo] = E[l] << 16 | R[l + 1] << 4 |
// (l + 1), R[l] = rank of the first code of length l among the codes this chain serves (those of kLo bits or more);
// the last register not above (c << 16 | 0xFFFF) names the length of the 15 bits c, the padded first code of that
// length and its rank.
template <int kLo>
struct Chain {
    uint32_t q[15 - kLo];
    uint32_t sel0;  // E[kLo - 1] << 16 | 0 << 4 | kLo
    uint32_t e15;   // E[15]: 15 bits at or above it are no code
    // c: the next 15 bits of the stream, first bit in bit 14.  Returns the rank; *len the code length; *bad: no code.
    TEC_HD uint32_t find(uint32_t c, uint32_t* len, bool* bad) const {
        const uint32_t cq = (c << 16) | 0xFFFFu;
        uint32_t sel = sel0;
#pragma unroll
        for (int i = 0; i < 15 - kLo; ++i) sel = cq >= q[i] ? q[i] : sel;
        *len = sel & 15u;
        *bad = c >= e15;
        return ((sel >> 4) & 0xFFFu) + ((c - (sel >> 16)) >> (15u - (sel & 15u)));
    }
};

// The (at most 32) distance symbols sorted by code: five bits each, six per register.  (Six scalars, not an array:
// an array picked by a computed index ends up in scratch memory.)
struct DistSyms {
    uint32_t w0, w1, w2, w3, w4, w5;
    TEC_HD void clear() { w0 = w1 = w2 = w3 = w4 = w5 = 0; }
    TEC_HD void set(uint32_t rank, uint32_t sym) {  // rank < 36, once per rank
        const uint32_t r = (rank * 43u) >> 8, v = sym << (5u * (rank - 6u * r));
        w0 |= r == 0 ? v : 0u;
        w1 |= r == 1 ? v : 0u;
        w2 |= r == 2 ? v : 0u;
        w3 |= r == 3 ? v : 0u;
        w4 |= r == 4 ? v : 0u;
        w5 |= r == 5 ? v : 0u;
    }
    TEC_HD uint32_t get(uint32_t rank) const {  // rank < 36
        const uint32_t r = (rank * 43u) >> 8;
        uint32_t v = w0;
        v = r == 1 ? w1 : v;
        TEC_OPAQUE(v);
        v = r == 2 ? w2 : v;
        v = r == 3 ? w3 : v;
        TEC_OPAQUE(v);
        v = r == 4 ? w4 : v;
        v = r == 5 ? w5 : v;
        return (v >> (5u * (rank - 6u * r))) & 31u;
    }
};

// Input: 64-bit bit buffer fed 32 bits at a time from a queue of 8-byte words: `cur` (being moved into the buffer),
// `nxt`, and the word behind it, which EVERY refill loads again -- whether or not the queue moves -- while the queue
// takes the value that an EARLIER refill loaded: a load is never waited for in the refill that issued it.  (A load placed
// under "if the queue moves" is waited for on the spot -- its value has to be merged with the old one where the branch
// ends -- and a lone wave then stands still for a memory latency in every trip of the symbol loop.)
//   refill()      takes the value of the refill before it (always the word at the current address: it was loaded after
//                 that refill had moved the queue, if it did);
//   refill_hot()  for the symbol loop, whose two refills per trip alternate between two slots, each taking the value
//                 loaded two refills earlier -- the queue moves at most every second refill (a word feeds two), so that
//                 value was loaded from the address that is still current.  hot_enter() / hot_exit() around the loop.
struct BitReader {
    const uint8_t* in;
    const uint8_t* in_end;
    const uint8_t* lp;  // address of the word behind `nxt`
    uint64_t buf, cur, nxt;
    uint64_t x0, x1;    // the word at lp, as loaded by earlier refills
    uint32_t cnt;       // valid bits in buf
    uint32_t cur_left;  // bytes of `cur` not yet moved into buf (8, 4 or 0)
    TEC_HD uint64_t load8() const {
        uint64_t v;
        memcpy(&v, lp < in_end ? lp : in_end, 8);  // never beyond in_end + 8
        return v;
    }
    TEC_HD void start(const uint8_t* from) {
        lp = from;
        buf = 0;
        cnt = 0;
        cur = load8();
        lp += 8;
        nxt = load8();
        lp += 8;
        x0 = x1 = load8();
        cur_left = 8;
    }
    TEC_HD void refill_hot(uint64_t& slot) {  // at least 33 valid bits afterwards
        const bool need = cnt <= 32u;
        const bool adv = need && cur_left == 0u;
        cur = adv ? nxt : cur;
        nxt = adv ? slot : nxt;
        // (the old value of the slot is dead before the load that refills it is issued: loaded value and slot can then
        // share a register, and nothing has to wait for the load where the loop closes)
        TEC_OPAQUE_ORDERED(nxt);
        lp += adv ? 8 : 0;
        cur_left = adv ? 8u : cur_left;
        slot = load8();
        buf |= need ? (cur & 0xFFFFFFFFull) << cnt : 0ull;
        cur = need ? cur >> 32 : cur;
        cur_left -= need ? 4u : 0u;
        cnt += need ? 32u : 0u;
    }
    TEC_HD void refill() { refill_hot(x0); }
    TEC_HD void hot_enter() { x1 = x0; }  // (the queue has not moved since x0 was loaded)
    TEC_HD void hot_exit() { x0 = x1; }   // (x1 is the younger load)
    TEC_HD uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    TEC_HD uint32_t ahead15() const { return __builtin_bitreverse32((uint32_t)buf) >> 17; }  // the next 15 bits, first bit on top
    TEC_HD void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
    }
    TEC_HD uint32_t get(uint32_t n) {  // n <= 16, after a refill
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    // bytes of the stream moved into buf so far (`nxt` and the unused part of `cur` are not)
    TEC_HD uint32_t taken() const { return (uint32_t)(lp - in) - 8u - cur_left; }
    // stream position (bytes) of the first bit that has not been consumed, rounded down
    TEC_HD uint32_t byte_pos() const { return taken() - (cnt >> 3); }
    TEC_HD bool overrun() const { return (uint64_t)taken() * 8u - cnt > (uint64_t)(in_end - in) * 8u; }
    TEC_HD bool far_overrun() const { return taken() > (uint32_t)(in_end - in) + 16u; }
};

TEC_HD uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32u - len); }

// Counts the code lengths lens(i), i < n, and checks the code as zlib does (inftrees.c): over-subscribed codes are
// refused, incomplete ones too unless there is no code at all, or the code is a literal/length or distance code of
// a single one-bit code.
template <class LenFn>
TEC_HD bool count_lengths(LenFn lens, uint32_t n, bool is_cl, Counts* cnt) {
    cnt->clear();
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t l = lens(i);
        if (l) cnt->add(l);
    }
    int32_t left = 1;
    uint32_t max_len = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = cnt->get(l);
        left = (left << 1) - (int32_t)c;
        if (left < 0) return false;
        max_len = c ? l : max_len;
    }
    return !(left > 0 && max_len > 0 && (is_cl || max_len != 1u));
}

// The chain registers of a counted code: codes of kLo bits or more.
template <int kLo>
TEC_HD void make_chain(const Counts& cnt, Chain<kLo>* ch) {
    uint32_t code = 0, rank = 0;
    ch->sel0 = (uint32_t)kLo;  // (E[0] = 0; overwritten below when kLo > 1)
#pragma unroll
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = cnt.get(l);
        code += c;                                  // one past the last code of length l
        const uint32_t e = code << (15u - l);       // <= 32768 for a code that is not over-subscribed
        if (l >= (uint32_t)kLo) rank += c;          // rank of the first code of length l + 1
        if (l + 1u == (uint32_t)kLo) ch->sel0 = (e << 16) | (uint32_t)kLo;
        if (l >= (uint32_t)kLo && l <= 14u) ch->q[l - kLo] = (e << 16) | (rank << 4) | (l + 1u);
        if (l == 15u) ch->e15 = e;
        code <<= 1;
    }
}

// Literal/length code (and the code-length code): first-level entries [0, 2^kLitRoot): (symbol << 4) | length for
// codes of at most kLitRoot bits, 0 for prefixes of longer codes and for unused patterns; the symbols of longer
// codes, sorted by (length, value), to m.lsym; the chain for them.  False: see count_lengths.
template <class Mem, class LenFn>
TEC_HD bool build_lit(Mem& m, LenFn lens, uint32_t n, bool is_cl, Chain<kLitRoot + 1>* ch) {
    Counts cnt;
    if (!count_lengths(lens, n, is_cl, &cnt)) return false;
    make_chain(cnt, ch);
    NextCodes next;
    next.clear();
    Counts long_offs;  // where the symbols of each length > root start in the long-symbol list
    long_offs.clear();
    uint32_t code = 0, n_long = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15u; ++l) {
        const uint32_t c = cnt.get(l);
        next.set(l, code);
        code = (code + c) << 1;
        if (l > (uint32_t)kLitRoot) {
            long_offs.add_n(l, n_long);
            n_long += c;
        }
    }
    for (uint32_t i = 0; i < kLitEntries; ++i) m.set_tab(i, 0u);
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens(s);
        if (!l) continue;
        if (l <= (uint32_t)kLitRoot) {
            const uint32_t e = (s << 4) | l;
            for (uint32_t x = bitrev(next.take(l), l); x < kLitEntries; x += 1u << l) m.set_tab(x, e);
        } else {  // first-level entries of its prefix stay 0: the chain decodes it
            m.set_lsym(long_offs.get(l), s);
            long_offs.add(l);
        }
    }
    return true;
}

// Distance code: chain over all lengths, the symbols sorted by (length, value) in registers.
template <class LenFn>
TEC_HD bool build_dist(LenFn lens, uint32_t n, Chain<1>* ch, DistSyms* ds) {
    Counts cnt;
    if (!count_lengths(lens, n, false, &cnt)) return false;
    make_chain(cnt, ch);
    Counts offs;
    offs.clear();
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 15u; ++l) {
        offs.add_n(l, rank);
        rank += cnt.get(l);
    }
    ds->clear();
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens(s);
        if (!l) continue;
        ds->set(offs.get(l), s);
        offs.add(l);
    }
    return true;
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
    uint32_t opb = 0;   // bytes of `out` stored (or left to phase 2) so far
    uint64_t acc = 0;   // literals decoded but not stored yet: out[opb, opb + nacc)
    uint32_t nacc = 0;
    uint32_t run = 0;   // bytes in place since the last token (nacc included)
    // stores the gathered literals: eight bytes at once (what lies above nacc is zero and belongs to bytes that this
    // lane or phase 2 writes later) unless that would touch the next block's bytes
    auto flush = [&]() {
        if (nacc == 0u) return;
        if (opb + 8u <= out_len) {
            memcpy(out + opb, &acc, 8);
        } else {
            uint64_t a = acc;
#pragma unroll 1
            for (uint32_t j = 0; j < nacc; ++j, a >>= 8) out[opb + j] = (uint8_t)a;  // (opb + nacc <= out_len)
        }
        opb += nacc;
        acc = 0;
        nacc = 0;
    };
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
            flush();
            if (out_len - opb < len) return kOutputOverrun;
            const uint8_t* src = in + pos;
            uint8_t* q = out + opb;
            uint32_t k = 0;
            for (; k + 8u <= len; k += 8u) {
                uint64_t v;
                memcpy(&v, src + k, 8);
                memcpy(q + k, &v, 8);
            }
            for (; k < len; ++k) q[k] = src[k];
            opb += len;
            run += len;
            br.start(src + len);
            continue;
        }
        uint32_t hlit = 288, hdist = 32;
        bool fixed = false;
        uint64_t dl_lo = 0x5555555555555555ull, dl_hi = 0x5555555555555555ull;  // distance code lengths, 4 bits each (fixed: all 5)
        Chain<kLitRoot + 1> ll;
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
            // its table sits in the (not yet built) literal/length table; no code is longer than 7 bits
            if (!build_lit(m, [&](uint32_t i) { return (uint32_t)(cl >> (3u * i)) & 7u; }, 19u, true, &ll)) return kBadCodeLengths;
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
        Chain<1> ld;
        DistSyms dsyms;
        uint64_t lcur = 0;  // both passes of the builder ask for the lengths in symbol order
        auto lit_len = [&](uint32_t i) -> uint32_t {
            if (fixed) return i < 144u ? 8u : (i < 256u ? 9u : (i < 280u ? 7u : 8u));
            if ((i & 15u) == 0u) lcur = m.lenw(i >> 4);
            return (uint32_t)(lcur >> (4u * (i & 15u))) & 15u;
        };
        if (!build_lit(m, lit_len, hlit, false, &ll)) return kBadCode;
        if (!build_dist([&](uint32_t i) { return (uint32_t)((i < 16u ? dl_lo : dl_hi) >> (4u * (i & 15u))) & 15u; }, hdist, &ld, &dsyms))
            return kBadCode;
        // ---- the block's symbols: one straight line per trip (see the head of this file)
        int err = kOk;
        bool eob = false;
        br.hot_enter();
        while (!eob && err == kOk) {
            br.refill_hot(br.x0);  // >= 33 bits
            // up to two literals straight from the first-level table (the second only if 20 bits stay for the full symbol)
#pragma unroll
            for (int stage = 0; stage < 2; ++stage) {
                const uint32_t e = m.tab(br.peek(kLitRoot));
                const uint32_t l = e & 15u;
                const bool ok = l != 0u && (e >> 4) < 256u && opb + nacc < out_len && (stage == 0 || br.cnt >= 20u + l);
                br.drop(ok ? l : 0u);
                acc |= (uint64_t)(ok ? e >> 4 : 0u) << (8u * nacc);
                nacc += ok ? 1u : 0u;
                run += ok ? 1u : 0u;
            }
            // one literal/length symbol, whatever its code
            const uint32_t e = m.tab(br.peek(kLitRoot));
            uint32_t llen;
            bool lbad;
            uint32_t rank = ll.find(br.ahead15(), &llen, &lbad);
            rank = rank < kLongSymsPerLane ? rank : kLongSymsPerLane - 1u;
            uint32_t slong = m.lsym_near(rank < kLongLds ? rank : 0u);
            TEC_OPAQUE(slong);  // (keeps this read apart from the rare one below: merged, they become one access to "any" memory)
            const bool direct = (e & 15u) != 0u;
            if (!direct && rank >= kLongLds) {  // (rare: more than kLongLds symbols with long codes)
                slong = m.lsym(rank);
                TEC_OPAQUE(slong);  // (the wait for this load stays inside the branch)
            }
            uint32_t s = direct ? e >> 4 : slong;
            br.drop(direct ? e & 15u : llen);
            err = !direct && lbad ? (int)kBadSymbol : err;
            const bool is_lit = s < 256u, room = opb + nacc < out_len;
            err = is_lit && !room ? (int)kOutputOverrun : err;
            acc |= (uint64_t)(is_lit && room ? s : 0u) << (8u * (nacc & 7u));
            nacc += is_lit && room ? 1u : 0u;
            run += is_lit && room ? 1u : 0u;
            eob = s == 256u;
            const bool is_len = s > 256u;
            s = is_len ? s - 257u : 0u;
            err = s >= 29u ? (int)kBadSymbol : err;
            // length: symbols 0..7 -> 3..10, then groups of four per extra bit, 28 -> 258
            const uint32_t lx = s < 8u ? 0u : (s >= 28u ? 0u : (s - 4u) >> 2);
            uint32_t length = s < 8u ? 3u + s : (s >= 28u ? 258u : 3u + ((4u + (s & 3u)) << lx));
            length += br.get(lx);  // (>= 33 bits at the top, at most 14 + 15 + 5 used: the second literal left 20 for this symbol)
            if (is_len || eob || nacc > 5u) flush();  // (the next trip may add three literals)
            br.refill_hot(br.x1);  // >= 33 bits again: a 15-bit distance code and 13 extra bits fit
            uint32_t dlen;
            bool dbad;
            uint32_t drank = ld.find(br.ahead15(), &dlen, &dbad);
            const uint32_t d = dsyms.get(drank < 35u ? drank : 35u);
            if (is_len) {
                br.drop(dlen);
                err = dbad || d >= 30u ? (int)kBadDistance : err;
                const uint32_t dc = d < 30u ? d : 0u;
                const uint32_t dx = dc < 4u ? 0u : (dc - 2u) >> 1;
                uint32_t dist = dc < 4u ? 1u + dc : 1u + ((2u + (dc & 1u)) << dx);
                dist += br.get(dx);
                err = dist > opb ? (int)kBadDistance : err;
                err = out_len - opb < length ? (int)kOutputOverrun : err;
                if (err == kOk) {
                    if (run > kTokMaxBefore) {  // a long run of bytes in place gets a token of its own
                        if (!sink.put(run)) err = kTokenOverrun;
                        run = 0;
                    }
                    if (!sink.put(kTokMatch | (run << 23) | ((length - 3u) << 15) | (dist - 1u))) err = kTokenOverrun;
                    run = 0;
                    opb += length;
                }
            }
            err = br.far_overrun() ? (int)kInputOverrun : err;
        }
        br.hot_exit();
        if (err != kOk) return err;
    }
    flush();
    if (br.overrun()) return kInputOverrun;
    if (run && !sink.put(run)) return kTokenOverrun;  // (run <= out_len <= 65 536)
    return opb == out_len ? kOk : kShortOutput;
}

}  // namespace tec_ginf
