// Native BGZF/BAM (and plain SAM) -> structure-of-arrays decoder: tec_bam_* entry points.
//
// The step BEFORE the counting path: replaces pysam.AlignmentFile(filename, 'r') + next(sam)
// (te_count/te_count.py:65,78-79 / 190,203 / 351,393) and the per-record attribute reads
// (te_count.py:81-98, 204-214, 394-438).  pysam / htslib are third-party code that is not part
// of the reference tree; this decoder follows the published SAM/BAM specification (SAMv1:
// BGZF section 4.1, BAM record layout section 4.2, CIGAR reference-consuming ops M/D/N/=/X).
// Host code only (no kernels); BGZF blocks are inflated by a pool of threads.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <sys/stat.h>
#include <vector>

#include "bamscan.h"
#include "ginflate.h"
#include "inflate_fast.h"
#include "tecount_internal.h"

// A growable byte window that is never zero-filled (std::vector::resize would touch every byte).
struct ByteBuf {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ~ByteBuf() { free(p); }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
    void drop_front(size_t k) {
        if (k == 0) return;
        memmove(p, p + k, n - k);
        n -= k;
    }
    bool grow(size_t add) {
        if (n + add > cap) {
            const size_t c = std::max(cap * 2, n + add + 4096);
            uint8_t* q = static_cast<uint8_t*>(realloc(p, c));
            if (!q) return false;
            p = q;
            cap = c;
        }
        n += add;
        return true;
    }
};

struct tec_bam {
    FILE* fh = nullptr;
    bool is_bam = false;
    std::string err;
    std::vector<std::string> ref_names;
    std::vector<uint32_t> ref_lens;
    // uncompressed stream window
    ByteBuf buf;
    size_t buf_pos = 0;
    bool eof = false;
    // compressed bytes (BAM only): the whole file mapped read-only, or -- when it cannot be
    // mapped (a pipe) -- a window read ahead with fread
    const uint8_t* map = nullptr;
    size_t map_len = 0;
    ByteBuf raw;
    size_t raw_pos = 0;
    bool file_eof = false;
    int n_threads = 4;
    uint64_t n_records = 0;
    // byte-range ownership (tec_bam_open_range): this reader delivers the records that START inside a
    // BGZF block whose file offset lies in [range_begin, range_end)
    bool ranged = false;
    uint64_t range_begin = 0, range_end = ~0ull;
    bool owned_done = false;     // every owned block has been inflated into the window
    bool in_epilogue = false;    // completing the last owned record from blocks beyond range_end
    bool finished = false;       // the end token has been computed
    struct WinBlock {
        int64_t win_off;    // of the block's first inflated byte relative to buf (negative: the block straddles the front)
        uint64_t file_off;  // of the BGZF block in the file
        uint32_t isize;
    };
    std::vector<WinBlock> wblocks;  // the blocks whose inflated bytes are (still) in the window
    uint64_t tok_first[2] = {0, 0}, tok_end[2] = {0, 0};  // {BGZF block file offset, offset inside its inflated bytes}
    bool have_first = false;
    // SAM text state
    std::string line;
};

namespace {

thread_local std::string g_bam_err;

int bam_fail(tec_bam* r, int code, const std::string& msg) {
    if (r) r->err = msg;
    g_bam_err = msg;
    return code;
}

struct RawBlock {
    uint64_t file_off = 0;  // of the BGZF block (mapped files)
    size_t off = 0;  // of the DEFLATE body inside r->raw
    size_t clen = 0;
    uint32_t isize = 0;
    size_t out_off = 0;
};

// One raw-DEFLATE block body -> exactly isize bytes.  The fast decoder first (inflate_fast.h, ~2x
// zlib on BAM payloads); anything it refuses goes to zlib, whose verdict is final.
bool inflate_one(const uint8_t* in, size_t in_len, uint8_t* out, size_t isize, tec_inflate::Tables* tabs) {
    if (tec_inflate::inflate_block(in, in_len, out, isize, tabs)) return true;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(in);
    zs.avail_in = (uInt)in_len;
    zs.next_out = out;
    zs.avail_out = (uInt)isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}

const uint8_t* raw_base(const tec_bam* r) { return r->map ? r->map : r->raw.data(); }

// At least `need` unread compressed bytes at raw_base() + raw_pos; false at end of file.
bool raw_ensure(tec_bam* r, size_t need) {
    if (r->map) return r->map_len - r->raw_pos >= need;
    while (r->raw.size() - r->raw_pos < need) {
        if (r->file_eof) return false;
        const size_t want = std::max<size_t>(need, 16u << 20);
        const size_t old = r->raw.size();
        if (!r->raw.grow(want)) return false;
        const size_t got = fread(r->raw.data() + old, 1, want, r->fh);
        r->raw.n = old + got;
        if (got < want) r->file_eof = true;
    }
    return true;
}
size_t raw_left(const tec_bam* r) { return (r->map ? r->map_len : r->raw.size()) - r->raw_pos; }

// Drops the consumed front of the inflate window.  Block positions are signed: a block that
// straddles the new front keeps a negative offset, so window byte p is byte p - win_off of its block.
void compact_window(tec_bam* r) {
    const size_t k = r->buf_pos;
    if (!k) return;
    r->buf.drop_front(k);
    r->buf_pos = 0;
    size_t keep = 0;
    for (const auto& w : r->wblocks) {
        if (w.win_off + (int64_t)w.isize <= (int64_t)k) continue;  // wholly consumed
        r->wblocks[keep] = w;
        r->wblocks[keep++].win_off -= (int64_t)k;
    }
    r->wblocks.resize(keep);
}

// Finds up to max_blocks BGZF blocks at the read position; *total = their inflated size.
int discover_blocks(tec_bam* r, size_t max_blocks, std::vector<RawBlock>* blocks, size_t* total) {
    blocks->clear();
    *total = 0;
    if (r->eof) return 0;
    if (!r->map && r->raw_pos) {
        r->raw.drop_front(r->raw_pos);
        r->raw_pos = 0;
    }
    while (blocks->size() < max_blocks) {
        if (r->ranged && !r->in_epilogue && r->raw_pos >= r->range_end) {  // the next block belongs to the next range
            r->owned_done = true;
            break;
        }
        if (!raw_ensure(r, 18)) {
            if (raw_left(r) == 0) {
                r->eof = true;
                r->owned_done = true;
                break;
            }
            return bam_fail(r, TEC_EINVAL, "not a BGZF block (truncated or not a BAM file)");
        }
        const uint8_t* h = raw_base(r) + r->raw_pos;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4))
            return bam_fail(r, TEC_EINVAL, "not a BGZF block (truncated or not a BAM file)");
        const unsigned xlen = h[10] | (h[11] << 8);
        if (!raw_ensure(r, 12 + (size_t)xlen)) return bam_fail(r, TEC_EINVAL, "truncated BGZF extra field");
        h = raw_base(r) + r->raw_pos;  // the window may have moved
        const uint8_t* extra = h + 12;
        // the BC subfield is normally first; walk the extra field to be safe
        int bsize = -1;
        for (size_t o = 0; o + 4 <= xlen;) {
            const unsigned slen = extra[o + 2] | (extra[o + 3] << 8);
            if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = extra[o + 4] | (extra[o + 5] << 8);
            o += 4 + slen;
        }
        if (bsize < 0) return bam_fail(r, TEC_EINVAL, "BGZF block without BC subfield");
        const long clen = (long)bsize + 1 - 12 - (long)xlen - 8;
        if (clen < 0) return bam_fail(r, TEC_EINVAL, "corrupt BGZF block size");
        if (!raw_ensure(r, (size_t)bsize + 1)) return bam_fail(r, TEC_EINVAL, "truncated BGZF block");
        const uint8_t* tail = raw_base(r) + r->raw_pos + 12 + xlen + (size_t)clen;
        RawBlock b;
        b.file_off = r->raw_pos;
        b.off = r->raw_pos + 12 + xlen;
        b.clen = (size_t)clen;
        b.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (b.isize > 65536u) return bam_fail(r, TEC_EINVAL, "corrupt BGZF block (ISIZE above 64 KiB)");  // SAMv1 4.1
        b.out_off = *total;
        *total += b.isize;
        r->raw_pos += (size_t)bsize + 1;
        blocks->push_back(b);
    }
    return blocks->empty() ? 0 : 1;
}

// Reads up to max_blocks BGZF blocks and appends their inflated bytes to r->buf (header parsing;
// the record path inflates inside read_chunk below).
int refill(tec_bam* r, size_t max_blocks) {
    compact_window(r);
    std::vector<RawBlock> blocks;
    size_t total = 0;
    const int k = discover_blocks(r, max_blocks, &blocks, &total);
    if (k <= 0) return k;
    const size_t base = r->buf.size();
    for (const RawBlock& b : blocks) r->wblocks.push_back({(int64_t)(base + b.out_off), b.file_off, b.isize});
    if (!r->buf.grow(total)) return bam_fail(r, TEC_EINVAL, "out of memory for the inflate window");
    std::unique_ptr<tec_inflate::Tables> tabs(new tec_inflate::Tables);
    for (const RawBlock& b : blocks)
        if (b.isize && !inflate_one(raw_base(r) + b.off, b.clen, r->buf.data() + base + b.out_off, b.isize, tabs.get())) {
            r->buf.n = base;  // the window holds nothing of this refill
            while (!r->wblocks.empty() && r->wblocks.back().win_off >= (int64_t)base) r->wblocks.pop_back();
            return bam_fail(r, TEC_EINVAL, "BGZF inflate failed (corrupt block)");
        }
    return 1;
}

// Window offset where inflated bytes of blocks beyond the owned range begin (SIZE_MAX: none in the window).
size_t own_limit(const tec_bam* r) {
    if (!r->ranged) return SIZE_MAX;
    for (const auto& w : r->wblocks)
        if (w.file_off >= r->range_end) return w.win_off > 0 ? (size_t)w.win_off : 0;
    return SIZE_MAX;
}

// refill() that keeps the window's front where it is (buf_pos is 0 while the first record of a range is searched)
int refill_keep(tec_bam* r, size_t max_blocks) {
    const size_t save = r->buf_pos;
    r->buf_pos = 0;
    const int k = refill(r, max_blocks);
    r->buf_pos = save;
    return k;
}

// Makes at least `need` unread bytes available; returns false at a clean end of stream.
bool ensure(tec_bam* r, size_t need, int* rc) {
    *rc = 0;
    while (r->buf.size() - r->buf_pos < need) {
        if (r->eof) return false;
        const int k = refill(r, 8);
        if (k < 0) {
            *rc = k;
            return false;
        }
    }
    return true;
}

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

uint64_t mate_key_hash(const char* q, size_t n) { return tec_bamscan::mate_key_hash((const uint8_t*)q, n); }

struct OutArrays {
    int32_t *tid, *pos, *end;
    uint16_t* flag;
    uint8_t* mapq;
    uint64_t* qkey;
    uint8_t *cb24, *ub24, *tagflags;
};

void put_tag(uint8_t* dst24, const char* v, size_t n, uint8_t* tf, uint8_t present_bit, uint8_t long_bit) {
    *tf |= present_bit;
    if (n > 24) {
        *tf |= long_bit;
        return;
    }
    memset(dst24, 0, 24);
    memcpy(dst24, v, n);
}

int read_bam_header(tec_bam* r) {
    int rc;
    if (!ensure(r, 12, &rc)) return bam_fail(r, rc ? rc : TEC_EINVAL, "empty BAM");
    const uint8_t* p = r->buf.data() + r->buf_pos;
    if (memcmp(p, "BAM\1", 4) != 0) return bam_fail(r, TEC_EINVAL, "bad BAM magic");
    const uint32_t l_text = le32(p + 4);
    if (!ensure(r, 12 + (size_t)l_text, &rc)) return bam_fail(r, TEC_EINVAL, "truncated BAM header");
    r->buf_pos += 8 + l_text;
    p = r->buf.data() + r->buf_pos;
    const uint32_t n_ref = le32(p);
    r->buf_pos += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!ensure(r, 4, &rc)) return bam_fail(r, TEC_EINVAL, "truncated BAM reference list");
        const uint32_t l_name = le32(r->buf.data() + r->buf_pos);
        if (!ensure(r, 8 + (size_t)l_name, &rc)) return bam_fail(r, TEC_EINVAL, "truncated BAM reference list");
        const uint8_t* q = r->buf.data() + r->buf_pos + 4;
        r->ref_names.emplace_back((const char*)q, l_name ? l_name - 1 : 0);
        r->ref_lens.push_back(le32(q + l_name));
        r->buf_pos += 8 + l_name;
    }
    return TEC_OK;
}

// ---------------------------------------------------------------- SAM text
bool sam_getline(tec_bam* r) {
    r->line.clear();
    int c;
    bool any = false;
    while ((c = fgetc(r->fh)) != EOF) {
        any = true;
        if (c == '\n') break;
        r->line.push_back((char)c);
    }
    if (!r->line.empty() && r->line.back() == '\r') r->line.pop_back();
    return any;
}

int read_sam_header(tec_bam* r) {
    for (;;) {
        const long at = ftell(r->fh);
        if (!sam_getline(r)) break;
        if (r->line.empty() || r->line[0] != '@') {
            fseek(r->fh, at, SEEK_SET);
            break;
        }
        if (r->line.compare(0, 3, "@SQ") == 0) {
            std::string name;
            uint32_t len = 0;
            size_t i = 0;
            while (i < r->line.size()) {
                size_t j = r->line.find('\t', i);
                if (j == std::string::npos) j = r->line.size();
                if (r->line.compare(i, 3, "SN:") == 0) name = r->line.substr(i + 3, j - i - 3);
                if (r->line.compare(i, 3, "LN:") == 0) len = (uint32_t)strtoul(r->line.c_str() + i + 3, nullptr, 10);
                i = j + 1;
            }
            r->ref_names.push_back(name);
            r->ref_lens.push_back(len);
        }
    }
    return TEC_OK;
}

int sam_record(tec_bam* r, uint64_t o, const OutArrays& a) {
    std::vector<std::pair<size_t, size_t>> f;  // field spans
    const std::string& s = r->line;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find('\t', i);
        if (j == std::string::npos) j = s.size();
        f.emplace_back(i, j - i);
        i = j + 1;
    }
    if (f.size() < 11) return bam_fail(r, TEC_EINVAL, "SAM line with fewer than 11 fields");
    auto str = [&](int k) { return s.substr(f[k].first, f[k].second); };
    const std::string rname = str(2);
    int tid = -1;
    if (rname != "*")
        for (size_t k = 0; k < r->ref_names.size(); ++k)
            if (r->ref_names[k] == rname) {
                tid = (int)k;
                break;
            }
    const long pos1 = strtol(s.c_str() + f[3].first, nullptr, 10);
    const std::string cigar = str(5);
    long reflen = 0;
    bool have_cigar = cigar != "*";
    if (have_cigar) {
        long num = 0;
        for (char c : cigar) {
            if (c >= '0' && c <= '9') {
                num = num * 10 + (c - '0');
            } else {
                if (c == 'M' || c == 'D' || c == 'N' || c == '=' || c == 'X') reflen += num;
                num = 0;
            }
        }
    }
    a.tid[o] = tid;
    a.pos[o] = (int32_t)(pos1 - 1);
    // pysam: reference_end is None without a CIGAR; htslib's bam_endpos counts a CIGAR that
    // consumes no reference base (only S/I/H/P) as one base
    a.end[o] = have_cigar ? (int32_t)(pos1 - 1 + (reflen ? reflen : 1)) : INT32_MIN;
    a.flag[o] = (uint16_t)strtoul(s.c_str() + f[1].first, nullptr, 10);
    a.mapq[o] = (uint8_t)strtoul(s.c_str() + f[4].first, nullptr, 10);
    if (a.qkey) a.qkey[o] = mate_key_hash(s.c_str() + f[0].first, f[0].second);
    if (a.tagflags) {
        uint8_t tf = 0;
        const char *cb = nullptr, *cr = nullptr, *ub = nullptr, *ur = nullptr;
        size_t ncb = 0, ncr = 0, nub = 0, nur = 0;
        for (size_t k = 11; k < f.size(); ++k) {
            if (f[k].second < 5) continue;
            const char* t = s.c_str() + f[k].first;
            if (t[2] != ':' || t[3] != 'Z' || t[4] != ':') continue;
            const char* v = t + 5;
            const size_t n = f[k].second - 5;
            if (t[0] == 'C' && t[1] == 'B') { cb = v; ncb = n; }
            else if (t[0] == 'C' && t[1] == 'R') { cr = v; ncr = n; }
            else if (t[0] == 'U' && t[1] == 'B') { ub = v; nub = n; }
            else if (t[0] == 'U' && t[1] == 'R') { ur = v; nur = n; }
        }
        memset(a.cb24 + 24 * o, 0, 24);
        memset(a.ub24 + 24 * o, 0, 24);
        if (cb) put_tag(a.cb24 + 24 * o, cb, ncb, &tf, 1, 4);
        else if (cr) put_tag(a.cb24 + 24 * o, cr, ncr, &tf, 1, 4);
        if (ub) put_tag(a.ub24 + 24 * o, ub, nub, &tf, 2, 8);
        else if (ur) put_tag(a.ub24 + 24 * o, ur, nur, &tf, 2, 8);
        a.tagflags[o] = tf;
    }
    return TEC_OK;
}

}  // namespace

extern "C" int tec_inflate_raw(const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_len, int use_zlib) {
    if ((!in && in_len) || (!out && out_len)) return TEC_EINVAL;
    if (use_zlib) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return TEC_EINVAL;
        zs.next_in = const_cast<Bytef*>(in);
        zs.avail_in = (uInt)in_len;
        zs.next_out = out;
        zs.avail_out = (uInt)out_len;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        return rc == Z_STREAM_END && zs.avail_out == 0 ? TEC_OK : TEC_EINVAL;
    }
    if (use_zlib == 2) {  // the GPU lane decoder (ginflate.h), host instantiation of the same source
        if (in_len > 0x7FFFFFFFull || out_len > 0x7FFFFFFFull) return TEC_EINVAL;
        std::vector<uint8_t> padded(in_len + tec_ginf::kInPad, 0);  // the decoder reads whole 8-byte words
        if (in_len) memcpy(padded.data(), in, in_len);
        std::unique_ptr<tec_ginf::HostMem> m(new tec_ginf::HostMem);
        memset(m.get(), 0x77, sizeof(tec_ginf::HostMem));
        // phase 1 (the lane's code) into a token array, phase 2 replayed sequentially
        std::vector<uint32_t> tok((size_t)out_len / 3 + out_len / 128 + 16);
        tec_ginf::VectorSink sink{tok.data(), (uint32_t)tok.size()};
        if (tec_ginf::inflate_stream(padded.data(), (uint32_t)in_len, out, (uint32_t)out_len, *m, sink) != tec_ginf::kOk) return TEC_EINVAL;
        return tec_ginf::resolve_tokens(tok.data(), sink.n, out, (uint32_t)out_len) ? TEC_OK : TEC_EINVAL;
    }
    std::unique_ptr<tec_inflate::Tables> tabs(new tec_inflate::Tables);
    return tec_inflate::inflate_block(in, in_len, out, out_len, tabs.get()) ? TEC_OK : TEC_EINVAL;
}

extern "C" const char* tec_bam_last_error(const tec_bam* r) { return r ? r->err.c_str() : g_bam_err.c_str(); }

namespace {
// First BGZF block at or after file offset `from`: the 16 fixed header bytes (magic, FEXTRA, XLEN = 6, 'BC', SLEN = 2) and
// a chain of three consecutive blocks that look the same (or the end of the file).
bool bgzf_header_at(const uint8_t* m, size_t len, size_t o, size_t* bsize) {
    if (len - o < 18) return false;
    const uint8_t* h = m + o;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || h[3] != 4) return false;
    // the BC subfield anywhere in the extra field (htslib writes it alone; the format allows others around it)
    const size_t xlen = h[10] | (h[11] << 8);
    if (xlen < 6 || len - o < 12 + xlen) return false;
    long bs = -1;
    for (size_t x = 0; x + 4 <= xlen;) {
        const size_t slen = h[12 + x + 2] | (h[12 + x + 3] << 8);
        if (h[12 + x] == 'B' && h[12 + x + 1] == 'C' && slen == 2 && x + 6 <= xlen) bs = h[12 + x + 4] | (h[12 + x + 5] << 8);
        x += 4 + slen;
    }
    if (bs < 0) return false;
    *bsize = (size_t)bs + 1;
    return *bsize >= 20 + xlen && o + *bsize <= len;
}
size_t find_bgzf_block(const uint8_t* m, size_t len, size_t from) {
    for (size_t o = from; o + 18 <= len; ++o) {
        if (m[o] != 31 || m[o + 1] != 139) continue;
        size_t p = o, bs = 0;
        int good = 0;
        while (good < 3 && p < len && bgzf_header_at(m, len, p, &bs)) {
            p += bs;
            ++good;
        }
        if (good == 3 || (good > 0 && p == len)) return o;
    }
    return len;
}
}  // namespace

// ---- BGZF helpers shared with the on-GPU decoder (gbam.hip)
size_t tec_bgzf_find_block(const uint8_t* m, size_t len, size_t from) { return find_bgzf_block(m, len, from); }

// Header of the BGZF block at m + o (general form: any extra subfields).  false: not a complete block.
bool tec_bgzf_parse(const uint8_t* m, size_t len, size_t o, size_t* bsize, size_t* body_off, size_t* body_len, uint32_t* isize) {
    if (len - o < 18) return false;
    const uint8_t* h = m + o;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
    const size_t xlen = h[10] | (h[11] << 8);
    if (len - o < 12 + xlen) return false;
    long bs = -1;
    for (size_t x = 0; x + 4 <= xlen;) {
        const size_t slen = h[12 + x + 2] | (h[12 + x + 3] << 8);
        if (h[12 + x] == 'B' && h[12 + x + 1] == 'C' && slen == 2 && x + 6 <= xlen) bs = h[12 + x + 4] | (h[12 + x + 5] << 8);
        x += 4 + slen;
    }
    if (bs < 0) return false;
    const long clen = bs + 1 - 12 - (long)xlen - 8;
    if (clen < 0 || len - o < (size_t)bs + 1) return false;
    *bsize = (size_t)bs + 1;
    *body_off = o + 12 + xlen;
    *body_len = (size_t)clen;
    const uint8_t* t = m + o + *bsize - 4;
    *isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
    return *isize <= 65536u;
}

// The referee: inflates one block body on the host (fast decoder, then zlib).
bool tec_bgzf_inflate_host(const uint8_t* body, size_t body_len, uint8_t* out, uint32_t isize) {
    if (!isize) return true;
    std::unique_ptr<tec_inflate::Tables> tabs(new tec_inflate::Tables);
    return inflate_one(body, body_len, out, isize, tabs.get());
}

static int bam_open_impl(const char* path, int n_threads, uint64_t range_begin, uint64_t range_end, bool ranged, tec_bam** out);

extern "C" int tec_bam_open(const char* path, int n_threads, tec_bam** out) {
    return bam_open_impl(path, n_threads, 0, ~0ull, false, out);
}

extern "C" int tec_bam_open_range(const char* path, int n_threads, uint64_t range_begin, uint64_t range_end, tec_bam** out) {
    return bam_open_impl(path, n_threads, range_begin, range_end, true, out);
}

extern "C" int tec_bam_range_tokens(tec_bam* r, uint64_t first[2], uint64_t end[2], int* has_first, int* finished) {
    if (!r) return TEC_EINVAL;
    if (first) { first[0] = r->tok_first[0]; first[1] = r->tok_first[1]; }
    if (end) { end[0] = r->tok_end[0]; end[1] = r->tok_end[1]; }
    if (has_first) *has_first = r->have_first ? 1 : 0;
    if (finished) *finished = r->finished ? 1 : 0;
    return TEC_OK;
}

extern "C" uint64_t tec_bam_file_size(const tec_bam* r) { return r && r->map ? (uint64_t)r->map_len : 0; }

static int bam_open_impl(const char* path, int n_threads, uint64_t range_begin, uint64_t range_end, bool ranged, tec_bam** out) {
    if (!path || !out) return bam_fail(nullptr, TEC_EINVAL, "tec_bam_open: NULL argument");
    *out = nullptr;
    FILE* fh = fopen(path, "rb");
    if (!fh) return bam_fail(nullptr, TEC_EINVAL, std::string("cannot open ") + path);
    tec_bam* r = new tec_bam();
    r->fh = fh;
    // n_threads <= 0: TEC_BAM_THREADS if set, else min(16, cores) (16 is one GPU's share of a host's cores)
    {
        const char* e = getenv("TEC_BAM_THREADS");
        const int env_threads = e ? atoi(e) : 0;
        r->n_threads = n_threads > 0 ? n_threads
                       : env_threads > 0 ? std::min(env_threads, 256)
                                         : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    }
    uint8_t m[4] = {0};
    const size_t got = fread(m, 1, 4, fh);
    fseek(fh, 0, SEEK_SET);
    r->is_bam = got >= 2 && m[0] == 31 && m[1] == 139;
    if (r->is_bam) {  // map the file when it can be mapped (no copy through the page cache into a user buffer)
        struct stat st;
        if (fstat(fileno(fh), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fh), 0);
            if (m != MAP_FAILED) {
                r->map = static_cast<const uint8_t*>(m);
                r->map_len = (size_t)st.st_size;
                (void)madvise(m, r->map_len, MADV_SEQUENTIAL);
            }
        }
    }
    int rc = r->is_bam ? read_bam_header(r) : read_sam_header(r);
    auto bail = [&](int code) {
        g_bam_err = r->err;
        if (r->map) munmap(const_cast<uint8_t*>(r->map), r->map_len);
        fclose(fh);
        delete r;
        return code;
    };
    if (rc != TEC_OK) return bail(rc);
    if (ranged) {
        if (!r->is_bam || !r->map) return bail(bam_fail(r, TEC_EINVAL, "byte ranges need a BAM file that can be mapped"));
        r->ranged = true;
        r->range_begin = range_begin;
        r->range_end = range_end;
    }
    if (r->is_bam && r->map) {
        if (ranged && range_begin > 0) {
            // not the first range: start at the first BGZF block inside it and GUESS the first record
            // start there (bamscan.h plausible_record); the guess is verified against the previous
            // range's end token by the caller
            r->buf.n = 0;
            r->buf_pos = 0;
            r->wblocks.clear();
            r->eof = false;
            r->owned_done = false;  // (reading the header of a file of a few blocks has already met the end of the file)
            r->raw_pos = find_bgzf_block(r->map, r->map_len, (size_t)std::min<uint64_t>(range_begin, r->map_len));
            std::vector<uint32_t> lens(r->ref_lens);
            size_t x = 0;
            bool found = false;
            for (;;) {
                const size_t n = r->buf.size();
                const size_t lim = std::min(n, own_limit(r));  // the record must start in an owned block
                for (; x < lim && x + 36 <= n && !found; ++x)
                    if (tec_bamscan::plausible_record(r->buf.data(), n, x, (int32_t)r->ref_names.size(), lens.data(), 3)) {
                        found = true;
                        break;
                    }
                if (found || r->eof || (r->owned_done && x >= lim)) break;
                // a start position in the last bytes of the owned blocks can only be judged with the bytes behind it
                if (r->owned_done) r->in_epilogue = true;
                rc = refill_keep(r, 4);
                if (rc < 0) return bail(rc);
                if (rc == 0) break;
            }
            r->in_epilogue = false;
            if (found) {
                r->buf_pos = x;
                bool in_owned = false;
                for (const auto& w : r->wblocks)
                    if ((int64_t)x >= w.win_off && (int64_t)x < w.win_off + (int64_t)w.isize) {
                        r->tok_first[0] = w.file_off;
                        r->tok_first[1] = (uint64_t)((int64_t)x - w.win_off);
                        in_owned = w.file_off < r->range_end;
                    }
                r->have_first = in_owned;
                if (!in_owned) found = false;
            }
            if (!found) {  // no record starts in this range
                r->buf.n = 0;
                r->buf_pos = 0;
                r->owned_done = true;
                r->finished = true;
                r->have_first = false;
            }
        } else {
            // the first range (or the whole file): records start right after the header
            for (const auto& w : r->wblocks)
                if ((int64_t)r->buf_pos >= w.win_off && (int64_t)r->buf_pos < w.win_off + (int64_t)w.isize) {
                    r->tok_first[0] = w.file_off;
                    r->tok_first[1] = (uint64_t)((int64_t)r->buf_pos - w.win_off);
                    r->have_first = true;
                }
            if (ranged && r->raw_pos >= r->range_end) r->owned_done = true;
        }
    }
    *out = r;
    return TEC_OK;
}

extern "C" void tec_bam_close(tec_bam* r) {
    if (!r) return;
    if (r->map) munmap(const_cast<uint8_t*>(r->map), r->map_len);
    if (r->fh) fclose(r->fh);
    delete r;
}

extern "C" int tec_bam_n_refs(const tec_bam* r) { return r ? (int)r->ref_names.size() : -1; }
extern "C" uint32_t tec_bam_ref_len(const tec_bam* r, int i) {
    return (r && i >= 0 && (size_t)i < r->ref_lens.size()) ? r->ref_lens[i] : 0;
}
extern "C" const char* tec_bam_ref_name(const tec_bam* r, int i) {
    return (r && i >= 0 && (size_t)i < r->ref_names.size()) ? r->ref_names[i].c_str() : nullptr;
}

// One record (rec points at its block_size field, the whole record is in memory) -> row w of the SoA.
// The field rules live in bamscan.h, shared with the on-GPU decoder.
bool extract_record(const uint8_t* rec, uint64_t w, const OutArrays& a) {
    tec_bamscan::Core c;
    if (!tec_bamscan::parse_core(rec, true, &c)) return false;
    a.tid[w] = c.tid;
    a.pos[w] = c.pos;
    a.end[w] = c.end;
    a.flag[w] = (uint16_t)c.flag;
    a.mapq[w] = (uint8_t)c.mapq;
    if (a.qkey) a.qkey[w] = tec_bamscan::mate_key_hash(c.name, c.l_name ? c.l_name - 1 : 0);
    if (a.tagflags) a.tagflags[w] = (uint8_t)tec_bamscan::scan_tags(c.aux, c.rec_end, a.cb24 + 24 * w, a.ub24 + 24 * w);
    return true;
}

// One chunk: 512 or more BGZF blocks (at least 32 per thread, so that a share stays cache-sized) in contiguous shares, one per thread.  A thread inflates its
// share, then -- holding the baton {next record start, records so far} from the thread before it --
// walks the record boundaries of its share while the bytes are still in its cache, hands the baton
// on, and extracts the fields of its own records.  Only the boundary walk is sequential, and it
// runs on cache-hot data (a single-threaded walk over the whole inflated window was memory-bound
// and capped the decoder at ~5 M records/s however many threads inflated).
int read_chunk(tec_bam* r, uint64_t max_left, uint64_t first_row, const OutArrays& a, uint64_t* got) {
    *got = 0;
    compact_window(r);
    std::vector<RawBlock> blocks;
    size_t total = 0;
    // chunk size grows with the thread count: the per-chunk costs (thread start, the baton chain,
    // block discovery) are sequential
    const int k = discover_blocks(r, std::max<size_t>(512, 32 * (size_t)r->n_threads), &blocks, &total);
    if (k < 0) return k;
    if (k == 0) return TEC_OK;
    const size_t base = r->buf.size();
    if (!r->buf.grow(total)) return bam_fail(r, TEC_EINVAL, "out of memory for the inflate window");
    for (const RawBlock& b : blocks) r->wblocks.push_back({(int64_t)(base + b.out_off), b.file_off, b.isize});
    const int nt = std::max(1, std::min<int>(r->n_threads, (int)blocks.size()));
    const size_t per = (blocks.size() + nt - 1) / nt;
    struct Baton {
        size_t pos = 0;
        uint64_t count = 0;
        bool ready = false;
        bool bad = false;  // a share before this one failed: nothing downstream is trusted
    };
    std::vector<Baton> baton(nt + 1);
    baton[0].pos = 0;  // the window was compacted: the first unparsed record starts at 0
    baton[0].ready = true;
    std::vector<int> status(nt, 0);  // 1 inflate failed, 2 corrupt record size, 3 corrupt record layout
    std::mutex mu;
    std::condition_variable cv;
    const uint8_t* raw = raw_base(r);
    uint8_t* win = r->buf.data();
    auto work = [&](int t) {
        const size_t b0 = (size_t)t * per, b1 = std::min(blocks.size(), b0 + per);
        if (b0 < b1) {
            std::unique_ptr<tec_inflate::Tables> tabs(new tec_inflate::Tables);
            for (size_t i = b0; i < b1 && !status[t]; ++i) {
                const RawBlock& b = blocks[i];
                if (b.isize && !inflate_one(raw + b.off, b.clen, win + base + b.out_off, b.isize, tabs.get())) status[t] = 1;
            }
        }
        // bytes [0, share_end) are inflated once the baton arrives (every earlier thread passes it after its own inflate)
        const size_t share_end = b1 < blocks.size() ? base + blocks[b1].out_off : base + total;
        Baton in;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return baton[t].ready; });
            in = baton[t];
        }
        std::vector<size_t> offs;
        size_t pos = in.pos;
        uint64_t count = in.count;
        if (!status[t] && !in.bad && b0 < b1) {
            while (count < max_left && pos + 4 <= share_end) {
                const uint32_t bs = le32(win + pos);
                if (bs < 32) {
                    status[t] = 2;
                    break;
                }
                if (pos + 4 + (size_t)bs > share_end) break;  // completed by a later share (or the next chunk)
                offs.push_back(pos);
                pos += 4 + (size_t)bs;
                ++count;
            }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            baton[t + 1].pos = pos;
            baton[t + 1].count = count;
            baton[t + 1].bad = in.bad || status[t] != 0;
            baton[t + 1].ready = true;
        }
        cv.notify_all();
        uint64_t w = first_row + in.count;
        for (size_t o : offs) {
            if (!extract_record(win + o, w++, a)) {
                status[t] = 3;
                return;
            }
        }
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (int t = 0; t < nt; ++t)
        if (status[t]) {  // a failed chunk leaves nothing in the window
            r->buf.n = base;
            while (!r->wblocks.empty() && r->wblocks.back().win_off >= (int64_t)base) r->wblocks.pop_back();
        }
    for (int t = 0; t < nt; ++t) {
        if (status[t] == 1) return bam_fail(r, TEC_EINVAL, "BGZF inflate failed (corrupt block)");
        if (status[t] == 2) return bam_fail(r, TEC_EINVAL, "corrupt BAM record size");
        if (status[t] == 3) return bam_fail(r, TEC_EINVAL, "corrupt BAM record layout");
    }
    r->buf_pos = baton[nt].pos;
    *got = baton[nt].count;
    return TEC_OK;
}

extern "C" int tec_bam_read(tec_bam* r, uint64_t max_records, int32_t* tid, int32_t* pos, int32_t* end, uint16_t* flag,
                            uint8_t* mapq, uint64_t* qkey, uint8_t* cb24, uint8_t* ub24, uint8_t* tagflags,
                            uint64_t* n_out) {
    if (!r || !n_out || !tid || !pos || !end || !flag || !mapq) return bam_fail(r, TEC_EINVAL, "tec_bam_read: NULL argument");
    if (tagflags && (!cb24 || !ub24)) return bam_fail(r, TEC_EINVAL, "tec_bam_read: tagflags without cb24/ub24");
    const OutArrays a{tid, pos, end, flag, mapq, qkey, cb24, ub24, tagflags};
    uint64_t o = 0;
    *n_out = 0;
    if (!r->is_bam) {
        while (o < max_records && sam_getline(r)) {
            if (r->line.empty()) continue;
            const int rc = sam_record(r, o, a);
            if (rc != TEC_OK) return rc;
            ++o;
        }
        *n_out = o;
        r->n_records += o;
        return TEC_OK;
    }
    // (A) complete records that are already in the window (a previous call stopped at max_records
    // in the middle of a chunk): sequential, at most one chunk's worth
    {
        size_t pos_b = r->buf_pos;
        const size_t have = std::min(r->buf.size(), own_limit(r));
        while (o < max_records && pos_b + 4 <= have) {
            const uint32_t bs = le32(r->buf.data() + pos_b);
            if (bs < 32) return bam_fail(r, TEC_EINVAL, "corrupt BAM record size");
            if (pos_b + 4 + (size_t)bs > have) break;
            if (!extract_record(r->buf.data() + pos_b, o, a)) return bam_fail(r, TEC_EINVAL, "corrupt BAM record layout");
            ++o;
            pos_b += 4 + (size_t)bs;
        }
        r->buf_pos = pos_b;
    }
    // (B) chunks of up to 512 BGZF blocks: inflate, boundary scan and field extraction fused per thread
    while (o < max_records && !r->eof && !r->owned_done) {
        uint64_t got = 0;
        const int rc = read_chunk(r, max_records - o, o, a, &got);
        if (rc != TEC_OK) return rc;
        o += got;
    }
    // (C) the owned blocks are exhausted: a record that STARTS in them but ends in a later range's
    // blocks is still ours -- complete it from the following blocks; then note where the next
    // range's first record starts (the end token, compared with that range's own guess)
    if ((r->eof || r->owned_done) && !r->finished && o < max_records) {
        r->in_epilogue = true;
        while (r->buf_pos < std::min(r->buf.size(), own_limit(r))) {  // else: the next record starts in a block that is not ours
            const size_t have = r->buf.size() - r->buf_pos;
            bool complete = false;
            if (have >= 4) {
                const uint32_t bs = le32(r->buf.data() + r->buf_pos);
                if (bs < 32) return bam_fail(r, TEC_EINVAL, "corrupt BAM record size");
                complete = have >= 4 + (size_t)bs;
            }
            if (complete) {
                if (!extract_record(r->buf.data() + r->buf_pos, o, a)) return bam_fail(r, TEC_EINVAL, "corrupt BAM record layout");
                r->buf_pos += 4 + (size_t)le32(r->buf.data() + r->buf_pos);
                ++o;
                break;
            }
            const int k = refill(r, 1);
            if (k < 0) return k;
            if (k == 0) return bam_fail(r, TEC_EINVAL, "truncated BAM record");
        }
        // end token: the block that holds window byte buf_pos (skipping empty blocks; end of file = {file size, 0})
        for (;;) {
            bool found = false;
            for (const auto& w : r->wblocks)
                if ((int64_t)r->buf_pos >= w.win_off && (int64_t)r->buf_pos < w.win_off + (int64_t)w.isize) {
                    r->tok_end[0] = w.file_off;
                    r->tok_end[1] = (uint64_t)((int64_t)r->buf_pos - w.win_off);
                    found = true;
                    break;
                }
            if (found) break;
            const int k = refill(r, 1);
            if (k < 0) return k;
            if (k == 0) {
                r->tok_end[0] = r->map ? (uint64_t)r->map_len : ~0ull;
                r->tok_end[1] = 0;
                break;
            }
        }
        r->finished = true;
    }
    *n_out = o;
    r->n_records += o;
    return TEC_OK;
}
