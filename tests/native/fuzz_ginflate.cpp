// Memory-safety fuzz of the lane decoder (te_counter_amd/csrc/ginflate.h, host instantiation: the code one GPU lane runs).
// Built by tests/test_inflate_cpu.py with g++ -fsanitize=address,undefined and run on the CPU.  Every buffer has its
// exact size (input + the kInPad readable bytes the decoder is promised, output, token array), and the per-lane memory
// checks every index against the sizes the device keeps in LDS / global scratch: damaged and random streams must end in
// an error code or in zlib's bytes, never outside a buffer.
//   fuzz_ginflate <iterations> <seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include <zlib.h>

#include "ginflate.h"

namespace {

[[noreturn]] void die(const char* what, uint32_t i) {
    fprintf(stderr, "fuzz_ginflate: %s (index %u)\n", what, i);
    abort();
}

struct CheckedMem {
    uint16_t t[tec_ginf::kLitEntries];
    uint16_t s[tec_ginf::kLongSymsPerLane];
    uint64_t lw[tec_ginf::kLenWords];
    uint64_t lenw(uint32_t k) const { if (k >= tec_ginf::kLenWords) die("lenw", k); return lw[k]; }
    void set_lenw(uint32_t k, uint64_t v) { if (k >= tec_ginf::kLenWords) die("set_lenw", k); lw[k] = v; }
    uint32_t tab(uint32_t i) const { if (i >= tec_ginf::kLitEntries) die("tab", i); return t[i]; }
    void set_tab(uint32_t i, uint32_t v) { if (i >= tec_ginf::kLitEntries) die("set_tab", i); t[i] = (uint16_t)v; }
    uint32_t lsym(uint32_t i) const { if (i >= tec_ginf::kLongSymsPerLane) die("lsym", i); return s[i]; }
    uint32_t lsym_near(uint32_t i) const { if (i >= tec_ginf::kLongLds) die("lsym_near", i); return s[i]; }
    void set_lsym(uint32_t i, uint32_t v) { if (i >= tec_ginf::kLongSymsPerLane) die("set_lsym", i); s[i] = (uint16_t)v; }
};

std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& data, int level, int strategy) {
    z_stream z{};
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) die("deflateInit2", (uint32_t)level);
    std::vector<uint8_t> out(2 * data.size() + 1024);  // (fixed codes spend nine bits on most literals: beyond deflateBound)
    z.next_in = const_cast<uint8_t*>(data.data());
    z.avail_in = (uInt)data.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("deflate", (uint32_t)data.size());
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// zlib's verdict: the inflated bytes if the stream is a complete raw DEFLATE stream of exactly n bytes
bool zlib_inflate(const std::vector<uint8_t>& comp, uint32_t n, std::vector<uint8_t>& out) {
    z_stream z{};
    if (inflateInit2(&z, -15) != Z_OK) die("inflateInit2", 0);
    out.assign(n + 1, 0);
    z.next_in = const_cast<uint8_t*>(comp.data());
    z.avail_in = (uInt)comp.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && z.total_out == n;
    inflateEnd(&z);
    out.resize(n);
    return ok;
}

}  // namespace

int main(int argc, char** argv) {
    const int iterations = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    auto below = [&](uint64_t n) { return (uint32_t)(rng() % n); };
    std::unique_ptr<CheckedMem> mem(new CheckedMem);
    int accepted = 0, agreed = 0;
    for (int it = 0; it < iterations; ++it) {
        // ---- a payload and its stream
        const uint32_t n = 1 + below(it % 7 == 0 ? 65536 : 6000);
        std::vector<uint8_t> data(n);
        const uint32_t alphabet = 1 + below(it % 5 == 0 ? 256 : 24), period = 1 + below(300);
        for (uint32_t i = 0; i < n; ++i) data[i] = (uint8_t)(below(4) ? below(alphabet) : (i >= period ? data[i - period] : 0));
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        std::vector<uint8_t> comp = deflate_raw(data, (int)below(10), strategies[below(5)]);
        uint32_t claimed = n;
        // ---- damage
        switch (it % 6) {
            case 0: break;  // intact
            case 1: comp[below(comp.size())] ^= (uint8_t)(1u << below(8)); break;
            case 2: comp.resize(below(comp.size())); break;
            case 3: for (int k = 0; k < 4 && !comp.empty(); ++k) comp[below(comp.size())] = (uint8_t)below(256); break;
            case 4: for (auto& b : comp) b = (uint8_t)below(256); break;  // noise
            case 5: claimed = below(2) ? n - below(n) : n + 1 + below(70000 - n); break;  // a wrong inflated size
        }
        // ---- exact-size buffers
        std::unique_ptr<uint8_t[]> in(new uint8_t[comp.size() + tec_ginf::kInPad]);
        memcpy(in.get(), comp.data(), comp.size());
        memset(in.get() + comp.size(), (int)below(256), tec_ginf::kInPad);
        std::unique_ptr<uint8_t[]> out(new uint8_t[claimed ? claimed : 1]);
        const uint32_t cap = claimed / 3 + claimed / 128 + 16;
        std::unique_ptr<uint32_t[]> tok(new uint32_t[cap]);
        memset(mem.get(), (int)below(256), sizeof(CheckedMem));
        tec_ginf::VectorSink sink{tok.get(), cap};
        const int rc = tec_ginf::inflate_stream(in.get(), (uint32_t)comp.size(), out.get(), claimed, *mem, sink);
        if (rc != tec_ginf::kOk) continue;
        if (!tec_ginf::resolve_tokens(tok.get(), sink.n, out.get(), claimed)) continue;
        ++accepted;
        std::vector<uint8_t> want;
        if (zlib_inflate(comp, claimed, want)) {
            if (memcmp(want.data(), out.get(), claimed) != 0) die("accepted a stream with other bytes than zlib's", (uint32_t)it);
            ++agreed;
        } else {
            die("accepted a stream that zlib refuses", (uint32_t)it);
        }
    }
    printf("fuzz_ginflate: %d streams, %d accepted, %d equal to zlib\n", iterations, accepted, agreed);
    return accepted > 0 ? 0 : 1;
}
