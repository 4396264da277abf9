// Memory-safety fuzz of the BAM record parser shared by both decoders (te_counter_amd/csrc/bamscan.h; on the GPU one lane
// per record runs it).  Built by tests/test_inflate_cpu.py with -fsanitize=address,undefined and run on the CPU.  A record
// lives in a heap block of exactly 4 + block_size bytes, a stream view in a block of exactly `total` bytes: whatever the
// bytes say, the parser must stay inside.
//   fuzz_bamscan <iterations> <seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "bamscan.h"

int main(int argc, char** argv) {
    const int iterations = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    auto below = [&](uint64_t n) { return (uint32_t)(rng() % n); };
    auto put32 = [](uint8_t* p, uint32_t v) { memcpy(p, &v, 4); };
    uint64_t parsed = 0, tagged = 0, plausible = 0;
    for (int it = 0; it < iterations; ++it) {
        // ---- a record that is well formed to begin with
        const uint32_t l_name = 1 + below(it % 9 == 0 ? 255 : 40), n_cigar = below(it % 11 == 0 ? 400 : 6), l_seq = below(it % 13 == 0 ? 3000 : 160);
        std::vector<uint8_t> aux;
        const int n_tags = (int)below(8);
        for (int k = 0; k < n_tags; ++k) {
            static const char* names[] = {"CB", "CR", "UB", "UR", "NH", "CG", "XX", "ZZ"};
            const char* nm = names[below(8)];
            static const char types[] = "AcCsSiIfZHB?";
            const char ty = types[below(12)];
            aux.push_back((uint8_t)nm[0]);
            aux.push_back((uint8_t)nm[1]);
            aux.push_back((uint8_t)ty);
            uint32_t sz = 0;
            switch (ty) {
                case 'A': case 'c': case 'C': sz = 1; break;
                case 's': case 'S': sz = 2; break;
                case 'i': case 'I': case 'f': sz = 4; break;
                case 'Z': case 'H': {
                    const uint32_t n = below(below(4) ? 30 : 300);
                    for (uint32_t j = 0; j < n; ++j) aux.push_back((uint8_t)('A' + below(26)));
                    if (below(20)) aux.push_back(0);  // (one in twenty: no terminator)
                    break;
                }
                case 'B': {
                    static const char sub[] = "cCsSiIf?";
                    const char st = sub[below(8)];
                    const uint32_t n = below(50), es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                    aux.push_back((uint8_t)st);
                    uint8_t w[4];
                    put32(w, below(10) ? n : below(1u << 31));  // (one in ten: a count that lies)
                    aux.insert(aux.end(), w, w + 4);
                    sz = n * es;
                    break;
                }
                default: sz = below(6);
            }
            for (uint32_t j = 0; j < sz; ++j) aux.push_back((uint8_t)below(256));
        }
        const uint32_t fixed = 32 + l_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq;
        uint32_t bs = fixed + (uint32_t)aux.size();
        std::vector<uint8_t> rec(4 + bs, 0);
        put32(&rec[0], bs);
        put32(&rec[4], below(30) - 1);  // refID
        put32(&rec[8], below(1u << 28));
        rec[12] = (uint8_t)l_name;
        rec[13] = (uint8_t)below(256);
        rec[16] = (uint8_t)n_cigar;
        rec[17] = (uint8_t)(n_cigar >> 8);
        rec[18] = (uint8_t)below(256);
        rec[19] = (uint8_t)below(16);
        put32(&rec[20], l_seq);
        put32(&rec[24], below(30) - 1);
        put32(&rec[28], below(1u << 28));
        for (uint32_t j = 0; j + 1 < l_name; ++j) rec[36 + j] = (uint8_t)(below(12) ? 'a' + below(26) : '/');
        for (uint32_t c = 0; c < n_cigar; ++c) put32(&rec[36 + l_name + 4 * c], (below(300) << 4) | below(it % 7 == 0 ? 16 : 9));
        if (n_cigar && below(6) == 0) put32(&rec[36 + l_name], (l_seq << 4) | 4u);  // the CG placeholder
        if (!aux.empty()) memcpy(rec.data() + 4 + fixed, aux.data(), aux.size());
        // ---- damage: bytes, and the sizes the layout hangs on
        switch (it % 5) {
            case 0: break;
            case 1: for (int k = 0; k < 3; ++k) rec[4 + below(bs)] = (uint8_t)below(256); break;
            case 2: rec[12] = (uint8_t)below(256); break;
            case 3: put32(&rec[20], below(2) ? below(1u << 31) : below(400)); break;
            case 4: rec[16] = (uint8_t)below(256); rec[17] = (uint8_t)below(256); break;
        }
        // (block_size itself is what the callers trust to lie inside the stream: they bound it before calling)
        {
            std::unique_ptr<uint8_t[]> exact(new uint8_t[rec.size()]);
            memcpy(exact.get(), rec.data(), rec.size());
            tec_bamscan::Core c;
            if (bs >= 32 && tec_bamscan::parse_core(exact.get(), true, &c)) {
                ++parsed;
                if (c.rec_end != exact.get() + rec.size() || c.aux > c.rec_end) {
                    fprintf(stderr, "fuzz_bamscan: layout outside the record (iteration %d)\n", it);
                    abort();
                }
                uint8_t cb[24], ub[24];
                tagged += tec_bamscan::scan_tags(c.aux, c.rec_end, cb, ub) != 0;
                (void)tec_bamscan::mate_key_hash(c.name, c.l_name ? c.l_name - 1u : 0u);
            }
        }
        // ---- the record-start guesser over a view of exactly `total` bytes, at every kind of offset
        {
            const uint32_t lead = below(40), total = lead + (uint32_t)rec.size() - below(below(3) ? 1 : (uint32_t)rec.size());
            std::unique_ptr<uint8_t[]> view(new uint8_t[total ? total : 1]);
            for (uint32_t j = 0; j < total; ++j) view[j] = j < lead ? (uint8_t)below(256) : rec[j - lead];
            uint32_t ref_len[32];
            for (auto& r : ref_len) r = below(1u << 29);
            for (int k = 0; k < 6; ++k) {
                const uint64_t x = k == 0 ? lead : below(total + 1);
                if (x <= total) plausible += tec_bamscan::plausible_record(view.get(), total, x, 29, below(2) ? ref_len : nullptr, (int)below(3));
            }
        }
    }
    printf("fuzz_bamscan: %d records, %llu parsed, %llu with barcode tags, %llu plausible starts\n", iterations,
           (unsigned long long)parsed, (unsigned long long)tagged, (unsigned long long)plausible);
    return parsed > 0 && plausible > 0 ? 0 : 1;
}
