/*
 * TEST INFRASTRUCTURE ONLY -- the C form of the CPU oracle for the bulk path.
 *
 * A literal restatement of the reference's data structure and loop, so that it is an
 * independent check of libtecount's fine-bin / closed-form-candidate kernel:
 *   - orc_index_build: buckets[chrom][k] = feature indices touching 10 kb bucket k
 *     (te_count/miniglbase/genelist.py:367-378: range(left//bs*bs, (right+bs)//bs*bs, bs))
 *   - orc_bulk_count: the per-read loop of measureTE.parse_bamse / parse_bampe
 *     (te_count/te_count.py:201-263 / 76-152) on pre-decoded units.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Parity status: pinned -- checked against the golden vectors captured from the reference
 * (tests/test_host_cpu.py::test_c_oracle_golden), and against oracle/te_oracle.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BS 10000

typedef struct {
    uint32_t n_chrom, n_names;
    uint64_t n_features;
    int32_t *left, *right;
    uint32_t *name, *cls;
    uint64_t *chrom_base;  /* first bucket of each chromosome in off[] */
    int64_t *chrom_nb;     /* number of buckets per chromosome */
    uint64_t *off;         /* CSR offsets */
    uint32_t *idx;         /* feature indices, in feature order inside a bucket */
} orc_index;

static int64_t floordiv(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

void orc_index_free(orc_index *ix) {
    if (!ix) return;
    free(ix->left); free(ix->right); free(ix->name); free(ix->cls); free(ix->chrom_base);
    free(ix->chrom_nb); free(ix->off); free(ix->idx);
    free(ix);
}

orc_index *orc_index_build(uint64_t n, const int32_t *chrom, const int32_t *left, const int32_t *right,
                           const uint32_t *name, const uint32_t *cls, uint32_t n_chrom, uint32_t n_names) {
    orc_index *ix = calloc(1, sizeof *ix);
    ix->n_chrom = n_chrom; ix->n_names = n_names; ix->n_features = n;
    ix->left = malloc((n + 1) * 4); ix->right = malloc((n + 1) * 4);
    ix->name = malloc((n + 1) * 4); ix->cls = malloc((n + 1) * 4);
    memcpy(ix->left, left, n * 4); memcpy(ix->right, right, n * 4);
    memcpy(ix->name, name, n * 4); memcpy(ix->cls, cls, n * 4);
    ix->chrom_base = calloc(n_chrom + 1, 8); ix->chrom_nb = calloc(n_chrom + 1, 8);
    for (uint64_t i = 0; i < n; ++i) {  /* buckets run from left//bs to right//bs inclusive */
        int64_t hi = floordiv((int64_t)right[i] + BS, BS);  /* exclusive */
        if (hi > ix->chrom_nb[chrom[i]]) ix->chrom_nb[chrom[i]] = hi;
    }
    uint64_t total = 0;
    for (uint32_t c = 0; c < n_chrom; ++c) { ix->chrom_base[c] = total; total += (uint64_t)ix->chrom_nb[c]; }
    ix->off = calloc(total + 2, 8);
    for (uint64_t i = 0; i < n; ++i) {
        int64_t lo = floordiv(left[i], BS), hi = floordiv((int64_t)right[i] + BS, BS);
        for (int64_t b = lo < 0 ? 0 : lo; b < hi; ++b) ix->off[ix->chrom_base[chrom[i]] + b + 1]++;
    }
    for (uint64_t b = 0; b < total; ++b) ix->off[b + 1] += ix->off[b];
    ix->idx = malloc((ix->off[total] + 1) * 4);
    uint64_t *cur = malloc((total + 1) * 8);
    memcpy(cur, ix->off, total * 8);
    for (uint64_t i = 0; i < n; ++i) {
        int64_t lo = floordiv(left[i], BS), hi = floordiv((int64_t)right[i] + BS, BS);
        for (int64_t b = lo < 0 ? 0 : lo; b < hi; ++b) ix->idx[cur[ix->chrom_base[chrom[i]] + b]++] = (uint32_t)i;
    }
    free(cur);
    return ix;
}

#define COUNTABLE ((1u << 1) | (1u << 2) | (1u << 3)) /* gene, TE, snRNA: te_count.py:134-144 */
#define ENHANCER (1u << 4)

/* stats: [assigned, lowq, badchrom, qcfail].  Returns 0, or 1 if a unit hit only 'enhancer'
 * features (the reference raises NameError there, te_count.py:145-147). */
int orc_bulk_count(orc_index *ix, uint64_t n, const int32_t *chrom, const int32_t *p1, const int32_t *p2,
                   const uint16_t *flag, const uint8_t *mapq, int q, uint64_t *counts, uint64_t *stats) {
    uint32_t hitbuf_cap = 64, *hits = malloc(hitbuf_cap * 4);
    int crashed = 0;
    /* the "already seen in this read" marks are private to the call, so that several threads may count
     * disjoint unit ranges against one index (bench.py's all-cores baseline) */
    uint64_t *stamp = calloc((size_t)ix->n_names + 1, 8), *fstamp = calloc((size_t)ix->n_features + 1, 8), tick = 0;
    for (uint64_t u = 0; u < n; ++u) {
        if (flag[u] & 0x604) { stats[3]++; continue; }
        if ((int)mapq[u] < q) { stats[1]++; continue; }
        int32_t c = chrom[u];
        if (c < 0 || (uint32_t)c >= ix->n_chrom) { stats[2]++; continue; }
        int64_t a = p1[u], b = p2[u];
        int64_t bk[2] = { floordiv(a - 1, BS), floordiv(b + 1, BS) };
        ++tick;
        uint32_t nh = 0, classes = 0;
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && bk[1] == bk[0]) break;
            if (bk[k] < 0 || bk[k] >= ix->chrom_nb[c]) continue;
            uint64_t g = ix->chrom_base[c] + (uint64_t)bk[k];
            for (uint64_t j = ix->off[g]; j < ix->off[g + 1]; ++j) {
                uint32_t f = ix->idx[j];
                if (fstamp[f] == tick) continue;  /* loc_ids is a set */
                fstamp[f] = tick;
                int64_t gl = ix->left[f], gr = ix->right[f];
                if ((a >= gl && a + 1 <= gr) || (b - 1 >= gl && b <= gr)) {
                    if (nh == hitbuf_cap) { hitbuf_cap *= 2; hits = realloc(hits, hitbuf_cap * 4); }
                    hits[nh++] = f;
                    classes |= 1u << ix->cls[f];
                }
            }
        }
        if (!nh) continue;
        if (classes & COUNTABLE) {
            for (uint32_t h = 0; h < nh; ++h) {
                uint32_t nm = ix->name[hits[h]];
                if (stamp[nm] != tick) { stamp[nm] = tick; counts[nm]++; }
            }
        } else if (classes & ENHANCER) {
            crashed = 1;
            break;
        }
        stats[0]++;
    }
    free(hits);
    free(stamp);
    free(fstamp);
    return crashed;
}
