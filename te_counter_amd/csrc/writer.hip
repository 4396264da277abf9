// Dense cell x feature TSV writer: the step AFTER the single-cell path.
//
// Replaces the densifying loops of measureTE.sc_save_result (te_count/te_count.py:744-754): for
// every kept barcode one row of `n_names` integers, almost all of them 0 (1e4 cells x 4e4
// features = 4e8 numbers of text).  Host code; bytes identical to the reference's
// '{}\n'.format('\t'.join([barcode] + [str(c) for c in counts])).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "tecount_internal.h"

// header: the complete first line including '\n'.  labels: row barcodes, '\0'-separated, in row
// order.  row_cell[i]: the barcode id of row i; coo_*: (cell, name, count) sorted by (cell, name).
extern "C" int tec_write_dense_tsv(const char* path, const char* header, uint64_t header_len, uint32_t n_names,
                                   uint32_t n_rows, const char* labels, const uint32_t* row_cell, uint64_t nnz,
                                   const uint32_t* coo_cell, const uint32_t* coo_name, const uint64_t* coo_count) {
    if (!path || !header || (n_rows && (!labels || !row_cell)) || (nnz && (!coo_cell || !coo_name || !coo_count)))
        return tec_fail(nullptr, TEC_EINVAL, "tec_write_dense_tsv: NULL argument");
    FILE* fh = fopen(path, "wb");
    if (!fh) return tec_fail(nullptr, TEC_EINVAL, "tec_write_dense_tsv: cannot open %s", path);
    std::vector<char> iobuf(8 << 20);
    setvbuf(fh, iobuf.data(), _IOFBF, iobuf.size());
    bool ok = fwrite(header, 1, header_len, fh) == header_len;
    std::string zeros;  // "\t0" repeated: runs of empty columns are copied from here
    zeros.reserve(2 * (size_t)n_names);
    for (uint32_t i = 0; i < n_names; ++i) zeros += "\t0";
    std::string row;
    const char* lab = labels;
    for (uint32_t r = 0; r < n_rows && ok; ++r) {
        const size_t ll = strlen(lab);
        row.assign(lab, ll);
        lab += ll + 1;
        // this cell's triplets
        const uint32_t cell = row_cell[r];
        uint64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (coo_cell[mid] < cell) lo = mid + 1; else hi = mid;
        }
        uint32_t col = 0;
        char num[24];
        for (uint64_t k = lo; k < nnz && coo_cell[k] == cell; ++k) {
            const uint32_t c = coo_name[k];
            if (c >= n_names || c < col) {
                fclose(fh);
                return tec_fail(nullptr, TEC_EINVAL, "tec_write_dense_tsv: triplets not sorted / out of range");
            }
            row.append(zeros, 0, 2 * (size_t)(c - col));
            const int n = snprintf(num, sizeof num, "\t%llu", (unsigned long long)coo_count[k]);
            row.append(num, (size_t)n);
            col = c + 1;
        }
        row.append(zeros, 0, 2 * (size_t)(n_names - col));
        row.push_back('\n');
        ok = fwrite(row.data(), 1, row.size(), fh) == row.size();
    }
    if (fclose(fh) != 0) ok = false;
    return ok ? TEC_OK : tec_fail(nullptr, TEC_EINVAL, "tec_write_dense_tsv: write to %s failed", path);
}
