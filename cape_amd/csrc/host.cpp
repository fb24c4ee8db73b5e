// Host-only entry points of libcape_hip.so.
#include "cape_hip.h"

extern "C" int cape_abi_version(void) { return CAPE_ABI_VERSION; }

extern "C" int cape_csr_validate(int32_t rows, int32_t cols, int64_t nnz, const int32_t *rowptr, const int32_t *colidx) {
    if (rows < 0 || cols < 0 || nnz < 0 || !rowptr || (nnz > 0 && !colidx)) return CAPE_EINVAL;
    if (rowptr[0] != 0 || rowptr[rows] != nnz) return CAPE_ERANGE;
    for (int32_t r = 0; r < rows; ++r) {
        const int32_t a = rowptr[r], b = rowptr[r + 1];
        if (b < a || b > nnz) return CAPE_ERANGE;
        for (int32_t e = a; e < b; ++e) {
            if (colidx[e] < 0 || colidx[e] >= cols) return CAPE_ERANGE;
            if (e > a && colidx[e] <= colidx[e - 1]) return CAPE_EUNSORTED;
        }
    }
    return CAPE_OK;
}
