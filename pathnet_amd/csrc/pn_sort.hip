// pn_sort.hip -- the one place the library leans on a ROCm primitive: a stable LSD radix sort of (int32 key, int32 value)
// pairs (hipCUB / rocPRIM DeviceRadixSort), used by the deterministic backward of the aggregator (pn_pagg.hip,
// pn_set_deterministic): the gather backward's contributions are ordered by destination row, ties in the order of the
// path steps, and then summed in that order.  Its own translation unit because the rocPRIM headers take ten seconds
// to compile.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include "pn_internal.h"

namespace pn {

// Bytes of temporary storage reserved in the aggregator workspace for a sort of n pairs.  rocPRIM can only be asked on
// a machine with a device, and pn_pagg_workspace_bytes must work without one; so the reservation is a bound (double
// buffers for keys and values, per-block digit histograms, look-back state) and sort_pairs_i32 checks the real need
// against it at every call.
size_t sort_temp_reserve(int64_t n) { return (size_t)(n < 0 ? 0 : n) * 20 + ((size_t)8 << 20); }

int sort_pairs_i32(void *tmp, size_t tmp_bytes, const int32_t *keys_in, int32_t *keys_out, const int32_t *vals_in,
                   int32_t *vals_out, int64_t n, int key_bits, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return PN_OK;
    if (n > 2000000000LL) PN_FAIL(PN_ERR_ARG, "sort: %lld pairs exceed int32", (long long)n);
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 31) key_bits = 31;
    size_t need = 0;
    PN_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (int)n, 0,
                                                    key_bits, stream));
    if (need > tmp_bytes)
        PN_FAIL(PN_ERR_CAPACITY, "sort of %lld pairs needs %zu bytes of temporary storage, %zu reserved", (long long)n,
                need, tmp_bytes);
    PN_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, need, keys_in, keys_out, vals_in, vals_out, (int)n, 0, key_bits,
                                                    stream));
    return PN_OK;
}

}  // namespace pn
