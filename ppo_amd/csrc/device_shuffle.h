// Device permutation generator: see device_shuffle.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ppoxp {
size_t device_shuffle_workspace_bytes(size_t M);
// perm_out[j] = index of the j-th smallest splitmix64(seed, i), ties by index.  Returns a hipError_t.
int device_shuffle(uint64_t seed, size_t M, uint32_t *perm_out, void *workspace, size_t workspace_bytes,
                   hipStream_t stream);
}  // namespace ppoxp
