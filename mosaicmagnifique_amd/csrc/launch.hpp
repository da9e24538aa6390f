// launch.hpp -- host-side launchers of the kernels that live in filter.hip (a
// separate translation unit because it is built with fp contraction enabled).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mm {

// A pixel pair the fp32 CIEDE2000 form must not judge (opposite-hue discontinuity).
struct EdgeEntry { int cell, cand, pixel; };

// Tier-1 filter scores f32 [n_cells][L] (see filter.hip).  Pairs on the discontinuity
// are logged to `edges` (capacity edge_cap, count in *edge_count, which may exceed
// the capacity: the caller must then not trust the scores) and patched with the
// fp64 form by a follow-up kernel on the same stream.
hipError_t launch_filter(int type, const float* cells, const uint8_t* cmask, const float* lib,
                         float* scores, int n_cells, int L, int ppad,
                         EdgeEntry* edges, unsigned int* edge_count, unsigned int edge_cap,
                         hipStream_t stream);

// fp32 filter form of the per-pair difference (accuracy harness).  flag_edges != 0
// writes -1 for pairs the filter hands to the fp64 form.
hipError_t launch_pair_fast(int type, const float* first, const float* second, size_t n,
                            float* out, int flag_edges, hipStream_t stream);

}  // namespace mm
