// Scalar-variable hot path (graph colouring / N-queens / job-shop machine assignment).
// Round-1 status: parameter block + placeholders; the fused scalar search kernel lands next.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_list_model.h"

namespace sf {

struct ScalarModel {
    int32_t n = 0;        // entities
    int32_t n_values = 0;
    int32_t allows_unassigned = 0;
    int32_t levels = 2;
    int32_t* vals = nullptr;       // [R][n]  (-1 = None)
    int64_t* score = nullptr;      // [R][4]
    int32_t* best_vals = nullptr;  // [R][n]
    int64_t* best_score = nullptr; // [R][4]
};

__global__ void k_scalar_evaluate_all(ScalarModel, int64_t*, int) {}
__global__ void k_scalar_evaluate_moves(ScalarModel, int, const int32_t*, int64_t, int64_t*, int32_t*) {}
__global__ void k_scalar_apply(ScalarModel, int, int, int, int, int, int32_t*) {}
__global__ void k_scalar_phase_start(ScalarModel, SearchParams) {}

}  // namespace sf
