// Tanner-graph and schedule handles of the generic BP engines (ldpc_bp_generic.hip: float32, ldpc_bp_f64.hip: float64).
#pragma once
#include "common.h"

#include <vector>

struct samd_ldpc_graph {
  int num_edges = 0, num_cn = 0, num_vn = 0;
  int max_dc = 0, max_dv = 0;
  int32_t* cn_ptr = nullptr;   // [num_cn+1]
  int32_t* cn_edge = nullptr;  // [E] VN-major edge id of the i-th edge of each CN
  int32_t* cn_vn = nullptr;    // [E] VN index of that edge
  int32_t* vn_ptr = nullptr;   // [num_vn+1] (edges of a VN are contiguous)
  std::vector<int32_t> h_cn_ptr, h_cn_vn;  // host copies for schedule construction
};

// CN schedule (decoding.py:252-270, 463-500): sub-iteration j updates the check nodes
// cn_list[j] and then the variable nodes adjacent to them (all other VN outputs are
// unchanged by construction, so recomputing them as the reference does is a no-op).
struct samd_ldpc_schedule {
  int num_sub = 0, width = 0, num_cn = 0;
  int32_t* cn_list = nullptr;            // device [num_sub][width]
  int32_t* vn_list = nullptr;            // device, concatenated per sub-iteration
  std::vector<int32_t> vn_off;           // host [num_sub+1]
  int32_t* first_mask = nullptr;         // device [num_cn]: 1 if CN is active in sub-iteration 0
};

