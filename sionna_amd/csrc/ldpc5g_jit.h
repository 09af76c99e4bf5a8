// Specialised 5G LDPC decoders: the per-wave work lists of the explicit-message engine (ldpc5g_onchip_bp.hip builds
// them, ldpc5g_decode_msg_kernel walks them with scalar code) turned into straight-line source for ONE code and
// compiled at run time with hipRTC for gfx950 (ldpc5g_jit.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

struct samd_ldpc5g;

namespace samd {

struct JitItem {
  int idx;    // base row (check-node item) or base column (variable-node item)
  int q;      // first 64-lane chunk of lifted copies
  int nch;    // chunks covered (1 or 2)
  int prio;   // issue priority 0..3 (ldpc5g.h: item_priorities)
};

// Host-side description of the schedule (kept with the handle; nothing here lives on the device)
struct JitPlan {
  int z = 0, edges = 0, ncu = 0, nbu = 0;
  std::vector<int32_t> row_off;                                     // [ncu] byte offset of the row's first edge block
  std::vector<int> row_deg, fused_col;                              // [ncu]; fused degree-1 column or -1
  std::vector<std::vector<std::pair<int32_t, int32_t>>> col_edges;  // [nbu] (edge block byte offset, 4 shift), rows ascending
  std::vector<char> col_fused;                                      // [nbu]
  std::vector<std::vector<JitItem>> cn, vn;                         // [16 waves], in issue order
  // general = 1: built for any code by build_jit_plan_general (graph data only: the generator makes its own schedule)
  int general = 0;
  int prune_row = -1, prune_z0 = 0;   // base row whose lifted copies z >= prune_z0 the rate matching pruned (decoding.py:1344-1373)
};

// how the any-lifting-size programs map lifted copies onto lanes (jit/ldpc5g_jit_templates.h, JIT_GENERAL)
struct JitGeometry {
  int G = 1;        // codewords per workgroup (pairx: PAIRS of codewords)
  int pairx = 0;    // 1: a slot's two values are copy z of TWO codewords (2 g, 2 g + 1) - any Z, odd ones too, a rotation never
                    // swaps the pair (no selections in the variable-node phase); 0: copies (z, z + Z / 2) of ONE codeword
  int H = 0;        // the rotation's period in lanes: Z / 2 (a lane owns copies (z, z + H)), or Z with pairx
  int P = 0;        // G H lanes in use
  int chunks = 0;   // 64-lane chunks per edge block
  int blk = 0;      // bytes per edge block (512 per chunk)
  int nw = 16;      // waves per workgroup
  int wgs = 1;      // workgroups per CU
  // codes whose messages exceed LDS: the edge blocks of the base rows >= spill_row live in the workgroup's row of a
  // caller-owned workspace (L2); e_lds = edges whose blocks stay in LDS (all of them without spill)
  int spill_row = -1, e_lds = 0;
  size_t ws_bytes = 0;   // bytes of one workgroup's workspace row
};

// development knobs of the generator (SAMD_JIT_* options, read when the source is generated)
struct JitState;

struct JitKnobs {
  int pipe = 1;        // items whose loads are in flight (1: load - update - store per item)
  int xor128 = 0;      // Z = 128: second chunk's block position recomputed in the loop (v_xor) instead of a register
  int layout = 1;      // 1 (Z = 128 only): the two chunks of an edge block interleaved (8-byte DS instructions in both phases)
  int prefetch = 0;    // next codeword's channel LLRs requested one codeword ahead.  Round 5: on (+).  End of round 6: OFF - the
                       // values wait in ~40 registers through a whole decode (23 -> 4 spilled registers without them); C2 min-sum
                       // 15.58 -> 15.28 ms, boxplus-phi unchanged, the other codes of the sweep -0.5 ... +4.5 % (profiles/r06zy)
  int prio = 1;        // s_setprio per item
  int vnrev = -1;      // VN lists assigned to the waves in reverse order (-1: by generator - 0 for the levelled Z = 128 schedule, else 1)
  int sched = -1;      // -1: by layout (interleaved 1, planar 0).  0: the generic kernel's lists (tuned on hardware over rounds 2-3: cut items, SIMD-aware order);
                       // 1: an own longest-processing-time assignment by instruction counts - measured 4-7 % slower
                       // for every cost model tried (profiles/r05c_jit_sched_sweep.txt): an item's cost is its latency
                       // chain, not its instruction count
  int rotate = -1;     // a wave's item order rotated by its index on its SIMD (-1: by generator - 0 for the levelled Z = 128 schedule, else 1)
  int cn_slope = 10, cn_ovh = -1, cn_fused = 6, vn_slope = 8, vn_ovh = 10, vn_pair_max = -1;   // cost model (-1: by layout)
  int cn_pair_max = 32;   // rows of higher degree are two single-chunk items
  int waves = 0;          // own schedule (sched = 1): waves per workgroup, 0 = the generic kernel's 16
  int cmp_ahead = 0;      // min-sum check node: comparisons issued this many edges ahead of the selections that read them
  int phi_rolled = 1;     // boxplus-phi: the check-node loops over a row's edges rolled (one phi body per pass) instead of unrolled
  int general = 0;        // 1: the any-lifting-size programs also for the codes of the Z = 128 class (A/B)
  int group = 0, wgs = 0; // any-lifting-size programs: codewords per workgroup / workgroups per CU (0: chosen by the generator)
  int pairx = -1;         // -1: by the generator; 0 / 1: pairs inside a codeword / across two codewords
  int spill = 1;          // codes beyond LDS: the last base rows' blocks in an L2 workspace row (0: such codes keep the generic engines)
  int a1 = 1;             // plain min-sum: no clip of the smallest magnitude (it cannot exceed llr_max; JIT_A1_NOCLIP)
  int phi_tab0 = 1;       // boxplus-phi: table of the logarithm at LDS address 0 (no v_or per lookup)
  int phi_tab32 = 1;      // boxplus-phi: the table as two planes read with 4-byte loads (no register moves; JIT_PHI_TAB32)
  int phi_lean = 1;       // boxplus-phi: clamp as one v_med3 with |x| folded, sign of a v2c from its sign bit (a v2c is never -0)
  int simdbal = 1;        // Z = 128 class: items exchanged between waves of different SIMDs to level the per-SIMD instruction sums
  int state = 0;          // the variant that takes / returns the message image of a workgroup pass (return_state / msg_v2c; set by the caller)
  int vst32 = 0;          // Z = 128 class: variable-node results stored in node order by two 4-byte stores (JIT_VN_ST32, templates)
  int abl = 0;         // -DSAMD_DEV builds: SAMD_JIT_ABL (see jit/ldpc5g_jit_templates.h)
  void capture();
};

// true when the code is in the class the generator covers (see jit_eligible in ldpc5g_jit.cpp)
bool jit_eligible(const samd_ldpc5g* h);
// the whole translation unit handed to hipRTC.  with_ops = false: without the gfx950 operation definitions and the
// __global__ entry (what tests/jit_emu compiles for the CPU)
// rule: 0 offset-min-sum, 1 min-sum, 2 boxplus-phi on the defined exp / log
std::string jit_generate_source(const samd_ldpc5g* h, int return_infobits, int rule, bool with_ops, const JitKnobs& knobs);
// SAMD_OK, SAMD_ERR_UNSUPPORTED (caller runs the generic kernel) or an error
int launch_onchip_jit(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                      float llr_max, float offset, int hard_out, int return_infobits, void* workspace, size_t workspace_bytes,
                      void* stream, const float* state_in = nullptr, float* state_out = nullptr);
// workspace the generated kernel of this code needs for `batch` codewords (0: none - the messages fit LDS - or no such kernel)
size_t jit_workspace_bytes(const samd_ldpc5g* h, int batch, int cn_mode);
// graph data of the generator for a code that build_onchip_bp_tables left without a plan (any even lifting size)
void build_jit_plan_general(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row);
// 0: no generated kernel; 1: the Z = 128 k class (whole chunks of one kind, constants); 2: any-lifting-size programs
int jit_class(const samd_ldpc5g* h, const JitKnobs& kn, bool phi);
bool jit_geometry(const samd_ldpc5g* h, const JitKnobs& kn, bool phi, JitGeometry* g);
JitState* new_jit_state();
void free_jit(samd_ldpc5g* h);

}  // namespace samd
