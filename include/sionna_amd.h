/*
 * sionna_amd.h - C-ABI of the MI355X (gfx950) hot-path library  libsionna_amd.so
 *
 * This is the drop-in boundary of the build: every entry point below replaces one
 * chain of TensorFlow ops of the reference (NVlabs/sionna v1.2.1, paths relative to
 * src/sionna/phy/).  The reference has NO native code and no FFI of its own
 * (SURVEY.md section 0, fact 1), so the functions are what a ctypes / pybind11 binding
 * inside the reference's Block.call() methods would bind.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no torch / TF types.
 *  - All data pointers are DEVICE pointers (HBM) unless the parameter is documented as
 *    "host".  The caller owns every buffer, including the workspace (query its size
 *    with the matching *_workspace_bytes call).  The library owns only opaque handles.
 *  - Tensors are batch-first, row-major, exactly like the reference API
 *    (bits = float32 0.0/1.0, LLRs = logits log p(1)/p(0), symbols = interleaved
 *    complex64).
 *  - Every launch is asynchronous on the caller's stream (hipStream_t passed as void*;
 *    NULL = default stream).  Functions are re-entrant per (handle, stream).
 *  - Return value: 0 = OK, negative = error (samd_last_error() gives a thread-local
 *    message).  Nothing throws across the ABI.
 */
#ifndef SIONNA_AMD_H
#define SIONNA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMD_OK 0
#define SAMD_ERR_INVALID (-1)
#define SAMD_ERR_HIP (-2)
#define SAMD_ERR_UNSUPPORTED (-3)
#define SAMD_ERR_WORKSPACE (-4)

/* check-node update rules of LDPCBPDecoder(cn_update=...)  fec/ldpc/decoding.py:295-312 */
#define SAMD_CN_BOXPLUS 0        /* cn_update_tanh          decoding.py:955-1043  */
#define SAMD_CN_BOXPLUS_PHI 1    /* cn_update_phi (default) decoding.py:1045-1166 */
#define SAMD_CN_MINSUM 2         /* cn_update_minsum        decoding.py:911-953   */
#define SAMD_CN_OFFSET_MINSUM 3  /* cn_update_offset_minsum decoding.py:755-909   */
/* SAMD_CN_BOXPLUS_PHI evaluates phi = log(e^x+1) - log(e^x-1) on a DEFINED float32 exp / log (a Cephes-style range
 * reduction + polynomial, one fixed sequence of IEEE operations: csrc/bp_math.h = oracle/ldpc_bp.c) - results are
 * bit-identical to that specification, which is NOT claimed to be TensorFlow's / Eigen's bits: against the reference's own
 * cn_update_phi executed on NumPy's float32 exp / log it agrees to 1e-5 (tests/test_oracle_ref_exec.py).  It is the
 * reference's DEFAULT rule and ~1.7x slower here than ..._PHI_FAST, the same rule on the GPU's transcendental unit
 * (v_exp_f32 / v_log_f32, ~1 ulp, unspecified last bits; soft outputs within 1e-5 of the defined form on
 * well-conditioned messages only, DESIGN.md "phi conditioning").  ..._PHI_FAST exists on the float32 explicit-message, state-passing, layered and
 * HBM-resident engines; the float64 decoder, the callback (torch) engine and the first on-chip boxplus engine
 * (csrc/ldpc5g_onchip_bp.hip, codes the grouped engine does not take) run the defined form under either name. */
#define SAMD_CN_BOXPLUS_PHI_FAST 4

const char* samd_last_error(void);
int samd_version(void);
/* number of visible HIP devices, or a negative error */
int samd_device_count(void);

/* Development switches (csrc/options.h).  The library never calls getenv on a compute path: the SAMD_* variables of
 * the process environment are copied into a registry once, when the library is loaded; this entry changes one value
 * afterwards (value NULL removes the key).  A switch affects handles CREATED after the call - a handle keeps the
 * options it was built with for its whole life, so handles built under different options can run concurrently from
 * several host threads - and the few handle-less entry points from their next call on.  Tests and the tools/ scripts
 * use it instead of setenv; a product integration never needs it.  Keys must start with "SAMD_". */
int samd_debug_set_option(const char* key, const char* value);
/* incremented by every samd_debug_set_option (host-side caches of handles key on it) */
int samd_debug_options_generation(void);
/* value of a switch: its length (copied NUL-terminated into buf when cap suffices), -1 when it is not set */
long samd_debug_get_option(const char* key, char* buf, size_t cap);

/* ------------------------------------------------------------------------------------
 * Generic LDPC flooding BP decoder (any parity-check matrix).
 * Replaces LDPCBPDecoder.call + _bp_iter      fec/ldpc/decoding.py:544-637, 416-524
 *          vn_update_sum                      fec/ldpc/decoding.py:681-732
 *          cn_update_{tanh,phi,minsum,offset_minsum}  decoding.py:755-1166
 * ---------------------------------------------------------------------------------- */
typedef struct samd_ldpc_graph samd_ldpc_graph_t;

/* Edge list in VN-major order (ascending VN, ascending CN inside a VN) - the edge order
 * of decoding.py:282-288 with a stable sort.  cn_idx / vn_idx: HOST int32[num_edges]. */
int samd_ldpc_graph_create(const int32_t* cn_idx, const int32_t* vn_idx, int num_edges,
                           int num_cn, int num_vn, samd_ldpc_graph_t** out);
void samd_ldpc_graph_destroy(samd_ldpc_graph_t* g);

size_t samd_ldpc_bp_workspace_bytes(const samd_ldpc_graph_t* g, int batch);

/* llr_in  [batch, num_vn] logits (clipped to +-llr_max inside, decoding.py:552-554)
 * out     [batch, out_cols]: columns 0..out_cols-1 of x_hat (out_cols <= num_vn);
 *         hard_out=1 -> 0/1 with "0 >= LLR_internal -> 1" (decoding.py:623), else logits
 * state   nullable [num_edges, batch] v2c messages in logit sign (IDD state,
 *         decoding.py:569-573, 636): read when state_in!=0, written when state_out!=0
 * offset  only used by SAMD_CN_OFFSET_MINSUM (reference default 0.5)                  */
int samd_ldpc_bp_decode_f32(const samd_ldpc_graph_t* g, const float* llr_in, float* out,
                            int out_cols, float* state, int state_in, int state_out,
                            int batch, int num_iter, int cn_mode, float llr_max,
                            float offset, int hard_out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Array CN schedule (`cn_schedule` argument of LDPCBPDecoder, decoding.py:252-270; "layered"
 * of LDPC5GDecoder, :1383-1389): sub-iteration j updates the check nodes cn_schedule[j][:]
 * (HOST int32 [num_sub][width], a node at most once per row) and then the variable nodes
 * adjacent to them, num_iter times over all rows (_bp_iter, :463-520). */
typedef struct samd_ldpc_schedule samd_ldpc_schedule_t;
int samd_ldpc_schedule_create(const samd_ldpc_graph_t* g, const int32_t* cn_schedule, int num_sub,
                              int width, samd_ldpc_schedule_t** out);
void samd_ldpc_schedule_destroy(samd_ldpc_schedule_t* s);

/* samd_ldpc_bp_decode_f32 under a schedule; same arguments, workspace and state semantics
 * (state_in: the given msg_v2c is what the first sub-iteration's check nodes read, all c2v
 * start at 0 - exactly the reference's behaviour). */
int samd_ldpc_bp_decode_scheduled_f32(const samd_ldpc_graph_t* g, const samd_ldpc_schedule_t* sched,
                                      const float* llr_in, float* out, int out_cols, float* state,
                                      int state_in, int state_out, int batch, int num_iter,
                                      int cn_mode, float llr_max, float offset, int hard_out,
                                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * 5G-NR LDPC (quasi-cyclic) code handle: encoder, rate matching / recovery, decoder.
 * Replaces LDPC5GEncoder.__init__/call   fec/ldpc/encoding.py:61-137, 599-668
 *          LDPC5GDecoder.__init__/call   fec/ldpc/decoding.py:1302-1403, 1427-1536
 * The handle stores the base-graph entries (38.212 Tab. 5.3.2-2/-3 for the chosen
 * i_LS) - rows/cols/shifts: HOST int16[num_entries], raw shift values (mod Z applied
 * inside, encoding.py:343-346).  num_bits_per_symbol = 0 disables the 38.212 5.4.2.2
 * output interleaver (encoding.py:196-246).  nb_pruned = pruned trailing rows/columns
 * of the PCM used by the DECODER (decoding.py:1344-1373; 0 = unpruned).
 * ---------------------------------------------------------------------------------- */
typedef struct samd_ldpc5g samd_ldpc5g_t;

int samd_ldpc5g_create(int bg, int z, const int16_t* rows, const int16_t* cols,
                       const int16_t* shifts, int num_entries, int k, int n,
                       int num_bits_per_symbol, int nb_pruned, samd_ldpc5g_t** out);
void samd_ldpc5g_destroy(samd_ldpc5g_t* h);

/* bits [batch,k] float32 0/1 -> out [batch,n] float32 0/1 (encode + rate-match +
 * interleave), bit-exact with u*G mod 2 of the reference's golden matrices. */
int samd_ldpc5g_encode_f32(const samd_ldpc5g_t* h, const float* bits, float* out,
                           int batch, void* stream);

/* Rate recovery only (decoding.py:1431-1475): llr [batch,n] -> out [batch,N_vn] logits,
 * N_vn = n_ldpc - nb_pruned; filler positions = -llr_max, punctured = 0. */
int samd_ldpc5g_rate_recover_f32(const samd_ldpc5g_t* h, const float* llr, float* out,
                                 int batch, float llr_max, void* stream);

/* Decoder output mapping (decoding.py:1486-1536): x_hat [batch,N_vn] -> out [batch,n]
 * (drop filler, drop first 2Z, keep n, re-interleave). */
int samd_ldpc5g_extract_codeword_f32(const samd_ldpc5g_t* h, const float* x_hat,
                                     float* out, int batch, void* stream);

size_t samd_ldpc5g_decode_workspace_bytes(const samd_ldpc5g_t* h, int batch, int cn_mode);
/* Which engine samd_ldpc5g_decode_f32 runs for this code and rule: 0 none (SAMD_ERR_UNSUPPORTED: use the
 * HBM-resident samd_ldpc_bp_decode_f32), 1 on chip with compressed check-node state (min-sum family, every
 * code), 2 on chip with one float per edge in LDS (all rules; codes whose messages fit in 160 KB), 3 the
 * same with the messages of the last base rows in the L2 workspace row (larger codes: boxplus rules always,
 * min-sum while at most ~27 % of the edges spill - beyond that engine 1 is faster). */
int samd_ldpc5g_decode_engine(const samd_ldpc5g_t* h, int cn_mode);

/* LDPC5GDecoder(cn_schedule="layered").call on chip (decoding.py:1383-1389: one sub-iteration per base row; _bp_iter
 * with an array schedule :463-520): rate recovery + num_iter layered iterations + output mapping in ONE kernel, the
 * sent check-node messages and the variable-node totals resident in LDS, channel LLRs and the state of the fused
 * degree-1 columns in `workspace` (samd_ldpc5g_decode_layered_workspace_bytes).  min-sum / offset-min-sum / boxplus-phi,
 * codes of any lifting size without a partially pruned base row whose state fits in LDS (config C2 does);
 * ..._supported() tells - the
 * scheduled HBM-resident engine (samd_ldpc_bp_decode_scheduled_f32) takes everything else.  Same bits as that engine and
 * as the oracle's literal "update every variable node after every layer". */
int samd_ldpc5g_decode_layered_supported(const samd_ldpc5g_t* h, int cn_mode);
size_t samd_ldpc5g_decode_layered_workspace_bytes(const samd_ldpc5g_t* h, int batch);
int samd_ldpc5g_decode_layered_f32(const samd_ldpc5g_t* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                                   float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* Whole LDPC5GDecoder.call on chip: rate recovery + num_iter flooding BP iterations +
 * output mapping in ONE kernel, one codeword per workgroup, messages resident in LDS.
 *   One float per edge in LDS (engine 2) when the code's E messages x 4 B fit in 160 KB - n=8448 rate 1/3
 *     and smaller; channel LLRs in `workspace` (one L2-resident row per workgroup) when they do not fit
 *     beside the messages.  All four check-node rules; boxplus / boxplus-phi use the arithmetic and summation
 *     order of samd_ldpc_bp_decode_f32 (identical bits), min-sum is bit-exact.
 *   SAMD_CN_MINSUM / SAMD_CN_OFFSET_MINSUM on larger codes (engine 1): compressed check-node state, every 5G
 *     code; codes whose state exceeds 160 KB keep the channel LLRs, then the VN totals, then the sign words
 *     in `workspace`; same results.
 * Returns SAMD_ERR_UNSUPPORTED when the code cannot run on chip (caller then uses rate_recover +
 * samd_ldpc_bp_decode_f32 + extract).  samd_ldpc5g_decode_workspace_bytes(h, batch, cn_mode) sizes `workspace`.
 * llr [batch,n] logits -> out [batch,k] (return_infobits=1) or [batch,n] (=0). */
int samd_ldpc5g_decode_f32(const samd_ldpc5g_t* h, const float* llr, float* out,
                           int batch, int num_iter, int cn_mode, float llr_max,
                           float offset, int hard_out, int return_infobits,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Specialised decoders (csrc/ldpc5g_jit.cpp).  For codes whose messages fit in LDS (round 5: lifting size a multiple of 128,
 * whole base rows / columns, k, n and the interleaver's row length multiples of 64 - BASELINE config C2; round 6: every even
 * lifting size, any k / n / pruning - BASELINE C4's and C1's codes), samd_ldpc5g_decode_f32 with SAMD_CN_MINSUM /
 * SAMD_CN_OFFSET_MINSUM runs a kernel GENERATED for that one code:
 * the per-wave work lists written out as straight-line source (block offsets, shifts and rate-matching offsets as
 * constants), compiled once per machine and code with hipRTC (libhiprtc.so, bound with dlopen) for gfx950.  Same
 * reference path (decoding.py:1427-1536, 416-524, 681-953), same bits as the generic kernel.  One kernel per (code,
 * return_infobits, rule): cn_mode below is SAMD_CN_MINSUM or SAMD_CN_OFFSET_MINSUM.  Compilation happens at the
 * first decode of at least 256 codewords (development options SAMD_LDPC_JIT = 0 off / 1 default / 2 any batch,
 * SAMD_LDPC_JIT_MIN_BATCH); whenever it is not possible the generic kernel runs.
 *   ..._supported: 1 when the handle's code is in that class.
 *   ..._prepare:   compile now; 1 ready, 0 not in the class / switched off, < 0 failed (samd_last_error = compiler log).
 *   ..._source:    the generated translation unit (with_ops = 0: without the gfx950 operation definitions and the kernel
 *                  entry - what tests/jit_emu runs on the CPU); returns its length, copies at most cap - 1 bytes.
 *   ..._code:      the compiled code object (for llvm-objdump); returns its size, copies when cap suffices.
 *   ..._launches:  how many decode calls of this handle ran on a specialised kernel so far.
 *   ..._cache_stats: process-wide counters of the code-object cache, what[0] = hipRTC compilations, what[1] = code objects
 *                  read from disk (<cache dir>/<sha256(source | hipRTC version | target)>.co; cache dir = option
 *                  SAMD_JIT_CACHE_DIR, else $XDG_CACHE_HOME/sionna_amd, else $HOME/.cache/sionna_amd; SAMD_JIT_CACHE=0: off).
 * Round 6: the class is every lifting size whose messages fit LDS (an 8-byte slot holds copy z of two codewords, or the copies
 * (z, z + Z/2) of one; several codewords per workgroup) and the codes with up to 30 % of their messages beyond it (the last base
 * rows' blocks in the caller's workspace: samd_ldpc5g_decode_workspace_bytes); the reference decodes every (k, n) through one path (encoding.py:248-282, decoding.py:1302-1403).
 * ..._source works on a handle created under the development option SAMD_HOST_ONLY=1 (no device needed; such a handle
 * builds tables and schedules only and refuses every launch). */
int samd_ldpc5g_jit_supported(const samd_ldpc5g_t* h);
int samd_ldpc5g_jit_prepare(const samd_ldpc5g_t* h, int return_infobits, int cn_mode);
long samd_ldpc5g_jit_source(const samd_ldpc5g_t* h, int return_infobits, int cn_mode, int with_ops, char* buf, size_t cap);
long samd_ldpc5g_jit_code(const samd_ldpc5g_t* h, int return_infobits, int cn_mode, char* buf, size_t cap);
long samd_ldpc5g_jit_launches(const samd_ldpc5g_t* h);
int samd_ldpc5g_jit_cache_stats(long* what);

/* LDPC5GDecoder.call with return_state=True / msg_v2c (fec/ldpc/decoding.py:573, 636-637; 5G wrapper :1427-1536) on the
 * generated kernels (round 6).  The kernel moves the message image of one workgroup pass - the LDS copy of the edge blocks of
 * the codeword(s) a workgroup decodes side by side - to and from image buffers, DEVICE float[ceil(batch / cw_per_pass)][img_floats]:
 * samd_ldpc5g_state_layout gives the two numbers (SAMD_ERR_UNSUPPORTED: the code's messages exceed LDS - use
 * samd_ldpc_bp_decode_f32 with its [num_edges, batch] state); samd_ldpc5g_state_map fills three HOST int32[img_floats] arrays:
 * for every float of the image the codeword of the pass, the check node and the variable node of the lifted (pruned) graph it
 * belongs to, or -1 (padding, pruned check nodes); the host sorts them into a DEVICE table int32[img_floats] =
 * codeword << 24 | edge index in the reference's order (edges sorted by variable node, then check node), or -1, and
 * samd_ldpc5g_state_convert_f32 moves image <-> the reference's state [num_edges, batch] (logit sign) with it.
 * samd_ldpc5g_decode_state_f32: image_in == NULL starts from the channel values, image_out == NULL returns no state;
 * llr / out as samd_ldpc5g_decode_f32. */
int samd_ldpc5g_state_layout(const samd_ldpc5g_t* h, int cn_mode, int* img_floats, int* cw_per_pass);
int samd_ldpc5g_state_map(const samd_ldpc5g_t* h, int cn_mode, int32_t* cw, int32_t* cn, int32_t* vn);
int samd_ldpc5g_state_convert_f32(const int32_t* table, int img_floats, int cw_per_pass, long batch, float* image, float* canonical,
                                  int to_canonical, void* stream);
int samd_ldpc5g_decode_state_f32(const samd_ldpc5g_t* h, const float* llr, float* out, const float* image_in, float* image_out,
                                 int batch, int num_iter, int cn_mode, float llr_max, float offset, int hard_out,
                                 int return_infobits, void* stream);

/* ------------------------------------------------------------------------------------
 * Mapping.  points: DEVICE complex64[2^m] (interleaved re,im), label of point i = binary
 * representation of i, MSB first (mapping.py:486-514).
 * ---------------------------------------------------------------------------------- */
/* Mapper.call mapping.py:497-519: bits [num_symbols*m] -> symbols [num_symbols] */
int samd_qam_map_c64(const float* bits, const float* points, int m, int64_t num_symbols,
                     float* out_symbols, void* stream);

/* Demapper.call + SymbolLogits2LLRs.call (no prior)  mapping.py:664-691, 927-967.
 * y [num_symbols] complex64; no: scalar (no_len=1, DEVICE pointer) or per symbol
 * (no_len=num_symbols); method 0 = "app" (logsumexp), 1 = "maxlog";
 * out [num_symbols*m] logits (or hard decisions llr>0 when hard_out, misc.py:270). */
int samd_qam_demap_f32(const float* y, const float* no, int64_t no_len, const float* points,
                       int m, int64_t num_symbols, int method, int hard_out, float* out,
                       void* stream);

/* SymbolDemapper.call  mapping.py:693-792: out [num_symbols, 2^m] = log_softmax over the constellation
 * points of -|y - c|^2 / no (+ prior log-probabilities, [2^m] or [num_symbols, 2^m], or NULL); with hard_out
 * instead out_idx [num_symbols] int32 = index of the most likely point (first maximum). */
int samd_symbol_demap_f32(const float* y, const float* no, int64_t no_len, const float* points, int m,
                          int64_t num_symbols, const float* prior, int64_t prior_len, int hard_out,
                          float* out, int32_t* out_idx, void* stream);

/* SymbolLogits2LLRs.call  mapping.py:794-967: logits [rows, 2^m] on the constellation points (labels = point index, MSB
 * first) -> out [rows, m] LLRs: reduce over the points whose bit i is 1 minus reduce over those whose bit is 0, reduce =
 * logsumexp (method 0, "app") or max (method 1, "maxlog"); prior NULL or LLRs [m] / [rows, m] whose term
 * sum_i log_sigmoid(+-prior_i) is added to every point's logit; hard_out: decisions llr > 0.  m in 1..8. */
int samd_symbol_logits2llrs_f32(const float* logits, int m, int64_t rows, const float* prior, int64_t prior_len,
                                int method, int hard_out, float* out, void* stream);

/* LLRs2SymbolLogits.call  mapping.py:969-1058: llrs [rows, m] -> out [rows, 2^m], logit of point c =
 * sum_j log_sigmoid(+-llr_j) (+ where bit j of c's label - the binary representation of c, MSB first - is 1);
 * hard_out: out_idx [rows] int32 = first maximum of the row (tf.argmax), out may be NULL.  m in 1..8. */
int samd_llrs2symbol_logits_f32(const float* llrs, int m, int64_t rows, int hard_out, float* out,
                                int32_t* out_idx, void* stream);

/* SymbolLogits2Moments.call  mapping.py:1061-1138: logits [rows, 2^m], points DEVICE complex64[2^m] ->
 * mean complex64 [rows] = sum_c softmax(logits)_c x_c, var float [rows] = sum_c softmax_c |x_c - mean|^2. */
int samd_symbol_logits2moments_c64(const float* logits, const float* points, int m, int64_t rows,
                                   float* mean, float* var, void* stream);

/* PAM2QAM.__call__ on logits (hard_in_out=False)  mapping.py:1234-1314: pam1, pam2 [rows, 2^(nb/2)] logits of
 * the real / imaginary PAM constellation -> out [rows, 2^nb]: the matrix pam1_i + pam2_j flattened and gathered
 * with the table of interleaved labels, out[i P + j] = (pam1 (+) pam2)[t(i, j)] - the reference's expression,
 * literally (see csrc/mapping.hip).  nb = num_bits_per_symbol of the QAM constellation, even, 2..10. */
int samd_pam2qam_logits_f32(const float* pam1, const float* pam2, int num_bits_per_symbol, int64_t rows,
                            float* out, void* stream);

/* Demapper.call with prior knowledge on the bits (mapping.py:664-691, 927-967): as samd_qam_demap_f32 with
 * the a-priori term sum_i log_sigmoid(+-prior_i) added to the exponent of every point.  prior DEVICE
 * float LLRs, [m] (shared by all symbols) or [num_symbols, m]. */
int samd_qam_demap_prior_f32(const float* y, const float* no, int64_t no_len, const float* points, int m,
                             int64_t num_symbols, const float* prior, int64_t prior_len, int method,
                             int hard_out, float* out, void* stream);

/* Same contract for a SQUARE QAM constellation whose label interleaves the bits of two
 * identical PAM axes (qam() of mapping.py:44-118: even label bits -> real axis, odd ->
 * imaginary): the per-bit sums factorise per axis, so only the 2^(m/2) PAM levels are needed.
 * levels: DEVICE float[2^(m/2)], level of the PAM label j (MSB first), already normalised. */
int samd_square_qam_demap_f32(const float* y, const float* no, int64_t no_len, const float* levels,
                              int m, int64_t num_symbols, int method, int hard_out, float* out,
                              void* stream);

/* ------------------------------------------------------------------------------------
 * Random sources and AWGN.  Counter-based Philox4x32-10 stream keyed by (seed, call);
 * executable specification: oracle/utils.py.
 * ---------------------------------------------------------------------------------- */
/* BinarySource.call mapping.py:1350-1352: out[n] float32 in {0,1} */
int samd_binary_source_f32(uint64_t seed, uint64_t call, int64_t n, float* out, void* stream);

/* AWGN.call channel/awgn.py:63-78 + complex_normal utils/misc.py:19-54:
 * y = x + sqrt(no) * CN(0,1); x,y complex64[n]; no DEVICE scalar (no_len=1) or [n]. */
int samd_awgn_c64(const float* x, const float* no, int64_t no_len, uint64_t seed,
                  uint64_t call, int64_t n, float* y, void* stream);

/* ------------------------------------------------------------------------------------
 * OFDM resource grid, frequency-domain channel, LS estimation, LMMSE equalisation
 * (config C4 of the north star).  S below = num_tx * num_streams_per_tx, RE index = t*F + f.
 * ---------------------------------------------------------------------------------- */
/* ResourceGridMapper.call  ofdm/resource_grid.py:394-412.
 * x [batch,S,num_data]; pilots DEVICE [S,num_pilots]; data_pos / pilot_pos DEVICE int32
 * [S,num_re]: index of the data / pilot symbol carried by that RE of the FULL grid
 * (num_re = num_ofdm_symbols*fft_size) or -1; out [batch,S,num_re]. */
int samd_rg_map_c64(const float* x, const float* pilots, const int32_t* data_pos,
                    const int32_t* pilot_pos, int batch, int num_streams, int num_re,
                    int num_data, int num_pilots, float* out, void* stream);

/* Generic index gather out[b,g,j] = in[b, src_group[g], idx[g,j]] on float32 (floats_per_elem
 * = 1), complex64 / float64 (= 2) or complex128 (= 4) elements.  Implements RemoveNulledSubcarriers.call
 * (ofdm/resource_grid.py:551-552) and ResourceGridDemapper.call (:466-520). */
int samd_gather3(const float* in, const int32_t* src_group, const int32_t* idx, int batch,
                 int groups_in, int n_in, int groups_out, int n_out, int floats_per_elem,
                 float* out, void* stream);

/* TDL.__call__  channel/tr38901/tdl.py:372-470 (no spatial correlation): sum-of-sinusoids
 * taps a [batch,1,num_rx_ant,1,num_tx_ant,num_paths,num_time_steps] complex64 on the Philox
 * stream (seed, call .. call+3) - layout: oracle/ofdm.py::tdl_cir.  mean_powers DEVICE
 * float[num_paths] (linear, normalised); los=1 adds the specular term to path 0. */
int samd_tdl_cir_c64(uint64_t seed, uint64_t call, int batch, int num_rx_ant, int num_tx_ant,
                     int num_paths, int num_time_steps, int num_sinusoids,
                     float sampling_frequency, const float* mean_powers, float min_doppler,
                     float max_doppler, int los, float los_power, float los_aoa, float* a,
                     void* stream);

/* CDL.__call__  channel/tr38901/cdl.py:258-333 + ChannelCoefficientsGenerator (TR 38.901 Sec. 7.5
 * steps 10-11, no sub-clustering)  channel/tr38901/channel_coefficients.py:173-194, 459-1031.
 * Model / array / orientation dependent factors are tabulated by the host for the 20 x 20 (zenith
 * ray, azimuth ray) pairs of every cluster (index ((n*20 + zenith)*20 + azimuth)); all DEVICE:
 *   f_rx, f_tx  float [N][20][20][2 pol][2]   GCS field (F_theta, F_phi) per polarisation
 *   a_rx / a_tx complex64 [N][20][20][ant]    array responses exp(j 2 pi r.d / lambda)
 *   r_rx        float [N][20][20][3]          arrival unit vectors (Doppler)
 *   pol_rx/tx   int32 [ant] polarisation index; order int32 [N] cluster of output tap n (ascending
 *   delay); amp float [N] sqrt(P_n/20) (x sqrt(1/(K+1)) with a LoS path)
 *   los         nullable float [8 + 2U + 2S + 4]: f_rx[2][2], f_tx[2][2], a_rx[U] c64, a_tx[S] c64,
 *               r_rx[3], sqrt(K/(K+1)) of the specular path added to tap 0
 * Random part on the Philox stream (seed, call .. call+7), layout: oracle/cdl.py.
 * -> a [batch, num_rx_ant, num_tx_ant, N, num_time_steps] complex64. */
size_t samd_cdl_workspace_bytes(int batch, int num_clusters);
int samd_cdl_cir_c64(uint64_t seed, uint64_t call, int batch, int num_clusters, int num_rx_ant,
                     int num_tx_ant, int num_time_steps, float sampling_frequency, const float* f_rx,
                     const float* f_tx, const float* a_rx, const float* a_tx, const float* r_rx,
                     const int32_t* pol_rx, const int32_t* pol_tx, const int32_t* order,
                     const float* amp, const float* los, float xpr_scale, float two_pi_over_lambda,
                     float min_speed, float max_speed, void* workspace, size_t workspace_bytes,
                     float* a, void* stream);
/* the same on float64 tables (precision = "double"): every float argument double, a [..] complex128; the random part uses the
 * float32 stream's 24-bit uniforms (exact in double), so a double run sees the float32 run's realisation */
int samd_cdl_cir_c128(uint64_t seed, uint64_t call, int batch, int num_clusters, int num_rx_ant,
                      int num_tx_ant, int num_time_steps, double sampling_frequency, const double* f_rx,
                      const double* f_tx, const double* a_rx, const double* a_tx, const double* r_rx,
                      const int32_t* pol_rx, const int32_t* pol_tx, const int32_t* order,
                      const double* amp, const double* los, double xpr_scale, double two_pi_over_lambda,
                      double min_speed, double max_speed, void* workspace, size_t workspace_bytes,
                      double* a, void* stream);

/* cir_to_ofdm_channel  channel/utils.py:180-253.  a [B,rx,ra,tx,ta,P,T], tau [B,rx,tx,P],
 * frequencies DEVICE float[F] -> h_freq [B,rx,ra,tx,ta,T,F]; normalize: unit mean energy over
 * (ra,ta,T,F) per (b,rx,tx). */
int samd_cir_to_ofdm_c64(const float* a, const float* tau, const float* frequencies, int batch,
                         int num_rx, int num_rx_ant, int num_tx, int num_tx_ant, int num_paths,
                         int num_time_steps, int num_freqs, int normalize, float* h_freq,
                         void* stream);

/* OFDMChannel.call  channel/ofdm_channel.py:109-115 = cir_to_ofdm_channel (channel/utils.py:180-253) + ApplyOFDMChannel
 * (channel/apply_ofdm_channel.py:70-80) + AWGN (channel/awgn.py:63-78) in ONE launch, for the case that h_freq itself is not
 * read (return_channel = False, or the returned tensor never used): a [B,rx,ra,1,ta,P,T], tau [B,rx,1,P], x [B,1,ta,T,F]
 * -> y [B,rx,ra,T,F]; no: DEVICE float[1] or NULL (no noise); (seed, call): the stream samd_awgn_c64 would be given.  The same
 * bits as the three separate entries; the link's frequency response never reaches memory.  SAMD_ERR_UNSUPPORTED (several
 * transmitters, shapes outside the staged-register kernel): run the separate entries. */
int samd_ofdm_channel_fused_c64(const float* a, const float* tau, const float* frequencies, const float* x, const float* no,
                                uint64_t seed, uint64_t call, int batch, int num_rx, int num_rx_ant, int num_tx,
                                int num_tx_ant, int num_paths, int num_time_steps, int num_freqs, int normalize, float* y,
                                void* stream);

/* ApplyOFDMChannel.call (noise-free part)  channel/apply_ofdm_channel.py:70-80.
 * x [B, num_tx*num_tx_ant, num_re], h_freq [B, num_rx*num_rx_ant, num_tx*num_tx_ant, num_re]
 * -> y [B, num_rx*num_rx_ant, num_re]. */
int samd_apply_ofdm_channel_c64(const float* x, const float* h_freq, int batch, int num_rx_x_ant,
                                int num_tx_x_ant, int num_re, float* y, void* stream);

/* MMSEPICDetector.call  mimo/detection.py:1496-1643, output="bit" ([CST2011] MMSE with parallel
 * interference cancellation and num_iter self-iterations) on n independent problems:
 * y [n,m], h [n,m,k], s [n,m,m] complex64, prior [n,k,num_bits_per_symbol] a-priori LLRs, points
 * DEVICE complex64[2^num_bits_per_symbol] -> out [n,k,num_bits_per_symbol] extrinsic LLRs (hard_out:
 * their hard decisions).  maxlog 0 = "app".  Same (m,k) support as samd_lmmse_equalizer_c64. */
int samd_mmse_pic_f32(const float* y, const float* h, const float* s, const float* prior,
                      const float* points, int64_t n, int m, int k, int num_bits_per_symbol,
                      int maxlog, int num_iter, int hard_out, float* out, void* stream);

/* ofdm.MMSEPICDetector.call  ofdm/detection.py:1062-1173 (OFDMDetectorWithPrior :320-560): the
 * pre-processing of samd_ofdm_lmmse_c64 + the detector above in one launch.  prior / out
 * [batch, num_streams_total, num_data * num_bits_per_symbol]. */
int samd_ofdm_mmse_pic_f32(const float* y, const float* h_hat, const float* err_var, int ev_mode,
                           const float* no, const float* prior, const float* points,
                           const int32_t* sc_ind, const int32_t* desired, const int32_t* undesired,
                           const int32_t* data_pos, int batch, int num_rx, int num_rx_ant,
                           int num_streams_total, int streams_per_rx, int num_undesired,
                           int num_ofdm_symbols, int num_eff_subcarriers, int fft_size, int num_data,
                           int num_bits_per_symbol, int maxlog, int num_iter, int hard_out, float* out,
                           void* stream);

/* LinearEncoder.call  fec/linear/encoding.py:122-140: c = (u G) mod 2 for a binary generator
 * matrix.  gm_cols DEVICE uint32 [n][ceil(k/32)]: bit (i & 31) of word (i >> 5) of row j = G[i][j]
 * (the columns of G, packed).  u [batch,k] float 0/1 -> out [batch,n] float 0/1. */
int samd_gf2_encode_f32(const float* u, const uint32_t* gm_cols, int64_t batch, int k, int n,
                        float* out, void* stream);

/* EPDetector.call  mimo/detection.py:1229-1312 (expectation propagation, l iterations, damping beta):
 * y [n,m], h [n,m,k], s [n,m,m] -> out [n,k,W].  hard_out selects the output of :1272-1312:
 *   0  max-log LLRs, W = num_bits_per_symbol          (output="bit")
 *   1  hard bits, W = num_bits_per_symbol              (output="bit", hard_out=True)
 *   2  logits of the real and the imaginary PAM constellation, W = 2 * 2^(num_bits_per_symbol/2): [2][P]
 *      (output="symbol": samd_pam2qam_logits_f32 forms the QAM logits from them)
 *   3  QAM point index (as a float) of the two PAM argmax decisions, W = 1  (output="symbol", hard_out=True)
 * pam_points DEVICE float[2^(num_bits_per_symbol/2)]: unit-energy Gray PAM points / sqrt(2) in label
 * order; es = their variance; prec = numerical floor (1e-6 in single precision). */
int samd_ep_f32(const float* y, const float* h, const float* s, const float* pam_points, int64_t n,
                int m, int k, int num_bits_per_symbol, int l, float beta, float es, float prec,
                int hard_out, float* out, void* stream);

/* ofdm.EPDetector.call  ofdm/detection.py (OFDMDetector pre-processing + the detector above).
 * out [batch, num_streams_total, num_data * W], W and hard_out as for samd_ep_f32. */
int samd_ofdm_ep_f32(const float* y, const float* h_hat, const float* err_var, int ev_mode,
                     const float* no, const float* pam_points, const int32_t* sc_ind,
                     const int32_t* desired, const int32_t* undesired, const int32_t* data_pos, int batch,
                     int num_rx, int num_rx_ant, int num_streams_total, int streams_per_rx,
                     int num_undesired, int num_ofdm_symbols, int num_eff_subcarriers, int fft_size,
                     int num_data, int num_bits_per_symbol, int l, float beta, float es, float prec,
                     int hard_out, float* out, void* stream);

/* KBestDetector.call  mimo/detection.py:815-1037 (complex representation) with List2LLRSimple
 * mimo/utils.py:539-578: y [n,m], h [n,m,k], s [n,m,m] -> out [n,k,num_bits_per_symbol] LLRs clipped to
 * +-llr_clip (hard_out: bits of the best path).  num_paths = the detector's "k" (<= 64). */
int samd_kbest_f32(const float* y, const float* h, const float* s, const float* points, int64_t n, int m,
                   int k, int num_bits_per_symbol, int num_paths, float llr_clip, int hard_out,
                   float* out, void* stream);

/* ofdm.KBestDetector.call (OFDMDetector pre-processing + the detector above).
 * out [batch, num_streams_total, num_data * num_bits_per_symbol]. */
int samd_ofdm_kbest_f32(const float* y, const float* h_hat, const float* err_var, int ev_mode,
                        const float* no, const float* points, const int32_t* sc_ind,
                        const int32_t* desired, const int32_t* undesired, const int32_t* data_pos,
                        int batch, int num_rx, int num_rx_ant, int num_streams_total, int streams_per_rx,
                        int num_undesired, int num_ofdm_symbols, int num_eff_subcarriers, int fft_size,
                        int num_data, int num_bits_per_symbol, int num_paths, float llr_clip,
                        int hard_out, float* out, void* stream);

/* KBestDetector(use_real_rep=True)  mimo/detection.py:705-727, 815-823, 1011-1030: the same tree search on the real-valued
 * equivalent of the channel (complex2real_channel, mimo/utils.py:13-190) - 2k real streams over the PAM levels of one axis,
 * distances halved for the LLRs (List2LLRSimple, mimo/utils.py:544-547).  Arguments of samd_kbest_f32 / samd_ofdm_kbest_f32
 * with pam_points DEVICE complex64 [2^(num_bits_per_symbol/2)] = the levels as (p, 0); num_bits_per_symbol even (square QAM);
 * out [n, k, num_bits_per_symbol] (resp. [batch, num_streams_total, num_data * num_bits_per_symbol]) in the QAM's bit order. */
int samd_kbest_real_f32(const float* y, const float* h, const float* s, const float* pam_points, int64_t n, int m, int k,
                        int num_bits_per_symbol, int num_paths, float llr_clip, int hard_out, float* out, void* stream);
int samd_ofdm_kbest_real_f32(const float* y, const float* h_hat, const float* err_var, int ev_mode, const float* no,
                             const float* pam_points, const int32_t* sc_ind, const int32_t* desired,
                             const int32_t* undesired, const int32_t* data_pos, int batch, int num_rx, int num_rx_ant,
                             int num_streams_total, int streams_per_rx, int num_undesired, int num_ofdm_symbols,
                             int num_eff_subcarriers, int fft_size, int num_data, int num_bits_per_symbol,
                             int num_paths, float llr_clip, int hard_out, float* out, void* stream);

/* MaximumLikelihoodDetector.call  mimo/detection.py:145-537 on n independent problems: whitening with the Cholesky factor of s,
 * then the exponent -||y~ - H~ x||^2 (+ prior logits of the symbols of x) of every candidate vector x in points^k (stream 0 =
 * most significant digit, _build_vecs :414-470), reduced per (stream, point) with logsumexp (maxlog = 0, "app") or max.
 * y [n,m], h [n,m,k], s [n,m,m] complex64; prior nullable [n,k,2^num_bits_per_symbol] logits; points DEVICE complex64
 * [2^num_bits_per_symbol] -> logits [n,k,2^num_bits_per_symbol].  Bit LLRs / hard decisions: samd_symbol_logits2llrs_f32 on the
 * result (what the reference block does, :531-536).  UNSUPPORTED beyond 65536 candidate vectors or k * 2^nb > 200. */
int samd_ml_detect_f32(const float* y, const float* h, const float* s, const float* prior, const float* points,
                       int64_t n, int m, int k, int num_bits_per_symbol, int maxlog, float* logits, void* stream);
/* ofdm.MaximumLikelihoodDetector(.WithPrior).call  ofdm/detection.py:524-738 (on OFDMDetector / OFDMDetectorWithPrior :21-510):
 * the arguments of samd_ofdm_kbest_f32; prior (nullable) and logits [batch, num_streams_total, num_data, 2^num_bits_per_symbol]. */
int samd_ofdm_ml_f32(const float* y, const float* h_hat, const float* err_var, int ev_mode, const float* no,
                     const float* prior, const float* points, const int32_t* sc_ind, const int32_t* desired,
                     const int32_t* undesired, const int32_t* data_pos, int batch, int num_rx, int num_rx_ant,
                     int num_streams_total, int streams_per_rx, int num_undesired, int num_ofdm_symbols,
                     int num_eff_subcarriers, int fft_size, int num_data, int num_bits_per_symbol, int maxlog,
                     float* logits, void* stream);

/* ---- scrambling --------------------------------------------------- */

/* Scrambler.call fec/scrambling.py:186-261 / TB5GScrambler.call :442-468:
 * out[i] = |x[i] - seq[i % period]| (binary) or x[i] * (1 - 2 seq[i % period]) (soft values).
 * seq DEVICE float32 0/1 [period]; period = numel of the trailing dims the sequence spans. */
int samd_scramble_f32(const float* x, const float* seq, int64_t total, int64_t period, int binary,
                      float* out, void* stream);
/* the same on float64 values (precision = "double"); the bit sequence stays float32 */
int samd_scramble_f64(const double* x, const float* seq, int64_t total, int64_t period, int binary,
                      double* out, void* stream);

/* generate_prng_seq nr/utils.py:14-78 (TS 38.211 5.2.1 length-31 Gold sequence, N_c = 1600):
 * out DEVICE float32 0/1 [length].  Init-time helper (host recurrence + one upload). */
int samd_nr_prng_seq_f32(uint32_t c_init, int64_t length, float* out, void* stream);

/* ---- time-domain variant of the OFDM chain -------------------------------------------- */

/* OFDMModulator.call  ofdm/modulator.py:97-124 (ifftshift -> ifft (signal/utils.py:205-262,
 * sqrt(N)-normalised) -> cyclic prefix).  x [rows, num_ofdm_symbols, fft_size] complex64 ->
 * out [rows, out_len], out_len = sum_s (fft_size + cp_len[s]).  cp_len / sym_off DEVICE
 * int32[num_ofdm_symbols] (prefix length and output offset of every symbol; max_cp = max
 * cp_len <= fft_size); work = rows*num_ofdm_symbols*fft_size complex64 scratch.  The batched
 * 1-D transform is rocFFT's (bound at first use; UNSUPPORTED if librocfft cannot be loaded). */
int samd_ofdm_modulate_c64(const float* x, int rows, int num_ofdm_symbols, int fft_size,
                           const int32_t* cp_len, const int32_t* sym_off, int max_cp, int out_len,
                           float* work, float* out, void* stream);

/* OFDMDemodulator.call  ofdm/demodulator.py:143-203 (cp removal -> fft -> exp(-j 2 pi l_min
 * k / N) phase compensation -> fftshift).  y [rows, in_len] -> out [rows, num_ofdm_symbols,
 * fft_size]; trailing samples beyond the last symbol are ignored. */
int samd_ofdm_demodulate_c64(const float* y, int rows, int in_len, int num_ofdm_symbols,
                             int fft_size, const int32_t* cp_len, const int32_t* sym_off,
                             int l_min, float* work, float* out, void* stream);

/* cir_to_time_channel  channel/utils.py:256-349.  a [B,rx,ra,tx,ta,P,T], tau [B,rx,tx,P] ->
 * h_time [B,rx,ra,tx,ta,T,l_max-l_min+1] = sum_p a_p(t) sinc(l - W tau_p); normalize: unit
 * mean (over ra,ta,T) total tap energy per (b,rx,tx).  norm_scale (nullable, only with normalize): instead of a second pass over
 * h_time the normalisation factor of every (b, rx, tx) link goes to norm_scale [B,rx,tx] and h_time stays
 * un-normalised - samd_apply_time_channel_c64(link_scale = norm_scale) applies it to the received signal
 * (TimeChannel without return_channel). */
int samd_cir_to_time_c64(float bandwidth, const float* a, const float* tau, int l_min, int l_max,
                         int batch, int num_rx, int num_rx_ant, int num_tx, int num_tx_ant,
                         int num_paths, int num_time_steps, int normalize, float* h_time,
                         float* norm_scale, void* stream);

/* ApplyTimeChannel.call (noise-free part)  channel/apply_time_channel.py:85-137.
 * x [B,tx,ta,num_time_samples], h_time [B,rx,ra,tx,ta,num_time_samples+l_tot-1,l_tot] ->
 * y [B,rx,ra,num_time_samples+l_tot-1].  link_scale: nullable [B,rx,tx] factor per link. */
int samd_apply_time_channel_c64(const float* x, const float* h_time, const float* link_scale, int batch,
                                int num_rx, int num_rx_ant, int num_tx, int num_tx_ant,
                                int num_time_samples, int l_tot, float* y, void* stream);

/* LSChannelEstimator (+ NearestNeighborInterpolator)  ofdm/channel_estimation.py:138-173,
 * 257-285, 364-435: out[r,s,j] = y[r, src[s,j]] * coef[s,j]; rows r = batch*num_rx*num_rx_ant,
 * y [rows,n_in] (full grid), src DEVICE int32 [S,n_out] = grid index of the (nearest) pilot RE,
 * coef DEVICE complex64 [S,n_out] = 1/pilot (0 for zero pilots: divide_no_nan). */
int samd_ls_gather_scale_c64(const float* y, const int32_t* src, const float* coef, int rows,
                             int num_streams, int n_out, int n_in, float* out, void* stream);

/* LinearInterpolator.__call__  ofdm/channel_estimation.py:437-733: linear interpolation of the
 * estimates at the pilots, first across subcarriers, then across OFDM symbols (time_avg: mean over
 * the pilot-carrying symbols instead).  hp [rows, num_streams, num_pilots] complex64; DEVICE index
 * tables built once per pilot pattern: fi0/fi1 int32 [S,T,F] = 1 + pilot number of the left/right
 * support (0 = zero pad), fx0/fx1 float [S,T,F] their subcarrier positions, t0/t1 int32 [S,T] the
 * supporting symbols, npil float [S] number of pilot-carrying symbols -> out [rows,S,T,F]. */
int samd_lin_interp_c64(const float* hp, const int32_t* fi0, const int32_t* fi1, const float* fx0,
                        const float* fx1, const int32_t* t0, const int32_t* t1, const float* npil,
                        int rows, int num_streams, int num_pilots, int num_ofdm_symbols,
                        int num_subcarriers, int time_avg, float* out, void* stream);

/* lmmse_equalizer  mimo/equalization.py:101-233 on n independent problems:
 * y [n,m], h [n,m,k], s [n,m,m] complex64 -> x_hat [n,k] complex64, no_eff [n,k] float32.
 * Supported (m,k): (1,1) (2,1) (2,2) (4,1) (4,2) (4,4) (8,1) (8,2) (8,4); else UNSUPPORTED.
 * whiten: 1 / 0 = lmmse_equalizer with / without whitening; 2 = zf_equalizer (:235-298); 3 = mf_equalizer
 * (:300-470) - same inputs and outputs. */
int samd_lmmse_equalizer_c64(const float* y, const float* h, const float* s, int64_t n, int m,
                             int k, int whiten, float* x_hat, float* no_eff, void* stream);

/* LMMSEEqualizer / OFDMEqualizer.call  ofdm/equalization.py:109-275 fused with the per-RE solve.
 * y [B,RX,M,T,FFT]; h_hat [B,RX,M,S,T,F]; err_var: ev_mode 0 none | 1 table [S,T*F] | 2 full
 * [B,RX,M,S,T*F]; no [B,RX,M]; sc_ind [F] effective subcarrier -> fft bin; desired [RX,K] /
 * undesired [RX,U] global stream ids; data_pos [S,T*F] (effective grid) -> data symbol index
 * or -1.  Outputs x_hat, no_eff [B,S,num_data].  All index tables DEVICE int32. */
int samd_ofdm_lmmse_c64(const float* y, const float* h_hat, const float* err_var, int ev_mode,
                        const float* no, const int32_t* sc_ind, const int32_t* desired,
                        const int32_t* undesired, const int32_t* data_pos, int batch, int num_rx,
                        int num_rx_ant, int num_streams_total, int streams_per_rx,
                        int num_undesired, int num_ofdm_symbols, int num_eff_subcarriers,
                        int fft_size, int num_data, int whiten, float* x_hat, float* no_eff,
                        void* stream);

/* LSChannelEstimator(interpolation_type="nn").call + LMMSEEqualizer.call (+ Demapper.call) in one launch
 *   ofdm/channel_estimation.py:175-285 (LS at the pilots) and :323-435 (nearest-neighbour interpolation),
 *   ofdm/equalization.py:107-275, ofdm/detection.py:740-847 (LinearDetector), mapping.py:664-691.
 * No interfering streams (num_interfering_streams_per_rx = 0), whitened LMMSE.  h_hat is never materialised: at every
 * RE it is y_ls[b, rx, m, ls_src[s, re]] * ls_coef[s, re] (the estimator's own product), the estimation-error variance
 * max(no_ls * ls_ev[s, re], 0).  y, y_ls [B,RX,M,T,FFT] complex64 (y_ls = the grid the estimator saw, normally y itself);
 * ls_src int32 / ls_coef complex64 / ls_ev float32 [S, T*F] DEVICE tables over the effective grid; no_ls: 1 or B*RX*M
 * values; no [B,RX,M]; the other tables as for samd_ofdm_lmmse_c64.
 * num_bits_per_symbol = 0: outputs x_hat, no_eff [B,S,num_data] (llr may be NULL).
 * num_bits_per_symbol = 2,4,6,8 (square QAM, pam_levels DEVICE float[2^(m/2)]): output llr [B,S,num_data*m]
 * (logits, or hard decisions when hard_out), x_hat / no_eff may be NULL.  Bit-identical to the three separate entries. */
int samd_ofdm_lsnn_lmmse_c64(const float* y, const float* y_ls, const int32_t* ls_src, const float* ls_coef,
                             const float* ls_ev, const float* no_ls, int64_t no_ls_len, const float* no,
                             const int32_t* sc_ind, const int32_t* desired, const int32_t* data_pos, int batch,
                             int num_rx, int num_rx_ant, int num_streams_total, int streams_per_rx,
                             int num_ofdm_symbols, int num_eff_subcarriers, int fft_size, int num_data,
                             int num_bits_per_symbol, int maxlog, int hard_out, const float* pam_levels,
                             float* x_hat, float* no_eff, float* llr, void* stream);

/* TDL spatial correlation  channel/tr38901/tdl.py:474-492: out[b, i, q] = sum_j mat[i][j] a[b, j, q] over the
 * num_rx_ant * num_tx_ant antenna pairs (rx major); a / out [batch, num_rx_ant * num_tx_ant, inner] complex64
 * (inner = num_paths * num_time_steps), mat [n, n] complex64 DEVICE (square root of the correlation matrix). */
int samd_spatial_corr_c64(const float* a, const float* mat, int batch, int num_rx_ant, int num_tx_ant,
                          int64_t inner, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * precision = "double" (reference block.py:25-52): float64 variants of the blocks whose
 * results depend on the arithmetic precision - the BP decoders and the LLR demapper.
 * ---------------------------------------------------------------------------------- */
/* LDPCBPDecoder.call in float64 (decoding.py:544-637; phi clip :1115-1116).  sched = NULL: flooding;
 * otherwise the array schedule of samd_ldpc_schedule_create.  Buffers like samd_ldpc_bp_decode_f32
 * but double: llr_in [batch, num_vn], out [batch, out_cols], state [num_edges, batch]. */
size_t samd_ldpc_bp_workspace_bytes_f64(const samd_ldpc_graph_t* g, int batch);
int samd_ldpc_bp_decode_f64(const samd_ldpc_graph_t* g, const samd_ldpc_schedule_t* sched,
                            const double* llr_in, double* out, int out_cols, double* state,
                            int state_in, int state_out, int batch, int num_iter, int cn_mode,
                            double llr_max, double offset, int hard_out, void* workspace,
                            size_t workspace_bytes, void* stream);
/* LDPC5GDecoder rate recovery / output mapping in float64 (decoding.py:1438-1475, 1508-1531). */
int samd_ldpc5g_rate_recover_f64(const samd_ldpc5g_t* h, const double* llr, double* out, int batch,
                                 double llr_max, void* stream);
int samd_ldpc5g_extract_codeword_f64(const samd_ldpc5g_t* h, const double* x_hat, double* out,
                                     int batch, void* stream);
/* Demapper.call in float64 (mapping.py:664-691, 927-967): y [num_symbols] complex128, points
 * [2^m] complex128, prior NULL | [m] | [num_symbols, m]; method 0 = app, 1 = maxlog. */
int samd_qam_demap_f64(const double* y, const double* no, int64_t no_len, const double* points,
                       int m, int64_t num_symbols, const double* prior, int64_t prior_len,
                       int method, int hard_out, double* out, void* stream);

/* complex128 / float64 variants of the OFDM link blocks of config C4 (csrc/f64_ofdm.hip, csrc/ofdm_time.hip): the same
 * layouts, index tables and Philox stream positions as the complex64 entries above (samd_awgn_c64, samd_rg_map_c64,
 * samd_tdl_cir_c64, samd_cir_to_ofdm_c64, samd_apply_ofdm_channel_c64, samd_ls_gather_scale_c64, samd_ofdm_modulate_c64,
 * samd_ofdm_demodulate_c64), every real argument double.  Random draws: the float32 stream's 24-bit uniforms (exact in
 * double), Box-Muller / affine maps evaluated in double - specification oracle/f64_ofdm.py.  samd_awgn_c128: x nullable
 * (= 0: complex_normal, utils/misc.py:19-54, with no = var).  samd_gather3 moves complex128 elements with
 * floats_per_elem = 4. */
int samd_awgn_c128(const double* x, const double* no, int64_t no_len, uint64_t seed, uint64_t call, int64_t n,
                   double* y, void* stream);
int samd_rg_map_c128(const double* x, const double* pilots, const int32_t* data_pos, const int32_t* pilot_pos,
                     int batch, int num_streams, int num_re, int num_data, int num_pilots, double* out, void* stream);
int samd_tdl_cir_c128(uint64_t seed, uint64_t call, int batch, int num_rx_ant, int num_tx_ant, int num_paths,
                      int num_time_steps, int num_sinusoids, double sampling_frequency, const double* mean_powers,
                      double min_doppler, double max_doppler, int los, double los_power, double los_aoa, double* a,
                      void* stream);
int samd_cir_to_ofdm_c128(const double* a, const double* tau, const double* frequencies, int batch, int num_rx,
                          int num_rx_ant, int num_tx, int num_tx_ant, int num_paths, int num_time_steps,
                          int num_freqs, int normalize, double* h_freq, void* stream);
int samd_apply_ofdm_channel_c128(const double* x, const double* h_freq, int batch, int num_rx_x_ant,
                                 int num_tx_x_ant, int num_re, double* y, void* stream);
int samd_ls_gather_scale_c128(const double* y, const int32_t* src, const double* coef, int rows, int num_streams,
                              int n_out, int n_in, double* out, void* stream);
int samd_ofdm_modulate_c128(const double* x, int rows, int num_ofdm_symbols, int fft_size, const int32_t* cp_len,
                            const int32_t* sym_off, int max_cp, int out_len, double* work, double* out, void* stream);
int samd_ofdm_demodulate_c128(const double* y, int rows, int in_len, int num_ofdm_symbols, int fft_size,
                              const int32_t* cp_len, const int32_t* sym_off, int l_min, double* work, double* out,
                              void* stream);

/* cir_to_time_channel / ApplyTimeChannel in complex128 (precision = "double"; csrc/f64_time.hip): the layouts of
 * samd_cir_to_time_c64 / samd_apply_time_channel_c64 without the deferred normalisation factor (h_time is normalised in place) */
int samd_cir_to_time_c128(double bandwidth, const double* a, const double* tau, int l_min, int l_max, int batch,
                          int num_rx, int num_rx_ant, int num_tx, int num_tx_ant, int num_paths, int num_time_steps,
                          int normalize, double* h_time, void* stream);
int samd_apply_time_channel_c128(const double* x, const double* h_time, int batch, int num_rx, int num_rx_ant,
                                 int num_tx, int num_tx_ant, int num_time_samples, int l_tot, double* y, void* stream);

/* samd_spatial_corr_c64 on complex128 coefficients (precision = "double") */
int samd_spatial_corr_c128(const double* a, const double* mat, int batch, int num_rx_ant, int num_tx_ant, int64_t inner,
                           double* out, void* stream);

/* samd_lin_interp_c64 on complex128 estimates (precision = "double"); index and position tables as for the complex64 entry */
int samd_lin_interp_c128(const double* hp, const int32_t* fi0, const int32_t* fi1, const float* fx0,
                         const float* fx1, const int32_t* t0, const int32_t* t1, const float* npil,
                         int rows, int num_streams, int num_pilots, int num_ofdm_symbols,
                         int num_subcarriers, int time_avg, double* out, void* stream);

/* float64 variants of the symbol-domain mapping entries (csrc/f64_mapping.hip): the argument layouts of samd_symbol_demap_f32,
 * samd_symbol_logits2llrs_f32, samd_llrs2symbol_logits_f32, samd_symbol_logits2moments_c64, samd_pam2qam_logits_f32 with every
 * real argument double (complex128 symbols / points / means). */
int samd_symbol_demap_f64(const double* y, const double* no, int64_t no_len, const double* points, int m,
                          int64_t num_symbols, const double* prior, int64_t prior_len, int hard_out, double* out,
                          int32_t* out_idx, void* stream);
int samd_symbol_logits2llrs_f64(const double* logits, int m, int64_t rows, const double* prior, int64_t prior_len,
                                int method, int hard_out, double* out, void* stream);
int samd_llrs2symbol_logits_f64(const double* llrs, int m, int64_t rows, int hard_out, double* out, int32_t* out_idx,
                                void* stream);
int samd_symbol_logits2moments_c128(const double* logits, const double* points, int m, int64_t rows, double* mean,
                                    double* var, void* stream);
int samd_pam2qam_logits_f64(const double* pam1, const double* pam2, int num_bits_per_symbol, int64_t rows, double* out,
                            void* stream);

/* lmmse_equalizer / zf_equalizer / mf_equalizer in complex128 (precision = "double", reference block.py:25-52;
 * mimo/equalization.py:101-463): y [n, M], h [n, M, K], s [n, M, M] complex128 -> x_hat [n, K] complex128, no_eff
 * [n, K] float64; mode 0 LMMSE without whitening, 1 LMMSE, 2 ZF, 3 MF; K <= 8, K <= M <= 16.  OFDMEqualizer.call with
 * precision="double" builds the per-resource-element inputs on the device and calls this. */
int samd_lmmse_equalizer_c128(const double* y, const double* h, const double* s, int64_t n, int m, int k, int mode,
                              double* x_hat, double* no_eff, void* stream);

/* MaximumLikelihoodDetector.call in float64 (precision = "double"; mimo/detection.py:463-537): y [n, M], h [n, M, K], s [n, M, M]
 * complex128, prior DEVICE float64 [n, K, 2^nb] logits on the points or NULL, points DEVICE complex128 [2^nb] -> out [n, K, 2^nb]
 * logits; maxlog 0 = "app".  K <= 8, M <= 16, (2^nb)^K <= 65536, K 2^nb <= 200.  The OFDM blocks build the per-resource-element
 * inputs on the device (as for samd_lmmse_equalizer_c128) and call this. */
int samd_ml_detect_f64(const double* y, const double* h, const double* s, const double* prior, const double* points, int64_t n,
                       int m, int k, int num_bits_per_symbol, int maxlog, double* out, void* stream);

/* EPDetector.call in float64 (precision = "double"; mimo/detection.py:1166-1312): arguments and output modes of samd_ep_f32
 * (hard_out 0 max-log LLRs [n, K, nb] / 1 hard bits, 2 the logits of the two PAM constellations [n, K, 2, 2^(nb/2)], 3 the QAM
 * index of their argmax decisions [n, K]) on complex128 / float64 buffers; pam DEVICE float64 [2^(nb/2)]; K <= 8, M <= 16. */
int samd_ep_f64(const double* y, const double* h, const double* s, const double* pam, int64_t n, int m, int k,
                int num_bits_per_symbol, int num_iter, double beta, double es, double prec, int hard_out, double* out,
                void* stream);

/* ------------------------------------------------------------------------------------
 * CRC and Polar codes (config C5 of the north star).
 * ---------------------------------------------------------------------------------- */
/* CRCEncoder.call / CRCDecoder.call  fec/crc.py:175-215, 289-321.  poly = coefficients of
 * x^(crc_len-1)..x^0 of the generator (bit crc_len-1 = x^(crc_len-1)).  check=0: bits
 * [n_words,k] -> out [n_words,k+crc_len] (bits followed by the parity); check=1: bits
 * [n_words,k] INCLUDING the received parity -> out [n_words] = 1.0 where the CRC holds. */
int samd_crc_f32(const float* bits, int64_t n_words, int k, uint32_t poly, int crc_len, int check,
                 float* out, void* stream);

/* PolarEncoder.call (+ rate-matching gather)  fec/polar/encoding.py:140-209, 732:
 * u [batch,k] placed at info_pos (DEVICE int32[k]) of a length-n word (n a power of two), polar
 * transform, out[b,i] = x[out_idx[i]] (DEVICE int32[n_out]). */
int samd_polar_encode_f32(const float* u, const int32_t* info_pos, const int32_t* out_idx, int batch,
                          int k, int n, int n_out, float* out, void* stream);

/* PolarSCDecoder / PolarSCLDecoder (default TF path, fast SCL)  fec/polar/decoding.py:122-263,
 * 525-723, 919-1045, 1345-1437.  llr [batch,n] logits (n <= 1024); ops DEVICE int32[num_ops]
 * packed decoding schedule built by the host (sionna_amd/phy/fec/polar/decoding.py::
 * build_schedule / pack_schedule: op | stage<<3 | side<<7 | (bit_index+2048)<<8 with op 0 f, 1 g,
 * 2 leaf, 3 rate-0, 4 repetition, 5 combine, 6 end, 7 = a complete subtree of `stage` whose first
 * bit is bit_index, result onto `side`, bit 20 = fast-SCL shortcuts inside it.  Every engine accepts
 * subtrees of stage 1 with two information leaves; subtrees of stage R and R + 1 are accepted iff
 * samd_polar_scl_register_stages(n, list_size, sc_mode) == R (-1: none - the generic engine runs;
 * R >= 1: the engine whose stages 0..R live in registers decodes such a subtree without further
 * schedule dispatch, from the frozen pattern it derives from info_pos)); info_pos
 * DEVICE int32[k]; iil_inv nullable DEVICE int32[k] (inverse input interleaver applied before the
 * CRC check); sc_mode=1 -> hard SC decisions (list_size must be 1); crc_len=0 disables the
 * CRC-aided selection.  u_hat [batch,k]; crc_status nullable [batch].  workspace: caller-owned
 * device scratch of samd_polar_scl_workspace_bytes() bytes (top LLR stage of the resident
 * codewords; it lives in L2 so that more codewords fit in LDS). */
int samd_polar_scl_register_stages(int n, int list_size, int sc_mode);
size_t samd_polar_scl_workspace_bytes(int batch, int n, int list_size);
int samd_polar_scl_decode_f32(const float* llr, const int32_t* ops, int num_ops, const int32_t* info_pos,
                              const int32_t* iil_inv, int batch, int n, int k, int list_size,
                              int sc_mode, uint32_t crc_poly, int crc_len, float* u_hat,
                              float* crc_status, void* workspace, size_t workspace_bytes,
                              void* stream);
/* Polar5GDecoder.call (polar/decoding.py:1999-2086) with its rate recovery (:2018-2052) inside the decoder's channel-LLR load:
 * llr [batch,n_in] as received; src_a / src_b DEVICE int32[n] (src_b nullable): position i of the mother code reads llr[src_a[i]]
 * (-1: 0 = punctured, -2: -rm_fill = shortened) + llr[src_b[i]] (repetition; -1: nothing).  The rest as samd_polar_scl_decode_f32.
 * SAMD_ERR_UNSUPPORTED when (n, list_size, sc_mode) runs the generic engine: gather on the host side and call that entry. */
int samd_polar5g_scl_decode_f32(const float* llr, int n_in, const int32_t* src_a, const int32_t* src_b, float rm_fill,
                                const int32_t* ops, int num_ops, const int32_t* info_pos, const int32_t* iil_inv, int batch,
                                int n, int k, int list_size, int sc_mode, uint32_t crc_poly, int crc_len, float* u_hat,
                                float* crc_status, void* workspace, size_t workspace_bytes, void* stream);
/* precision = "double" (reference block.py:25-52): the same decoder on float64 LLRs with the arithmetic of the reference's own
 * float64 NumPy twin (decoding.py:1113-1149: literal log(1 + e^x) and log(1 + e^(x+y)) - log(e^x + e^y), libm) - oracle/polar_scl.c
 * precision 1.  Always the generic engine: the schedule must use stage-1 SUBTREE records only (what
 * samd_polar_scl_register_stages() = -1 asks for).  u_hat / crc_status double. */
size_t samd_polar_scl_workspace_bytes_f64(int batch, int n, int list_size);
int samd_polar_scl_decode_f64(const double* llr, const int32_t* ops, int num_ops, const int32_t* info_pos,
                              const int32_t* iil_inv, int batch, int n, int k, int list_size, int sc_mode,
                              uint32_t crc_poly, int crc_len, double* u_hat, double* crc_status,
                              void* workspace, size_t workspace_bytes, void* stream);

/* PolarBPDecoder.call  fec/polar/decoding.py:1587-1771 (and the decoder Polar5GDecoder(dec_type="BP")
 * instantiates, :1896-1912): num_iter flooding iterations on the polar factor graph - per iteration a
 * left-to-right sweep over the log2(n) butterfly stages (R messages) and a right-to-left sweep (L
 * messages), boxplus log(1+exp(x+y)) - log(exp(x)+exp(y)) on inputs clipped to +-19.3 (:1587-1603) in
 * float32 on the library's defined exp / log.  llr [batch,n] logits (n a power of two); prior DEVICE
 * float[n]: the right-going messages entering column 0 (19.3 at frozen positions, 0 elsewhere, :1632-1636);
 * info_pos DEVICE int32[k].  out [batch,k]: hard_out!=0 -> bits (1 where the final LLR <= 0), else soft
 * logits (:1719-1723).  A codeword's messages ((2 log2(n) - 1) n floats) live in LDS for n <= 1024
 * (samd_polar_bp_workspace_bytes() == 0, workspace may be NULL); longer codes keep them in the
 * caller-owned device workspace. */
size_t samd_polar_bp_workspace_bytes(int batch, int n);
int samd_polar_bp_decode_f32(const float* llr, const float* prior, const int32_t* info_pos, int batch,
                             int n, int k, int num_iter, int hard_out, float* out, void* workspace,
                             size_t workspace_bytes, void* stream);
/* the same decoder on float64 (precision = "double"): libm exp / log in the literal boxplus; message columns that exceed the
 * LDS in double (n >= 1024) go to the workspace */
size_t samd_polar_bp_workspace_bytes_f64(int batch, int n);
int samd_polar_bp_decode_f64(const double* llr, const double* prior, const int32_t* info_pos, int batch, int n,
                             int k, int num_iter, int hard_out, double* out, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-GPU: the one collective of the path.  One process per GPU; the Monte-Carlo
 * batches are independent, so the only exchange is the SUM of the int64 error counters
 * {bit_errors, block_errors, num_bits, num_blocks} over the ranks - one RCCL allReduce of 32 bytes
 * per iteration instead of the full-tensor gathers of the reference's tf.distribute path
 * (utils/misc.py:546-547, 616-655).  Python hosts use torch.distributed (sim_ber(distribute=...));
 * these entries give a C host the same collective.  RCCL is loaded on first use (dlopen).
 *   samd_comm_unique_id      rank 0 fills SAMD_COMM_ID_BYTES host bytes and hands them to the other
 *                            ranks out of band (file, socket, MPI, environment)
 *   samd_comm_create         every rank, with the SAME id; binds the calling thread's current HIP
 *                            device (call hipSetDevice(local_rank) first); collective: returns when
 *                            all world_size ranks have called it
 *   samd_comm_allreduce_sum_i64   counters DEVICE int64[count], in place, on `stream`
 * ---------------------------------------------------------------------------------- */
#define SAMD_COMM_ID_BYTES 128
typedef struct samd_comm samd_comm_t;
int samd_comm_unique_id(void* id_out);
int samd_comm_create(const void* id, int rank, int world_size, samd_comm_t** out);
int samd_comm_rank(const samd_comm_t* c);
int samd_comm_world_size(const samd_comm_t* c);
int samd_comm_allreduce_sum_i64(samd_comm_t* c, int64_t* counters, int64_t count, void* stream);
void samd_comm_destroy(samd_comm_t* c);

/* ------------------------------------------------------------------------------------
 * The per-item linear-algebra helpers of the MIMO blocks as calls of their own (csrc/mimo_linalg.hip): n independent
 * problems, interleaved complex, row-major matrices, 1 <= M, K <= 16; _c64 = complex64, _c128 = complex128
 * (precision = "double").  The receiver kernels carry the same algebra fused; these are for host code that calls the
 * helpers directly.
 *   samd_inv_cholesky    utils/linalg.py:8-32      a [n,M,M] Hermitian positive definite = L L^H -> out [n,M,M] = L^-1
 *   samd_matrix_pinv     utils/linalg.py:35-59     a [n,M,K] of full column rank (K <= M) -> out [n,K,M] = (A^H A)^-1 A^H
 *   samd_whiten_channel  mimo/utils.py:292-356     y [n,M], h [n,M,K], s [n,M,M] = L L^H -> yw = L^-1 y, hw = L^-1 h
 *   samd_lmmse_matrix    mimo/equalization.py:11-99  h [n,M,K], s [n,M,M] or NULL -> g [n,K,M] = H^H (H H^H + S)^-1,
 *                        with s == NULL (white unit-variance noise) (H^H H + I)^-1 H^H
 * ---------------------------------------------------------------------------------- */
int samd_inv_cholesky_c64(const float* a, int64_t n, int m, float* out, void* stream);
int samd_inv_cholesky_c128(const double* a, int64_t n, int m, double* out, void* stream);
int samd_matrix_pinv_c64(const float* a, int64_t n, int m, int k, float* out, void* stream);
int samd_matrix_pinv_c128(const double* a, int64_t n, int m, int k, double* out, void* stream);
int samd_whiten_channel_c64(const float* y, const float* h, const float* s, int64_t n, int m, int k, float* yw, float* hw,
                            void* stream);
int samd_whiten_channel_c128(const double* y, const double* h, const double* s, int64_t n, int m, int k, double* yw,
                             double* hw, void* stream);
int samd_lmmse_matrix_c64(const float* h, const float* s, int64_t n, int m, int k, float* g, void* stream);
int samd_lmmse_matrix_c128(const double* h, const double* s, int64_t n, int m, int k, double* g, void* stream);

/* ------------------------------------------------------------------------------------
 * Error counting  utils/metrics.py:94-144 (count_errors, count_block_errors).
 * b, b_hat [num_blocks, block_len] float32; counters: DEVICE int64[2], ADDED to:
 * counters[0] += #(b != b_hat), counters[1] += #blocks with any mismatch.
 * soft!=0 applies hard_decisions (llr > 0, misc.py:270) to b_hat first.
 * ---------------------------------------------------------------------------------- */
int samd_count_errors_f32(const float* b, const float* b_hat, int64_t num_blocks,
                          int64_t block_len, int soft, int64_t* counters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIONNA_AMD_H */
