// TEST INFRASTRUCTURE: runs the per-wave programs libsionna_amd.so generated for one 5G LDPC code (JIT_EMU_SRC, written by
// tests/test_jit_emu.py from samd_ldpc5g_jit_source) on the CPU - JIT_NWAVES host threads = the waves of a workgroup (16 unless SAMD_JIT_WAVES says otherwise), one
// workgroup after the other.  See jit_emu_ops.h.
#include "jit_emu_ops.h"
#include JIT_EMU_SRC

#include <thread>
#include <vector>

#ifndef JIT_WS_FLOATS
#define JIT_WS_FLOATS 0
#endif
// st_in / st_out (state variant of the generated source, SAMD_JIT_STATE=1): the message images, [passes][JIT image bytes]
extern "C" int jit_emu_decode_state(const float* llr_in, float* out, int batch, int num_iter, float llr_max, float offset,
                                    int hard_out, int grid, const float* st_in, float* st_out) {
  std::vector<unsigned char> lds((size_t)JIT_LDS_FLOATS * 4);
  std::vector<float> ws((size_t)JIT_WS_FLOATS * grid + 16, std::numeric_limits<float>::quiet_NaN());   // one row per workgroup
  for (int blk = 0; blk < grid; ++blk) {
    memset(lds.data(), 0xFF, lds.size());                    // NaN pattern: a slot read before it was written shows up
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, JIT_NWAVES);
    std::vector<std::thread> th;
    for (int w = 0; w < JIT_NWAVES; ++w)
      th.emplace_back([&, w]() {
        jit_emu_ctx = JitEmuCtx{lds.data(), lds.size(), &bar, blk, grid, (size_t)JIT_WS_FLOATS * 4};
#define JIT_EMU_CASE(W) case W: jit_wave_##W(llr_in, out, batch, num_iter, llr_max, offset, hard_out, ws.data(), st_in, st_out); break;
        switch (w) {
#if JIT_NWAVES > 0
          JIT_EMU_CASE(0)
#endif
#if JIT_NWAVES > 1
          JIT_EMU_CASE(1)
#endif
#if JIT_NWAVES > 2
          JIT_EMU_CASE(2)
#endif
#if JIT_NWAVES > 3
          JIT_EMU_CASE(3)
#endif
#if JIT_NWAVES > 4
          JIT_EMU_CASE(4)
#endif
#if JIT_NWAVES > 5
          JIT_EMU_CASE(5)
#endif
#if JIT_NWAVES > 6
          JIT_EMU_CASE(6)
#endif
#if JIT_NWAVES > 7
          JIT_EMU_CASE(7)
#endif
#if JIT_NWAVES > 8
          JIT_EMU_CASE(8)
#endif
#if JIT_NWAVES > 9
          JIT_EMU_CASE(9)
#endif
#if JIT_NWAVES > 10
          JIT_EMU_CASE(10)
#endif
#if JIT_NWAVES > 11
          JIT_EMU_CASE(11)
#endif
#if JIT_NWAVES > 12
          JIT_EMU_CASE(12)
#endif
#if JIT_NWAVES > 13
          JIT_EMU_CASE(13)
#endif
#if JIT_NWAVES > 14
          JIT_EMU_CASE(14)
#endif
#if JIT_NWAVES > 15
          JIT_EMU_CASE(15)
#endif
        }
      });
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&bar);
  }
  return 0;
}

extern "C" int jit_emu_decode(const float* llr_in, float* out, int batch, int num_iter, float llr_max, float offset,
                              int hard_out, int grid) {
  return jit_emu_decode_state(llr_in, out, batch, num_iter, llr_max, offset, hard_out, grid, nullptr, nullptr);
}
