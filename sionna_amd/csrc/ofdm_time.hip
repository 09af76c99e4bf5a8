// Time-domain variant of the C4 chain (SURVEY.md §8a last row): OFDM modulator / demodulator
// around rocFFT, the discrete-time channel taps and the time-varying FIR channel.
//
//   OFDMModulator.call      /root/reference/src/sionna/phy/ofdm/modulator.py:97-124
//   OFDMDemodulator.call    /root/reference/src/sionna/phy/ofdm/demodulator.py:143-203
//   fft / ifft              /root/reference/src/sionna/phy/signal/utils.py:150-262
//   cir_to_time_channel     /root/reference/src/sionna/phy/channel/utils.py:256-349
//   ApplyTimeChannel.call   /root/reference/src/sionna/phy/channel/apply_time_channel.py:85-137
//
// The batched 1-D transforms are rocFFT's (north_star: "OFDM FFT goes to rocFFT"); the
// library is bound lazily with dlopen so that libsionna_amd.so itself carries no link-time
// dependency on it (inside a PyTorch process the already-mapped librocfft.so.0 is reused).
// The kernels here are the HBM-bound glue around the transform: (i)fftshift, cyclic-prefix
// insertion / removal, the 1/sqrt(N) scaling and the l_min phase compensation are folded into
// one gather pass before and one scatter pass after the transform.
#include "common.h"

#include <dlfcn.h>
#include <rocfft/rocfft.h>

#include <map>
#include <mutex>
#include <tuple>

namespace samd {
namespace {

// ---------------------------------------------------------------- rocFFT binding
struct RocfftApi {
  void* handle = nullptr;
  decltype(&rocfft_setup) setup = nullptr;
  decltype(&rocfft_plan_create) plan_create = nullptr;
  decltype(&rocfft_plan_destroy) plan_destroy = nullptr;
  decltype(&rocfft_plan_get_work_buffer_size) plan_get_work_buffer_size = nullptr;
  decltype(&rocfft_execute) execute = nullptr;
  decltype(&rocfft_execution_info_create) execution_info_create = nullptr;
  decltype(&rocfft_execution_info_set_work_buffer) execution_info_set_work_buffer = nullptr;
  decltype(&rocfft_execution_info_set_stream) execution_info_set_stream = nullptr;
  bool ok = false;
  std::string err;
};

struct FftPlan {
  rocfft_plan plan = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
};

std::mutex g_fft_mutex;
RocfftApi g_fft;
std::map<std::tuple<int, int, int, int>, FftPlan> g_plans;  // (device, n, batch, inverse)

template <typename F>
bool bind(void* h, const char* name, F* f, std::string* err) {
  *f = reinterpret_cast<F>(dlsym(h, name));
  if (!*f) {
    *err = std::string("rocFFT symbol missing: ") + name;
    return false;
  }
  return true;
}

bool load_rocfft_locked() {
  if (g_fft.ok) return true;
  if (g_fft.handle == nullptr) {
    const char* names[] = {"librocfft.so.0", "librocfft.so", "/opt/rocm/lib/librocfft.so.0"};
    for (const char* n : names) {
      g_fft.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (g_fft.handle) break;
    }
    if (!g_fft.handle) {
      g_fft.err = std::string("cannot load librocfft.so.0: ") + dlerror();
      return false;
    }
  }
  void* h = g_fft.handle;
  std::string* e = &g_fft.err;
  if (!(bind(h, "rocfft_setup", &g_fft.setup, e) && bind(h, "rocfft_plan_create", &g_fft.plan_create, e) &&
        bind(h, "rocfft_plan_destroy", &g_fft.plan_destroy, e) &&
        bind(h, "rocfft_plan_get_work_buffer_size", &g_fft.plan_get_work_buffer_size, e) &&
        bind(h, "rocfft_execute", &g_fft.execute, e) &&
        bind(h, "rocfft_execution_info_create", &g_fft.execution_info_create, e) &&
        bind(h, "rocfft_execution_info_set_work_buffer", &g_fft.execution_info_set_work_buffer, e) &&
        bind(h, "rocfft_execution_info_set_stream", &g_fft.execution_info_set_stream, e)))
    return false;
  if (g_fft.setup() != rocfft_status_success) {
    g_fft.err = "rocfft_setup failed";
    return false;
  }
  g_fft.ok = true;
  return true;
}

// Batched in-place 1-D complex64 (dbl: complex128) transform of `batch` contiguous rows of length n.
int fft_inplace(void* data, int n, int batch, bool inverse, hipStream_t stream, bool dbl = false) {
  std::lock_guard<std::mutex> lock(g_fft_mutex);
  if (!load_rocfft_locked()) {
    set_error(g_fft.err);
    return SAMD_ERR_UNSUPPORTED;
  }
  int dev = 0;
  SAMD_HIP_CHECK(hipGetDevice(&dev));
  const auto key = std::make_tuple(dev, n, batch, (inverse ? 1 : 0) | (dbl ? 2 : 0));
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    FftPlan p;
    const size_t len = (size_t)n;
    if (g_fft.plan_create(&p.plan, rocfft_placement_inplace,
                          inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                          dbl ? rocfft_precision_double : rocfft_precision_single, 1, &len, (size_t)batch, nullptr) != rocfft_status_success) {
      set_error("rocfft_plan_create failed");
      return SAMD_ERR_HIP;
    }
    size_t wbytes = 0;
    g_fft.plan_get_work_buffer_size(p.plan, &wbytes);
    if (g_fft.execution_info_create(&p.info) != rocfft_status_success) {
      set_error("rocfft_execution_info_create failed");
      return SAMD_ERR_HIP;
    }
    if (wbytes) {
      SAMD_HIP_CHECK(hipMalloc(&p.work, wbytes));
      g_fft.execution_info_set_work_buffer(p.info, p.work, wbytes);
    }
    it = g_plans.emplace(key, p).first;
  }
  FftPlan& p = it->second;
  if (g_fft.execution_info_set_stream(p.info, (void*)stream) != rocfft_status_success) {
    set_error("rocfft_execution_info_set_stream failed");
    return SAMD_ERR_HIP;
  }
  void* in[1] = {data};
  if (g_fft.execute(p.plan, in, nullptr, p.info) != rocfft_status_success) {
    set_error("rocfft_execute failed");
    return SAMD_ERR_HIP;
  }
  return SAMD_OK;
}

// ---------------------------------------------------------------- kernels
// work[r, s, k] = x[r, s, (k + n/2 [ceil for ifftshift]) mod n]: ifftshift moves the DC
// subcarrier (index n//2) to bin 0 (modulator.py:100).
__device__ __forceinline__ void sincos_r(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void sincos_r(double x, double* s, double* c) { sincos(x, s, c); }

template <typename R2, typename R>
__global__ void ifftshift_kernel(const R2* __restrict__ x, long long total, int n, R2* __restrict__ work) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % n);
  const long long row = i / n;
  const int src = (k + n / 2) % n;  // np.fft.ifftshift: out[k] = in[(k + n//2) % n]
  work[i] = x[row * n + src];
}

// out[r, off[s] + t] = work[r, s, (t - cp[s]) mod n] / sqrt(n), t in [0, n + cp[s])
// (modulator.py:103-124 with ifft = sqrt(n) * IDFT_normalised, signal/utils.py:246-262).
template <typename R2, typename R>
__global__ void add_cp_kernel(const R2* __restrict__ work, const int32_t* __restrict__ cp,
                              const int32_t* __restrict__ off, int nsym, int n, int out_len, R scale,
                              int ntb, R2* __restrict__ out) {
  const int s = (blockIdx.x / ntb) % nsym;
  const long long r = blockIdx.x / ntb / nsym;
  const int t = (blockIdx.x % ntb) * blockDim.x + threadIdx.x;
  const int c = cp[s];
  if (t >= n + c) return;
  int src = t - c;
  if (src < 0) src += n;
  const R2 v = work[(r * nsym + s) * n + src];
  out[r * out_len + off[s] + t] = R2{v.x * scale, v.y * scale};
}

// work[r, s, t] = y[r, off[s] + cp[s] + t]  (demodulator.py:184-195)
template <typename R2, typename R>
__global__ void remove_cp_kernel(const R2* __restrict__ y, const int32_t* __restrict__ cp,
                                 const int32_t* __restrict__ off, int nsym, int n, int in_len,
                                 int ntb, R2* __restrict__ work) {
  const int s = (blockIdx.x / ntb) % nsym;
  const long long r = blockIdx.x / ntb / nsym;
  const int t = (blockIdx.x % ntb) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  work[(r * nsym + s) * n + t] = y[r * in_len + off[s] + cp[s] + t];
}

// out[r, s, k'] = work[r, s, k] * exp(j * phase_step * k) / sqrt(n), k = (k' - n//2) mod n
// i.e. fftshift after the phase compensation (demodulator.py:197-203).
template <typename R2, typename R>
__global__ void demod_post_kernel(const R2* __restrict__ work, long long total, int n, R phase_step, R scale,
                                  R2* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int kp = (int)(i % n);
  const long long row = i / n;
  int k = kp - n / 2;  // np.fft.fftshift: out[k'] = in[(k' - n//2) % n]
  if (k < 0) k += n;
  const R2 v = work[row * n + k];
  R sn, cs;
  sincos_r(phase_step * (R)k, &sn, &cs);
  const R re = (v.x * cs - v.y * sn) * scale;
  const R im = (v.x * sn + v.y * cs) * scale;
  out[i] = R2{re, im};
}

__device__ __forceinline__ float sincf(float x) {
  // tf.experimental.numpy.sinc: sin(pi x) / (pi x), 1 at x = 0
  if (x == 0.0f) return 1.0f;
  const float y = 3.14159265358979323846f * x;
  return sinf(y) / y;
}

// One workgroup per (b, rx, tx): g[p][l] = sinc(l - tau_p W) in LDS, then
// h[ra, ta, t, l] = sum_p a[ra, ta, p, t] g[p][l]; optional in-block normalisation
// (utils.py:337-347: c = mean over (ra, ta, t) of sum_l |h|^2).
__global__ void __launch_bounds__(256) cir_to_time_kernel(const float2* __restrict__ a, const float* __restrict__ tau,
                                                          float bandwidth, int l_min, int L, int num_rx, int RA,
                                                          int num_tx, int TA, int P, int T, int normalize,
                                                          float2* __restrict__ h, float* __restrict__ scale_out) {
  extern __shared__ __attribute__((aligned(16))) float g[];  // [P][L], then the per-wave output stage
  __shared__ float red[256];
  const int grp = blockIdx.x;  // (b, rx, tx)
  const int tx = grp % num_tx;
  const int rx = (grp / num_tx) % num_rx;
  const long long b = grp / (num_tx * num_rx);
  const float* tau_g = tau + (long long)grp * P;
  for (int i = threadIdx.x; i < P * L; i += blockDim.x) {
    const int p = i / L, l = i % L;
    g[i] = sincf((float)(l_min + l) - tau_g[p] * bandwidth);
  }
  __syncthreads();
  // one lane per time step, kTapTile taps at a time in registers: every path coefficient a[p][t] is read once
  // per tile (coalesced over t), the sinc weights are LDS broadcasts, no per-element index arithmetic;
  // the sum over paths keeps its order (p ascending), so h is what the element-per-thread form produced.
  // A wave's 64 x L results go through an LDS stage so that h[t][l] leaves as contiguous 512-byte stores
  // (one lane per t would scatter every store instruction over 64 cache lines).
  constexpr int kTapTile = 9;
  float2* stage = reinterpret_cast<float2*>(g + ((P * L + 1) & ~1)) + (size_t)(threadIdx.x >> 6) * 64 * L;   // [64][L] per wave
  const int lane = threadIdx.x & 63;
  float energy = 0.0f;
  for (int lk = 0; lk < RA * TA; ++lk) {
    const int ra = lk / TA, ta = lk - ra * TA;
    const long long link = ((((b * num_rx + rx) * RA + ra) * num_tx + tx) * TA + ta);
    const float2* ap = a + link * P * T;
    float2* hp = h + link * T * L;
    for (int t0 = (threadIdx.x >> 6) * 64; t0 < T; t0 += blockDim.x) {      // 64 time steps per wave and pass
      const int t = t0 + lane;
      for (int l0 = 0; l0 < L; l0 += kTapTile) {
        float re[kTapTile], im[kTapTile];
#pragma unroll
        for (int j = 0; j < kTapTile; ++j) { re[j] = 0.0f; im[j] = 0.0f; }
        if (t < T)
          for (int p = 0; p < P; ++p) {
            const float2 v = ap[(long long)p * T + t];
            const float* gp = g + p * L + l0;
#pragma unroll
            for (int j = 0; j < kTapTile; ++j)
              if (l0 + j < L) { const float w = gp[j]; re[j] += v.x * w; im[j] += v.y * w; }
          }
#pragma unroll
        for (int j = 0; j < kTapTile; ++j)
          if (l0 + j < L) {
            stage[lane * L + l0 + j] = make_float2(re[j], im[j]);
            energy += re[j] * re[j] + im[j] * im[j];                       // zero for t >= T
          }
      }
      __builtin_amdgcn_wave_barrier();
      const int cnt = min(64, T - t0) * L;
      for (int i = lane; i < cnt; i += 64) hp[(long long)t0 * L + i] = stage[i];
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (!normalize) return;
  red[threadIdx.x] = energy;
  __syncthreads();                                                          // also orders the h writes above
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float c = sqrtf(red[0] / (float)(RA * TA * T));
  const float inv = c > 0.0f ? 1.0f / c : 0.0f;  // divide_no_nan
  if (scale_out) {                                // the caller applies the factor to the received signal
    if (threadIdx.x == 0) scale_out[grp] = inv;
    return;
  }
  for (int ra = 0; ra < RA; ++ra) {               // the TA links of one receive antenna are contiguous in h
    const long long link0 = ((((b * num_rx + rx) * RA + ra) * num_tx + tx) * TA);
    float2* hp = h + link0 * T * L;
    const long long cnt = (long long)TA * T * L;
    for (long long i = threadIdx.x; i < cnt; i += blockDim.x) {
      const float2 v = hp[i];
      hp[i] = make_float2(v.x * inv, v.y * inv);
    }
  }
}

// y[b, rxa, t] = sum_{txa} sum_l h[b, rxa, txa, t, l] x[b, txa, t - l], 0 <= t - l < Tn
// (apply_time_channel.py:121-132); rxa = rx * RA + ra, txa = tx * TA + ta.
__global__ void __launch_bounds__(256) apply_time_kernel(const float2* __restrict__ x, const float2* __restrict__ h,
                                                         const float* __restrict__ link_scale, int num_rx, int RA,
                                                         int num_tx, int TA, int Tn, int L, int ntb,
                                                         float2* __restrict__ y) {
  // the 256 x L taps of a block's time steps are one contiguous piece of h per link: staged through LDS with
  // coalesced loads (a lane reading its own L taps directly would touch 64 cache lines per load instruction)
  extern __shared__ __attribute__((aligned(16))) float2 hs[];   // [256][L]
  const int Tout = Tn + L - 1;
  const int t0 = (blockIdx.x % ntb) * blockDim.x;
  const int t = t0 + threadIdx.x;
  const int rxa = (blockIdx.x / ntb) % (num_rx * RA);
  const long long b = blockIdx.x / ntb / (num_rx * RA);
  const int rx = rxa / RA, ra = rxa % RA;
  const int cnt = min((int)blockDim.x, Tout - t0) * L;
  float re = 0.0f, im = 0.0f;
  for (int tx = 0; tx < num_tx; ++tx) {
    // deferred channel normalisation (samd_cir_to_time_c64 norm_scale): one factor per (b, rx, tx)
    const float sc = link_scale ? link_scale[(b * num_rx + rx) * num_tx + tx] : 1.0f;
    for (int ta = 0; ta < TA; ++ta) {
      const long long link = ((((b * num_rx + rx) * RA + ra) * num_tx + tx) * TA + ta);
      const float2* hp = h + (link * Tout + t0) * L;
      __syncthreads();
      for (int i = threadIdx.x; i < cnt; i += blockDim.x) hs[i] = hp[i];
      __syncthreads();
      if (t < Tout) {
        const float2* xp = x + (b * num_tx * TA + tx * TA + ta) * Tn;
        const int lo = t - (Tn - 1) > 0 ? t - (Tn - 1) : 0;
        const int hi = t < L - 1 ? t : L - 1;
        for (int l = lo; l <= hi; ++l) {
          float2 hv = hs[threadIdx.x * L + l];
          if (link_scale) { hv.x *= sc; hv.y *= sc; }      // the value the normalising second pass would have stored
          const float2 xv = xp[t - l];
          re += hv.x * xv.x - hv.y * xv.y;
          im += hv.x * xv.y + hv.y * xv.x;
        }
      }
    }
  }
  if (t < Tout) y[(b * num_rx * RA + rxa) * Tout + t] = make_float2(re, im);
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_ofdm_modulate_c64(const float* x, int rows, int num_ofdm_symbols, int fft_size,
                                      const int32_t* cp_len, const int32_t* sym_off, int max_cp, int out_len,
                                      float* work, float* out, void* stream) {
  SAMD_REQUIRE(x && cp_len && sym_off && work && out, "null argument");
  SAMD_REQUIRE(rows >= 0 && num_ofdm_symbols > 0 && fft_size > 0 && max_cp >= 0 && max_cp <= fft_size, "bad shape");
  if (rows == 0) return SAMD_OK;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)rows * num_ofdm_symbols * fft_size;
  ifftshift_kernel<float2, float><<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const float2*)x, total, fft_size, (float2*)work);
  if (int rc = launch_status()) return rc;
  if (int rc = fft_inplace(work, fft_size, rows * num_ofdm_symbols, true, st)) return rc;
  const int ntb = (fft_size + max_cp + 255) / 256;
  SAMD_REQUIRE((long long)ntb * num_ofdm_symbols * rows < (1ll << 31), "grid too large");
  add_cp_kernel<float2, float><<<(unsigned)((long long)ntb * num_ofdm_symbols * rows), 256, 0, st>>>(
      (const float2*)work, cp_len, sym_off, num_ofdm_symbols, fft_size, out_len, 1.0f / sqrtf((float)fft_size), ntb,
      (float2*)out);
  return launch_status();
}

extern "C" int samd_ofdm_demodulate_c64(const float* y, int rows, int in_len, int num_ofdm_symbols, int fft_size,
                                        const int32_t* cp_len, const int32_t* sym_off, int l_min, float* work,
                                        float* out, void* stream) {
  SAMD_REQUIRE(y && cp_len && sym_off && work && out, "null argument");
  SAMD_REQUIRE(rows >= 0 && num_ofdm_symbols > 0 && fft_size > 0 && l_min <= 0, "bad shape");
  if (rows == 0) return SAMD_OK;
  hipStream_t st = (hipStream_t)stream;
  const int ntb = (fft_size + 255) / 256;
  SAMD_REQUIRE((long long)ntb * num_ofdm_symbols * rows < (1ll << 31), "grid too large");
  remove_cp_kernel<float2, float><<<(unsigned)((long long)ntb * num_ofdm_symbols * rows), 256, 0, st>>>(
      (const float2*)y, cp_len, sym_off, num_ofdm_symbols, fft_size, in_len, ntb, (float2*)work);
  if (int rc = launch_status()) return rc;
  if (int rc = fft_inplace(work, fft_size, rows * num_ofdm_symbols, false, st)) return rc;
  const long long total = (long long)rows * num_ofdm_symbols * fft_size;
  // demodulator.py:143-146: -2 * PI * l_min / fft_size * range(fft_size), float32 left to right
  const float phase_step = (-2.0f * 3.14159265358979323846f * (float)l_min) / (float)fft_size;
  demod_post_kernel<float2, float><<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const float2*)work, total, fft_size, phase_step,
                                                                     1.0f / sqrtf((float)fft_size), (float2*)out);
  return launch_status();
}

// precision = "double" (reference block.py:25-52): the same passes on complex128 with a double-precision rocFFT plan
extern "C" int samd_ofdm_modulate_c128(const double* x, int rows, int num_ofdm_symbols, int fft_size, const int32_t* cp_len,
                                       const int32_t* sym_off, int max_cp, int out_len, double* work, double* out, void* stream) {
  SAMD_REQUIRE(x && cp_len && sym_off && work && out, "null argument");
  SAMD_REQUIRE(rows >= 0 && num_ofdm_symbols > 0 && fft_size > 0 && max_cp >= 0 && max_cp <= fft_size, "bad shape");
  if (rows == 0) return SAMD_OK;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)rows * num_ofdm_symbols * fft_size;
  ifftshift_kernel<double2, double><<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const double2*)x, total, fft_size, (double2*)work);
  if (int rc = launch_status()) return rc;
  if (int rc = fft_inplace(work, fft_size, rows * num_ofdm_symbols, true, st, true)) return rc;
  const int ntb = (fft_size + max_cp + 255) / 256;
  SAMD_REQUIRE((long long)ntb * num_ofdm_symbols * rows < (1ll << 31), "grid too large");
  add_cp_kernel<double2, double><<<(unsigned)((long long)ntb * num_ofdm_symbols * rows), 256, 0, st>>>(
      (const double2*)work, cp_len, sym_off, num_ofdm_symbols, fft_size, out_len, 1.0 / sqrt((double)fft_size), ntb, (double2*)out);
  return launch_status();
}

extern "C" int samd_ofdm_demodulate_c128(const double* y, int rows, int in_len, int num_ofdm_symbols, int fft_size,
                                         const int32_t* cp_len, const int32_t* sym_off, int l_min, double* work, double* out,
                                         void* stream) {
  SAMD_REQUIRE(y && cp_len && sym_off && work && out, "null argument");
  SAMD_REQUIRE(rows >= 0 && num_ofdm_symbols > 0 && fft_size > 0 && l_min <= 0, "bad shape");
  if (rows == 0) return SAMD_OK;
  hipStream_t st = (hipStream_t)stream;
  const int ntb = (fft_size + 255) / 256;
  SAMD_REQUIRE((long long)ntb * num_ofdm_symbols * rows < (1ll << 31), "grid too large");
  remove_cp_kernel<double2, double><<<(unsigned)((long long)ntb * num_ofdm_symbols * rows), 256, 0, st>>>(
      (const double2*)y, cp_len, sym_off, num_ofdm_symbols, fft_size, in_len, ntb, (double2*)work);
  if (int rc = launch_status()) return rc;
  if (int rc = fft_inplace(work, fft_size, rows * num_ofdm_symbols, false, st, true)) return rc;
  const long long total = (long long)rows * num_ofdm_symbols * fft_size;
  const double phase_step = (-2.0 * 3.14159265358979323846 * (double)l_min) / (double)fft_size;
  demod_post_kernel<double2, double><<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const double2*)work, total, fft_size, phase_step,
                                                                                     1.0 / sqrt((double)fft_size), (double2*)out);
  return launch_status();
}

extern "C" int samd_cir_to_time_c64(float bandwidth, const float* a, const float* tau, int l_min, int l_max,
                                    int batch, int num_rx, int num_rx_ant, int num_tx, int num_tx_ant, int num_paths,
                                    int num_time_steps, int normalize, float* h_time, float* norm_scale, void* stream) {
  SAMD_REQUIRE(a && tau && h_time && batch > 0 && l_max >= l_min && num_paths > 0 && num_time_steps > 0,
               "bad argument");
  const int L = l_max - l_min + 1;
  const size_t lds = (size_t)((num_paths * L + 1) & ~1) * sizeof(float) + (size_t)4 * 64 * L * sizeof(float2);   // sinc table + per-wave stage
  SAMD_REQUIRE(lds <= 64 * 1024, "num_paths * l_tot too large for the LDS sinc table");
  cir_to_time_kernel<<<batch * num_rx * num_tx, 256, lds, (hipStream_t)stream>>>(
      (const float2*)a, tau, bandwidth, l_min, L, num_rx, num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps,
      normalize, (float2*)h_time, normalize ? norm_scale : nullptr);
  return launch_status();
}

extern "C" int samd_apply_time_channel_c64(const float* x, const float* h_time, const float* link_scale, int batch, int num_rx, int num_rx_ant,
                                           int num_tx, int num_tx_ant, int num_time_samples, int l_tot, float* y,
                                           void* stream) {
  SAMD_REQUIRE(x && h_time && y && batch > 0 && num_time_samples > 0 && l_tot > 0, "bad argument");
  const int Tout = num_time_samples + l_tot - 1;
  const int ntb = (Tout + 255) / 256;
  SAMD_REQUIRE((long long)ntb * num_rx * num_rx_ant * batch < (1ll << 31), "grid too large");
  SAMD_REQUIRE((size_t)256 * l_tot * sizeof(float2) <= 64 * 1024, "l_tot too large for the LDS tap stage");
  apply_time_kernel<<<(unsigned)((long long)ntb * num_rx * num_rx_ant * batch), 256, (size_t)256 * l_tot * sizeof(float2),
                      (hipStream_t)stream>>>(
      (const float2*)x, (const float2*)h_time, link_scale, num_rx, num_rx_ant, num_tx, num_tx_ant, num_time_samples, l_tot, ntb,
      (float2*)y);
  return launch_status();
}
