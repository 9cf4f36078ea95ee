// api.hip — error plumbing, device query, and the host-side filter-matrix builder.
#include <math.h>
#include <stdarg.h>

#include <vector>

#include "common.hpp"

namespace afldm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return AFLDM_ELAUNCH;
  }
  return AFLDM_OK;
}

// 1-D masks in FFT-bin order; the rules (incl. the N%4 Nyquist special case) follow the
// reference's create_lpf_rect (ideal_lpf.py:12-24) and create_recon_rect (ideal_lpf.py:38-49).
static void mask_1d(int N, double cutoff, bool recon, std::vector<double>& r) {
  r.assign(N, 1.0);
  int lo = (int)floor((N * cutoff) / 2.0);
  int hi = N - lo;
  for (int k = lo + 1; k < hi; ++k) r[k] = 0.0;
  if (N % 4 == 0 && lo >= 0 && hi < N) {
    r[lo] = recon ? 0.5 : 0.0;
    r[hi] = recon ? 0.5 : 0.0;
  }
}

// h = real(ifft(mask)); circulant C[i][j] = h[(i - j) mod M]  
static void impulse_response(const std::vector<double>& mask, std::vector<double>& h) {
  const int M = (int)mask.size();
  h.assign(M, 0.0);
  for (int n = 0; n < M; ++n) {
    double s = 0.0;
    for (int k = 0; k < M; ++k) s += mask[k] * cos(2.0 * M_PI * (double)k * (double)n / (double)M);
    h[n] = s / M;
  }
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_version(void) { return 100; }

extern "C" const char* afldm_last_error(void) { return g_err; }

extern "C" int afldm_device_info(char* name, int name_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("afldm_device_info: no HIP device");
    return AFLDM_ELAUNCH;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    set_error("afldm_device_info: hipGetDeviceProperties failed");
    return AFLDM_ELAUNCH;
  }
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return p.multiProcessorCount;
}

extern "C" int afldm_filter_matrix(int kind, int N, int up, float* out) {
  AFLDM_REQUIRE(out != nullptr, AFLDM_ENULL, "afldm_filter_matrix: out is NULL");
  std::vector<double> mask, h;
  if (kind == 0) {  // U [up*N x N] = up * C(recon(up*N, 1/up))[:, ::up]
    AFLDM_REQUIRE(N >= 1 && up >= 2 && up <= 16, AFLDM_ESHAPE, "afldm_filter_matrix: bad N=%d up=%d", N, up);
    const int M = N * up;
    const int lo = (int)floor((M * (1.0 / up)) / 2.0);
    AFLDM_REQUIRE(!(M % 4 == 0 && lo == 0), AFLDM_ESHAPE,
                  "afldm_filter_matrix: N=%d up=%d has no valid recon mask (reference raises too)", N, up);
    mask_1d(M, 1.0 / up, true, mask);
    impulse_response(mask, h);
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) out[(size_t)i * N + j] = (float)(up * h[((i - j * up) % M + M) % M]);
    return AFLDM_OK;
  }
  if (kind == 1) {  // D [N/2 x N] = C(lpf(N, 1/2))[::2, :]
    AFLDM_REQUIRE(N >= 2 && N % 2 == 0, AFLDM_ESHAPE, "afldm_filter_matrix: LPF plane size %d must be even", N);
    mask_1d(N, 0.5, false, mask);
    impulse_response(mask, h);
    for (int i = 0; i < N / 2; ++i)
      for (int j = 0; j < N; ++j) out[(size_t)i * N + j] = (float)h[((2 * i - j) % N + N) % N];
    return AFLDM_OK;
  }
  if (kind == 2) {  // L [N x N] = C(lpf(N, 1/2))
    AFLDM_REQUIRE(N >= 2, AFLDM_ESHAPE, "afldm_filter_matrix: LPF plane size %d too small", N);
    const int lo = (int)floor((N * 0.5) / 2.0);
    AFLDM_REQUIRE(!(N % 4 == 0 && lo == 0), AFLDM_ESHAPE, "afldm_filter_matrix: N=%d has no valid LPF mask", N);
    mask_1d(N, 0.5, false, mask);
    impulse_response(mask, h);
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) out[(size_t)i * N + j] = (float)h[((i - j) % N + N) % N];
    return AFLDM_OK;
  }
  set_error("afldm_filter_matrix: unknown kind %d", kind);
  return AFLDM_ESHAPE;
}
