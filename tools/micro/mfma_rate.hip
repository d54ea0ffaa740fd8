// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 (and the 32x32x2 f32 form) on gfx950,
// waves per SIMD 1 / 2, independent accumulators.  Prints cycles per MFMA per SIMD and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int NACC> __global__ void k_f64(double *out, int iters) {
  v4d acc[NACC]; for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k_f32(float *out, int iters) {
  v4f acc[NACC]; for (int i = 0; i < NACC; ++i) acc[i] = v4f{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k_f32_32(float *out, int iters) {
  v16f acc[NACC]; for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  float s = 0; for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> double timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e-3;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
  printf("%s: %d CUs, clock %.0f MHz\n", p.name, cus, clk / 1e6);
  double *o; hipMalloc(&o, 1 << 26);
  const int iters = 20000;
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;   // 4 SIMDs x wps waves
    double t = timeit([&] { hipLaunchKernelGGL(k_f64<8>, dim3(cus), dim3(threads), 0, 0, o, iters); });
    double n = (double)iters * 8 * wps;   // MFMAs per SIMD
    printf("f64 16x16x4, %d waves/SIMD: %.1f ns per MFMA per SIMD (%.1f cycles at nominal clock), %.1f TFLOP/s\n", wps, t / n * 1e9, t / n * clk, 2048.0 * n * 4 * cus / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(k_f32<8>, dim3(cus), dim3(threads), 0, 0, (float *)o, iters); });
    printf("f32 16x16x4, %d waves/SIMD: %.1f ns per MFMA per SIMD (%.1f cycles), %.1f TFLOP/s\n", wps, t / n * 1e9, t / n * clk, 2048.0 * n * 4 * cus / t / 1e12);
    double n2 = (double)iters * 4 * wps;
    t = timeit([&] { hipLaunchKernelGGL(k_f32_32<4>, dim3(cus), dim3(threads), 0, 0, (float *)o, iters); });
    printf("f32 32x32x2, %d waves/SIMD: %.1f ns per MFMA per SIMD (%.1f cycles), %.1f TFLOP/s\n", wps, t / n2 * 1e9, t / n2 * clk, 4096.0 * n2 * 4 * cus / t / 1e12);
  }
  return 0;
}
