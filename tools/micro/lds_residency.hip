// micro-benchmark: how many workgroups of T threads with L bytes of static LDS are RESIDENT on one CU of gfx950 at the same time (census), against what
// hipOccupancyMaxActiveBlocksPerMultiprocessor answers.  Every block stamps its start / end (s_memrealtime, 100 MHz) and its hardware id, spins ~30 us, and the
// host sweeps the intervals per CU.  Build: hipcc --offload-arch=gfx950 -O2 -o lds_residency lds_residency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
template <int LDS_BYTES, int NV> __global__ void k_census(Rec *rec, double *sink, int spin) {
  __shared__ double L[LDS_BYTES / 8];
  if constexpr (NV == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79"); if constexpr (NV == 88) asm volatile("v_mov_b32 v87, 0" ::: "v87"); if constexpr (NV == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if constexpr (NV == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127"); if constexpr (NV == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63"); if constexpr (NV == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < LDS_BYTES / 8; i += blockDim.x) L[i] = i;
  __syncthreads();
  double s = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) s += L[(threadIdx.x * 7 + (int)s) % (LDS_BYTES / 8)];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    rec[blockIdx.x] = Rec{t0, __builtin_amdgcn_s_memrealtime(), hw, xcc};
  }
  if (s == 12345.678) sink[0] = s;
}
template <int LDS_BYTES, int NV> void run(int threads) {
  const int blocks = 2048, spin = 3000;   // 30 us
  Rec *d; double *sink; (void)hipMalloc(&d, blocks * sizeof(Rec)); (void)hipMalloc(&sink, 8);
  int api = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, (const void *)k_census<LDS_BYTES, NV>, threads, 0);
  hipLaunchKernelGGL((k_census<LDS_BYTES, NV>), dim3(blocks), dim3(threads), 0, 0, d, sink, spin);
  (void)hipDeviceSynchronize();
  std::vector<Rec> h(blocks); (void)hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (+ higher SE bits on big chips): key = everything above the simd / wave / pipe fields
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (auto &r : h) { const unsigned long long key = ((unsigned long long)(r.xcc & 0xf) << 32) | (r.hw & 0xff00u); /* HW_ID: cu 11:8, sh 12, se 15:13 (tg_id 19:16 differs between resident workgroups) */ ev[key].push_back({r.t0, +1}); ev[key].push_back({r.t1, -1}); }
  unsigned long long tmin = ~0ull, tmax = 0, busy = 0; for (auto &r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); busy += r.t1 - r.t0; }
  const double span_us = (tmax - tmin) / 100.0, mean_resident = (double)busy / (double)(tmax - tmin) / (double)ev.size();
  int worst = 0; double mean = 0;
  for (auto &kv : ev) { auto &v = kv.second; std::sort(v.begin(), v.end()); int cur = 0, mx = 0; for (auto &e : v) { cur += e.second; mx = std::max(mx, cur); } worst = std::max(worst, mx); mean += mx; }
  std::printf("LDS %6d B, >= %3d VGPRs, %4d threads: occupancy API %d blocks / CU; census: %zu distinct CUs, max resident per CU %d, mean of the per-CU maxima %.2f; %d blocks of ~30 us: span %.1f us, time-averaged resident blocks per CU %.2f\n", LDS_BYTES, NV, threads, api, ev.size(), worst, mean / ev.size(), blocks, span_us, mean_resident);
  (void)hipFree(d); (void)hipFree(sink);
}
int main() {
  run<52480, 0>(320); run<52480, 64>(320); run<52480, 80>(320); run<52480, 88>(320); run<52480, 96>(320); run<52480, 128>(320); run<52480, 168>(320);
  run<26624, 64>(320); run<26624, 80>(320); run<26624, 96>(320); run<26624, 128>(320);
  run<52480, 88>(256); run<52480, 128>(256); run<26624, 128>(256); run<26624, 96>(512);
  run<54272, 0>(320); run<40960, 96>(320);
  return 0;
}
