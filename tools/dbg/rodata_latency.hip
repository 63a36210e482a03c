// Is a table in the code object's read-only data slower to gather from than one in hipMalloc'd memory?  (the tetrahedron region table of hfcl_gjk.hpp:
// the waves of k_gjk_cvx wait ~6 us behind its one global_load_ubyte in their first rank-4 trip, profiles/r06_e)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct Lut { uint8_t v[4096]; constexpr Lut() : v() { for (unsigned m = 0; m < 4096; ++m) v[m] = uint8_t((m * 2654435761u) >> 24); } };
__device__ static const Lut g_lut = Lut();
template <int MODE> __global__ void __launch_bounds__(256) k(const uint8_t* tbl, uint32_t* out, int chain) {
  uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 37u & 4095u;
  for (int i = 0; i < chain; ++i) {
    const uint32_t b = MODE == 0 ? g_lut.v[idx] : tbl[idx];
    idx = (b * 131u + idx * 7u + 1u) & 4095u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = idx;
}
int main() {
  static constexpr Lut h = Lut();
  uint8_t* d_tbl; uint32_t* d_out;
  const int blocks = 2048;
  hipMalloc(&d_tbl, 4096); hipMalloc(&d_out, blocks * 256 * 4);
  hipMemcpy(d_tbl, h.v, 4096, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int chain : {1, 8, 64}) for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_tbl, d_out, chain);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_tbl, d_out, chain);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    std::printf("chain %2d  %s  %.4f ms  (%.1f ns per dependent gather per wave, %d waves)\n", chain, mode == 0 ? "rodata  " : "hipMalloc", best, best * 1e6 / chain / (blocks * 4 / 2048.0) , blocks * 4);
  }
  return 0;
}
