// mfma_support_probe.hip -- a number for north_star's MFMA clause (DESIGN.md "What MFMA can do for this path").
//
// The only GEMM-shaped piece of GJK / EPA on 32-vertex hulls is the support scan: 32 dot products of one direction with the hull's
// vertices and an arg-max (getShapeSupportLinear, src/narrowphase/support_functions.cpp:400-421).  If the 32 pairs of a wave share the
// hull (a batch sorted by hull id: cfg3 draws its pairs from a 4 096-hull library), the 32 x 32 dot products of a wave are one
// [32 vertices x 3] x [3 x 32 directions] product = two v_mfma_f32_32x32x2_f32 (K = 2: x,y then z,0).  This program measures exactly
// that piece in isolation, both ways, inside the dependent loop it lives in (the next direction depends on the support found):
//   valu : the form of k_gjk_cvx<2, ., .>: two lanes per pair, 16 vertices of the pair's OWN hull per lane in registers, 16 dot products per
//          lane (3 FMA each), a running arg-max, one exchange with the partner lane, the winner's vertex broadcast;
//   mfma : 32 directions against ONE hull per wave: two MFMAs give every lane 16 of the 32 dot products of its direction (C layout:
//          col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), a running arg-max over the 16, one exchange with lane ^ 32,
//          the winner's vertex from an LDS copy of the hull.  Lanes j and j + 32 carry direction j redundantly, as the two lanes of a
//          pair do in the valu form.
// Both forms take the first vertex among equal products and evaluate a product as fma(z, dz, fma(y, dy, x * dx)) -- the MFMA's
// accumulation order (the guide: exact f32, bitwise an fmaf chain) -- so their supports must agree index for index: checked.
// Output: supports per second chip-wide, wave-instructions are counted by the profiler if wanted; build + run on the GPU box:
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o /tmp/mfma_probe tools/mfma_support_probe.hip && /tmp/mfma_probe
// (-ffp-contract=off: the direction update is amplifying, a product contracted in one kernel and not in the other sends the two forms
// down different sequences; the dot products themselves are explicit FMAs)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#define HIP_OK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NV = 32;

// the serial work between two supports, kept small and identical in both forms: the next direction from the support found
__device__ __forceinline__ void next_direction(float& dx, float& dy, float& dz, float sx, float sy, float sz, int t) {
#pragma clang fp contract(off)  // (the same roundings in both kernels: the update amplifies a last-bit difference into another support)
  // a rotation-ish update that keeps |d| bounded and depends on the support point (as GJK's v -> -v does)
  const float nx = __fadd_rn(__fadd_rn(__fsub_rn(__fmul_rn(dy, sz), __fmul_rn(dz, sy)), __fmul_rn(0.37f, sx)), __fmul_rn(0.01f, float(t & 7)));
  const float ny = __fadd_rn(__fsub_rn(__fmul_rn(dz, sx), __fmul_rn(dx, sz)), __fmul_rn(0.41f, sy));
  const float nz = __fadd_rn(__fsub_rn(__fmul_rn(dx, sy), __fmul_rn(dy, sx)), __fmul_rn(0.29f, sz));
  const float l2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(nx, nx), __fmul_rn(ny, ny)), __fmul_rn(nz, nz)), 1e-20f);
  const float inv = __frsqrt_rn(l2);
  dx = __fmul_rn(nx, inv);
  dy = __fmul_rn(ny, inv);
  dz = __fmul_rn(nz, inv);
}

// ---- valu: two lanes per pair, the pair's own hull in registers (k_gjk_cvx's W = 2 form)
__global__ void __launch_bounds__(256) k_valu(const float* __restrict__ hulls, const uint32_t* __restrict__ hull_of_pair, const float* __restrict__ dirs,
                                              uint32_t n_pairs, int iters, uint32_t* __restrict__ out_idx_sum, uint8_t* __restrict__ trace, uint32_t n_trace) {
  const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pair = gl >> 1;
  const int half = gl & 1;
  if (pair >= n_pairs) return;
  const float* h = hulls + size_t(hull_of_pair[pair]) * NV * 3;
  float vx[16], vy[16], vz[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {  // lane `half` keeps vertices half * 16 ... half * 16 + 15
    vx[k] = h[3 * (half * 16 + k) + 0];
    vy[k] = h[3 * (half * 16 + k) + 1];
    vz[k] = h[3 * (half * 16 + k) + 2];
  }
  float dx = dirs[3 * pair], dy = dirs[3 * pair + 1], dz = dirs[3 * pair + 2];
  uint32_t acc = 0;
  for (int t = 0; t < iters; ++t) {
    float best = -3.4e38f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float p = __fmaf_rn(vz[k], dz, __fmaf_rn(vy[k], dy, __fmul_rn(vx[k], dx)));
      const bool take = p > best;  // strict: the first index among equals
      best = take ? p : best;
      bi = take ? k : bi;
    }
    bi += half * 16;
    const float ob = __shfl_xor(best, 1);
    const int oi = __shfl_xor(bi, 1);
    const bool other = ob > best || (ob == best && oi < bi);
    const int wi = other ? oi : bi;
    // the winner's vertex: from the lane that holds it
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const bool m = (wi & 15) == k;
      sx = m ? vx[k] : sx;
      sy = m ? vy[k] : sy;
      sz = m ? vz[k] : sz;
    }
    const int src = (threadIdx.x & ~1) | (wi >> 4);
    sx = __shfl(sx, src);
    sy = __shfl(sy, src);
    sz = __shfl(sz, src);
    acc = acc * 31u + uint32_t(wi);
    if (half == 0 && pair < n_trace) trace[size_t(pair) * iters + t] = uint8_t(wi);
    next_direction(dx, dy, dz, sx, sy, sz, t);
  }
  if (half == 0) out_idx_sum[pair] = acc;
}

// ---- mfma: 32 directions of a wave against one hull
__global__ void __launch_bounds__(256) k_mfma(const float* __restrict__ hulls, const uint32_t* __restrict__ hull_of_wave, const float* __restrict__ dirs,
                                              uint32_t n_pairs, int iters, uint32_t* __restrict__ out_idx_sum, uint8_t* __restrict__ trace, uint32_t n_trace) {
  __shared__ float lds_hull[4][NV * 3 + 4];
  const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t pair = wave * 32u + uint32_t(lane & 31);
  if (wave * 32u >= n_pairs) return;
  const float* h = hulls + size_t(hull_of_wave[wave]) * NV * 3;
  float* const lh = lds_hull[wave_in_block];
  for (int k = lane; k < NV * 3; k += 64) lh[k] = h[k];
  // A operands of the two MFMAs: lane l holds A[i = l & 31][k = l >> 5]
  const int vi = lane & 31, hi = lane >> 5;
  const float a0 = h[3 * vi + hi];                 // x (k = 0) / y (k = 1)
  const float a1 = hi == 0 ? h[3 * vi + 2] : 0.f;  // z (k = 0) / 0 (k = 1)
  const bool valid = pair < n_pairs;
  float dx = valid ? dirs[3 * pair] : 1.f, dy = valid ? dirs[3 * pair + 1] : 0.f, dz = valid ? dirs[3 * pair + 2] : 0.f;
  __builtin_amdgcn_s_barrier();
  uint32_t acc = 0;
  for (int t = 0; t < iters; ++t) {
    // B operands: lane l holds B[k = l >> 5][j = l & 31]
    const float b0 = hi == 0 ? dx : dy;
    const float b1 = hi == 0 ? dz : 0.f;
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c, 0, 0, 0);  // x dx + y dy
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c, 0, 0, 0);  // + z dz (+ 0)
    // this lane's 16 rows, in increasing row order: row = (r & 3) + 8 (r >> 2) + 4 hi
    float best = -3.4e38f;
    int bi = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = c[r];
      const bool take = p > best;
      best = take ? p : best;
      bi = take ? ((r & 3) + 8 * (r >> 2)) : bi;
    }
    bi += 4 * hi;
    const float ob = __shfl_xor(best, 32);
    const int oi = __shfl_xor(bi, 32);
    const bool other = ob > best || (ob == best && oi < bi);
    const int wi = other ? oi : bi;
    const float sx = lh[3 * wi], sy = lh[3 * wi + 1], sz = lh[3 * wi + 2];
    acc = acc * 31u + uint32_t(wi);
    if (hi == 0 && valid && pair < n_trace) trace[size_t(pair) * iters + t] = uint8_t(wi);
    next_direction(dx, dy, dz, sx, sy, sz, t);
  }
  if (hi == 0 && valid) out_idx_sum[pair] = acc;
}

int main(int argc, char** argv) {
  const uint32_t n_pairs = argc > 1 ? uint32_t(std::atol(argv[1])) : 1u << 20;
  const int iters = argc > 2 ? std::atoi(argv[2]) : 64;
  const int n_hulls = 4096;
  std::vector<float> hulls(size_t(n_hulls) * NV * 3), dirs(size_t(n_pairs) * 3);
  uint64_t rng = 99;
  auto uni = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return float(double(rng >> 11) / double(1ull << 53)); };
  for (auto& x : hulls) x = 2.f * uni() - 1.f;
  for (uint32_t p = 0; p < n_pairs; ++p) {
    float x = 2.f * uni() - 1.f, y = 2.f * uni() - 1.f, z = 2.f * uni() - 1.f;
    const float l = std::sqrt(x * x + y * y + z * z) + 1e-9f;
    dirs[3 * p] = x / l; dirs[3 * p + 1] = y / l; dirs[3 * p + 2] = z / l;
  }
  // the batch sorted by hull: the 32 pairs of a wave share one
  std::vector<uint32_t> hull_of_wave((n_pairs + 31) / 32), hull_of_pair(n_pairs);
  for (size_t w = 0; w < hull_of_wave.size(); ++w) hull_of_wave[w] = uint32_t(w % n_hulls);
  for (uint32_t p = 0; p < n_pairs; ++p) hull_of_pair[p] = hull_of_wave[p / 32];
  float *d_hulls, *d_dirs;
  uint32_t *d_how, *d_hop, *d_o1, *d_o2;
  HIP_OK(hipMalloc(&d_hulls, hulls.size() * 4)); HIP_OK(hipMalloc(&d_dirs, dirs.size() * 4));
  HIP_OK(hipMalloc(&d_how, hull_of_wave.size() * 4)); HIP_OK(hipMalloc(&d_hop, hull_of_pair.size() * 4));
  HIP_OK(hipMalloc(&d_o1, n_pairs * 4)); HIP_OK(hipMalloc(&d_o2, n_pairs * 4));
  HIP_OK(hipMemcpy(d_hulls, hulls.data(), hulls.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_dirs, dirs.data(), dirs.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_how, hull_of_wave.data(), hull_of_wave.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_hop, hull_of_pair.data(), hull_of_pair.size() * 4, hipMemcpyHostToDevice));
  const uint32_t n_trace = n_pairs < 65536u ? n_pairs : 65536u;
  uint8_t *d_t1, *d_t2;
  HIP_OK(hipMalloc(&d_t1, size_t(n_trace) * iters)); HIP_OK(hipMalloc(&d_t2, size_t(n_trace) * iters));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const int grid_valu = int((size_t(n_pairs) * 2 + 255) / 256), grid_mfma = int((size_t(n_pairs) * 2 + 255) / 256);  // 2 lanes per direction in both
  float ms_valu = 1e30f, ms_mfma = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    float ms;
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_valu, dim3(grid_valu), dim3(256), 0, 0, d_hulls, d_hop, d_dirs, n_pairs, iters, d_o1, d_t1, n_trace);
    HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1)); HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms_valu = std::fmin(ms_valu, ms);
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mfma, dim3(grid_mfma), dim3(256), 0, 0, d_hulls, d_how, d_dirs, n_pairs, iters, d_o2, d_t2, n_trace);
    HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1)); HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms_mfma = std::fmin(ms_mfma, ms);
  }
  HIP_OK(hipGetLastError());
  std::vector<uint32_t> o1(n_pairs), o2(n_pairs);
  HIP_OK(hipMemcpy(o1.data(), d_o1, n_pairs * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(o2.data(), d_o2, n_pairs * 4, hipMemcpyDeviceToHost));
  size_t differ = 0;
  for (uint32_t p = 0; p < n_pairs; ++p) differ += o1[p] != o2[p];
  {  // where the traced sequences part: the iteration of the first different support
    std::vector<uint8_t> t1(size_t(n_trace) * iters), t2(size_t(n_trace) * iters);
    HIP_OK(hipMemcpy(t1.data(), d_t1, t1.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(t2.data(), d_t2, t2.size(), hipMemcpyDeviceToHost));
    std::vector<size_t> first(size_t(iters) + 1, 0);
    for (uint32_t p = 0; p < n_trace; ++p) {
      int t = 0;
      while (t < iters && t1[size_t(p) * iters + t] == t2[size_t(p) * iters + t]) ++t;
      ++first[size_t(t)];
    }
    std::printf("  first differing support, by iteration (of %u traced pairs; last column = never):", n_trace);
    for (int t = 0; t <= iters; ++t)
      if (first[size_t(t)]) std::printf(" %d:%zu", t, first[size_t(t)]);
    std::printf("\n");
  }
  const double sup = double(n_pairs) * iters;
  std::printf("pairs %u, %d dependent supports each, 32-vertex hulls, fp32\n", n_pairs, iters);
  std::printf("  valu (2 lanes per pair, own hull in registers): %8.3f ms  %7.2f G supports/s\n", ms_valu, sup / ms_valu * 1e-6);
  std::printf("  mfma (32 directions per hull, 2 x v_mfma_f32_32x32x2_f32): %8.3f ms  %7.2f G supports/s   (%.2fx)\n", ms_mfma, sup / ms_mfma * 1e-6,
              ms_valu / ms_mfma);
  std::printf("  support sequences that differ between the forms: %zu of %u\n", differ, n_pairs);
  return differ ? 2 : 0;
}
