// occupancy_probe.hip -- how many single-wave workgroups does a CU of this chip really hold at once, as a function of the
// workgroup's LDS size (and with / without a private segment)?  The EPA kernels are sized by this: 8 polytopes per
// 64-thread workgroup, and the number of resident workgroups per CU is what their run time scales with.
//
// Method: every block bumps a per-CU counter (CU identified by HW_ID: XCC, SE, CU), records the maximum it ever saw,
// spins ~200 us so that all blocks of the launch overlap, and leaves.  Reported: max over CUs and the mean of the per-CU
// maxima, next to what hipOccupancyMaxActiveBlocksPerMultiprocessor predicts.
// Build: hipcc -O3 --offload-arch=gfx950 tools/occupancy_probe.hip -o build/occupancy_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

extern __shared__ char dyn_lds[];

template <int SCRATCH_WORDS>
__global__ void __launch_bounds__(64) k_probe(int* active, int* peak, int* which_cu, unsigned long long spin_ticks) {
  // HW_ID (hwreg 4): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; XCC_ID (hwreg 20)[3:0]
  unsigned hw = 0, xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
  volatile int priv[SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1];
  if (SCRATCH_WORDS > 0)
    for (int i = 0; i < SCRATCH_WORDS; ++i) priv[(i * 7 + threadIdx.x) % SCRATCH_WORDS] = i;  // dynamic index: stays in scratch
  if (threadIdx.x == 0) {
    dyn_lds[0] = 1;
    const int now = atomicAdd(&active[cu], 1) + 1;
    atomicMax(&peak[cu], now);
    which_cu[blockIdx.x] = int(cu);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) {
    }
    atomicSub(&active[cu], 1);
  }
  if (SCRATCH_WORDS > 0 && priv[threadIdx.x % SCRATCH_WORDS] == -12345) peak[0] = 0;
}

template <int SW>
static void probe(size_t lds, int n_cus, int* d_active, int* d_peak, int* d_which) {
  const int slots = 1 << 13;
  CK(hipMemset(d_active, 0, slots * sizeof(int)));
  CK(hipMemset(d_peak, 0, slots * sizeof(int)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<SW>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  int predicted = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&predicted, reinterpret_cast<const void*>(k_probe<SW>), 64, lds));
  const int blocks = n_cus * 40;
  hipLaunchKernelGGL(k_probe<SW>, dim3(blocks), dim3(64), lds, 0, d_active, d_peak, d_which, 20000ull /* 200 us at 100 MHz */);
  CK(hipDeviceSynchronize());
  std::vector<int> peak(slots);
  CK(hipMemcpy(peak.data(), d_peak, slots * sizeof(int), hipMemcpyDeviceToHost));
  int mx = 0, used = 0;
  double sum = 0;
  for (int v : peak)
    if (v > 0) {
      mx = std::max(mx, v);
      sum += v;
      ++used;
    }
  printf("LDS %6zu B/workgroup  scratch %3d B/lane : resident single-wave workgroups per CU: max %2d  mean %.2f over %d CUs   (runtime API predicts %d)\n",
         lds, SW * 4, mx, used ? sum / used : 0.0, used, predicted);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int *d_active, *d_peak, *d_which;
  CK(hipMalloc(&d_active, (1 << 13) * sizeof(int)));
  CK(hipMalloc(&d_peak, (1 << 13) * sizeof(int)));
  CK(hipMalloc(&d_which, prop.multiProcessorCount * 40 * sizeof(int)));
  for (size_t lds : {size_t(1024), size_t(8192), size_t(10240), size_t(12288), size_t(13184), size_t(13312), size_t(13824), size_t(14336), size_t(15360),
                     size_t(16000), size_t(16384), size_t(17408), size_t(19456), size_t(20480)}) {
    probe<0>(lds, prop.multiProcessorCount, d_active, d_peak, d_which);
  }
  probe<28>(13184, prop.multiProcessorCount, d_active, d_peak, d_which);
  probe<28>(1024, prop.multiProcessorCount, d_active, d_peak, d_which);
  return 0;
}
