// valu_peak.hip -- calibration microbenchmark for the issue-side bound bench.py reports
// ("the figure was never calibrated with a microbenchmark on the box").
//
// Measures, on the box, how many wave64 VALU instructions one SIMD retires per shader clock as a function of
// the number of resident waves per SIMD (1 / 2 / 4 / 8) for
//   fma32      v_fma_f32        8 independent chains per lane (throughput)
//   fma32dep   v_fma_f32        1 dependent chain per lane    (issue-to-issue latency of a dependent VALU op)
//   pkfma32    v_pk_fma_f32     8 independent chains (two fp32 FMAs per lane and instruction)
//   fma64      v_fma_f64        8 independent chains
//   mixed      v_fma_f32 + v_cndmask_b32 + v_cmp_lt_f32 (the select-heavy mix of the GJK/EPA kernels)
//   lds        ds_read_b128 (uniform address, the broadcast reads of the EPA polytope blocks) latency chain
// and the H2D / D2H bandwidth of pinned and pageable host buffers (for the host-buffer entry points).
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o gpurun_out/valu_peak ; run on the GPU box.
// Output: one line per (kind, waves/SIMD): cycles per instruction per SIMD and chip-wide G wave-inst/s.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;  // instructions per chain and loop trip

struct Out {
  unsigned long long clk0, clk1, rt0, rt1;
};

extern __shared__ char dyn_lds[];

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(Out* out, float seed, int iters) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  const float x = 1.0000001f, y = 1e-9f;
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
  const double dx = 1.0000001, dy = 1e-9;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
  const f2 px = {x, x}, py = {y, y};
  if (KIND == 5) {
    unsigned* l = reinterpret_cast<unsigned*>(dyn_lds);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
      l[4 * i] = unsigned(i * 37 + 11) & 255u;  // a permutation cycle of the 256 records
      l[4 * i + 1] = i;
    }
    __syncthreads();
  }
  unsigned addr = 0;
  const unsigned long long c0 = __builtin_readcyclecounter();  // s_memtime: shader clock
  const unsigned long long r0 = wall_clock64();                // s_memrealtime: constant 100 MHz
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (KIND == 0) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(x), "v"(y));
      } else if (KIND == 1) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0)
                     : "v"(x), "v"(y));
      } else if (KIND == 2) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                     : "v"(px), "v"(py));
      } else if (KIND == 3) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                     : "v"(dx), "v"(dy));
      } else if (KIND == 4) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %2, %2, %3, vcc\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(x), "v"(y)
                     : "vcc");
      } else {
        // dependent uniform-address 16-byte LDS reads (the broadcast reads of the EPA polytope blocks):
        // the index of the next read comes out of the previous one
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 q = reinterpret_cast<const u4*>(dyn_lds)[addr];
        addr = q.x;
        a0 += float(q.y);
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + float(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.y + p2.x + p3.y + p4.x +
            p5.y + p6.x + p7.y;
  if (s == 12345.678f) out[0].clk0 = 1;  // keep the chains alive
  if ((threadIdx.x & 63) == 0) {
    Out o = {c0, c1, r0, r1};
    out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = o;
  }
}

template <int KIND>
static void run(const char* name, int insts_per_unroll_step, int n_cus, Out* d_out, std::vector<Out>& h_out) {
  for (int wps : {1, 2, 4, 8}) {
    // 256-thread blocks = one wave per SIMD; `wps` blocks resident per CU, enforced by the dynamic LDS size
    const size_t lds = (size_t(160) * 1024 / wps) - 512;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_valu<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const int blocks = n_cus * wps;  // exactly one resident set: no tail, no second round
    const int iters = KIND == 5 ? ITERS / 8 : ITERS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), lds, 0, d_out, 1.0f, 10);  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), lds, 0, d_out, 1.0f, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = blocks * 4;
    CK(hipMemcpy(h_out.data(), d_out, waves * sizeof(Out), hipMemcpyDeviceToHost));
    double clk = 0, rt = 0;
    for (int w = 0; w < waves; ++w) {
      clk += double(h_out[w].clk1 - h_out[w].clk0);
      rt += double(h_out[w].rt1 - h_out[w].rt0);
    }
    clk /= waves;
    rt /= waves;
    const double insts_per_wave = double(iters) * UNROLL * insts_per_unroll_step;
    const double ghz = (clk / rt) * 0.1;                       // shader clocks per 100 MHz tick
    const double cyc_per_inst_simd = clk / (insts_per_wave * wps);  // one SIMD retires wps waves' instructions
    const double chip = double(waves) * insts_per_wave / (ms * 1e-3) / 1e9;
    printf("%-9s waves/SIMD %d  blocks %5d  %.3f ms  shader clock %.3f GHz (s_memtime/s_memrealtime)  %.3f cycles per wave-inst per SIMD"
           "  chip %.1f G wave-inst/s\n",
           name, wps, blocks, ms, ghz, cyc_per_inst_simd, chip);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
  }
}

static void pcie(size_t bytes) {
  void* d = nullptr;
  CK(hipMalloc(&d, bytes));
  void* pinned = nullptr;
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
  void* pageable = malloc(bytes);
  memset(pinned, 1, bytes);
  memset(pageable, 1, bytes);
  auto timeit = [&](const char* what, void* dst, const void* src, hipMemcpyKind kind) {
    CK(hipMemcpy(dst, src, bytes, kind));
    const int reps = 5;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipMemcpy(dst, src, bytes, kind));
    CK(hipDeviceSynchronize());
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("pcie %-22s %8.1f MB  %7.2f GB/s\n", what, bytes / 1e6, bytes / s / 1e9);
  };
  timeit("H2D pinned", d, pinned, hipMemcpyHostToDevice);
  timeit("D2H pinned", pinned, d, hipMemcpyDeviceToHost);
  timeit("H2D pageable", d, pageable, hipMemcpyHostToDevice);
  timeit("D2H pageable", pageable, d, hipMemcpyDeviceToHost);
  // both directions at once on two streams (what a chunked H2D | kernels | D2H pipeline sees)
  {
    void* d2 = nullptr;
    void* pinned2 = nullptr;
    CK(hipMalloc(&d2, bytes));
    CK(hipHostMalloc(&pinned2, bytes, hipHostMallocDefault));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int reps = 5;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      CK(hipMemcpyAsync(d, pinned, bytes, hipMemcpyHostToDevice, s1));
      CK(hipMemcpyAsync(pinned2, d2, bytes, hipMemcpyDeviceToHost, s2));
    }
    CK(hipDeviceSynchronize());
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("pcie %-22s %8.1f MB  %7.2f GB/s per direction (both at once)\n", "H2D+D2H pinned", bytes / 1e6, bytes / s / 1e9);
    // host-side memcpy pageable -> pinned (the staging copy a host-buffer entry point has to do), 1 thread
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) memcpy(pinned, pageable, bytes);
    const double sm = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("host memcpy pageable->pinned, 1 thread: %7.2f GB/s\n", bytes / sm / 1e9);
    CK(hipFree(d2));
    CK(hipHostFree(pinned2));
  }
  free(pageable);
  CK(hipHostFree(pinned));
  CK(hipFree(d));
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clockRate %.0f MHz  wallClock %.0f kHz\n", prop.name, n_cus, prop.clockRate / 1e3, 1e5);
  Out* d_out = nullptr;
  CK(hipMalloc(&d_out, size_t(n_cus) * 8 * 4 * sizeof(Out)));
  std::vector<Out> h(size_t(n_cus) * 8 * 4);
  run<0>("fma32", 8, n_cus, d_out, h);
  run<1>("fma32dep", 8, n_cus, d_out, h);
  run<2>("pkfma32", 8, n_cus, d_out, h);
  run<3>("fma64", 8, n_cus, d_out, h);
  run<4>("mixed", 8, n_cus, d_out, h);
  run<5>("lds_dep", 1, n_cus, d_out, h);
  pcie(size_t(256) << 20);
  return 0;
}
