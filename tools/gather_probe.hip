// Gather-rate probe for the BVH traversal's access pattern (MI355X): every lane fetches 128-byte records at random,
// unrelated addresses -- how many records per second does the chip deliver, per lane (8 loads of 16 bytes, each load
// instruction touching 64 different lines) or cooperatively (the 8 lanes of a group fetch each other's records: each
// load instruction covers 8 whole records)?  Dependent mode: the next index depends on the record just read (the
// traversal's chain); independent mode: indices from a counter-based generator (throughput only).
//   hipcc -O3 --offload-arch=gfx950 -o build/gather_probe tools/gather_probe.hip ; gpurun -- build/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { uint4 v[8]; };  // 128 bytes

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE, bool DEP>
__global__ void __launch_bounds__(128) k_probe(const Rec* recs, uint32_t mask, int steps, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  uint32_t idx = mix(blockIdx.x * blockDim.x + threadIdx.x) & mask;
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    uint32_t got;
    if (MODE == 0) {  // the lane's own record
      const uint4* p = recs[idx].v;
      uint32_t a = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const uint4 q = p[k]; a += q.x ^ q.y ^ q.z ^ q.w; }
      got = a;
    } else {  // 8-lane groups fetch each other's records; a lane ends up with one piece of each (enough for a checksum)
      const int g0 = lane & ~7, piece = lane & 7;
      uint32_t a[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t want = __shfl(idx, g0 + r, 64);
        const uint4 q = recs[want].v[piece];
        a[r] = q.x ^ q.y ^ q.z ^ q.w;
      }
      // hand every piece checksum to the lane that asked (3 butterfly steps of adds over the group would do; here a
      // shuffle per round, like the exchange through LDS the traversal would need)
      uint32_t mine = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint32_t t = a[r];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        if (piece == r) mine = t;
      }
      got = mine;
    }
    acc += got;
    idx = DEP ? mix(got + s) & mask : mix(idx + 0x9E3779B9u * (s + 1)) & mask;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, bool DEP>
static double run(const Rec* d, uint32_t mask, int blocks, int steps, uint32_t* d_out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<MODE, DEP>), dim3(blocks), dim3(128), 0, 0, d, mask, 8, d_out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<MODE, DEP>), dim3(blocks), dim3(128), 0, 0, d, mask, steps, d_out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return double(blocks) * 128 * steps / (ms * 1e-3);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint32_t* d_out; hipMalloc(&d_out, size_t(cus) * 16 * 128 * 4);
  for (int log2n : {17, 21}) {  // 16 MB (the eight cfg4 models: 20 MB) and 256 MB of records
    const size_t n = size_t(1) << log2n;
    std::vector<uint32_t> h(n * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = uint32_t(i * 2654435761u);
    Rec* d; hipMalloc(&d, n * sizeof(Rec)); hipMemcpy(d, h.data(), n * sizeof(Rec), hipMemcpyHostToDevice);
    for (int wpc : {2, 4, 8, 16}) {  // waves per CU
      const int blocks = cus * wpc / 2;
      const int steps = 2000;
      printf("%4zu MB of records, %2d waves per CU:  per lane  dependent %6.2f G rec/s  independent %6.2f | cooperative  dependent %6.2f  independent %6.2f\n",
             n * sizeof(Rec) >> 20, wpc, run<0, true>(d, uint32_t(n - 1), blocks, steps, d_out) / 1e9, run<0, false>(d, uint32_t(n - 1), blocks, steps, d_out) / 1e9,
             run<1, true>(d, uint32_t(n - 1), blocks, steps, d_out) / 1e9, run<1, false>(d, uint32_t(n - 1), blocks, steps, d_out) / 1e9);
    }
    hipFree(d);
  }
  return 0;
}
