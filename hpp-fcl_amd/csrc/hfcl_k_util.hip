// hfcl_k_util.hip -- record utilities that are pure data movement (HBM bound, no geometry).
//   k_compact_records<R,C>  full result records -> compact records for the multi-GPU exchange (hfcl_result_compact):
//                           one record per lane; a lane touches 24 of the 96 (8 of the 44) bytes of its record, the
//                           stores are lane-contiguous.
#include "hfcl_dev.hpp"
#include "hfcl_launch.hpp"

__device__ __forceinline__ void compact_one(const hfcl_result& r, hfcl_result_compact& c) {
  c.distance = r.distance;
  // b1, b2, status, num_contacts are the last 16 bytes of the record: one 16-byte load
  const uint4 tail = *reinterpret_cast<const uint4*>(&r.b1);
  c.b1 = int32_t(tail.x);
  c.b2 = int32_t(tail.y);
  c.status = tail.z;
  c.num_contacts = int32_t(tail.w);
}
__device__ __forceinline__ void compact_one(const hfcl_result_f32& r, hfcl_result_compact_f32& c) {
  c.distance = r.distance;
  c.status = r.status;
}

template <typename R, typename C>
__global__ void __launch_bounds__(256) k_compact_records(const R* __restrict__ in, C* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    C c;
    compact_one(in[i], c);
    out[i] = c;
  }
}

void launch_compact_records(hipStream_t st, const hfcl_result* in, hfcl_result_compact* out, uint32_t n) {
  const uint32_t grid = std::min<uint32_t>((n + 255u) / 256u, 256u * 16u);
  hipLaunchKernelGGL((k_compact_records<hfcl_result, hfcl_result_compact>), dim3(grid ? grid : 1u), dim3(256), 0, st, in, out, n);
}
void launch_compact_records(hipStream_t st, const hfcl_result_f32* in, hfcl_result_compact_f32* out, uint32_t n) {
  const uint32_t grid = std::min<uint32_t>((n + 255u) / 256u, 256u * 16u);
  hipLaunchKernelGGL((k_compact_records<hfcl_result_f32, hfcl_result_compact_f32>), dim3(grid ? grid : 1u), dim3(256), 0, st, in, out, n);
}
