#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* h, const float* d, uint32_t* cnt) {
  const int lane = threadIdx.x & 63, vi = lane & 31, hi = lane >> 5;
  const float* hh = h + blockIdx.x * 96; const float* dd = d + blockIdx.x * 96;
  const float a0 = hh[3 * vi + hi], a1 = hi == 0 ? hh[3 * vi + 2] : 0.f;
  const float b0 = dd[3 * vi + hi], b1 = hi == 0 ? dd[3 * vi + 2] : 0.f;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c, 0, 0, 0);
  uint32_t e[6] = {0,0,0,0,0,0};
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi, col = lane & 31;
    const float x = hh[3*row], y = hh[3*row+1], z = hh[3*row+2], dx = dd[3*col], dy = dd[3*col+1], dz = dd[3*col+2];
    const float v0 = __fmaf_rn(z, dz, __fmaf_rn(y, dy, __fmul_rn(x, dx)));
    const float v1 = __fmaf_rn(z, dz, __fmaf_rn(x, dx, __fmul_rn(y, dy)));
    const float v2 = __fadd_rn(__fadd_rn(__fmul_rn(x, dx), __fmul_rn(y, dy)), __fmul_rn(z, dz));
    const float v3 = (float)((double)x*dx + (double)y*dy + (double)z*dz);
    const float v4 = __fmaf_rn(z, dz, (float)((double)x*dx + (double)y*dy));
    const float v5 = __fadd_rn((float)((double)x*dx + (double)y*dy), __fmul_rn(z,dz));
    e[0] += c[r] == v0; e[1] += c[r] == v1; e[2] += c[r] == v2; e[3] += c[r] == v3; e[4] += c[r] == v4; e[5] += c[r] == v5;
  }
  for (int i = 0; i < 6; ++i) atomicAdd(&cnt[i], e[i]);
}
int main() {
  const int B = 4096; std::vector<float> h(B*96), d(B*96); uint64_t rng=7;
  auto uni=[&](){rng=rng*6364136223846793005ull+1442695040888963407ull; return float(double(rng>>11)/double(1ull<<53))*2.f-1.f;};
  for (auto&x:h) x=uni(); for (auto&x:d) x=uni();
  float *dh,*dd; uint32_t* dc; hipMalloc(&dh,h.size()*4); hipMalloc(&dd,d.size()*4); hipMalloc(&dc,24); hipMemset(dc,0,24);
  hipMemcpy(dh,h.data(),h.size()*4,hipMemcpyHostToDevice); hipMemcpy(dd,d.data(),d.size()*4,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k,dim3(B),dim3(64),0,0,dh,dd,dc); uint32_t c[6]; hipMemcpy(c,dc,24,hipMemcpyDeviceToHost);
  printf("of %d products: fma(z,fma(y,x*dx)) %u | fma(z,fma(x,y*dy)) %u | (x*dx+y*dy)+z*dz rounded each %u | exact sum rounded once %u | fma(z,dz,round(exact xy)) %u | round(exact xy)+round(z dz) %u\n", B*1024, c[0],c[1],c[2],c[3],c[4],c[5]);
}
