// link_probe.hip -- what the host link delivers to the host-buffer entry points (hfcl_*_batch): copies between PAGEABLE host
// arrays and the device, one direction and both at once from two host threads, in pieces of the sizes the pipeline uses.
// Build: hipcc -O3 --offload-arch=gfx950 tools/link_probe.hip -o gpurun_out/link_probe -lpthread ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t total = size_t(256) << 20;
  char *h_in = (char*)malloc(total), *h_out = (char*)malloc(total), *p_in = nullptr, *p_out = nullptr, *d_in = nullptr, *d_out = nullptr;
  CK(hipSetDevice(0));
  memset(h_in, 1, total); memset(h_out, 2, total);
  CK(hipHostMalloc((void**)&p_in, total)); CK(hipHostMalloc((void**)&p_out, total));
  memset(p_in, 1, total); memset(p_out, 2, total);
  CK(hipMalloc((void**)&d_in, total)); CK(hipMalloc((void**)&d_out, total));
  printf("h %p %p p %p %p d %p %p\n", h_in, h_out, p_in, p_out, d_in, d_out);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t pieces[4] = {size_t(3) << 20, size_t(12) << 20, size_t(48) << 20, total};
  for (size_t piece : pieces) {
    for (int pinned = 0; pinned < 2; ++pinned) {
      char* src = pinned ? p_in : h_in; char* dst = pinned ? p_out : h_out;
      auto h2d = [&]() { for (size_t o = 0; o + piece <= total; o += piece) CK(hipMemcpyAsync(d_in + o, src + o, piece, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); };
      auto d2h = [&]() { for (size_t o = 0; o + piece <= total; o += piece) CK(hipMemcpyAsync(dst + o, d_out + o, piece, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); };
      h2d(); d2h();
      double t0 = now(); h2d(); double t_in = now() - t0;
      t0 = now(); d2h(); double t_out = now() - t0;
      t0 = now();
      { std::thread a(h2d), b(d2h); a.join(); b.join(); }
      double t_both = now() - t0;
      printf("%-8s pieces of %3zu MB: H2D %5.1f GB/s  D2H %5.1f GB/s  both at once (two threads) %5.1f GB/s per direction\n", pinned ? "pinned" : "pageable",
             piece >> 20, total / t_in / 1e9, total / t_out / 1e9, total / t_both / 1e9);
    }
  }
  // ---- the pattern of the host-buffer pipeline: chunks of 16 MB in / 8 MB out, a kernel per chunk between them, three
  // streams, host threads for the two directions; printed: total time and the rate of each direction
  {
    hipStream_t sc;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    const int NC = 12;
    const size_t cin = size_t(16) << 20, cout = size_t(8) << 20;
    std::vector<hipEvent_t> ein(NC), edone(NC);
    for (int k = 0; k < NC; ++k) { CK(hipEventCreateWithFlags(&ein[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&edone[k], hipEventDisableTiming)); }
    for (int variant = 0; variant < 3; ++variant) {
      // 0: every array its own copy (4 per chunk), 1: one copy per chunk, 2: one copy per chunk and no kernel between
      for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        std::thread a([&]() {
          for (int k = 0; k < NC; ++k) {
            if (variant == 0) {
              for (int j = 0; j < 4; ++j) CK(hipMemcpyAsync(d_in + k * cin + j * (cin / 4), h_in + k * cin + j * (cin / 4), cin / 4, hipMemcpyHostToDevice, s1));
            } else {
              CK(hipMemcpyAsync(d_in + k * cin, h_in + k * cin, cin, hipMemcpyHostToDevice, s1));
            }
            CK(hipEventRecord(ein[k], s1));
          }
          CK(hipStreamSynchronize(s1));
        });
        std::thread b([&]() {
          for (int k = 0; k < NC; ++k) {
            CK(hipStreamWaitEvent(sc, ein[k], 0));
            if (variant != 2) CK(hipMemsetAsync(d_out + k * cout, 7, cout, sc));  // (stands for the kernels of the chunk)
            CK(hipEventRecord(edone[k], sc));
            CK(hipStreamWaitEvent(s2, edone[k], 0));
            CK(hipMemcpyAsync(h_out + k * cout, d_out + k * cout, cout, hipMemcpyDeviceToHost, s2));
          }
          CK(hipStreamSynchronize(s2));
        });
        a.join(); b.join();
        double t = now() - t0;
        if (rep) printf("pipeline pattern %d: %d chunks of 16 MB in + 8 MB out: %.3f ms  (in %.1f GB/s, out %.1f GB/s)\n", variant, NC, 1e3 * t, NC * cin / t / 1e9, NC * cout / t / 1e9);
      }
    }
  }
  return 0;
}
