// hfcl_{distance,collide}_batch_multi_device over EVERY visible device (include/hppfcl_amd.h "several devices in one process"): the
// in-place ncclAllGather of hpp-fcl_amd/csrc/hfcl_multi.hip with more than one rank.  Every device's gathered buffer is compared with the
// records of ONE library on device 0 over the whole list (byte for byte; the queries are independent, a shard's records do not depend
// on which replica computed them), and the exchange reports what it did (ranks seen by the communicator, bus rate).
// Exit code 0 = passed, 77 = SKIPPED (fewer than two devices: nothing here can run -- it does not pass), 1 = failed, 3 = no GPU at all.
// Plain C ABI + the HIP runtime for the device buffers: what a C++ integrator writes.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hppfcl_amd.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, hfcl_last_error()); ++failures; } } while (0)
#define HIP_OK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { std::printf("FAILED %s:%d: %s: %s\n", __FILE__, __LINE__, #c, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  const int n_dev = hfcl_device_count();
  if (n_dev < 1) {
    std::printf("no GPU\n");
    return 3;
  }
  int G = n_dev;
  if (argc > 1) G = std::atoi(argv[1]);  // (a subset of the devices)
  if (G < 2 || G > n_dev) {
    std::printf("SKIPPED: %d device(s) visible, the all-gather needs two or more distinct devices\n", n_dev);
    return 77;
  }
  // a small mixed library: box, capsule, sphere, an 8-vertex hull
  std::vector<hfcl_shape> shapes(4);
  std::memset(shapes.data(), 0, shapes.size() * sizeof(hfcl_shape));
  shapes[0].type = HFCL_GEOM_BOX;
  shapes[0].params[0] = 1.0; shapes[0].params[1] = 0.7; shapes[0].params[2] = 0.5;  // half sides
  shapes[1].type = HFCL_GEOM_CAPSULE;
  shapes[1].params[0] = 0.4; shapes[1].params[1] = 0.6;                              // radius, half length
  shapes[2].type = HFCL_GEOM_SPHERE;
  shapes[2].params[0] = 0.7;
  shapes[3].type = HFCL_GEOM_CONVEX;
  shapes[3].vertex_offset = 0; shapes[3].num_points = 8;
  std::vector<double> verts;
  for (int i = 0; i < 8; ++i) { verts.push_back((i & 1) ? 0.8 : -0.8); verts.push_back((i & 2) ? 0.6 : -0.6); verts.push_back((i & 4) ? 0.9 : -0.9); }

  const size_t n = 200003;  // ragged: the last shard is short
  std::vector<uint32_t> s1(n), s2(n);
  std::vector<double> tf1(12 * n), tf2(12 * n);
  uint64_t rng = 2024;
  auto uni = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return double(rng >> 11) / double(1ull << 53); };
  auto pose = [&](double* t, double spread) {  // column-major R (a rotation about a random axis) then T: the image of Transform3f
    double ax = uni() - 0.5, ay = uni() - 0.5, az = uni() - 0.5;
    const double l = std::sqrt(ax * ax + ay * ay + az * az) + 1e-12;
    ax /= l; ay /= l; az /= l;
    const double a = 6.283185307179586 * uni(), c = std::cos(a), s = std::sin(a), v = 1 - c;
    const double R[9] = {c + ax * ax * v, ay * ax * v + az * s, az * ax * v - ay * s, ax * ay * v - az * s, c + ay * ay * v, az * ay * v + ax * s,
                         ax * az * v + ay * s, ay * az * v - ax * s, c + az * az * v};
    for (int k = 0; k < 9; ++k) t[k] = R[k];
    for (int k = 0; k < 3; ++k) t[9 + k] = spread * (uni() - 0.5);
  };
  for (size_t k = 0; k < n; ++k) {
    s1[k] = uint32_t(k % 4);
    s2[k] = uint32_t((k / 4 + k) % 4);
    pose(&tf1[12 * k], 0.5);
    pose(&tf2[12 * k], 4.0);
  }

  // ---- the reference records: one library on device 0, the whole list, host buffers
  hfcl_lib* single = hfcl_lib_create(shapes.data(), shapes.size(), verts.data(), verts.size() / 3, 0);
  CHECK(single != nullptr);
  if (!single) return 1;
  hfcl_distance_request dq;
  hfcl_distance_request_init(&dq);
  hfcl_collision_request cq;
  hfcl_collision_request_init(&cq);
  std::vector<hfcl_result> ref_d(n), ref_c(n);
  CHECK(hfcl_distance_batch(single, s1.data(), s2.data(), tf1.data(), tf2.data(), n, &dq, ref_d.data(), nullptr, nullptr) == HFCL_OK);
  CHECK(hfcl_collide_batch(single, s1.data(), s2.data(), tf1.data(), tf2.data(), n, &cq, ref_c.data(), nullptr, nullptr) == HFCL_OK);
  hfcl_lib_destroy(single);
  size_t separated = 0;
  for (size_t k = 0; k < n; ++k) separated += ref_d[k].distance > 0;
  CHECK(separated > n / 20 && separated < n - n / 20);  // (the list exercises both outcomes)

  // ---- G replicas, every shard resident on its device
  std::vector<int> devices(static_cast<size_t>(G));
  for (int g = 0; g < G; ++g) devices[size_t(g)] = g;
  int dev_before = -1;
  HIP_OK(hipSetDevice(0));
  hfcl_multi* m = hfcl_multi_create(devices.data(), G, shapes.data(), shapes.size(), verts.data(), verts.size() / 3);
  CHECK(m != nullptr);
  if (!m) return 1;
  CHECK(hfcl_multi_size(m) == G);
  const size_t per = (n + size_t(G) - 1) / size_t(G);
  std::vector<uint32_t*> d_s1(size_t(G), nullptr), d_s2(size_t(G), nullptr);
  std::vector<double*> d_t1(size_t(G), nullptr), d_t2(size_t(G), nullptr);
  std::vector<hfcl_result*> d_all(size_t(G), nullptr);
  std::vector<void*> streams(size_t(G), nullptr);
  for (int g = 0; g < G; ++g) {
    size_t lo, hi;
    hfcl_shard_range(n, g, G, &lo, &hi);
    const size_t cnt = hi - lo;
    HIP_OK(hipSetDevice(g));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    streams[size_t(g)] = st;
    HIP_OK(hipMalloc(&d_s1[size_t(g)], (cnt + 1) * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&d_s2[size_t(g)], (cnt + 1) * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&d_t1[size_t(g)], (cnt + 1) * 12 * sizeof(double)));
    HIP_OK(hipMalloc(&d_t2[size_t(g)], (cnt + 1) * 12 * sizeof(double)));
    HIP_OK(hipMalloc(&d_all[size_t(g)], size_t(G) * per * sizeof(hfcl_result)));
    HIP_OK(hipMemcpy(d_s1[size_t(g)], s1.data() + lo, cnt * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_s2[size_t(g)], s2.data() + lo, cnt * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_t1[size_t(g)], tf1.data() + 12 * lo, cnt * 12 * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_t2[size_t(g)], tf2.data() + 12 * lo, cnt * 12 * sizeof(double), hipMemcpyHostToDevice));
  }
  HIP_OK(hipSetDevice(G - 1));
  HIP_OK(hipGetDevice(&dev_before));
  std::vector<hfcl_result> got(size_t(G) * per);
  for (int pass = 0; pass < 2; ++pass) {  // distance(), then collide()
    const std::vector<hfcl_result>& ref = pass == 0 ? ref_d : ref_c;
    for (int rep = 0; rep < 3; ++rep) {  // (the first call creates the communicators; the last one is the one reported)
      for (int g = 0; g < G; ++g) {
        HIP_OK(hipSetDevice(g));
        HIP_OK(hipMemsetAsync(d_all[size_t(g)], 0xEE, size_t(G) * per * sizeof(hfcl_result), static_cast<hipStream_t>(streams[size_t(g)])));
      }
      HIP_OK(hipSetDevice(G - 1));
      const int rc = pass == 0 ? hfcl_distance_batch_multi_device(m, d_s1.data(), d_s2.data(), d_t1.data(), d_t2.data(), n, &dq, d_all.data(), streams.data())
                               : hfcl_collide_batch_multi_device(m, d_s1.data(), d_s2.data(), d_t1.data(), d_t2.data(), n, &cq, d_all.data(), streams.data());
      CHECK(rc == HFCL_OK);
      if (rc != HFCL_OK) return 1;
      int dev_after = -1;
      HIP_OK(hipGetDevice(&dev_after));
      CHECK(dev_after == dev_before);  // the caller's current device is put back
    }
    int ranks = 0;
    double ms = -1;
    size_t bytes = 0;
    CHECK(hfcl_multi_last_gather(m, &ranks, &ms, &bytes) == HFCL_OK);
    CHECK(ranks == G);
    CHECK(bytes == per * sizeof(hfcl_result));
    CHECK(ms > 0);
    std::printf("%s: %d ranks, all-gather of %zu B per rank in %.3f ms = %.1f GB/s received per rank (bus), %.1f GB/s algorithmic over all ranks\n",
                pass == 0 ? "distance" : "collide", ranks, bytes, ms, double(G - 1) * double(bytes) / ms * 1e-6, double(G) * double(G - 1) * double(bytes) / ms * 1e-6);
    for (int g = 0; g < G; ++g) {
      HIP_OK(hipSetDevice(g));
      HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(streams[size_t(g)])));
      HIP_OK(hipMemcpy(got.data(), d_all[size_t(g)], size_t(G) * per * sizeof(hfcl_result), hipMemcpyDeviceToHost));
      size_t differ = 0;
      for (int r = 0; r < G; ++r) {  // rank r's shard sits at slot r of every device's buffer
        size_t lo, hi;
        hfcl_shard_range(n, r, G, &lo, &hi);
        for (size_t k = lo; k < hi; ++k) differ += std::memcmp(&got[size_t(r) * per + (k - lo)], &ref[k], sizeof(hfcl_result)) != 0;
      }
      if (differ) std::printf("device %d: %zu of %zu gathered records differ from the single-library records\n", g, differ, n);
      CHECK(differ == 0);
    }
  }
  // a tuning option reaches every replica; an unknown one is refused
  CHECK(hfcl_multi_set_option(m, "split", "1") == HFCL_OK);
  CHECK(hfcl_multi_set_option(m, "no_such_option", "1") == HFCL_ERR_INVALID_ARGUMENT);
  for (int g = 0; g < G; ++g) {
    (void)hipSetDevice(g);
    (void)hipFree(d_s1[size_t(g)]); (void)hipFree(d_s2[size_t(g)]); (void)hipFree(d_t1[size_t(g)]); (void)hipFree(d_t2[size_t(g)]); (void)hipFree(d_all[size_t(g)]);
    (void)hipStreamDestroy(static_cast<hipStream_t>(streams[size_t(g)]));
  }
  hfcl_multi_destroy(m);
  std::printf(failures ? "FAILED (%d checks)\n" : "all checks passed\n", failures);
  return failures ? 1 : 0;
}
