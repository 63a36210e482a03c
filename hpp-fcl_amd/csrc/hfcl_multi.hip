// hfcl_multi.hip -- several devices in one process (include/hppfcl_amd.h: hfcl_multi_*; SURVEY.md 8e "one process, G
// streams").  Host code over the single-library C ABI: a replica of the library per device, the pair list cut into
// contiguous shards (hfcl_shard_range = hpp-fcl_amd/sharding.py: shard_range), a host thread per shard for the host-buffer
// entry points (each replica has its own H2D | kernels | D2H pipeline), and for device-resident batches an in-place all-gather of
// the fixed-size records through librccl.so (RCCL over xGMI), loaded with dlopen on first use so that a caller who never
// shards pays nothing for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hppfcl_amd.h"

void hfcl_internal_set_error(const char* msg);  // hfcl_host.hip: the calling thread's hfcl_last_error()

namespace {
// the five entry points of RCCL this file uses (rccl/rccl.h; ncclChar = 0: the records travel as bytes)
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
  int (*CommCount)(void* comm, int* count) = nullptr;  // (optional: diagnostics)
  const char* (*GetErrorString)(int) = nullptr;
  bool load(std::string& err) {
    if (handle) return true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) {
      err = std::string("librccl.so: ") + dlerror();
      return false;
    }
    auto sym = [&](const char* n) { return dlsym(handle, n); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !AllGather) {
      err = "librccl.so lacks ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd / ncclCommDestroy";
      dlclose(handle);
      handle = nullptr;
      return false;
    }
    return true;
  }
};
}  // namespace

struct hfcl_multi {
  std::vector<int> devices;
  std::vector<hfcl_lib*> libs;
  Rccl rccl;
  std::vector<void*> comms;  // one communicator per replica (created with the first device-resident batch)
  // a registration (shapes, adjacency, model) that reached some replicas and failed on another leaves them different for good: every later
  // call says so instead of computing with tables that disagree
  std::string poisoned;
  // the last device-resident batch: ranks the communicator reports (ncclCommCount; 1 without a collective), the all-gather's duration on
  // replica 0's stream (events around the grouped call) and the bytes each rank contributed
  hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;
  int gather_ranks = 0;
  bool gather_timed = false;
  size_t gather_bytes_per_rank = 0;
};

namespace {
// the caller's current device, put back when a multi-device call returns
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceGuard() {
    if (dev >= 0) hipSetDevice(dev);
  }
};
int check_usable(const hfcl_multi* m, const char* who) {
  if (!m) {
    hfcl_internal_set_error((std::string(who) + ": null hfcl_multi").c_str());
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (!m->poisoned.empty()) {
    hfcl_internal_set_error((std::string(who) + ": the replicas differ since " + m->poisoned + " -- destroy this hfcl_multi and create it again").c_str());
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return HFCL_OK;
}
}  // namespace

extern "C" {

void hfcl_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
  const size_t w = world > 0 ? size_t(world) : 1, per = (n + w - 1) / w;
  const size_t a = std::min(n, size_t(rank < 0 ? 0 : rank) * per);
  if (lo) *lo = a;
  if (hi) *hi = std::min(n, a + per);
}

hfcl_multi* hfcl_multi_create(const int* devices, int n_devices, const hfcl_shape* shapes, size_t n_shapes, const double* vertices,
                              size_t n_vertices) {
  if (!devices || n_devices < 1) {
    hfcl_internal_set_error("hfcl_multi_create: at least one device");
    return nullptr;
  }
  DeviceGuard guard;
  hfcl_multi* m = new hfcl_multi;
  for (int i = 0; i < n_devices; ++i) {
    hfcl_lib* lib = hfcl_lib_create(shapes, n_shapes, vertices, n_vertices, devices[i]);
    if (!lib) {  // (hfcl_last_error() is the replica's)
      hfcl_multi_destroy(m);
      return nullptr;
    }
    m->devices.push_back(devices[i]);
    m->libs.push_back(lib);
  }
  return m;
}

void hfcl_multi_destroy(hfcl_multi* m) {
  if (!m) return;
  DeviceGuard guard;
  for (size_t g = 0; g < m->comms.size(); ++g)
    if (m->comms[g]) m->rccl.CommDestroy(m->comms[g]);
  if (m->ev_g0 || m->ev_g1) {
    if (!m->devices.empty()) hipSetDevice(m->devices[0]);
    if (m->ev_g0) hipEventDestroy(m->ev_g0);
    if (m->ev_g1) hipEventDestroy(m->ev_g1);
  }
  for (hfcl_lib* lib : m->libs) hfcl_lib_destroy(lib);
  delete m;
}

int hfcl_multi_size(const hfcl_multi* m) { return m ? int(m->libs.size()) : 0; }
hfcl_lib* hfcl_multi_replica(hfcl_multi* m, int i) { return (m && i >= 0 && size_t(i) < m->libs.size()) ? m->libs[size_t(i)] : nullptr; }

// Registrations go to every replica.  A failure on replica g > 0 leaves replicas 0 .. g-1 changed and the others not: the hfcl_multi is
// marked and refuses further work (the caller's tables were rejected by a device -- out of memory, a lost device -- there is no state
// to roll back to that the replicas could be brought to without the same call succeeding).
int hfcl_multi_set_shapes(hfcl_multi* m, const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices) {
  if (int rc = check_usable(m, "hfcl_multi_set_shapes")) return rc;
  DeviceGuard guard;
  for (size_t g = 0; g < m->libs.size(); ++g)
    if (int rc = hfcl_lib_set_shapes(m->libs[g], shapes, n_shapes, vertices, n_vertices)) {
      if (g > 0) m->poisoned = "hfcl_multi_set_shapes failed on replica " + std::to_string(g);
      return rc;
    }
  return HFCL_OK;
}
int hfcl_multi_set_convex_neighbors(hfcl_multi* m, uint32_t shape_id, const uint32_t* offsets, const uint32_t* neighbors) {
  if (int rc = check_usable(m, "hfcl_multi_set_convex_neighbors")) return rc;
  DeviceGuard guard;
  for (size_t g = 0; g < m->libs.size(); ++g)
    if (int rc = hfcl_lib_set_convex_neighbors(m->libs[g], shape_id, offsets, neighbors)) {
      if (g > 0) m->poisoned = "hfcl_multi_set_convex_neighbors failed on replica " + std::to_string(g);
      return rc;
    }
  return HFCL_OK;
}
int hfcl_multi_add_bvh(hfcl_multi* m, const hfcl_bvh_node* nodes, size_t n_nodes, const double* vertices, size_t n_vertices,
                       const uint32_t* triangles, size_t n_tris) {
  if (check_usable(m, "hfcl_multi_add_bvh")) return -1;
  DeviceGuard guard;
  int index = -1;
  for (size_t g = 0; g < m->libs.size(); ++g) {
    const int k = hfcl_lib_add_bvh(m->libs[g], nodes, n_nodes, vertices, n_vertices, triangles, n_tris);
    if (k < 0) {
      if (g > 0) m->poisoned = "hfcl_multi_add_bvh failed on replica " + std::to_string(g);
      return k;
    }
    if (index >= 0 && k != index) {
      m->poisoned = "hfcl_multi_add_bvh gave the model different indices (a model was registered on one replica only)";
      hfcl_internal_set_error("hfcl_multi_add_bvh: the replicas disagree on the model's index (a model was registered on one replica only)");
      return -1;
    }
    index = k;
  }
  return index;
}
// hfcl_lib_set_option on every replica (an unknown key fails on replica 0, before anything changed)
int hfcl_multi_set_option(hfcl_multi* m, const char* key, const char* value) {
  if (int rc = check_usable(m, "hfcl_multi_set_option")) return rc;
  for (hfcl_lib* lib : m->libs)
    if (int rc = hfcl_lib_set_option(lib, key, value)) return rc;
  return HFCL_OK;
}
// The last device-resident batch: ranks the communicator reports, milliseconds of the all-gather on replica 0's stream (waits for it;
// < 0 when there was no collective), bytes each rank contributed.
int hfcl_multi_last_gather(hfcl_multi* m, int* ranks, double* ms, size_t* bytes_per_rank) {
  if (!m) return HFCL_ERR_INVALID_ARGUMENT;
  if (ranks) *ranks = m->gather_ranks;
  if (bytes_per_rank) *bytes_per_rank = m->gather_bytes_per_rank;
  if (ms) {
    *ms = -1.0;
    if (m->gather_timed && m->ev_g0 && m->ev_g1) {
      DeviceGuard guard;
      float t = 0.f;
      if (hipSetDevice(m->devices[0]) == hipSuccess && hipEventSynchronize(m->ev_g1) == hipSuccess && hipEventElapsedTime(&t, m->ev_g0, m->ev_g1) == hipSuccess)
        *ms = double(t);
    }
  }
  return HFCL_OK;
}

}  // extern "C"

// every shard through `call(replica, lo, hi)`, shard 0 on the calling thread; the first failure is the call's (with its message)
template <class Call>
static int run_sharded(hfcl_multi* m, size_t n, const char* who, Call&& call) {
  if (int rc = check_usable(m, who)) return rc;
  DeviceGuard guard;
  const int G = int(m->libs.size());
  std::vector<int> rc(size_t(G), 0);
  std::vector<std::string> err(static_cast<size_t>(G));
  auto one = [&](int g) {
    size_t lo, hi;
    hfcl_shard_range(n, g, G, &lo, &hi);
    if (hi <= lo) return;
    rc[size_t(g)] = call(m->libs[size_t(g)], lo, hi);
    if (rc[size_t(g)]) err[size_t(g)] = hfcl_last_error();
  };
  std::vector<std::thread> workers;
  int inline_from = G;  // shards whose thread could not be started run on the calling thread
  try {
    workers.reserve(size_t(G));
    for (int g = 1; g < G; ++g) workers.emplace_back(one, g);
  } catch (...) {
    inline_from = 1 + int(workers.size());
  }
  one(0);
  for (int g = inline_from; g < G; ++g) one(g);
  for (std::thread& t : workers) t.join();
  for (int g = 0; g < G; ++g)
    if (rc[size_t(g)]) {
      hfcl_internal_set_error(("replica " + std::to_string(g) + " (device " + std::to_string(m->devices[size_t(g)]) + "): " + err[size_t(g)]).c_str());
      return rc[size_t(g)];
    }
  return HFCL_OK;
}

// the device-resident form: each replica's shard into its slot of the gathered buffer, then the in-place all-gather
template <class Launch>
static int run_gathered(hfcl_multi* m, size_t n, const char* who, hfcl_result* const* d_gathered, void* const* streams, const void* const* const* arrays,
                        int n_arrays, Launch&& launch) {
  if (int rc = check_usable(m, who)) return rc;
  const int G = int(m->libs.size());
  bool null_arg = !d_gathered;
  for (int a = 0; a < n_arrays && !null_arg; ++a) null_arg = !arrays[a];
  for (int g = 0; g < G && !null_arg; ++g) {
    null_arg = !d_gathered[g];
    for (int a = 0; a < n_arrays && !null_arg; ++a) null_arg = n > 0 && !arrays[a][g];
  }
  if (null_arg) {
    hfcl_internal_set_error((std::string(who) + ": null pointer array or entry (one device pointer per replica)").c_str());
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard guard;  // (the replicas' calls and the collective switch the current device)
  size_t per, dummy;
  hfcl_shard_range(n, 0, G, &dummy, &per);  // ceil(n / G)
  m->gather_ranks = 1;
  m->gather_timed = false;
  m->gather_bytes_per_rank = per * sizeof(hfcl_result);
  if (G > 1) {
    if (std::set<int>(m->devices.begin(), m->devices.end()).size() != size_t(G)) {
      hfcl_internal_set_error("hfcl_*_batch_multi_device: a device is listed twice (the all-gather needs one rank per device)");
      return HFCL_ERR_INVALID_ARGUMENT;
    }
    if (m->comms.empty()) {
      std::string err;
      if (!m->rccl.load(err)) {
        hfcl_internal_set_error(err.c_str());
        return HFCL_ERR_HIP;
      }
      m->comms.assign(size_t(G), nullptr);
      if (int e = m->rccl.CommInitAll(m->comms.data(), G, m->devices.data())) {
        hfcl_internal_set_error((std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(e) : "error")).c_str());
        m->comms.clear();
        return HFCL_ERR_HIP;
      }
    }
    int count = G;
    if (m->rccl.CommCount && m->rccl.CommCount(m->comms[0], &count) == 0) m->gather_ranks = count;
    else m->gather_ranks = G;
    if (!m->ev_g0 && hipSetDevice(m->devices[0]) == hipSuccess) {
      if (hipEventCreate(&m->ev_g0) != hipSuccess) m->ev_g0 = nullptr;
      if (hipEventCreate(&m->ev_g1) != hipSuccess) m->ev_g1 = nullptr;
    }
  }
  // every shard into its slot; a shard that cannot be launched ends the call AFTER the shards in front of it have run (nothing of this
  // call is in flight on the caller's buffers when it returns an error) and without a collective
  for (int g = 0; g < G; ++g) {
    size_t lo, hi;
    hfcl_shard_range(n, g, G, &lo, &hi);
    if (hi <= lo) continue;
    if (int rc = launch(g, hi - lo, d_gathered[g] + size_t(g) * per, streams ? streams[g] : nullptr)) {
      const std::string msg = hfcl_last_error();
      for (int k = 0; k < g; ++k)
        if (hipSetDevice(m->devices[size_t(k)]) == hipSuccess) hipStreamSynchronize(static_cast<hipStream_t>(streams ? streams[k] : nullptr));
      hfcl_internal_set_error(("replica " + std::to_string(g) + " (device " + std::to_string(m->devices[size_t(g)]) + "): " + msg).c_str());
      return rc;
    }
  }
  if (G > 1) {
    const bool timed = m->ev_g0 && m->ev_g1 && hipSetDevice(m->devices[0]) == hipSuccess &&
                       hipEventRecord(m->ev_g0, static_cast<hipStream_t>(streams ? streams[0] : nullptr)) == hipSuccess;
    int e = m->rccl.GroupStart();
    bool dev_failed = false;
    if (!e) {  // (a group that was opened is closed on every path)
      for (int g = 0; g < G && !e && !dev_failed; ++g) {
        if (hipSetDevice(m->devices[size_t(g)]) != hipSuccess) {
          dev_failed = true;
          break;
        }
        e = m->rccl.AllGather(d_gathered[g] + size_t(g) * per, d_gathered[g], per * sizeof(hfcl_result), /* ncclChar */ 0, m->comms[size_t(g)],
                              static_cast<hipStream_t>(streams ? streams[g] : nullptr));
      }
      const int e2 = m->rccl.GroupEnd();
      if (!e) e = e2;
    }
    if (dev_failed) {
      hfcl_internal_set_error("hfcl_*_batch_multi_device: hipSetDevice failed while the all-gather was being enqueued");
      return HFCL_ERR_HIP;
    }
    if (e) {
      hfcl_internal_set_error((std::string("ncclAllGather: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(e) : "error")).c_str());
      return HFCL_ERR_HIP;
    }
    if (timed && hipSetDevice(m->devices[0]) == hipSuccess && hipEventRecord(m->ev_g1, static_cast<hipStream_t>(streams ? streams[0] : nullptr)) == hipSuccess)
      m->gather_timed = true;
  }
  return HFCL_OK;
}

extern "C" {

int hfcl_collide_batch_multi(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const double* tf1, const double* tf2, size_t n,
                             const hfcl_collision_request* req, hfcl_result* out, const hfcl_guess* guess_in, hfcl_guess* guess_out) {
  return run_sharded(m, n, "hfcl_collide_batch_multi", [&](hfcl_lib* lib, size_t lo, size_t hi) {
    return hfcl_collide_batch(lib, shape1 + lo, shape2 + lo, tf1 + 12 * lo, tf2 + 12 * lo, hi - lo, req, out + lo, guess_in ? guess_in + lo : nullptr,
                              guess_out ? guess_out + lo : nullptr);
  });
}
int hfcl_distance_batch_multi(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const double* tf1, const double* tf2, size_t n,
                              const hfcl_distance_request* req, hfcl_result* out, const hfcl_guess* guess_in, hfcl_guess* guess_out) {
  return run_sharded(m, n, "hfcl_distance_batch_multi", [&](hfcl_lib* lib, size_t lo, size_t hi) {
    return hfcl_distance_batch(lib, shape1 + lo, shape2 + lo, tf1 + 12 * lo, tf2 + 12 * lo, hi - lo, req, out + lo, guess_in ? guess_in + lo : nullptr,
                               guess_out ? guess_out + lo : nullptr);
  });
}
int hfcl_collide_batch_multi_f32(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                                 const hfcl_collision_request* req, hfcl_result_f32* out) {
  return run_sharded(m, n, "hfcl_collide_batch_multi_f32", [&](hfcl_lib* lib, size_t lo, size_t hi) {
    return hfcl_collide_batch_f32(lib, shape1 + lo, shape2 + lo, pose1 + 7 * lo, pose2 + 7 * lo, hi - lo, req, out + lo);
  });
}
int hfcl_distance_batch_multi_f32(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                                  const hfcl_distance_request* req, hfcl_result_f32* out) {
  return run_sharded(m, n, "hfcl_distance_batch_multi_f32", [&](hfcl_lib* lib, size_t lo, size_t hi) {
    return hfcl_distance_batch_f32(lib, shape1 + lo, shape2 + lo, pose1 + 7 * lo, pose2 + 7 * lo, hi - lo, req, out + lo);
  });
}
int hfcl_collide_batch_multi_device(hfcl_multi* m, const uint32_t* const* d_shape1, const uint32_t* const* d_shape2, const double* const* d_tf1,
                                    const double* const* d_tf2, size_t n, const hfcl_collision_request* req, hfcl_result* const* d_gathered,
                                    void* const* streams) {
  const void* const* const arrays[4] = {reinterpret_cast<const void* const*>(d_shape1), reinterpret_cast<const void* const*>(d_shape2),
                                       reinterpret_cast<const void* const*>(d_tf1), reinterpret_cast<const void* const*>(d_tf2)};
  return run_gathered(m, n, "hfcl_collide_batch_multi_device", d_gathered, streams, arrays, 4, [&](int g, size_t count, hfcl_result* d_out, void* st) {
    return hfcl_collide_batch_device(m->libs[size_t(g)], d_shape1[g], d_shape2[g], d_tf1[g], d_tf2[g], count, req, d_out, nullptr, nullptr, st);
  });
}
int hfcl_distance_batch_multi_device(hfcl_multi* m, const uint32_t* const* d_shape1, const uint32_t* const* d_shape2, const double* const* d_tf1,
                                     const double* const* d_tf2, size_t n, const hfcl_distance_request* req, hfcl_result* const* d_gathered,
                                     void* const* streams) {
  const void* const* const arrays[4] = {reinterpret_cast<const void* const*>(d_shape1), reinterpret_cast<const void* const*>(d_shape2),
                                       reinterpret_cast<const void* const*>(d_tf1), reinterpret_cast<const void* const*>(d_tf2)};
  return run_gathered(m, n, "hfcl_distance_batch_multi_device", d_gathered, streams, arrays, 4, [&](int g, size_t count, hfcl_result* d_out, void* st) {
    return hfcl_distance_batch_device(m->libs[size_t(g)], d_shape1[g], d_shape2[g], d_tf1[g], d_tf2[g], count, req, d_out, nullptr, nullptr, st);
  });
}

}  // extern "C"
