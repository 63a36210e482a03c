"""The N > 1 path: launcher, ragged shards, padded all-gather, exchange formats.

CPU: bench.py's launcher and multigpu.RecordExchange over gloo with two spawned ranks.
GPU (-m gpu): the REAL bench.py code path with two ranks sharing device 0 over gloo (RCCL refuses duplicate devices): what
the ranks gather must equal, byte for byte, the records of the whole list computed by one rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_command_is_the_documented_launch(pkg):
    cmd = pkg.multigpu.rank_command("bench.py", ["--gpus", "8"], 8, port=29999, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", "29999", "bench.py", "--gpus", "8"]
    assert pkg.multigpu.device_of_rank(3) == 3
    assert [pkg.multigpu.device_of_rank(r, "0,0") for r in range(2)] == [0, 0]


def test_spawned_ranks_exchange_cpu(pkg, tmp_path):
    out = str(tmp_path / "gathered.npy")
    rc = pkg.multigpu.spawn_ranks(os.path.join(ROOT, "tests", "multigpu_worker.py"), [out, "1001"], 2, timeout=300)
    assert rc == 0
    ok, world, recv = np.load(out)
    assert ok == 1 and world == 2
    assert recv == 501 * 96  # a rank receives the other rank's padded shard: ceil(1001 / 2) records of 96 B


def test_compact_records_host_image(pkg):
    abi = pkg.abi
    r = np.zeros(5, dtype=abi.RESULT_DTYPE)
    r["distance"] = np.arange(5) - 2.5
    r["b1"], r["b2"], r["status"], r["num_contacts"] = 7, -1, 0x80 | 3, 1
    c = abi.compact_records(r)
    assert c.dtype.itemsize == 24 and np.array_equal(c["distance"], r["distance"]) and np.all(c["status"] == 0x83)
    f = np.zeros(3, dtype=abi.RESULT_F32_DTYPE)
    f["distance"], f["status"] = 1.5, 5
    assert abi.compact_records(f).dtype.itemsize == 8


def test_bench_refuses_missing_gpus_instead_of_hanging():
    """`python bench.py --gpus 2` as a plain process is the launcher; without enough devices it says so and exits != 0."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "GPU(s) visible" in (p.stderr + p.stdout)


def _bench(args, timeout=900):
    """Run bench.py; returns its FULL record (bench_full.json) after checking that the last stdout line is the compact form of it."""
    import tempfile
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    with tempfile.TemporaryDirectory() as d:
        env["HFCL_BENCH_FULL_DIR"] = d
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) <= 4096, p.stdout[-2000:]
        compact = json.loads(lines[0])
        full = json.load(open(os.path.join(d, "bench_full.json")))
    assert abs(compact["value"] - full["value"]) <= 1e-4 * full["value"] and compact["n_gpus"] == full["n_gpus"]
    return full


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["full", "compact"])
def test_two_ranks_on_one_gpu_strong_cfg5(torch_cuda, gather):
    """One ragged broadphase pair list cut over 2 ranks (sharding.shard_range), records exchanged: the gathered buffer
    equals the single-rank records of the whole list byte for byte (full and 24-B compact form)."""
    line = _bench(["--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--workload", "cfg5", "--scaling", "strong",
                   "--pairs", "100001", "--steps", "3", "--warmup", "1", "--gather", gather, "--verify-gather",
                   "--no-cpu-baseline"])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert cfg["shard"] == [0, 50001] and cfg["pairs_per_step_all_gpus"] == 100001
    assert cfg["gather"] == gather and cfg["all_gather_results"] is True
    assert cfg["gather_check"] == {"block_checksums": True, "equals_single_rank_run": True}
    words = 6 if gather == "compact" else 24
    assert cfg["gather_bytes_per_rank_per_step"] == {"sent": 50001 * words * 4, "received": 50001 * words * 4}
    assert line["value"] > 0


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_weak_cfg3(torch_cuda):
    """The headline's N > 1 form as a driver launches it (no extra flag): own batch per rank, fp32 records all-gathered; block checksums
    of both ranks agree, and every rank's block equals that rank's batch run alone on rank 0, byte for byte."""
    line = _bench(["--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--workload", "cfg3", "--pairs", "65537",
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and cfg["pairs_per_step_all_gpus"] == 2 * 65537
    assert cfg["gather_check"] == {"block_checksums": True, "equals_single_rank_run": True}
    assert cfg["gather_bytes_per_rank_per_step"]["received"] == 65537 * 44
    # what the first run on several GPUs needs to be believed: the ranks the backend reached, the exchange timed alone, every rank's step
    # without exchange (weak scaling: the N = 1 step)
    ex, pr = cfg["exchange"], cfg["per_rank_ms_no_exchange"]
    assert ex["ranks_seen"] == 2 and ex["ms"] > 0 and ex["bus_GBps"] > 0 and ex["bytes_received_per_rank"] == 65537 * 44
    assert 0 < pr["min"] <= pr["max"] and pr["max"] < 50 * line["ms_per_step"]


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["full", "compact"])
def test_rccl_exchange_branch_with_one_rank(torch_cuda, gather, monkeypatch):
    """The branch a driver meets with 8 ranks -- process group "nccl" (= RCCL), records all-gathered with async_op on the
    communication stream behind an event of the launch stream, Work.wait() on the launch stream two steps later -- executed
    on this box with a world of one (HFCL_BENCH_FORCE_DIST=1: every N > 1 code path of bench.py, one rank)."""
    monkeypatch.setenv("HFCL_BENCH_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_PORT", "29577")
    line = _bench(["--workload", "cfg3", "--pairs", "65537", "--steps", "5", "--warmup", "1", "--backend", "nccl", "--gather", gather,
                   "--no-cpu-baseline"])
    cfg = line["config"]
    assert cfg["backend"] == "nccl" and cfg["all_gather_results"] is True and cfg["gather"] == gather
    assert cfg["gather_check"]["block_checksums"] is True
    assert cfg["gather_bytes_per_rank_per_step"]["sent"] == 65537 * (8 if gather == "compact" else 44)
    assert cfg["exchange"]["ranks_seen"] == 1 and cfg["exchange"]["ms"] > 0  # (the all-gather alone, device events on the communication stream)
    assert line["value"] > 0


@pytest.mark.gpu
def test_compact_records_device(torch_cuda, pkg):
    """hfcl_compact_results_device{,_f32}: every field of a compact record is a bit copy of the full record's."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=30011, seed=3)
    req = wl.make_request(b, abi)
    lib = pkg.Library(b.lib, device=0)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    out = torch.zeros(len(b) * 24, dtype=torch.int32, device=dev)
    lib.collide_device(t(b.s1.astype(np.int32)), t(b.s2.astype(np.int32)), t(b.tf1), t(b.tf2), len(b), req, out)
    comp = torch.zeros(len(b) * 6, dtype=torch.int32, device=dev)
    lib.compact_results_device(out, len(b), comp)
    torch.cuda.synchronize()
    full = out.cpu().numpy().view(abi.RESULT_DTYPE)
    got = comp.cpu().numpy().view(abi.RESULT_COMPACT_DTYPE)
    assert got.tobytes() == abi.compact_records(full).tobytes()
    b3 = wl.cfg3_convex_convex(n=20003, seed=2)
    lib3 = pkg.Library(b3.lib, device=0)
    out3 = torch.zeros(len(b3) * 11, dtype=torch.int32, device=dev)
    lib3.distance_device_f32(t(b3.s1.astype(np.int32)), t(b3.s2.astype(np.int32)), t(b3.pose1_f32), t(b3.pose2_f32), len(b3),
                             wl.make_request(b3, abi), out3)
    comp3 = torch.zeros(len(b3) * 2, dtype=torch.int32, device=dev)
    lib3.compact_results_device(out3, len(b3), comp3, f32=True)
    torch.cuda.synchronize()
    assert comp3.cpu().numpy().view(abi.RESULT_COMPACT_F32_DTYPE).tobytes() == \
        abi.compact_records(out3.cpu().numpy().view(abi.RESULT_F32_DTYPE)).tobytes()
    lib.close()
    lib3.close()


# ---- the C-ABI multi-device entry points (include/hppfcl_amd.h: hfcl_multi_*) ----
def test_c_abi_shard_range_equals_python(pkg):
    """hfcl_shard_range (host code, no GPU) = sharding.shard_range: ceil(n / world) pairs per rank, ragged and empty tails."""
    for n in (0, 1, 7, 8, 9, 1000, 100001):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                lo, hi = pkg.engine.shard_range(n, r, world)
                assert (lo, hi) == pkg.sharding.shard_range(n, r, world)
                assert lo == min(n, covered)
                covered = hi
            assert covered == n


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cfg5_mixed", "meshes"])
def test_c_abi_multi_equals_single_library(pkg, case):
    """hfcl_{collide,distance}_batch_multi over two replicas (both on device 0: two libraries, two host threads, two pipelines) against
    the single-library call on the same 100 001-pair list: every record and every cached guess, byte for byte."""
    abi, wl = pkg.abi, pkg.workloads
    if case == "meshes":
        b = wl.mesh_vs_shapes(n=20001, seed=4)
    else:
        b = wl.cfg5_mixed(n=100001, seed=2)
    single = wl.make_library(pkg, b)
    multi = pkg.MultiLibrary(b.lib, devices=(0, 0))
    assert len(multi) == 2
    for m in getattr(b, "meshes", []) or []:
        multi.add_bvh(m)
    try:
        for kind in ("collide", "distance"):
            req = abi.default_collision_request() if kind == "collide" else abi.default_distance_request()
            want_guess = case != "meshes"
            ref = getattr(single, kind)(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=want_guess)
            got = getattr(multi, kind)(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=want_guess)
            if want_guess:
                assert ref[1].tobytes() == got[1].tobytes(), kind + ": cached guesses"
                ref, got = ref[0], got[0]
            # (mesh distance() included: which walks share a wave differs between the two calls, the records do not -- a walk whose
            # reported pair could depend on it is walked again in the reference's order, include/hppfcl_amd.h)
            assert got.tobytes() == ref.tobytes(), kind
    finally:
        single.close()
        multi.close()


@pytest.mark.gpu
def test_c_abi_multi_f32_equals_single_library(pkg, torch_cuda):
    """hfcl_{collide,distance}_batch_multi_f32 over two replicas on device 0 against hfcl_*_batch_f32 of one library on the same 100 001-pair
    list (fp32 path, host arrays): byte for byte."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg3_convex_convex(n=100001, seed=6)
    single = wl.make_library(pkg, b)
    multi = pkg.MultiLibrary(b.lib, devices=(0, 0))
    try:
        for kind in ("collide", "distance"):
            req = abi.default_collision_request() if kind == "collide" else abi.default_distance_request()
            ref = getattr(single, kind + "_f32")(b.s1, b.s2, b.pose1_f32, b.pose2_f32, req)
            got = getattr(multi, kind + "_f32")(b.s1, b.s2, b.pose1_f32, b.pose2_f32, req)
            assert got.tobytes() == ref.tobytes(), kind
    finally:
        single.close()
        multi.close()


@pytest.mark.gpu
def test_c_abi_multi_device_resident(pkg, torch_cuda):
    """The device-resident form: one replica gathers nothing and equals hfcl_distance_batch_device; a device listed twice is refused
    (one communicator rank per device) with the reason in hfcl_last_error()."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=30001, seed=3)
    b.kind = "distance"
    req = abi.default_distance_request()
    dev = torch.device("cuda:0")
    d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
    d_t1, d_t2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
    one = pkg.MultiLibrary(b.lib, devices=(0,))
    single = pkg.Library(b.lib)
    try:
        out1 = torch.zeros(len(b) * 24, dtype=torch.int32, device=dev)
        out2 = torch.zeros(len(b) * 24, dtype=torch.int32, device=dev)
        one.distance_device_gathered([d_s1], [d_s2], [d_t1], [d_t2], len(b), req, [out1])
        single.distance_device(d_s1, d_s2, d_t1, d_t2, len(b), req, out2)
        torch.cuda.synchronize()
        assert torch.equal(out1, out2)
    finally:
        one.close()
        single.close()
    two = pkg.MultiLibrary(b.lib, devices=(0, 0))
    try:
        per = (len(b) + 1) // 2
        g = [torch.zeros(2 * per * 24, dtype=torch.int32, device=dev) for _ in range(2)]
        with pytest.raises(pkg.EngineError) as e:
            two.distance_device_gathered([d_s1, d_s1], [d_s2, d_s2], [d_t1, d_t1], [d_t2, d_t2], len(b), req, g)
        assert e.value.code == abi.ERR_INVALID_ARGUMENT and "listed twice" in str(e.value)
    finally:
        two.close()


@pytest.mark.gpu
def test_cpp_multi_device_all_gather(pkg):
    """tests/cpp/test_multi_device.cpp: hfcl_{distance,collide}_batch_multi_device over every visible device -- the in-place ncclAllGather
    of csrc/hfcl_multi.hip with more than one rank -- every device's gathered buffer against the single-library records, the ranks the
    communicator reports, the caller's device restored.  SKIPS (does not pass) where fewer than two devices are visible."""
    import os
    import subprocess
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-s", "-C", d])
    r = subprocess.run([os.path.join(d, "test_multi_device")], capture_output=True, text=True, timeout=600)
    if r.returncode == 77:
        pytest.skip(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_abi_multi_hardening(pkg, torch_cuda):
    """hfcl_multi_*: null handles and null pointer arrays are refused with HFCL_ERR_INVALID_ARGUMENT (not dereferenced), the caller's
    current device survives a call, an option reaches every replica, and the gather statistics of a one-replica batch say "no collective"."""
    import ctypes as C
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    d = pkg.engine.dll()
    req = abi.default_distance_request()
    for fn in (d.hfcl_distance_batch_multi, d.hfcl_collide_batch_multi):
        assert fn(None, None, None, None, None, C.c_size_t(0), C.byref(req), None, None, None) == abi.ERR_INVALID_ARGUMENT
    assert d.hfcl_distance_batch_multi_device(None, None, None, None, None, C.c_size_t(0), C.byref(req), None, None) == abi.ERR_INVALID_ARGUMENT
    assert d.hfcl_multi_set_shapes(None, None, C.c_size_t(0), None, C.c_size_t(0)) == abi.ERR_INVALID_ARGUMENT
    assert d.hfcl_multi_add_bvh(None, None, C.c_size_t(0), None, C.c_size_t(0), None, C.c_size_t(0)) < 0
    b = wl.cfg5_mixed(n=5001, seed=3)
    one = pkg.MultiLibrary(b.lib, devices=(0,), options={"split": 1})
    try:
        assert d.hfcl_distance_batch_multi_device(one._h, None, None, None, None, C.c_size_t(len(b)), C.byref(req), None, None) == abi.ERR_INVALID_ARGUMENT
        assert "null pointer array" in pkg.engine.last_error()
        one.set_option("HFCL_CVX_W", 4)
        with pytest.raises(pkg.EngineError):
            one.set_option("no_such_option", 1)
        with pytest.raises(pkg.EngineError):
            one.set_option("cvx_w", 3)
        dev = torch.device("cuda:0")
        t = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.tf1, b.tf2)]
        out = torch.zeros(len(b) * 24, dtype=torch.int32, device=dev)
        one.distance_device_gathered([t[0]], [t[1]], [t[2]], [t[3]], len(b), req, [out])
        torch.cuda.synchronize()
        assert torch.cuda.current_device() == 0
        g = one.last_gather()
        assert g["ranks"] == 1 and g["ms"] is None and g["bytes_per_rank"] == len(b) * 96
    finally:
        one.close()
