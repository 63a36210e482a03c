"""The reference's known-answer tests run against BOTH solvers: `oracle` (the CPU restatement, `-m "not gpu"`) and `gpu`
(the HIP path through the C ABI, `-m gpu`).  A test module imports the `oracle` fixture below; it shadows conftest's and
hands the test either the oracle module or an adapter with the oracle's distance_batch / collide_batch signature on top of
engine.Library.  KATs on oracle-only entry points (raw GJK on a Minkowski difference, simplex projection) have no device
entry point and skip on the gpu leg -- the device's projection / GJK core is pinned on them through tests/hostsim."""
import numpy as np
import pytest


class _RawLibrary:
    def __init__(self, shapes, verts):
        self._s, self._v = np.ascontiguousarray(shapes), np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)

    def shapes_array(self):
        return self._s

    def vertices_array(self):
        return self._v


class GpuSolver:
    """oracle_binding's batch signatures on the HIP engine (device 0).  No fallback: raises without a GPU."""

    def __init__(self, pkg):
        self.pkg = pkg
        self._libs = {}

    def _lib(self, shapes, verts):
        raw = _RawLibrary(shapes, np.zeros((0, 3)) if verts is None else verts)
        key = (raw.shapes_array().tobytes(), raw.vertices_array().tobytes())
        if key not in self._libs:
            if len(self._libs) > 8:
                for lib in self._libs.values():
                    lib.close()
                self._libs = {}
            self._libs[key] = self.pkg.Library(raw, device=0)
        return self._libs[key]

    def _run(self, which, shapes, verts, s1, s2, tf1, tf2, req, **kw):
        unknown = set(kw) - {"n_threads"}
        assert not unknown, "GpuSolver: unsupported keyword(s) %s" % unknown
        lib = self._lib(shapes, verts)
        tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
        tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
        try:
            return getattr(lib, which)(np.asarray(s1), np.asarray(s2), tf1, tf2, req)
        except self.pkg.EngineError as e:
            # the C ABI reports what the reference throws as std::invalid_argument (collision.cpp:82-85,95-100) as error
            # codes; the oracle binding raises ValueError for them, like include/hppfcl_amd_compat.hpp rethrows
            if e.code in (self.pkg.abi.ERR_INVALID_ARGUMENT, self.pkg.abi.ERR_UNSUPPORTED_PAIR):
                raise ValueError(str(e)) from e
            raise

    def distance_batch(self, shapes, verts, s1, s2, tf1, tf2, req=None, **kw):
        return self._run("distance", shapes, verts, s1, s2, tf1, tf2, req, **kw)

    def collide_batch(self, shapes, verts, s1, s2, tf1, tf2, req=None, **kw):
        return self._run("collide", shapes, verts, s1, s2, tf1, tf2, req, **kw)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        pytest.skip("oracle.%s has no device entry point: this known-answer test pins the oracle only" % name)

    def close(self):
        for lib in self._libs.values():
            lib.close()
        self._libs = {}


@pytest.fixture(params=["oracle", pytest.param("gpu", marks=pytest.mark.gpu)])
def oracle(request, oracle, pkg):  # noqa: F811 -- shadows conftest's fixture on purpose and wraps it
    if request.param == "oracle":
        yield oracle
        return
    s = GpuSolver(pkg)
    yield s
    s.close()
