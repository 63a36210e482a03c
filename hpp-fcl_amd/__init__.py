"""hpp-fcl_amd: MI355X-native batched narrow-phase collision/distance engine.

Host-side Python surface over the C ABI (include/hppfcl_amd.h).  The package directory name
contains a hyphen (it is the project name); import it through `tests/conftest.py:load_pkg()` /
`__graft_entry__.load_pkg()` which register it as module `hppfcl_amd`.

This module holds plumbing only.  All compute happens in csrc/libhppfcl_amd.so (HIP kernels,
gfx950).  There is no CPU fallback: without the built library or without a GPU every compute
call raises."""
from . import abi, bvh_builder, compat, engine, geometry, multigpu, sharding, workloads  # noqa: F401
from .engine import EngineError, Library, MultiLibrary  # noqa: F401
from .geometry import ShapeLibrary, make_pose, quat_to_matrix  # noqa: F401
