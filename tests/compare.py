"""Parity checks shared by the CPU (hostsim) and GPU tests: engine records vs oracle records."""
import numpy as np


def check_parity(abi, got, ref, *, dist_tol, point_tol, flag_band, name="", allow_bad_frac=0.0, fp32=False, collect_only=False):
    """got/ref: result record arrays (RESULT_DTYPE or RESULT_F32_DTYPE).  Asserts:
       - contact flags, GJK status and EPA status equal, except where |d_ref| <= flag_band
         (decision boundary) -- bit-exact integer outputs;
       - |d - d_ref| <= dist_tol * (1 + |d_ref|);
       - witness points / normals within point_tol where both are finite and the witness is unique enough.
    Returns a dict of statistics."""
    n = len(ref)
    d_ref = ref["distance"].astype(np.float64)
    d_got = got["distance"].astype(np.float64)
    st_r, st_g = ref["status"], got["status"]
    near = np.abs(d_ref) <= flag_band
    contact_eq = abi.status_contact(st_r) == abi.status_contact(st_g)
    gjk_eq = abi.status_gjk(st_r) == abi.status_gjk(st_g)
    epa_eq = abi.status_epa(st_r) == abi.status_epa(st_g)
    skipped_eq = abi.status_skipped(st_r) == abi.status_skipped(st_g)
    assert skipped_eq.all(), name + ": skipped-record flags differ"
    bad_flags = ~(contact_eq | near)
    finite = np.isfinite(d_ref) & (np.abs(d_ref) < 1e300)
    dd = np.where(finite, np.abs(d_got - d_ref), 0.0)
    same_inf = np.where(~finite, (d_got == d_ref) | (fp32 & (np.abs(d_got) > 1e37)), True)
    bad_d = (dd > dist_tol * (1 + np.abs(np.where(finite, d_ref, 0)))) | ~same_inf
    # GJK early stop (status 2): the reported distance is only a lower bound above
    # distance_upper_bound (narrowphase.h:589-608); its value depends on the iteration at which the
    # bound was crossed, so only require both to be early-stopped lower bounds of the same sign.
    early = (abi.status_gjk(st_r) == 2) & (abi.status_gjk(st_g) == 2)
    bad_d &= ~(early & (d_got > 0) & (d_ref > 0))
    nan_r = np.isnan(ref["p1"]).any(axis=1)
    nan_g = np.isnan(got["p1"]).any(axis=1)
    bad_nan = (nan_r != nan_g) & ~near
    fin = ~nan_r & ~nan_g & finite
    # p2 - p1 = d * n is unique even when the witness pair is not: compare that, and the normal
    sep_r = (ref["p2"] - ref["p1"]).astype(np.float64)
    sep_g = (got["p2"] - got["p1"]).astype(np.float64)
    dsep = np.where(fin[:, None], np.abs(sep_g - sep_r), 0).max(axis=1)
    bad_sep = dsep > point_tol * (1 + np.abs(np.where(finite, d_ref, 0)))
    stats = dict(n=n, contact_frac=float(abi.status_contact(st_r).mean()),
                 flag_mismatch=int(bad_flags.sum()), gjk_status_mismatch=int((~gjk_eq & ~near).sum()),
                 epa_status_mismatch=int((~epa_eq & ~near).sum()), dist_bad=int(bad_d.sum()),
                 nan_mismatch=int(bad_nan.sum()), sep_bad=int(bad_sep.sum()),
                 max_dd=float(dd.max()) if n else 0.0,
                 p999_dd=float(np.quantile(dd, 0.999)) if n else 0.0, max_dsep=float(dsep.max()) if n else 0.0)
    if collect_only:  # the caller enumerates the violating records itself (per-record masks instead of assertions)
        stats.update(flag_bad_mask=bad_flags, dist_bad_mask=bad_d, nan_bad_mask=bad_nan, sep_bad_mask=bad_sep)
        return stats
    allowed = int(allow_bad_frac * n)
    assert stats["flag_mismatch"] <= allowed, "%s: contact flags differ outside the decision band: %s" % (name, stats)
    assert stats["dist_bad"] <= allowed, "%s: distances out of tolerance: %s" % (name, stats)
    assert stats["nan_mismatch"] <= allowed, "%s: NaN pattern differs: %s" % (name, stats)
    assert stats["sep_bad"] <= allowed, "%s: separation vectors out of tolerance: %s" % (name, stats)
    if not fp32:
        assert stats["gjk_status_mismatch"] <= allowed, "%s: GJK statuses differ: %s" % (name, stats)
        assert stats["epa_status_mismatch"] <= allowed, "%s: EPA statuses differ: %s" % (name, stats)
    return stats


def check_properties(abi, res, *, tol, name=""):
    """Size-independent invariants of the reference's own property test
    (test/normal_and_nearest_points.cpp:60-73): p2 = p1 + d*n, |n| = 1, |p2-p1| = |d|."""
    d = res["distance"].astype(np.float64)
    ok = np.isfinite(res["p1"]).all(axis=1) & np.isfinite(res["normal"]).all(axis=1) & (np.abs(d) < 1e30)
    p1, p2, n = (res[k][ok].astype(np.float64) for k in ("p1", "p2", "normal"))
    d = d[ok]
    scale = 1 + np.abs(d)
    assert np.all(np.abs(np.linalg.norm(n, axis=1) - 1) < tol), name + ": |normal| != 1"
    assert np.all(np.abs(p1 + d[:, None] * n - p2).max(axis=1) < tol * scale), name + ": p2 != p1 + d n"
    assert np.all(np.abs(np.linalg.norm(p2 - p1, axis=1) - np.abs(d)) < tol * scale), name + ": |p2-p1| != |d|"
    return int(ok.sum())
