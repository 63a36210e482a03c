"""The order logic of the wave continuation kernels (k_bvh_coop / k_bvh_shape_coop / k_bvh_distance_coop, DESIGN.md section 3
item 6c), as a model: a window of stack entries applied by scans must leave the walk in the state the sequential visit of the
same entries leaves it in -- bound, recorded distance, witness, first contact, nearest triangle -- ties included.  Pure numpy /
Python: this pins the algorithm the kernels implement with ballots and shuffles, independently of the GPU."""
import numpy as np

BIG = np.finfo(np.float64).max


def collide_sequential(kinds, vals, recs, contacts, dlb, rec, wit):
    """kinds: 'b' disjoint box (bound vals[i], recorded distance recs[i]) or 'l' leaf (vals[i] = distance - margin, recs[i] = distance,
    contacts[i]).  Returns (dlb, rec, witness index or the one passed in, index of the contact that ended the walk or -1)."""
    for i, k in enumerate(kinds):
        if k == "b":
            if not (dlb <= 0) and vals[i] < dlb:  # updateDistanceLowerBoundFromBV
                dlb, rec = vals[i], recs[i]
        else:
            if vals[i] < dlb:  # updateDistanceLowerBoundFromLeaf
                dlb, rec, wit = vals[i], recs[i], i
            if contacts[i]:
                return dlb, rec, wit, i
    return dlb, rec, wit, -1


def collide_scans(kinds, vals, recs, contacts, dlb, rec, wit):
    """The kernels' form: first contact by ballot, exclusive prefix minimum, 'lowered' mask, the FIRST entry that attains the
    window's minimum sets the recorded distance, the LAST leaf that lowered the bound is the witness."""
    n = len(kinds)
    cidx = [i for i in range(n) if kinds[i] == "l" and contacts[i]]
    c = cidx[0] if cidx else n
    visit = np.arange(n) <= c
    bnd = np.where(visit, vals, BIG)
    pre = np.minimum.accumulate(np.concatenate([[BIG], bnd[:-1]]))  # exclusive prefix minimum
    before = np.minimum(dlb, pre)
    lowered = visit & (bnd < before)
    if lowered.any():
        wmin = bnd.min()
        src = int(np.nonzero(visit & (bnd == wmin))[0][0])
        dlb, rec = wmin, recs[src]
    leaf_low = [i for i in range(n) if lowered[i] and kinds[i] == "l"]
    if leaf_low:
        wit = leaf_low[-1]
    return dlb, rec, wit, (c if c < n else -1)


def test_collide_window_equals_sequential_visit():
    rng = np.random.default_rng(7)
    grid = np.array([0.0, 0.125, 0.25, 0.25, 0.5, 0.5, 0.5, 1.0, 2.0])  # few distinct values: ties everywhere
    for trial in range(20000):
        n = int(rng.integers(1, 65))
        kinds = rng.choice(["b", "l"], n)
        vals = rng.choice(grid, n).astype(np.float64)
        vals = np.where(kinds == "l", vals - rng.choice([0.0, 0.0, 0.75], n), vals)  # some leaves penetrate
        margin = float(rng.choice([0.0, 0.03]))
        recs = np.where(kinds == "b", vals + margin, vals + margin)
        contacts = (kinds == "l") & (vals <= 0.0)
        dlb0 = float(rng.choice([BIG, 1.0, 0.5, 0.25, 0.0, -0.1]))
        a = collide_sequential(kinds, vals, recs, contacts, dlb0, dlb0 + margin, -7)
        b = collide_scans(kinds, vals, recs, contacts, dlb0, dlb0 + margin, -7)
        assert a == b, (trial, kinds, vals, dlb0, a, b)


def distance_sequential(vals, bounds, epa, mind, src):
    """Leaves in order: skipped when canStop (bound >= 0 and bound >= the minimum at its turn); the minimum is lowered by strictly
    smaller distances; a visited leaf that needs EPA ends the walk.  Returns (mind, index of the leaf that set it, index that ended or -1)."""
    for i in range(len(vals)):
        if bounds[i] >= 0 and bounds[i] >= mind:  # (a NaN bound never skips)
            continue
        if epa[i]:
            return mind, src, i
        if mind > vals[i]:
            mind, src = vals[i], i
    return mind, src, -1


def distance_segments(vals, bounds, epa, mind, src):
    """The kernels' form: one record-setting leaf at a time (first candidate by ballot), segments between the leaves that need EPA."""
    n = len(vals)
    start, run = 0, mind
    while True:
        cs = [i for i in range(start, n) if epa[i]]
        c = cs[0] if cs else n
        while True:
            cand = [i for i in range(start, c) if not epa[i] and not (bounds[i] >= 0 and bounds[i] >= run) and vals[i] < run]
            if not cand:
                break
            src = cand[0]
            run = vals[src]
            start = src + 1
        if c >= n:
            return run, src, -1
        if not (bounds[c] >= 0 and bounds[c] >= run):
            return run, src, c
        start = c + 1
        if start >= n:
            return run, src, -1


def test_distance_window_equals_sequential_visit():
    rng = np.random.default_rng(11)
    grid = np.array([-0.5, -0.25, 0.0, 0.0, 0.125, 0.25, 0.25, 0.5, 1.0])
    for trial in range(20000):
        n = int(rng.integers(1, 65))
        vals = rng.choice(grid, n).astype(np.float64)
        # bounds: clamped at 0 (so NOT below a penetrating distance), sometimes NaN (unbounded solids), sometimes loose
        bounds = np.maximum(vals - rng.choice([0.0, 0.0, 0.125, 0.5], n), 0.0)
        bounds = np.where(rng.random(n) < 0.1, np.nan, bounds)
        epa = (vals < 0) & (rng.random(n) < 0.5)
        mind0 = float(rng.choice([BIG, 1.0, 0.5, 0.25, 0.0, -0.25]))
        with np.errstate(invalid="ignore"):
            a = distance_sequential(vals, bounds, epa, mind0, -7)
            b = distance_segments(vals, bounds, epa, mind0, -7)
        assert a == b, (trial, vals, bounds, epa, mind0, a, b)
