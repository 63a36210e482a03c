"""The fold that lets a collide() walk be cut into pieces (k_bvh_collide's task levels, BvhSplit::cut_ticks of k_bvh_coop / k_bvh_shape_coop;
k_bvh_combine) as a model on the CPU.

The sequential walk keeps a running lower bound over its events (a box pair found disjoint: its bound, unless the running bound is
already <= 0; a triangle test: its distance minus the margin), reports the witness of the LAST triangle that lowered the bound when it
was visited (updateDistanceLowerBoundFromLeaf, collision_data.h:1186-1197 of the reference), and ends at the first contact.  A piece
of the walk starts from an empty state and reports (minimum, value and witness of its own last bound-lowering triangle, first contact);
the pieces are folded in DFS order: a piece's triangle becomes the witness when its value also lies below the bound as it stood before
the piece, the minimum is the minimum, the first contact ends the fold.  A piece can itself be a fold of pieces (cut again), and the
unit that was cut contributes the state it had reached.  This file checks that rule against the sequential walk on random event
sequences drawn from a handful of values (ties everywhere), cut at random places, two levels deep -- the device runs exactly this
(tests/test_gpu_parity.py::test_bvh_collide_forms_agree[cut], tests/test_bvh_shape.py::test_gpu_mesh_solid_long_walks compare its
records with the oracle's and with the uncut walks')."""
import random

INF = float("inf")
THRESHOLD = 0.0


def sequential(events, state=None):
    """-> (bound, witness id or None, contact id or None) after `events`, starting from `state` = (bound, witness)"""
    dlb, wit = state if state else (INF, None)
    for kind, v, ident in events:
        if kind == "bv":
            if not dlb <= 0 and v < dlb:  # updateDistanceLowerBoundFromBV: a box cannot find a negative distance
                dlb = v
        else:
            if v < dlb:  # updateDistanceLowerBoundFromLeaf
                dlb, wit = v, ident
            if v <= THRESHOLD:  # leafCollides: a contact; canStop() with one contact asked for
                return dlb, wit, ident
    return dlb, wit, None


def summary(events):
    """A piece walked from an empty state: (minimum, cand value, cand id, contact)."""
    dlb, cand_val, cand, contact = INF, INF, None, None
    for kind, v, ident in events:
        if kind == "bv":
            if not dlb <= 0 and v < dlb:
                dlb = v
        else:
            if v < dlb:
                dlb, cand_val, cand = v, v, ident
            if v <= THRESHOLD:
                contact = ident
                break
    return dlb, cand_val, cand, contact


def fold(own, children):
    """k_bvh_combine: `own` = (minimum, cand value, cand id, contact) of the unit that was cut, children in DFS order."""
    dlb, cand_val, cand, contact = own
    for c_dlb, c_cand_val, c_cand, c_contact in children:
        if c_cand_val < dlb:  # the child's last bound-lowering triangle also lowers the bound as it stood before the child
            cand_val, cand = c_cand_val, c_cand
        if c_dlb < dlb:
            dlb = c_dlb
        if c_contact is not None:
            contact = c_contact
            break
    return dlb, cand_val, cand, contact


def cut_walk(events, rng, depth):
    """The unit walks a prefix, is cut, and its rest -- in chunks -- is walked by units that may be cut again."""
    if depth == 0 or len(events) < 2 or rng.random() < 0.2:
        return summary(events)
    k = rng.randint(0, len(events) - 1)  # events the unit itself gets through before its time is up
    own = summary(events[:k])
    if own[3] is not None:  # it met a contact first: never cut
        return own
    rest = events[k:]
    n_chunks = rng.randint(1, min(6, len(rest)))
    bounds = sorted(rng.sample(range(1, len(rest)), n_chunks - 1)) if n_chunks > 1 else []
    pieces = [rest[a:b] for a, b in zip([0] + bounds, bounds + [len(rest)])]
    return fold(own, [cut_walk(p, rng, depth - 1) for p in pieces])


def test_fold_of_cut_walks_is_the_sequential_walk():
    rng = random.Random(3)
    values = [2.0, 1.5, 1.5, 1.0, 1.0, 0.5, 0.25, 0.25, 0.0, -0.5]
    n_contact = n_witness_from_piece = 0
    for trial in range(4000):
        n = rng.randint(1, 40)
        p_contact = rng.choice([0.0, 0.02, 0.2])
        events = []
        for i in range(n):
            if rng.random() < 0.5:
                events.append(("bv", rng.choice([v for v in values if v >= 0]), None))
            else:
                v = rng.choice(values[:8]) if rng.random() >= p_contact else rng.choice(values[8:])
                events.append(("leaf", v, i))
        want = sequential(events)
        got = cut_walk(events, rng, depth=3)
        assert (got[0], got[2], got[3]) == want, (trial, events, got, want)
        n_contact += want[2] is not None
        n_witness_from_piece += want[1] is not None
    assert n_contact > 500 and n_witness_from_piece > 2500


def test_taking_any_lowering_triangle_of_a_piece_would_not_do():
    """The rule is about the piece's LAST own lowering triangle: with its first one the fold reports another witness than the walk."""
    events = [("leaf", 1.5, 0), ("bv", 1.0, None), ("leaf", 0.5, 2), ("leaf", 0.25, 3)]
    want = sequential(events)
    assert want == (0.25, 3, None)
    own = summary(events[:1])
    piece = summary(events[1:])
    assert fold(own, [piece])[2] == 3
    first_lowering = (piece[0], 0.5, 2, None)  # (what a piece that reported its first lowering triangle would hand over)
    assert fold(own, [first_lowering])[2] != want[1]
