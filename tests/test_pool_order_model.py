"""The order logic of k_bvh_distance_pool as a model on the CPU.

The kernel evaluates the triangle pairs and box pairs of a distance() walk in whatever order its windows, its pooled rounds and
its deferrals produce, and must still report what distanceRecurse reports: the minimal distance and the FIRST triangle pair in
DFS order that attains it (DistanceResult::update lowers on `<` only; an entry is skipped when its bound is >= the minimum of
that moment).  It does so with a marker per walk: p = how many entries of the (DFS-ordered) stack stand behind the pair that
holds the minimum.  This file restates exactly that bookkeeping -- the window scan, the rule for dropping entries, the rule for
replacing the minimum, the write-back and the marker's update, the hand-over from the lanes' sequential walk -- in plain Python,
drives it with RANDOM schedules (which window entries get their turn in a trip, whether triangle pairs are evaluated or left on
the stack), and compares with the recursive walk on random pair trees whose bounds and distances come from a handful of values:
ties between distances, between bounds, and between a bound and the minimum, everywhere.

What the model assumes and the kernel cannot: bounds that never exceed a distance beneath them.  (In floating point they can,
by an ulp; that is the enumerated class of tests/test_gpu_parity.py: _check_distance_records and the `margin` of the kernel.)"""
import random

BIG = float("inf")


class Tree:
    """A random tree of node pairs: inner nodes carry (bound of child a, bound of child c), leaves a distance and an id."""

    def __init__(self, rng, depth, values):
        self.rng, self.values, self.n_leaves = rng, values, 0
        self.root = self._make(depth)
        self._bounds(self.root)

    def _make(self, depth):
        if depth == 0 or (depth < 4 and self.rng.random() < 0.25):
            self.n_leaves += 1
            return {"leaf": True, "d": self.rng.choice(self.values), "id": self.n_leaves - 1}
        return {"leaf": False, "a": self._make(depth - 1), "c": self._make(depth - 1)}

    def _bounds(self, node):
        """min distance beneath; a node's bound is any of the values that does not exceed it (valid, not monotone along a path)"""
        if node["leaf"]:
            node["min"] = node["d"]
        else:
            node["min"] = min(self._bounds(node["a"]), self._bounds(node["c"]))
        ok = [v for v in self.values if v <= node["min"]] + [0.0]
        node["bound"] = self.rng.choice(ok)
        return node["min"]


def sequential(tree, seed_d):
    """distanceRecurse (traversal_recurse.cpp:153-203) with the minimum seeded by preprocess(): returns (distance, leaf id or -1)."""
    best = [seed_d, -1]

    def rec(node):
        if node["leaf"]:
            if node["d"] < best[0]:  # DistanceResult::update
                best[0], best[1] = node["d"], node["id"]
            return
        a, c = node["a"], node["c"]
        d1, d2 = a["bound"], c["bound"]
        order = (c, a) if d2 < d1 else (a, c)
        for ch in order:
            if not (ch["bound"] >= best[0]):  # canStop
                rec(ch)

    rec(tree.root)
    return best[0], best[1]


def lanes_then_pool(tree, seed_d, rng, budget, win):
    """The lanes' sequential walk for `budget` steps on an explicit stack (top = next), then the pool's trips."""
    mind, best = seed_d, -1
    stack = [tree.root]  # bottom first; the root's bound is never tested (-1 in the kernel)
    first = True
    steps = 0
    while stack and steps < budget:
        node = stack.pop()
        steps += 1
        if not first and node["bound"] >= mind:
            continue
        first = False
        if node["leaf"]:
            if node["d"] < mind:
                mind, best = node["d"], node["id"]
            continue
        a, c = node["a"], node["c"]
        c_first = c["bound"] < a["bound"]
        stack += [a, c] if c_first else [c, a]  # the one visited first on top
    root_pending = first  # (budget 0: the root entry is still the untested one)
    # ---- hand-over: everything on the stack stands behind what the lane has visited
    p = len(stack)
    while stack:
        sp = len(stack)
        w = min(win, sp)
        base = sp - w
        window = [stack[sp - 1 - j] for j in range(w)]  # j = 0: top of the stack
        idx = [sp - 1 - j for j in range(w)]
        alive = []
        for j, node in enumerate(window):
            if root_pending and node is tree.root:
                alive.append(True)
                continue
            b = node["bound"]
            alive.append(not (b >= mind if idx[j] < p else b > mind))  # behind the minimum: canStop; in front: a tie is kept
        root_pending = False
        # the schedule: which of the live entries get their turn (pooled rounds are full, triangle pairs wait for a pass)
        do_leaves = rng.random() < 0.6
        turn = [alive[j] and (rng.random() < 0.7 if not window[j]["leaf"] else do_leaves and rng.random() < 0.8) for j in range(w)]
        if not any(turn) and any(alive):  # (the kernel always makes progress)
            turn = list(alive)
        # ---- triangle pairs evaluated in this trip: the smallest value, the first in DFS order among equals, and against the
        # standing minimum a tie wins only in front of it
        jw = -1
        for j in range(w):
            if turn[j] and window[j]["leaf"]:
                d = window[j]["d"]
                cand = d < mind or (d == mind and idx[j] >= p)
                if cand and (jw < 0 or d < window[jw]["d"]):
                    jw = j
        if jw >= 0:
            mind, best = window[jw]["d"], window[jw]["id"]
        # ---- write-back in order (deeper entries first), the marker counted over what the entries behind it became
        out, later_cnt = [], 0
        for j in range(w - 1, -1, -1):
            node = window[j]
            later = (j > jw) if jw >= 0 else (idx[j] < p)
            if not alive[j] or (turn[j] and node["leaf"]):
                items = []
            elif turn[j]:
                a, c = node["a"], node["c"]
                c_first = c["bound"] < a["bound"]
                items = [a, c] if c_first else [c, a]  # the one visited first on top
            else:
                items = [node]
            out += items
            later_cnt += len(items) if later else 0
        p = (base if jw >= 0 else min(p, base)) + later_cnt
        stack = stack[:base] + out
    return mind, best


def test_pool_reports_the_sequential_walks_pair():
    rng = random.Random(11)
    values = [0.25, 0.5, 0.5, 0.75, 1.0, 1.0, 1.5, 2.0]
    n_ties = 0
    for trial in range(1500):
        tree = Tree(rng, rng.randint(1, 9), values)
        seed_d = rng.choice([3.0, 1.0, 0.5])
        want = sequential(tree, seed_d)
        for budget in (0, 1, 5, 40):
            for win in (1, 4, 16):
                got = lanes_then_pool(tree, seed_d, rng, budget, win)
                assert got == want, (trial, budget, win, got, want)
        leaves = []

        def collect(node):
            if node["leaf"]:
                leaves.append(node["d"])
            else:
                collect(node["a"])
                collect(node["c"])

        collect(tree.root)
        n_ties += sum(1 for d in leaves if d == want[0]) > 1
    assert n_ties > 500  # the trees are full of ties at the minimum: the marker decides


def test_plain_minimum_would_not_do():
    """Without the marker (any pair that attains the minimum) the schedules DO change the reported pair: the test above is not
    vacuous."""
    rng = random.Random(5)
    values = [0.5, 0.5, 1.0]
    differ = 0
    for trial in range(200):
        tree = Tree(rng, 6, values)
        want = sequential(tree, 3.0)
        leaves = []

        def collect(node):
            if node["leaf"]:
                leaves.append((node["d"], node["id"]))
            else:
                collect(node["a"])
                collect(node["c"])

        collect(tree.root)
        others = [i for d, i in leaves if d == want[0] and i != want[1]]
        differ += bool(others)
    assert differ > 100
