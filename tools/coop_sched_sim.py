"""Model of the cooperative (one-launch) Cholesky + triangular-inverse schedule of csrc/coop.hip.

Workers (workgroups of one matrix's cluster) claim tiles from ONE ordered list (a linear extension of the tile DAG, so
any number of resident workers makes progress); a claimed tile walks its k-blocks in order and blocks on the flags of
tiles that are not finished yet.  This script evaluates candidate orders for (nblk, G) with rough per-phase times and
prints the makespan, so the order builder in csrc/coop.hip (same rules, in C++) can be chosen on paper first.
(Round 4's task set: since round 5 the library fuses tiles (j, j-1) and (j, j) into one task, gives the diagonal tile's
early sums a task of their own and announces a diagonal block's inverse ahead of its other stores -- build_order in
coop.hip models that; the order it produces is checked by tests/test_abi.py, not by this script.)

    python tools/coop_sched_sim.py 16 8
"""
import sys, heapq, itertools

T_OP, T_P, T_D, T_FLAG, T_CLAIM, T_PARK = 14.5, 17.0, 42.0, 2.0, 1.0, 3.0


def tasks_chol(n):
    return [("C", i, j) for j in range(n) for i in range(j, n)]


def deps_steps(task, n):
    """-> list of (needed finished tasks, duration) steps in execution order."""
    kind, i, j = task
    st = []
    if kind == "C":
        for k in range(j):
            need = [("C", i, k)] + ([("C", j, k)] if i != j else [])
            st.append((need, T_OP))
        if i == j:
            st.append(([], T_D))
        else:
            st.append(([("C", j, j)], T_P))
    else:  # "X", i > j : X[i][j] = -Dinv[i] * sum_{k=j}^{i-1} L[i][k] X[k][j]
        for k in range(j, i):
            need = [("C", i, k)] + ([("X", k, j)] if k > j else [("C", j, j)])
            st.append((need, T_OP))
        st.append(([("C", i, i)], T_P))
    return st


def simulate(order, n, G, verbose=False):
    fin = {}
    free = [(0.0, w) for w in range(G)]
    heapq.heapify(free)
    busy = 0.0
    for task in order:
        t, w = heapq.heappop(free)
        t += T_CLAIM
        for need, dur in deps_steps(task, n):
            for d in need:
                assert d in fin, (task, d)
                t = max(t, fin[d] + T_FLAG)
            t += dur
            busy += dur
        fin[task] = t
        heapq.heappush(free, (t, w))
    mk = max(fin.values())
    return mk, busy / (mk * G), fin


def order_colmajor(n, inv):
    o = tasks_chol(n)
    if inv:
        o += [("X", i, j) for j in range(n) for i in range(j + 1, n)]
    return o


def order_greedy(n, G, inv, slack=6.0):
    """List built by simulating greedy claims: a free worker takes, among the tiles whose dependencies are all claimed,
    the one with the largest bottom level (longest remaining path) among those that would block for at most `slack` us
    in total; if none, the one that blocks least."""
    allt = tasks_chol(n) + ([("X", i, j) for j in range(n) for i in range(j + 1, n)] if inv else [])
    steps = {t: deps_steps(t, n) for t in allt}
    preds = {t: set(d for need, _ in steps[t] for d in need) for t in allt}
    succs = {t: [] for t in allt}
    for t in allt:
        for d in preds[t]:
            succs[d].append(t)
    # bottom level with only the LAST step of each tile on the chain (earlier steps can be pre-accumulated)
    last = {t: steps[t][-1][1] + (steps[t][-2][1] if len(steps[t]) > 1 else 0) for t in allt}
    bl = {}
    def blevel(t):
        if t in bl: return bl[t]
        bl[t] = last[t] + max((blevel(s) for s in succs[t]), default=0.0)
        return bl[t]
    sys.setrecursionlimit(100000)
    for t in allt: blevel(t)
    fin, order, claimed = {}, [], set()
    free = [(0.0, w) for w in range(G)]
    heapq.heapify(free)
    remaining = set(allt)
    while remaining:
        t0, w = heapq.heappop(free)
        cands = [t for t in remaining if preds[t] <= claimed]
        best, bestkey = None, None
        for c in cands:
            t = t0 + T_CLAIM
            work = 0.0
            for need, dur in steps[c]:
                for d in need:
                    t = max(t, fin[d] + T_FLAG)
                t += dur
                work += dur
            blocked = t - t0 - T_CLAIM - work
            key = (0, -bl[c]) if blocked <= slack else (1, blocked)
            if bestkey is None or key < bestkey:
                best, bestkey, bestfin = c, key, t
        order.append(best)
        claimed.add(best)
        remaining.discard(best)
        fin[best] = bestfin
        heapq.heappush(free, (bestfin, w))
    return order


if __name__ == "__main__":
    n, G = int(sys.argv[1]), int(sys.argv[2])
    for inv in (False, True):
        work = sum(d for t in order_colmajor(n, inv) for _, d in deps_steps(t, n))
        print(f"nblk={n} G={G} inverse={inv}: work {work:.0f} us  / G = {work / G:.0f} us")
        mk, u, _ = simulate(order_colmajor(n, inv), n, G)
        print(f"  column-major        makespan {mk:8.1f} us  util {u:.2f}")
        for slack in (0.0, 3.0, 6.0, 15.0, 30.0, 60.0):
            o = order_greedy(n, G, inv, slack)
            mk, u, _ = simulate(o, n, G)
            print(f"  greedy slack {slack:5.1f}  makespan {mk:8.1f} us  util {u:.2f}")
