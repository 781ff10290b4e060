import numpy as np
def reorder_global(nodes, roots, K):
    """roots first, then every mesh's top-K-level child pairs breadth-first, then the remaining pairs in the old order. Returns (nodes, roots, T)."""
    M = len(roots); n = len(nodes)
    lf = nodes["leftFirst"]; tc = nodes["triCount"]
    top = []
    for r in roots:
        frontier = [int(r)]
        for _ in range(K):
            nxt = []
            for x in frontier:
                if tc[x] == 0:
                    l = int(lf[x]); top.append(l); nxt += [l, l + 1]
            frontier = nxt
    topset = set(top)
    rootset = set(int(r) for r in roots)
    # all pair starts in old order: inner nodes' leftFirst values, sorted
    inner = tc == 0
    starts_all = np.sort(lf[inner].astype(np.int64))
    rest = [int(l) for l in starts_all if int(l) not in topset]
    new_of = np.zeros(n, np.int64)
    for m, r in enumerate(roots): new_of[int(r)] = m
    pos = M
    order = top + rest
    for l in order:
        new_of[l] = pos; new_of[l + 1] = pos + 1; pos += 2
    assert pos == n, (pos, n)
    out = np.zeros_like(nodes)
    out[new_of] = nodes
    inn = out["triCount"] == 0
    out["leftFirst"][inn] = new_of[out["leftFirst"][inn]].astype(np.uint32)
    T = (M + 2 * len(top)) >> 1
    return out, np.arange(M, dtype=np.uint32), T
