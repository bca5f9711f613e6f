"""Independent second restatement of the reference algorithm in numpy scalar arithmetic
(np.float32 scalars: every +, -, *, / rounds once to binary32, nothing fuses).  Written
from the reference sources separately from oracle/vq_oracle.c and used only to
cross-check the C oracle on small inputs (tests/test_oracle_numpy.py).  Pure-Python loops:
keep inputs tiny.  Citations are into /root/reference.
"""
import numpy as np

F = np.float32
ZERO = F(0.0)


def distance2(a, b):  # src/core/vector.rs:135-143
    acc = F(0.0)
    for x, y in zip(a, b):
        diff = F(x) - F(y)
        acc = F(acc + F(diff * diff))
    return acc


def sq_euclid(a, b):  # src/core/distance.rs:76-82
    acc = F(-0.0)
    for x, y in zip(a, b):
        diff = F(x) - F(y)
        acc = F(acc + F(diff * diff))
    return acc


def manhattan(a, b):  # src/core/distance.rs:94
    acc = F(-0.0)
    for x, y in zip(a, b):
        acc = F(acc + np.abs(F(F(x) - F(y))))
    return acc


def cosine(a, b):  # src/core/distance.rs:107-119
    dot, na, nb = F(-0.0), F(-0.0), F(-0.0)
    for x, y in zip(a, b):
        dot = F(dot + F(F(x) * F(y)))
    for x in a:
        na = F(na + F(F(x) * F(x)))
    for y in b:
        nb = F(nb + F(F(y) * F(y)))
    na, nb = np.sqrt(na), np.sqrt(nb)
    eps = F(1e-10)
    if na < eps or nb < eps:
        return F(1.0)
    with np.errstate(all="ignore"):
        v = F(F(1.0) - F(dot / F(na * nb)))
    if v < F(0.0):
        return F(0.0)
    if v > F(1.0):
        return F(1.0)
    return v


def distance(metric, a, b):  # src/core/distance.rs:48-64
    if metric == 0:
        return sq_euclid(a, b)
    if metric == 1:
        return np.sqrt(sq_euclid(a, b))
    if metric == 2:
        return manhattan(a, b)
    return cosine(a, b)


def argmin_first(dists):  # strict '<', first minimum wins (vector.rs:352-363, pq.rs:183-191)
    best, bi = dists[0], 0
    for j in range(1, len(dists)):
        if dists[j] < best:
            best, bi = dists[j], j
    return bi


def lloyd(data, k, max_iters, init_rows, reseed_rows=()):  # src/core/vector.rs:390-461
    data = np.asarray(data, F)
    n, sd = data.shape
    cent = data[list(init_rows)].copy()
    reseeds = list(reseed_rows)
    iters = 0
    for _ in range(max_iters):
        iters += 1
        assign = [argmin_first([distance2(v, c) for c in cent]) for v in data]
        members = [[] for _ in range(k)]
        for i, a in enumerate(assign):
            members[a].append(i)
        changed = False
        for j in range(k):
            if members[j]:
                s = np.zeros(sd, F)
                for i in members[j]:
                    for t in range(sd):
                        s[t] = F(s[t] + data[i, t])
                new = np.array([F(s[t] / F(len(members[j]))) for t in range(sd)], F)
                if not all(np.abs(F(new[t] - cent[j, t])) < F(1e-6) for t in range(sd)):
                    changed = True
                cent[j] = new
            else:
                cent[j] = data[reseeds.pop(0)]
        if not changed:
            break
    return cent, iters


def pq_encode(metric, rows, codebooks):  # src/pq.rs:167-199
    rows = np.asarray(rows, F)
    m, k, sd = codebooks.shape
    codes = np.zeros((rows.shape[0], m), np.uint32)
    out = np.zeros(rows.shape, np.float16)
    for i, v in enumerate(rows):
        for s in range(m):
            sub = v[s * sd:(s + 1) * sd]
            b = argmin_first([distance(metric, sub, c) for c in codebooks[s]])
            codes[i, s] = b
            out[i, s * sd:(s + 1) * sd] = codebooks[s, b].astype(np.float16)
    return codes, out


def _total_key(x):
    b = np.array([x], F).view(np.int32)[0]
    return int(b ^ ((b >> 31) & 0x7FFFFFFF))


def tsvq_build(rows, depth):  # src/tsvq.rs:31-115 ; returns nested dict
    rows = np.asarray(rows, F)
    n, d = rows.shape
    mu = np.zeros(d, F)
    for v in rows:
        for t in range(d):
            mu[t] = F(mu[t] + v[t])
    mu = np.array([F(mu[t] / F(n)) for t in range(d)], F)
    node = dict(centroid=mu, left=None, right=None, n=n)
    if depth == 0 or n <= 1:
        return node
    var = []
    for t in range(d):
        acc = F(-0.0)
        for v in rows:
            diff = F(v[t] - mu[t])
            acc = F(acc + F(diff * diff))
        var.append(acc)
    split, best = 0, None
    for t, v in enumerate(var):
        if np.isnan(v):
            continue
        if best is None or not (v < best):
            best, split = v, t
    vals = sorted([x for x in rows[:, split] if not np.isnan(x)], key=_total_key)
    if len(vals) % 2 == 0:
        med = F(F(vals[len(vals) // 2 - 1] + vals[len(vals) // 2]) / F(2.0))
    else:
        med = vals[len(vals) // 2]
    li = [i for i in range(n) if rows[i, split] <= med]
    ri = [i for i in range(n) if not (rows[i, split] <= med)]
    if li and len(li) < n:
        node["left"] = tsvq_build(rows[li], depth - 1)
    if ri and len(ri) < n:
        node["right"] = tsvq_build(rows[ri], depth - 1)
    return node


def tsvq_flatten(node):
    """pre-order arrays like the C oracle"""
    cents, left, right = [], [], []

    def rec(nd):
        i = len(cents)
        cents.append(nd["centroid"])
        left.append(-1)
        right.append(-1)
        if nd["left"] is not None:
            left[i] = rec(nd["left"])
        if nd["right"] is not None:
            right[i] = rec(nd["right"])
        return i

    rec(node)
    return np.array(cents, F), np.array(left, np.int32), np.array(right, np.int32)


def tsvq_find_leaf(metric, node, v):  # src/tsvq.rs:117-132
    while True:
        l, r = node["left"], node["right"]
        if l is not None and r is not None:
            node = l if distance(metric, v, l["centroid"]) <= distance(metric, v, r["centroid"]) else r
        elif l is not None:
            node = l
        elif r is not None:
            node = r
        else:
            return node
