# fit a cost model of the ring kernel's range length q against measured launch times (gpurun_out/ring_q_alone.log)
import re, math, itertools, sys
import numpy as np
def cdiv(a, b): return -(-a // b)
def layout(B, n_out, n_in, bias=True):
    n0, t0 = cdiv(B, 32), cdiv(n_out, 64) * cdiv(n_in, 64)
    n1, t1 = cdiv(n_out, 32), cdiv(B, 64) * cdiv(n_in, 64)
    nbc, cu = (cdiv(n_out, 32), 4 + cdiv(B, 512)) if bias else (0, 0)
    return n0, t0, n1, t1, nbc, cu
def default_q(S, slots=512, minq=8):
    G = max(1, min(S // minq, slots, 768)); q = cdiv(S, G); return q
def features(B, n_out, n_in, q):
    n0, t0, n1, t1, nbc, cu = layout(B, n_out, n_in)
    Sg = t0 * n0 + t1 * n1; S = Sg + nbc * cu
    G = cdiv(S, q)
    # per workgroup: number of segments, partial segments; per tile: pieces
    maxpart, maxseg, maxpieces, tot_part = 0, 0, 1, 0
    bounds = [0]
    for _ in range(t0): bounds.append(bounds[-1] + n0)
    for _ in range(t1): bounds.append(bounds[-1] + n1)
    bset = set(bounds)
    import bisect
    for v in range(G):
        lo, hi = v * q, min((v + 1) * q, Sg)
        if lo >= Sg: break
        i = bisect.bisect_right(bounds, lo) - 1
        nseg = npart = 0
        pos = lo
        while pos < hi:
            tend = bounds[i + 1]
            e = min(tend, hi)
            whole = (pos == bounds[i] and e == tend)
            nseg += 1; npart += (0 if whole else 1)
            pos = e; i += 1
        maxseg = max(maxseg, nseg); maxpart = max(maxpart, npart); tot_part += npart
    for n, t, base in ((n0, t0, 0), (n1, t1, t0 * n0)):
        for k in range(min(t, 64)):
            s, e = base + k * n, base + (k + 1) * n
            maxpieces = max(maxpieces, (e - 1) // q - s // q + 1)
    return dict(q=q, G=G, occ=max(1.0, G / 256.0), maxpart=maxpart, maxseg=maxseg, maxpieces=maxpieces, tot_part=tot_part / max(1, G))
data = {}
cur = None
for line in open(sys.argv[1]):
    m = re.match(r"== Q=(\d+)", line)
    if m: cur = int(m.group(1)); continue
    m = re.match(r"\s*(\d+) x\s+(\d+) x\s+(\d+): main\s+([\d.]+)", line)
    if m and cur is not None:
        B, no, ni, us = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
        n0, t0, n1, t1, nbc, cu = layout(B, no, ni); S = t0 * n0 + t1 * n1 + nbc * cu
        q = cur if cur > 0 else default_q(S)
        if cur > 0 and cdiv(S, cur) > 768: q = default_q(S)
        data[(B, no, ni, q)] = us
rows = []
for (B, no, ni, q), us in data.items():
    f = features(B, no, ni, q); rows.append((B, no, ni, q, us, f))
X = np.array([[1.0, r[5]["q"] * r[5]["occ"], r[5]["maxpart"], r[5]["maxpieces"] - 1, r[5]["tot_part"]] for r in rows]); y = np.array([r[4] for r in rows])
coef, *_ = np.linalg.lstsq(X, y, rcond=None)
print("coef [fixed, per q*occ, per maxpart, per extra piece, avg partial/WG]:", np.round(coef, 3))
pred = X @ coef
print("rms err", np.sqrt(np.mean((pred - y) ** 2)))
shapes = sorted({(r[0], r[1], r[2]) for r in rows})
for sh in shapes:
    rr = [(r[3], r[4], float(p)) for r, p in zip(rows, pred) if (r[0], r[1], r[2]) == sh]
    rr.sort()
    best_meas = min(rr, key=lambda t: t[1]); best_pred = min(rr, key=lambda t: t[2])
    print(sh, "best measured q=%d %.1f us; model picks q=%d (measured %.1f us)" % (best_meas[0], best_meas[1], best_pred[0], best_pred[1]))
