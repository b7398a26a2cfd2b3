"""Marching squares + ordered contour assembly -- CPU restatement (TEST INFRASTRUCTURE).

Restates ``skimage.measure.find_contours(image, level)`` with the defaults the
reference uses (``fully_connected='low'``, ``positive_orientation='low'``, no
mask) -- call site ``wavebreaking/indices/contour_index.py:103``.  scikit-image
is a third-party dependency that is not under /root/reference and is unpinned
there (``setup.py:13-24``); the pin is scikit-image 0.18.3 as installed under
/opt/conda in the build container, against which the golden vectors
``tests/golden/contours_*.npz`` were generated (``tests/golden/make_golden.py``).

Algorithm:

* every 2x2 square, raster order (r0 outer, c0 inner); skipped if a corner is NaN;
  ``case = (ul>lvl) + 2(ur>lvl) + 4(ll>lvl) + 8(lr>lvl)`` (strict ``>``);
* crossing coordinate on an edge ``from + (lvl - v_from)/(v_to - v_from)`` in
  float64, always evaluated left->right on horizontal and top->bottom on vertical
  edges, so both squares sharing an edge get bit-identical coordinates;
* scikit-image joins segments through a coordinate dictionary; away from the
  degenerate ``value == level`` case that is the same as linking through integer
  edge ids, which is what is done here (and on the GPU);
* ordering rules: contours sorted by their raster-first segment; an open contour
  starts at its predecessor-less segment; a closed contour starts (and ends) at
  the ``to`` point of its raster-last segment.
"""
import numpy as np

TOP, RIGHT, BOTTOM, LEFT = 0, 1, 2, 3
# case -> tuple of (from_side, to_side); probed from skimage 0.18.3 on the 16 binary squares
SEGMENT_TABLE = {
    0: (), 15: (),
    1: ((TOP, LEFT),), 2: ((RIGHT, TOP),), 3: ((RIGHT, LEFT),), 4: ((LEFT, BOTTOM),),
    5: ((TOP, BOTTOM),), 6: ((RIGHT, TOP), (LEFT, BOTTOM)), 7: ((RIGHT, BOTTOM),),
    8: ((BOTTOM, RIGHT),), 9: ((TOP, LEFT), (BOTTOM, RIGHT)), 10: ((BOTTOM, TOP),),
    11: ((BOTTOM, LEFT),), 12: ((LEFT, RIGHT),), 13: ((TOP, RIGHT),), 14: ((LEFT, TOP),),
}
_NSEG = np.array([len(SEGMENT_TABLE[c]) for c in range(16)])
_FROM = np.full((16, 2), -1)
_TO = np.full((16, 2), -1)
for _c, _segs in SEGMENT_TABLE.items():
    for _s, (_f, _t) in enumerate(_segs):
        _FROM[_c, _s] = _f
        _TO[_c, _s] = _t


def _fraction(a, b, level):
    with np.errstate(divide="ignore", invalid="ignore"):
        f = (level - a) / (b - a)
    return np.where(a == b, 0.0, f)


def contour_segments(image, level):
    """All segments in raster order.

    Returns dict with per-segment arrays: r, c (square), side_from, side_to,
    edge_from, edge_to (integer edge ids), p_from, p_to ((n,2) float64 (row, col))."""
    a = np.asarray(image).astype(np.float64)
    level = float(level)
    nr, nc = a.shape
    ul, ur, ll, lr = a[:-1, :-1], a[:-1, 1:], a[1:, :-1], a[1:, 1:]
    valid = ~(np.isnan(ul) | np.isnan(ur) | np.isnan(ll) | np.isnan(lr))
    with np.errstate(invalid="ignore"):
        case = ((ul > level).astype(int) + 2 * (ur > level) + 4 * (ll > level) + 8 * (lr > level))
    case = np.where(valid, case, 0)
    nseg = _NSEG[case]
    rr, cc = np.nonzero(nseg)            # raster order
    cs = case[rr, cc]
    two = nseg[rr, cc] == 2
    # expand saddles into two consecutive entries
    rep = np.where(two, 2, 1)
    r = np.repeat(rr, rep)
    c = np.repeat(cc, rep)
    k = np.repeat(cs, rep)
    sub = np.zeros(r.size, dtype=int)
    first = np.cumsum(rep) - rep
    sub[first[two] + 1] = 1
    sf = _FROM[k, sub]
    st = _TO[k, sub]

    def point(side):
        vul, vur, vll, vlr = a[r, c], a[r, c + 1], a[r + 1, c], a[r + 1, c + 1]
        pr = np.empty(r.size)
        pc = np.empty(r.size)
        m = side == TOP
        pr[m] = r[m]
        pc[m] = c[m] + _fraction(vul[m], vur[m], level)
        m = side == BOTTOM
        pr[m] = r[m] + 1
        pc[m] = c[m] + _fraction(vll[m], vlr[m], level)
        m = side == LEFT
        pr[m] = r[m] + _fraction(vul[m], vll[m], level)
        pc[m] = c[m]
        m = side == RIGHT
        pr[m] = r[m] + _fraction(vur[m], vlr[m], level)
        pc[m] = c[m] + 1
        return np.stack([pr, pc], axis=1)

    def edge(side):
        # H(r,c) = 2*(r*nc+c) ; V(r,c) = 2*(r*nc+c)+1
        e = np.empty(r.size, dtype=np.int64)
        m = side == TOP
        e[m] = 2 * (r[m] * nc + c[m])
        m = side == BOTTOM
        e[m] = 2 * ((r[m] + 1) * nc + c[m])
        m = side == LEFT
        e[m] = 2 * (r[m] * nc + c[m]) + 1
        m = side == RIGHT
        e[m] = 2 * (r[m] * nc + c[m] + 1) + 1
        return e

    return dict(r=r, c=c, sub=sub, case=k, side_from=sf, side_to=st,
                edge_from=edge(sf), edge_to=edge(st), p_from=point(sf), p_to=point(st),
                level_hits=int(np.count_nonzero(a == level)))


def link_segments(seg):
    """succ / pred segment indices (-1 = none) through shared integer edge ids."""
    ef, et = seg["edge_from"], seg["edge_to"]
    n = ef.size
    succ = np.full(n, -1, dtype=np.int64)
    pred = np.full(n, -1, dtype=np.int64)
    if n == 0:
        return succ, pred
    order = np.argsort(ef, kind="stable")
    pos = np.searchsorted(ef[order], et)
    pos = np.minimum(pos, n - 1)
    hit = ef[order][pos] == et
    succ[hit] = order[pos[hit]]
    pred[succ[hit]] = np.nonzero(hit)[0]
    return succ, pred


def assemble(seg):
    """Ordered contours as lists of segment indices + closed flag, in scikit-image order."""
    succ, pred = link_segments(seg)
    n = succ.size
    visited = np.zeros(n, dtype=bool)
    chains = []
    succ_l = succ.tolist()
    # open chains: start at predecessor-less segments
    for s in np.nonzero(pred < 0)[0].tolist():
        chain = []
        i = s
        while i >= 0:
            visited[i] = True
            chain.append(i)
            i = succ_l[i]
        chains.append((chain, False))
    # closed loops: whatever is left; start right after the raster-last member
    for s in range(n):
        if visited[s]:
            continue
        loop = []
        i = s
        while not visited[i]:
            visited[i] = True
            loop.append(i)
            i = succ_l[i]
        last = max(loop)
        k = loop.index(last)
        loop = loop[k + 1:] + loop[:k + 1]
        chains.append((loop, True))
    chains.sort(key=lambda ch: min(ch[0]))
    return chains


def find_contours(image, level):
    """Restatement of skimage.measure.find_contours(image, level): list of (n,2) float64 (row, col)."""
    seg = contour_segments(image, level)
    out = []
    for chain, _closed in assemble(seg):
        idx = np.asarray(chain)
        pts = np.concatenate([seg["p_from"][idx[:1]], seg["p_to"][idx]], axis=0)
        out.append(pts)
    return out
