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
* scikit-image joins segments through a coordinate dictionary, sequentially in raster
  order (``_assemble_contours``).  While no cell equals the level every crossing
  coordinate belongs to exactly one grid edge and that is the same as linking through
  integer edge ids (``assemble``; ordering rules: contours sorted by their raster-first
  segment; an open contour starts at its predecessor-less segment; a closed contour
  starts and ends at the ``to`` point of its raster-last segment);
* a cell EQUAL to the level (it counts as "not above") puts every crossing of its
  incident edges ON the grid vertex: up to four arrivals and four departures then share
  one dictionary key, zero-length segments are skipped, a later contour end overwrites an
  earlier one under the same key, and in rare constellations the third-party code raises
  KeyError.  ``join_by_coordinates`` restates that sequential walk literally and is what
  ``find_contours`` uses whenever a cell equals the level (pinned by
  tests/golden/contours_levelhit.npz, KeyError cases included); ``vertex_links`` is the
  order-independent formulation the HIP kernel uses (a four-square state machine per
  vertex), valid whenever no contour end gets overwritten.
"""
from collections import deque

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


def join_by_coordinates(p_from, p_to):
    """skimage ``_assemble_contours`` (site-packages/skimage/measure/_find_contours.py, 0.18.3): the segments are taken
    in raster order and joined through two dictionaries keyed by the EXACT float coordinates of the pieces' first and
    last points; a segment whose two ends coincide is skipped; when two pieces are joined the one created first
    survives.  Raises KeyError exactly where the third-party routine does (a piece end whose dictionary entry was
    overwritten by another piece ending on the same grid vertex).  Returns the pieces in creation order."""
    pieces, first_at, last_at = {}, {}, {}
    created = 0
    for f, t in zip(map(tuple, np.asarray(p_from).reshape(-1, 2).tolist()),
                    map(tuple, np.asarray(p_to).reshape(-1, 2).tolist())):
        if f == t:
            continue
        after, n_after = first_at.pop(t, (None, None))       # the piece that starts where this segment ends
        before, n_before = last_at.pop(f, (None, None))       # the piece that ends where this segment starts
        if before is not None and after is not None:
            if before is after:
                before.append(t)                               # the piece closes
            elif n_after > n_before:
                before.extend(after)
                last_at.pop(after[-1])
                pieces.pop(n_after, None)
                first_at[before[0]] = (before, n_before)
                last_at[before[-1]] = (before, n_before)
            else:
                after.extendleft(reversed(before))
                first_at.pop(before[0])
                pieces.pop(n_before, None)
                first_at[after[0]] = (after, n_after)
                last_at[after[-1]] = (after, n_after)
        elif before is None and after is None:
            d = deque((f, t))
            pieces[created] = d
            first_at[f] = (d, created)
            last_at[t] = (d, created)
            created += 1
        elif before is None:
            after.appendleft(f)
            first_at[f] = (after, n_after)
        else:
            before.append(t)
            last_at[t] = (before, n_before)
    return [np.array(c) for _, c in sorted(pieces.items())]


_CORNER_OF = {TOP: ((0, 0), (0, 1)), BOTTOM: ((1, 0), (1, 1)), LEFT: ((0, 0), (1, 0)), RIGHT: ((0, 1), (1, 1))}


def _end_vertex(a, level, r, c, side):
    """grid vertex (row, col) the crossing on `side` of square (r, c) sits on (one end of the edge equals the level), or None"""
    for dr, dc in _CORNER_OF[int(side)]:
        if a[r + dr, c + dc] == level:
            return (r + dr, c + dc)
    return None


def vertex_links(image, level, seg):
    """succ / pred / dead (zero-length) per segment with the crossings that sit ON a grid vertex resolved by a small
    state machine per vertex: the segment ends on the vertex are visited in raster order (at most the two segments of
    each of the four squares around it); an arrival takes the pending departure or becomes the pending arrival, a
    departure takes the pending arrival or becomes the pending departure.  Equal to ``join_by_coordinates`` unless a
    pending end gets overwritten (``overwritten`` > 0: the sequential walk may then join differently or raise)."""
    a = np.asarray(image).astype(np.float64)
    succ, pred = link_segments(seg)
    n = succ.size
    fv = [_end_vertex(a, level, seg["r"][i], seg["c"][i], seg["side_from"][i]) for i in range(n)]
    tv = [_end_vertex(a, level, seg["r"][i], seg["c"][i], seg["side_to"][i]) for i in range(n)]
    dead = np.array([fv[i] is not None and fv[i] == tv[i] for i in range(n)], dtype=bool)
    events = {}
    for i in range(n):
        if tv[i] is not None:
            succ[i] = -1
        if fv[i] is not None:
            pred[i] = -1
        if dead[i]:
            continue
        if tv[i] is not None:
            events.setdefault(tv[i], []).append((i, 0))
        if fv[i] is not None:
            events.setdefault(fv[i], []).append((i, 1))
    overwritten = 0
    for ev in events.values():
        ev.sort()
        arrival = departure = None
        for i, leaving in ev:
            if not leaving:
                if departure is not None:
                    succ[i], pred[departure], departure = departure, i, None
                else:
                    overwritten += arrival is not None
                    arrival = i
            else:
                if arrival is not None:
                    succ[arrival], pred[i], arrival = i, arrival, None
                else:
                    overwritten += departure is not None
                    departure = i
    return succ, pred, dead, overwritten


def assemble_links(succ, pred, dead):
    """chains of segment indices in scikit-image order from successor links (dead segments left out)"""
    n = succ.size
    visited = dead.copy()
    succ_l = succ.tolist()
    chains = []
    for s in np.nonzero((pred < 0) & ~dead)[0].tolist():
        chain, i = [], s
        while i >= 0 and not visited[i]:
            visited[i] = True
            chain.append(i)
            i = succ_l[i]
        chains.append(chain)
    for s in range(n):
        if visited[s]:
            continue
        loop, i = [], s
        while not visited[i]:
            visited[i] = True
            loop.append(i)
            i = succ_l[i]
        k = loop.index(max(loop))
        chains.append(loop[k + 1:] + loop[:k + 1])
    chains.sort(key=min)
    return chains


def find_contours(image, level):
    """Restatement of skimage.measure.find_contours(image, level): list of (n,2) float64 (row, col)."""
    seg = contour_segments(image, level)
    if seg["level_hits"]:
        return join_by_coordinates(seg["p_from"], seg["p_to"])
    out = []
    for chain, _closed in assemble(seg):
        idx = np.asarray(chain)
        pts = np.concatenate([seg["p_from"][idx[:1]], seg["p_to"][idx]], axis=0)
        out.append(pts)
    return out


def find_contours_vertex_machine(image, level):
    """the same contours from ``vertex_links`` (what the HIP link kernel computes); also returns the number of
    overwritten pending ends (0: guaranteed equal to ``find_contours``)"""
    seg = contour_segments(image, level)
    succ, pred, dead, overwritten = vertex_links(image, level, seg)
    out = []
    for chain in assemble_links(succ, pred, dead):
        idx = np.asarray(chain)
        out.append(np.concatenate([seg["p_from"][idx[:1]], seg["p_to"][idx]], axis=0))
    return out, overwritten
