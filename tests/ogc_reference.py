"""An INDEPENDENT statement of the two GEOS predicates the streamer index calls (streamer_index.py:185-251), written from the
OGC Simple Features / DE-9IM text in exact rational arithmetic -- it shares no code and no formulation with oracle/geom.py
or csrc/wb_core.h (orientation determinants there; explicit intersection POINT SETS here), so that the randomized
differential tests (tests/test_geos_hardening.py) compare two derivations, not one derivation with itself.
TEST INFRASTRUCTURE.  GEOS itself is not installable in this image: the rows stay unpinned.

Definitions used (OGC 06-103r4, 6.1.6.1 / 6.1.15.3):
  * the boundary of a non-closed LineString is the SET of its two end points, its interior every other point of the curve
    (a POINT set: a location that is an end point is boundary wherever else the curve passes through it);
    a closed LineString (first == last) has an empty boundary;
  * a.touches(b)  <=>  a and b meet, and interior(a) and interior(b) have no point in common;
  * a.intersects(b)  <=>  a and b have a point in common.
"""
from fractions import Fraction


def _seg_seg_points(p, q, a, b):
    """the intersection of the closed segments [p, q] and [a, b] as a point set: () / (X,) / (X0, X1) -- a single point or the
    two end points of a common stretch of positive length.  Exact: parametric solution in Fractions."""
    px, py, qx, qy = (Fraction(int(v)) for v in (p[0], p[1], q[0], q[1]))
    ax, ay, bx, by = (Fraction(int(v)) for v in (a[0], a[1], b[0], b[1]))
    ux, uy = qx - px, qy - py                      # direction of [p, q]
    vx, vy = bx - ax, by - ay                      # direction of [a, b]
    den = ux * vy - uy * vx
    wx, wy = ax - px, ay - py
    if den != 0:
        # p + s u = a + t v
        s = (wx * vy - wy * vx) / den
        t = (wx * uy - wy * ux) / den
        if 0 <= s <= 1 and 0 <= t <= 1:
            return ((px + s * ux, py + s * uy),)
        return ()
    # parallel (or a degenerate segment)
    if ux == 0 and uy == 0:                        # [p, q] is a point
        pts = _point_on_segment((px, py), (ax, ay), (bx, by))
        return ((px, py),) if pts else ()
    if wx * uy - wy * ux != 0:                     # different lines
        return ()
    # same line: parameters of a and b along p + s u
    uu = ux * ux + uy * uy
    sa = (wx * ux + wy * uy) / uu
    sb = ((bx - px) * ux + (by - py) * uy) / uu
    lo, hi = max(Fraction(0), min(sa, sb)), min(Fraction(1), max(sa, sb))
    if lo > hi:
        return ()
    if lo == hi:
        return ((px + lo * ux, py + lo * uy),)
    return ((px + lo * ux, py + lo * uy), (px + hi * ux, py + hi * uy))


def _point_on_segment(x, a, b):
    (xx, xy), (ax, ay), (bx, by) = x, a, b
    if (bx - ax) * (xy - ay) - (by - ay) * (xx - ax) != 0:
        return False
    return min(ax, bx) <= xx <= max(ax, bx) and min(ay, by) <= xy <= max(ay, by)


def segments_intersect(p1, q1, p2, q2):
    """LineString([p1, q1]).intersects(LineString([p2, q2])) (streamer_index.py:233-236): the closed segments share a point"""
    return len(_seg_seg_points(p1, q1, p2, q2)) > 0


def chord_touches_linestring(p, q, contour):
    """LineString([p, q]).touches(LineString(contour)) (streamer_index.py:192-198) for a chord whose end points are vertices of
    the contour (so the two always meet): True iff no point lies in BOTH interiors.  interior(chord) = the open segment (p, q);
    interior(contour) = the curve minus the point set {first vertex, last vertex} (empty boundary if the contour is closed)."""
    pts = [(Fraction(int(v[0])), Fraction(int(v[1]))) for v in contour]
    P, Q = (Fraction(int(p[0])), Fraction(int(p[1]))), (Fraction(int(q[0])), Fraction(int(q[1])))
    boundary = set() if pts[0] == pts[-1] else {pts[0], pts[-1]}
    if P == Q:
        return True                                # (a degenerate chord has no interior)
    for k in range(len(pts) - 1):
        common = _seg_seg_points(p, q, contour[k], contour[k + 1])
        if len(common) == 2:
            # a stretch of positive length in common: infinitely many points of the open chord, at most two of them boundary
            return False
        for X in common:
            if X == P or X == Q:                   # the chord's own boundary: not in its interior
                continue
            if X in boundary:                      # the contour's boundary (as a point): not in its interior
                continue
            return False
    return True
