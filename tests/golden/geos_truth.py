"""Hand-derived truth tables for the GEOS predicates the reference calls (DATA, not code
from the reference).  shapely/GEOS cannot be installed in the build container, so these
rows were worked out on paper from the OGC Simple Features / DE-9IM definitions:

* ``touches(A, B)``  <=>  A and B share a point and Int(A) does not meet Int(B);
  Int(LineString) = the line minus its two end points.
* ``intersects(A, B)`` <=> A and B share a point (closed segments).
* ``Polygon.buffer(r).contains(Point)`` for lattice polygons, r = 1/2 or 1/8.
"""

# (contour vertices, index of p, index of q, expected chord.touches(contour), why)
TOUCHES = [
    # U-shaped contour, chord closes the U without meeting anything else
    ([(0, 0), (0, 3), (1, 4), (3, 4), (4, 3), (4, 0)], 1, 4, True, "clean chord across the mouth of a U"),
    # chord between neighbours on the contour coincides with a contour segment
    ([(0, 0), (0, 3), (1, 4), (3, 4), (4, 3), (4, 0)], 1, 2, False, "chord lies on a contour segment"),
    # chord properly crosses a later part of the contour
    ([(0, 0), (4, 0), (4, 4), (2, 4), (2, -2), (6, -2)], 0, 2, False, "contour segment (2,4)-(2,-2) crosses the chord"),
    # a contour vertex sits exactly on the open chord
    ([(0, 0), (0, 2), (2, 2), (2, 4), (4, 4), (4, 0), (6, 0)], 1, 4, False, "vertex (2,2)...(2,4): segment passes through chord interior at (2,3)"),
    ([(0, 0), (1, 3), (2, 2), (3, 3), (4, 0)], 0, 4, True, "nothing between the end points reaches the chord y=0"),
    ([(0, 0), (1, 3), (2, 0), (3, 3), (4, 0)], 0, 4, False, "interior vertex (2,0) lies on the open chord"),
    # the only vertex on the open chord is an END vertex of the contour -> boundary, does not count
    ([(2, 0), (2, 3), (0, 3), (0, 0), (1, -1), (3, -1), (4, 0), (4, 3)], 3, 6, True,
     "contour starts at (2,0), which is on the open chord (0,0)-(4,0); its first segment leaves upward"),
    # same, but the first contour segment runs ALONG the chord -> interiors overlap
    ([(2, 0), (3, 0), (3, 3), (0, 3), (0, 0), (1, -1), (5, -1), (6, 0)], 4, 7, False,
     "first contour segment (2,0)-(3,0) is collinear with and inside the chord"),
    # chord end point q lies in the interior of another contour segment: q is chord BOUNDARY -> still touches
    ([(0, 0), (0, 2), (3, 2), (3, 0), (1, 0), (1, 1), (2, 1)], 1, 2, False, "chord (0,2)-(3,2) is a contour segment"),
]

# (p1, q1, p2, q2, expected intersects)
INTERSECTS = [
    ((0, 0), (4, 4), (0, 4), (4, 0), True),      # proper crossing
    ((0, 0), (2, 2), (2, 2), (5, 0), True),      # shared end point
    ((0, 0), (4, 0), (2, 0), (2, 3), True),      # T junction
    ((0, 0), (4, 0), (5, 0), (7, 0), False),     # collinear, disjoint
    ((0, 0), (4, 0), (3, 0), (7, 0), True),      # collinear, overlapping
    ((0, 0), (4, 0), (0, 1), (4, 1), False),     # parallel
    ((0, 0), (1, 3), (2, 0), (3, 3), False),     # skew, disjoint
    ((0, 0), (4, 4), (0, 0), (4, 4), True),      # a chord with itself
]

# (polygon vertices, r, expected lattice cells as sorted (x, y) list)
CELLS = [
    # unit-lattice square 0..2: all 9 points; nothing outside is closer than 1
    ([(0, 0), (2, 0), (2, 2), (0, 2)], 0.5,
     [(x, y) for y in range(3) for x in range(3)]),
    # right triangle (0,0),(4,0),(0,4): points with x + y <= 4, x,y >= 0; nearest outside points are at
    # distance 1/sqrt(2) > 1/2 from the hypotenuse
    ([(0, 0), (4, 0), (0, 4)], 0.5,
     [(x, y) for y in range(5) for x in range(5) if x + y <= 4]),
    # thin triangle (0,0),(2,1),(0,1): the edge (0,0)-(2,1) passes (1,0) at distance 1/sqrt(5) < 1/2 -> included
    # at r = 1/2, excluded at r = 1/8
    ([(0, 0), (2, 1), (0, 1)], 0.5, [(0, 0), (1, 0), (0, 1), (1, 1), (2, 1)]),
    ([(0, 0), (2, 1), (0, 1)], 0.125, [(0, 0), (0, 1), (1, 1), (2, 1)]),
    # degenerate box of zero height (overturning box with min_lat == max_lat): the points on the segment
    ([(3, 1), (3, 1), (0, 1), (0, 1)], 0.5, [(0, 1), (1, 1), (2, 1), (3, 1)]),
    # long edge of slope 1/9 at r = 1/8: (1,0) is 1/sqrt(82)=0.110 below it -> in; (2,0) is 0.221 away -> out
    ([(0, 0), (9, 1), (9, 3), (0, 3)], 0.125,
     [(0, 0), (1, 0)] + [(x, 1) for x in range(10)] + [(x, 2) for x in range(10)] + [(x, 3) for x in range(10)]),
]

# to_xarray (events.py:69-79): polygon.buffer(r).contains(Point(lon_i, lat_j)) on the grid lon = 0..7, lat = 0..5
# (unit spacing, r = 1/2).  (rings, expected flagged (lon, lat) points, why)
FLAGS = [
    ([[(1, 1), (4, 1), (4, 3), (1, 3)]],
     [(x, y) for y in (1, 2, 3) for x in (1, 2, 3, 4)],
     "rectangle: its lattice points; every other grid point is >= 1 away"),
    ([[(0, 0), (4, 0), (0, 4)]],
     [(x, y) for y in range(5) for x in range(5) if x + y <= 4],
     "triangle: points on or under the hypotenuse; the nearest others are 1/sqrt(2) > 1/2 from it"),
    ([[(0, 0), (2, 1), (0, 1)]],
     [(0, 0), (1, 0), (0, 1), (1, 1), (2, 1)],
     "thin triangle: (1,0) is 1/sqrt(5) < 1/2 from the edge (0,0)-(2,1)"),
    ([[(1, 1), (2, 1), (2, 2), (1, 2)], [(5, 3), (7, 3), (7, 5), (5, 5)]],
     [(1, 1), (2, 1), (1, 2), (2, 2)] + [(x, y) for y in (3, 4, 5) for x in (5, 6, 7)],
     "MultiPolygon of two parts (a polygon split at the last meridian): union of the parts"),
    ([[(0, 0), (6, 0), (6, 5), (0, 5)], [(2, 1), (5, 1), (5, 4), (2, 4)]],
     [(x, y) for y in range(6) for x in range(7) if not (2 < x < 5 and 1 < y < 4)],
     "polygon with a hole: the two points strictly inside the hole, (3,2),(4,2),(3,3),(4,3), are 1 from its rim -> out"),
]
