"""Minimal stand-ins for xarray.DataArray and shapely geometries.

The reference takes ``xarray.DataArray`` in and hands ``geopandas.GeoDataFrame`` out.  When those
packages are importable this package uses them (same types as the reference).  They are not part of the
build image, so everything also works with the two small classes below, which expose exactly the
attributes the hot path touches (``dims``, ``values``, ``field[dim]`` -> coordinate with ``values``,
``data``, ``attrs``, ``encoding``, ``dtype``; geometries with ``coords`` / ``exterior`` / ``geoms`` /
``area``).
"""
import numpy as np

try:  # pragma: no cover - not in the build image
    import xarray as _xr
except Exception:  # pragma: no cover
    _xr = None

try:  # pragma: no cover
    import shapely.geometry as _sg
except Exception:  # pragma: no cover
    _sg = None

try:  # pragma: no cover
    import geopandas as _gpd
except Exception:  # pragma: no cover
    _gpd = None


class Coord:
    def __init__(self, values, attrs=None, encoding=None):
        self.values = np.asarray(values)
        self.attrs = dict(attrs or {})
        self.encoding = dict(encoding or {})

    @property
    def data(self):
        return self.values

    @property
    def dtype(self):
        return self.values.dtype

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __iter__(self):
        return iter(self.values)


class GridField:
    """(time, lat, lon)-like labelled array: the part of xarray.DataArray the indices use."""

    def __init__(self, values, dims=("time", "lat", "lon"), coords=None, attrs=None, name=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        if self.values.ndim != len(self.dims):
            raise ValueError("values has %d dimensions, dims names %d" % (self.values.ndim, len(self.dims)))
        self.coords = {}
        for d, n in zip(self.dims, self.values.shape):
            c = (coords or {}).get(d, np.arange(n))
            c = c if isinstance(c, Coord) else Coord(c)
            if len(c) != n:
                raise ValueError("coordinate %r has length %d, dimension has %d" % (d, len(c), n))
            self.coords[d] = c
        self.attrs = dict(attrs or {})
        self.name = name

    def __getitem__(self, dim):
        return self.coords[dim]

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def shape(self):
        return self.values.shape

    def sortby(self, dim, ascending=True):
        ax = self.dims.index(dim)
        order = np.argsort(self.coords[dim].values, kind="stable")
        if not ascending:
            order = order[::-1]
        coords = dict(self.coords)
        c = self.coords[dim]
        coords[dim] = Coord(c.values[order], c.attrs, c.encoding)
        return GridField(np.take(self.values, order, axis=ax), self.dims, coords, self.attrs, self.name)


def data_array_types():
    """Types accepted where the reference demands xarray.DataArray (data_utils.py:28-50)."""
    return (GridField,) if _xr is None else (_xr.DataArray, GridField)


# ------------------------------------------------------------------------------------------------ geometry
class _CoordSeq:
    def __init__(self, a):
        self._a = np.asarray(a, dtype=np.float64).reshape(-1, 2)

    @property
    def xy(self):
        return self._a[:, 0].copy(), self._a[:, 1].copy()

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._a, dtype=dtype)

    def __len__(self):
        return len(self._a)

    def __iter__(self):
        return iter(map(tuple, self._a))

    def __getitem__(self, i):
        return tuple(self._a[i])


class LineString:
    geom_type = "LineString"

    def __init__(self, coords):
        self.coords = _CoordSeq(coords)

    @property
    def is_empty(self):
        return len(self.coords) == 0


class _Ring:
    """a polygon ring; the closed coordinate sequence shapely would hold is built when somebody asks for it (an index
    call makes thousands of these and most frames are only filtered or written out)"""
    __slots__ = ("_raw", "_seq")

    def __init__(self, coords):
        self._raw = coords
        self._seq = None

    @property
    def coords(self):
        if self._seq is None:
            a = np.asarray(self._raw, dtype=np.float64).reshape(-1, 2)
            if len(a) and not (a[0, 0] == a[-1, 0] and a[0, 1] == a[-1, 1]):
                a = np.concatenate([a, a[:1]], axis=0)     # shapely closes rings
            self._seq = _CoordSeq(a)
            self._raw = None
        return self._seq


def _shoelace(a):
    x, y = a[:, 0], a[:, 1]
    return 0.5 * float(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))


class Polygon:
    geom_type = "Polygon"

    def __init__(self, shell=None):
        self.exterior = _Ring(shell if shell is not None else np.zeros((0, 2)))

    @property
    def is_empty(self):
        return len(self.exterior.coords) == 0

    @property
    def area(self):
        a = np.asarray(self.exterior.coords)
        return abs(_shoelace(a)) if len(a) else 0.0

    @property
    def geoms(self):
        return [self]


class MultiPolygon:
    geom_type = "MultiPolygon"

    def __init__(self, polygons):
        self.geoms = list(polygons)

    @property
    def is_empty(self):
        return len(self.geoms) == 0

    @property
    def area(self):
        return float(sum(p.area for p in self.geoms))


def make_linestring(coords):
    return _sg.LineString(coords) if _sg is not None else LineString(coords)


def make_polygon(rings):
    """rings: list of (n,2) arrays -> Polygon (one ring), MultiPolygon (several), empty Polygon (none);
    the three cases of index_utils.py:167-173."""
    P = _sg.Polygon if _sg is not None else Polygon
    M = _sg.MultiPolygon if _sg is not None else MultiPolygon
    if len(rings) == 0:
        return P()
    if len(rings) == 1:
        return P(rings[0])
    return M([P(r) for r in rings])


def make_frame(columns, geometry):
    """GeoDataFrame when geopandas is importable, else a pandas DataFrame with a `geometry` column."""
    import pandas as pd
    df = pd.DataFrame(columns)
    if _gpd is not None:  # pragma: no cover
        return _gpd.GeoDataFrame(df, geometry=list(geometry))
    df["geometry"] = list(geometry)
    return df


def empty_frame():
    import pandas as pd
    return _gpd.GeoDataFrame() if _gpd is not None else pd.DataFrame()


def frame_types():
    import pandas as pd
    return (pd.DataFrame,) if _gpd is None else (_gpd.GeoDataFrame,)
