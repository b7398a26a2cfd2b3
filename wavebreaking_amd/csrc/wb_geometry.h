// wb_geometry.h -- host side: event polygons in the data's own coordinates (transform_polygons, index_utils.py:129-184).
//
// An event's polygon is a list of index-lattice vertices in the periodically extended grid.  The reference maps it to
// (lon[x % nlon], lat[y]) (:139-146) and, when it reaches x >= nlon, first removes the strip nlon - 1 < x < nlon between the
// last meridian and its periodic copy (:148-173, GEOS linemerge / unary_union / polygonize / make_valid there -- unpinned
// here; see oracle/properties.py for an independent exact-arithmetic statement the tests compare ring sets with), keeping
// what lies on either side as separate rings, vertices truncated to the lattice (:135).  This is per-event work on a few
// hundred vertices: done on the host, in C++, for all events of a table at once, so that the Python layer only wraps arrays.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

struct WbRingSink {
    std::vector<int64_t> ring_offset;   // [events + 1] first ring of an event
    std::vector<int64_t> vert_offset;   // [rings + 1]  first vertex of a ring
    std::vector<double> x, y;
};

struct WbPt { double x, y; };

// pieces of the closed ring v on one side of the vertical line x = L (closed half-plane): boundary chains on the kept side,
// re-connected along the line by pairing the line points sorted by y
static void wb_clip_halfplane(const std::vector<WbPt>& v, double L, bool keep_le, std::vector<std::vector<WbPt>>& pieces) {
    const size_t n = v.size();
    std::vector<char> ins(n);
    size_t nin = 0;
    for (size_t k = 0; k < n; ++k) { ins[k] = keep_le ? v[k].x <= L : v[k].x >= L; nin += ins[k]; }
    if (nin == n) { pieces.push_back(v); return; }
    if (nin == 0) return;
    size_t s = 0;
    while (ins[s]) ++s;                               // start on an outside vertex
    std::vector<std::vector<WbPt>> chains;
    std::vector<WbPt> cur;
    bool open = false;
    for (size_t k = 0; k < n; ++k) {
        const size_t ia = (s + k) % n, ib = (s + k + 1) % n;
        const WbPt &a = v[ia], &b = v[ib];
        if (!ins[ia] && ins[ib]) {                    // entering
            const double t = (L - a.x) / (b.x - a.x);
            cur.clear();
            cur.push_back({L, a.y + t * (b.y - a.y)});
            cur.push_back(b);
            open = true;
        } else if (ins[ia] && ins[ib]) {
            if (open) cur.push_back(b);
        } else if (ins[ia] && !ins[ib]) {             // leaving
            const double t = (L - a.x) / (b.x - a.x);
            cur.push_back({L, a.y + t * (b.y - a.y)});
            chains.push_back(cur);
            open = false;
        }
    }
    struct LinePt { double y; int kind; int chain; };
    std::vector<LinePt> pts;
    for (size_t c = 0; c < chains.size(); ++c) {
        pts.push_back({chains[c].front().y, 0, (int)c});
        pts.push_back({chains[c].back().y, 1, (int)c});
    }
    std::stable_sort(pts.begin(), pts.end(), [](const LinePt& p, const LinePt& q) { return p.y < q.y || (p.y == q.y && p.kind < q.kind); });
    // partner of a chain's exit point: the point it is paired with (consecutive pairs of the sorted list)
    std::vector<int> next_chain(chains.size(), -1);
    for (size_t k = 0; k + 1 < pts.size(); k += 2) {
        const LinePt &p = pts[k], &q = pts[k + 1];
        if (p.kind == 1) next_chain[p.chain] = q.chain;
        if (q.kind == 1) next_chain[q.chain] = p.chain;
    }
    std::vector<char> used(chains.size(), 0);
    for (size_t c0 = 0; c0 < chains.size(); ++c0) {
        if (used[c0]) continue;
        std::vector<WbPt> piece;
        int c = (int)c0;
        size_t guard = 0;
        while (c >= 0 && !used[c] && guard <= chains.size()) {
            used[c] = 1;
            piece.insert(piece.end(), chains[c].begin(), chains[c].end());
            c = next_chain[c] >= 0 ? next_chain[c] : c;
            ++guard;
        }
        double a2 = 0.0;
        for (size_t k = 0; k < piece.size(); ++k) {
            const WbPt &p = piece[k], &q = piece[(k + 1) % piece.size()];
            a2 += p.x * q.y - q.x * p.y;
        }
        if (piece.size() >= 3 && std::fabs(a2) > 0.0) pieces.push_back(piece);
    }
}

// one event: index polygon -> rings in (lon, lat); index_space: the rings as they are cut, in index coordinates, before the
// truncation and the look-up (what the tests measure areas on)
static void wb_event_rings(const int32_t* px, const int32_t* py, int64_t n, const double* lon, const double* lat, int nlon,
                           WbRingSink& out, bool index_space = false) {
    bool split = false;
    for (int64_t k = 0; k < n; ++k) split |= px[k] >= nlon;                 // index_utils.py:136
    auto emit = [&](const std::vector<WbPt>& ring) {
        for (const WbPt& p : ring) {
            if (index_space) { out.x.push_back(p.x); out.y.push_back(p.y); continue; }
            const long long xi = (long long)p.x, yi = (long long)p.y;       // :135 truncation
            out.x.push_back(lon[((xi % nlon) + nlon) % nlon]);              // :139-146
            out.y.push_back(lat[yi]);
        }
        out.vert_offset.push_back((int64_t)out.x.size());
    };
    std::vector<WbPt> v((size_t)n);
    for (int64_t k = 0; k < n; ++k) v[(size_t)k] = {(double)px[k], (double)py[k]};
    if (!split) {
        emit(v);
    } else {
        std::vector<std::vector<WbPt>> pieces;
        wb_clip_halfplane(v, (double)(nlon - 1), true, pieces);
        wb_clip_halfplane(v, (double)nlon, false, pieces);
        for (const auto& p : pieces) emit(p);
    }
    out.ring_offset.push_back((int64_t)out.vert_offset.size() - 1);
}
