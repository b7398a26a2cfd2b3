// wb_pairs.h -- the all-pairs test of the streamer index (streamer_index.py:130-157) with the contour resident in LDS.
//
// wb_k_streamer_pairs (wb_indices.h) reads everything a tile needs from global memory: the block boxes, the tile's rows, the
// candidates' six per-point values, each a dependent round trip to L2, and the per-point tables of a batch (60 B x 7.3 M
// points) do not stay in any cache: 2.5 GB fetched per launch for 0.1 GB of algorithmic input, 0.7 GB written for 64 MB of
// pairs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r02_hbm_traffic_pmc.txt).  Here a workgroup loads ONE contour
// once -- its points (4 B each) and the prefix of the along-contour distance (8 B each) -- and every wave then sweeps
// 16-row tiles of it against the candidate blocks entirely out of LDS:
//   * no per-point trigonometric table: on a regular grid the great-circle term
//         h = sin^2(dphi / 2) + cos(phi_i) cos(phi_j) sin^2(dlam / 2)
//     needs sin^2 of a multiple of dlat / 2 and of dlon / 2 -- two tables of dy_max + 1 and 121 entries (|dx| <= 120,
//     streamer_index.py:157) from the HOST's math library -- and cos(phi) per latitude row (nlat entries);
//     the comparison with sin^2(geo_dis / 2R) keeps its guard band, widened by what the grid's coordinates deviate from
//     lat[0] + j dlat / lon[0] + i dlon (a float32 coordinate axis is off by 1e-6 degrees): inside it the pair is evaluated
//     as sklearn evaluates it, outside it no decision can depend on how h was formed;
//   * a tile's rows sit in the first 16 lanes' registers and are broadcast by v_readlane; a lane owns one candidate point j
//     of one of TWO candidate blocks (half a wave each), keeps its data in registers and sweeps the rows;
//   * pairs that pass are collected in a per-wave LDS buffer and go to the contour's list WB_P2_STAGE at a time with one
//     returning atomic and full-line stores.
// Contours longer than the LDS holds (a.p2_cap points) stay with the tiled kernel (wb_k_sel_scan leaves them in its list).
#pragma once

#define WB_P2_THREADS 1024         // 16 waves share one copy of the contour; two workgroups per CU = 8 waves per SIMD
#define WB_P2_STAGE 64             // pairs a wave collects before they are appended to the contour's list
#define WB_P2_MAXDX 120            // streamer_index.py:157

// bytes of dynamic LDS for contours of up to `cap` points (cap a multiple of WB_BLK)
WB_HD inline size_t wb_p2_lds_bytes(int cap, int nlat, int dy_max) {
    const size_t nb = (size_t)cap / WB_BLK + 1, ntl = (size_t)cap / WB_PAIR_ROWS + 1;
    return (size_t)cap * 8 + (size_t)nlat * 8 + (size_t)(dy_max + 2) * 8 + (WB_P2_MAXDX + 2) * 8 + nb * 8 +
           (size_t)(WB_P2_THREADS / 64) * WB_P2_STAGE * 8 + (size_t)cap * 4 + (size_t)nlat * 4 + nb * 16 + ntl * 16 + 64;
}

// A pair whose great-circle term or along-contour distance falls inside a guard band is not decided in the sweep: it goes
// to the batch's recheck list (contour, i, j) and wb_k_streamer_recheck evaluates it as the reference does.  (The exact
// evaluation -- sin, cos, asin in float64, a sequential sum -- costs the sweep its registers; the list is empty on a
// regular float64 grid and a few thousand entries long when the coordinates went through float32.)
WB_DEV inline void wb_p2_defer(const WbStreamerArgs& a, int s, int i, int j) {
    const int slot = wb_atomic_add(&a.ctl->n_recheck, 1);
    if (slot < a.recheck_cap) { int* r = a.recheck + 4 * (size_t)slot; r[0] = s; r[1] = i; r[2] = j; r[3] = 0; }
}

// the complete decision for the listed pairs, exact where a band is hit (streamer_index.py:130-139)
WB_KERNEL(wb_k_streamer_recheck, WbStreamerArgs a) {
    const int n = a.ctl->n_recheck < a.recheck_cap ? a.ctl->n_recheck : a.recheck_cap;
    for (int q0 = WB_BLOCK_ID * 64; q0 < n; q0 += WB_GRID_DIM * 64) {
        WB_PFOR(t, 64)
            const int q = q0 + (int)t;
            if (q < n) {
                const int s = a.recheck[4 * (size_t)q], i = a.recheck[4 * (size_t)q + 1], j = a.recheck[4 * (size_t)q + 2];
                const int job = a.sel_job[s], k = a.sel_cont[s];
                const size_t jb = (size_t)job * a.cap_seg;
                const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
                const long long po = a.poff[s];
                const double* on = a.on + po;
                const double* pre = a.pre + po;
                const uint32_t pi = pts[i], pj = pts[j];
                const int yi = wb_pt_y(pi), yj = wb_pt_y(pj), xi = wb_pt_x(pi), xj = wb_pt_x(pj);
                const int dx = xi > xj ? xi - xj : xj - xi, dy = yi > yj ? yi - yj : yj - yi;
                const double h = a.sl_tab[dy] + a.cphi[yi] * a.cphi[yj] * a.sx_tab[dx];
                bool near_geo = h < a.hthr;
                if (fabs(h - a.hthr) <= a.geo_band) {                     // evaluate as sklearn does
                    wb_atomic_add(&a.ctl->band[0], 1);
                    near_geo = wb_haversine(a.lat_rad[yi], a.lon_rad[xi % a.nlon], a.lat_rad[yj], a.lon_rad[xj % a.nlon]) * 6371.0 < a.geo_dis;
                }
                if (near_geo) {
                    double cont = pre[j] - pre[i];
                    if (fabs(cont - a.cont_dis) <= WB_CONT_BAND) {        // sequential row cumsum (streamer_index.py:135)
                        wb_atomic_add(&a.ctl->band[1], 1);
                        double acc = 0.0;
                        for (int m = i + 1; m <= j; ++m) acc = acc + on[m];
                        cont = acc;
                    }
                    if (cont > a.cont_dis) {
                        const int slot = wb_atomic_add(&a.npairs[s], 1);
                        if (slot < a.cap_pairs) a.pairs[(size_t)s * a.cap_pairs + slot] = ((unsigned long long)i << 32) | (unsigned long long)j;
                    }
                }
            }
        WB_END_PFOR
    }
}

#if defined(WB_HOSTSIM)
// host simulation: the same tiles, block filter, tables and guard bands, one pair after the other
WB_KERNEL(wb_k_streamer_pairs2, WbStreamerArgs a) {
    for (int item = WB_BLOCK_ID; item < a.ctl->nsel * a.p2_split; item += WB_GRID_DIM) {      // (the launch is persistent: a grid of resident workgroups)
    const int s = item / a.p2_split, part = item - s * a.p2_split;
    if (part != 0) continue;
    const int job = a.sel_job[s], k = a.sel_cont[s];
    const size_t jb = (size_t)job * a.cap_seg;
    const uint32_t* pts = a.cpts + 2 * jb + a.c_off[jb + k];
    const int N = a.c_len[jb + k];
    if (N > a.p2_cap) continue;
    const long long po = a.poff[s];
    const double* pre = a.pre + po;
    const int* bbx = a.bbx + 4 * ((po >> 5) + s);
    const int* tbx = a.tbx + 4 * ((po >> 4) + s);
    const int nblk = (N + WB_BLK - 1) / WB_BLK;
    unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    long long evaluated = 0;
    for (int i0 = 0; i0 < N - 1; i0 += WB_PAIR_ROWS) {
        const int nrows = N - 1 - i0 < WB_PAIR_ROWS ? N - 1 - i0 : WB_PAIR_ROWS;
        const int* tb4 = tbx + 4 * (i0 / WB_PAIR_ROWS);                 // box of rows i0 .. i0 + 15
        const int ya = tb4[2] < 0 ? 0 : (tb4[2] > a.nlat - 1 ? a.nlat - 1 : tb4[2]), yb = tb4[3] < 0 ? 0 : (tb4[3] > a.nlat - 1 ? a.nlat - 1 : tb4[3]);
        const int ml = a.dxlim[ya] > a.dxlim[yb] ? a.dxlim[ya] : a.dxlim[yb];
        const int t0 = tb4[0] - ml, t1 = tb4[1] + ml, t2 = tb4[2] - a.dy_max, t3 = tb4[3] + a.dy_max;
        for (int b = i0 / WB_BLK; b < nblk; ++b) {
            const int jlo = b * WB_BLK, jhi = jlo + WB_BLK - 1 < N - 1 ? jlo + WB_BLK - 1 : N - 1;
            if (jhi <= i0) continue;
            const int* bb = bbx + 4 * b;
            if (bb[0] > t1 || bb[1] < t0 || bb[2] > t3 || bb[3] < t2) continue;
            if (pre[jhi] - pre[i0] <= a.cont_dis - WB_CONT_BAND) continue;
            for (int j = jlo > i0 + 1 ? jlo : i0 + 1; j <= jhi; ++j) {
                const int xj = wb_pt_x(pts[j]), yj = wb_pt_y(pts[j]);
                for (int ii = 0; ii < nrows; ++ii) {
                    const int i = i0 + ii;
                    if (j <= i) break;
                    const int xi = wb_pt_x(pts[i]), yi = wb_pt_y(pts[i]);
                    const int dx = xi > xj ? xi - xj : xj - xi, dy = yi > yj ? yi - yj : yj - yi;
                    if (dx > a.dxlim[yi] || dy > a.dy_max) continue;
                    const double h = a.sl_tab[dy] + a.cphi[yi] * a.cphi[yj] * a.sx_tab[dx];
                    ++evaluated;
                    const double cont = pre[j] - pre[i];
                    const bool near_geo = h < a.hthr;
                    if (fabs(h - a.hthr) <= a.geo_band || (near_geo && fabs(cont - a.cont_dis) <= WB_CONT_BAND)) { wb_p2_defer(a, s, i, j); continue; }
                    if (!near_geo || !(cont > a.cont_dis)) continue;
                    const int slot = wb_atomic_add(&a.npairs[s], 1);
                    if (slot < a.cap_pairs) pairs[slot] = ((unsigned long long)i << 32) | (unsigned long long)j;
                }
            }
        }
    }
    wb_atomic_add((unsigned long long*)&a.ctl->evaluated_pairs, (unsigned long long)evaluated);
    }
}
#else
__device__ inline double wb_readlane_f64(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__global__ __launch_bounds__(WB_P2_THREADS, 8) void wb_k_streamer_pairs2(WbStreamerArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wb_p2_lds[];
    WB_PSTAMP_BEGIN
    const int cap = a.p2_cap, nbmax = cap / WB_BLK + 1, ntmax = cap / WB_PAIR_ROWS + 1, nlat = a.nlat, dy_max = a.dy_max;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = WB_P2_THREADS / 64;
    // LDS: 8-byte arrays first
    double* lpre = (double*)wb_p2_lds;                                   // [cap] prefix of the along-contour distance
    double* lcphi = lpre + cap;                                          // [nlat] cos(latitude)
    double* lsl = lcphi + nlat;                                          // [dy_max + 2] sin^2(k dlat / 2)
    double* lsx = lsl + dy_max + 2;                                      // [122] sin^2(k dlon / 2)
    double* lbpre = lsx + WB_P2_MAXDX + 2;                               // [nbmax] prefix at the last point of a block
    unsigned long long* lstage = (unsigned long long*)(lbpre + nbmax);   // [nwaves][WB_P2_STAGE]
    uint32_t* lpts = (uint32_t*)(lstage + nwaves * WB_P2_STAGE);         // [cap] x | y << 16
    int* ldxl = (int*)(lpts + cap);                                      // [nlat] reach in x of a row
    int* lbb = ldxl + nlat;                                              // [nbmax][4] block boxes
    int* ltb = lbb + 4 * nbmax;                                          // [ntmax][4] tile boxes
    int* ltile = ltb + 4 * ntmax;                                        // [1] next tile of this workgroup
    // the grid's tables once per workgroup; the workgroups are persistent (two per CU) and take (contour, part) items from
    // a device counter: launched one per contour, 2 920 workgroups of 1024 threads and 64 KB of LDS kept the device a third
    // full (178 ms of workgroup time in a 1.05 ms launch = 170 of 512 possible workgroups resident on average)
    for (int y = tid; y < nlat; y += WB_P2_THREADS) { lcphi[y] = a.cphi[y]; ldxl[y] = a.dxlim[y]; }
    for (int q = tid; q <= dy_max; q += WB_P2_THREADS) lsl[q] = a.sl_tab[q];
    for (int q = tid; q <= WB_P2_MAXDX; q += WB_P2_THREADS) lsx[q] = a.sx_tab[q];
    const int nitems = a.ctl->nsel * a.p2_split;
    long long evaluated = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const double hthr = a.hthr, geo_band = a.geo_band, cont_dis = a.cont_dis;
    // An item's descriptor -- the queue's atomic, then contour -> offset / length: a chain of three dependent loads, 8 us before
    // the first point could be asked for -- is fetched ONE ITEM AHEAD by a single lane of the last wave, while the workgroup
    // sweeps the present item; it goes through two LDS rows in turn.
    __shared__ long long wb_p2_desc[2][4];                               // item, points of the contour, offset of its points, offset of its prefix
    auto fetch = [&](int row) {
        const int it = atomicAdd(&a.ctl->p2_cursor, 1);
        long long nn = 0, off = 0, pp = 0;
        if (it < nitems) {
            const int ss = it / a.p2_split;
            const int jj = a.sel_job[ss], kk = a.sel_cont[ss];
            const size_t jb2 = (size_t)jj * a.cap_seg;
            off = (long long)(2 * jb2) + a.c_off[jb2 + kk];
            nn = a.c_len[jb2 + kk];
            pp = a.poff[ss];
        }
        wb_p2_desc[row][0] = it; wb_p2_desc[row][1] = nn; wb_p2_desc[row][2] = off; wb_p2_desc[row][3] = pp;
    };
    if (tid == 0) fetch(0);
    int row = 0;
    for (;;) {
    __syncthreads();                                                     // (the previous item's LDS is done with; this item's descriptor stands)
    const int item = (int)wb_p2_desc[row][0];
    if (item >= nitems) break;
    const int N = (int)wb_p2_desc[row][1];
    const uint32_t* gpts = a.cpts + wb_p2_desc[row][2];
    const long long po = wb_p2_desc[row][3];
    if (tid == WB_P2_THREADS - 64) fetch(row ^ 1);                       // (the row read before the barrier above; read next behind the next one)
    row ^= 1;
    const int s = item / a.p2_split, part = item - s * a.p2_split;
    if (N > a.p2_cap) continue;                                          // the tiled kernel's list holds this contour
    const double* gpre = a.pre + po;
    const int* gbbx = a.bbx + 4 * ((po >> 5) + s);
    const int* gtbx = a.tbx + 4 * ((po >> 4) + s);
    const int nblk = (N + WB_BLK - 1) / WB_BLK;
    unsigned long long* pairs = a.pairs + (size_t)s * a.cap_pairs;
    const int ntile = (N - 1 + WB_PAIR_ROWS - 1) / WB_PAIR_ROWS;
    for (int i = tid; i < N; i += WB_P2_THREADS) { lpts[i] = gpts[i]; lpre[i] = gpre[i]; }
    for (int q = tid; q < 4 * nblk; q += WB_P2_THREADS) lbb[q] = gbbx[q];
    for (int q = tid; q < 4 * ntile && q < 4 * ntmax; q += WB_P2_THREADS) ltb[q] = gtbx[q];
    if (tid == 0) ltile[0] = 0;
    for (int b = tid; b < nblk; b += WB_P2_THREADS) {
        const int jhi = b * WB_BLK + WB_BLK - 1 < N - 1 ? b * WB_BLK + WB_BLK - 1 : N - 1;
        lbpre[b] = gpre[jhi];
    }
    WB_PSTAMP(a.ctl, 6);                                                 // (development stamps: 6 load, 0 barrier, 1 tile set-up, 2 block
    __syncthreads();                                                     //  filter, 3 sweep, 4 append, 5 tail)
    WB_PSTAMP(a.ctl, 0);

    unsigned long long* stage = lstage + wave * WB_P2_STAGE;
    int fill = 0;                                                        // pairs waiting in this wave's buffer (wave-uniform)
    auto flush = [&]() {
        // the buffer's pairs go to the contour's list: one returning atomic, consecutive slots
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        int base = 0;
        if (lane == 0) base = atomicAdd(&a.npairs[s], fill);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int q = lane; q < fill; q += 64)
            if (base + q < a.cap_pairs) pairs[base + q] = stage[q];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        fill = 0;
    };
    // tiles are taken one at a time from a counter in LDS (the part'th of every p2_split tiles belongs to this workgroup): a
    // tile that meets a streamer costs many sweeps, and the workgroup's 16 waves leave together
    for (;;) {
        int mine = 0;
        if (lane == 0) mine = atomicAdd(ltile, 1);
        const int tile = __builtin_amdgcn_readfirstlane(mine) * a.p2_split + part;
        if (tile >= ntile) break;
        const int i0 = tile * WB_PAIR_ROWS;
        const int nrows = N - 1 - i0 < WB_PAIR_ROWS ? N - 1 - i0 : WB_PAIR_ROWS;     // rows i0 .. i0 + nrows - 1
        // the rows: lane q holds row min(q & 15, nrows - 1); their packed points go to SGPRs once per tile (v_readlane)
        const int rq = (lane & 15) < nrows ? (lane & 15) : nrows - 1;
        const uint32_t rp = lpts[i0 + rq];
        int sxy[WB_PAIR_ROWS];
#pragma unroll
        for (int q = 0; q < WB_PAIR_ROWS; ++q) sxy[q] = __builtin_amdgcn_readlane((int)rp, q);
        // box of the rows (wb_k_streamer_prep; rows i0 .. i0 + 15, a superset), widened by the reach of a pair.  dxlim grows
        // with |latitude|: over the tile's rows it is largest at one of the two ends of their y range -- that reach also
        // serves as the |dx| limit of every row of the tile (each row's own limit is a necessary condition for a great-circle
        // distance below geo_dis and nothing more; the cap of 120 columns, streamer_index.py:157, is in every entry)
        const int* tb4 = ltb + 4 * tile;
        const int bx0 = tb4[0], bx1 = tb4[1], by0 = tb4[2], by1 = tb4[3];
        const int ya = by0 < 0 ? 0 : (by0 > nlat - 1 ? nlat - 1 : by0), yb = by1 < 0 ? 0 : (by1 > nlat - 1 ? nlat - 1 : by1);
        const int la = ldxl[ya], lb = ldxl[yb];
        const int ml = __builtin_amdgcn_readfirstlane(la > lb ? la : lb);
        const int t0 = bx0 - ml, t1 = bx1 + ml, t2 = by0 - dy_max, t3 = by1 + dy_max;
        const double pre_i0 = lpre[i0];
        WB_PSTAMP(a.ctl, 1);
        for (int b0 = i0 / WB_BLK; b0 < nblk; b0 += 64) {
            const int b = b0 + lane;
            bool ok = b < nblk;
            if (ok) {
                const int jhi = b * WB_BLK + WB_BLK - 1 < N - 1 ? b * WB_BLK + WB_BLK - 1 : N - 1;
                const int* bb = lbb + 4 * b;
                ok = jhi > i0 && !(bb[0] > t1 || bb[1] < t0 || bb[2] > t3 || bb[3] < t2) &&
                     !(lbpre[b] - pre_i0 <= cont_dis - WB_CONT_BAND);         // cont(i, j) <= pre[jhi] - pre[i0]
            }
            unsigned long long todo = __ballot(ok);
            WB_PSTAMP(a.ctl, 2);
            while (todo) {
                // two candidate blocks per sweep: the lower half of the wave takes one, the upper half the next
                const int ba = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                int bbk = -1;
                if (todo) { bbk = __ffsll((long long)todo) - 1; todo &= todo - 1ull; }
                const int myb = lane < 32 ? ba : bbk;
                const int j = (b0 + (myb < 0 ? 0 : myb)) * WB_BLK + (lane & 31);
                // a point that is within cont_dis of the tile's FIRST row along the contour is within it of every row
                const double prej = lpre[j < N ? j : 0];
                const bool live = myb >= 0 && j > i0 && j < N && !(prej - pre_i0 <= cont_dis - WB_CONT_BAND);
                const uint32_t pj = lpts[live ? j : 0];
                const uint32_t xj = (uint32_t)wb_pt_x(pj), yj = (uint32_t)wb_pt_y(pj);
                const double cj = lcphi[yj];
                // only j > i: matters where the block overlaps the tile's own rows (wave-uniform test on the first block)
                const bool beyond = (b0 + ba) * WB_BLK >= i0 + WB_PAIR_ROWS;
                const int dj = beyond ? WB_PAIR_ROWS : j - i0;               // rows 0 .. dj - 1 have i < j
                uint32_t passed = 0, inband = 0;                          // bit ii: (i0 + ii, j) is a candidate / goes to the recheck list
                // Row by row (their points are in SGPRs): |dx| <= reach (streamer_index.py:157: <= 120), |dy| <= dy_max (R |dphi|
                // alone within geo_dis) -- four VALU operations (v_sad_u16 on the zero-extended halves) -- and only where some
                // lane passes, the great-circle term from the tables and the along-contour distance
#pragma unroll
                for (int q = 0; q < WB_PAIR_ROWS; ++q) {
                    const uint32_t xi = (uint32_t)sxy[q] & 0xFFFFu, yi = (uint32_t)sxy[q] >> 16;
                    const uint32_t dx = __builtin_amdgcn_sad_u16(xj, xi, 0u), dy = __builtin_amdgcn_sad_u16(yj, yi, 0u);
                    const bool in = live && dx <= (uint32_t)ml && dy <= (uint32_t)dy_max && q < dj && q < nrows;
                    if (__ballot(in) == 0ull) continue;
                    const double ci = lcphi[yi], prei = lpre[i0 + q];     // (uniform addresses: broadcast reads, in flight with the two below)
                    const double h = lsl[in ? dy : 0u] + ci * cj * lsx[in ? dx : 0u];
                    const double cont = prej - prei;
                    if (in) {
                        ++evaluated;
                        const bool near_geo = h < hthr;
                        if (fabs(h - hthr) <= geo_band || (near_geo && fabs(cont - cont_dis) <= WB_CONT_BAND)) inband |= 1u << q;
                        else if (near_geo && cont > cont_dis) passed |= 1u << q;
                    }
                }
                WB_PSTAMP(a.ctl, 3);
                if (__any(inband != 0)) {
                    while (inband) {
                        const int ii = __ffs((int)inband) - 1;
                        inband &= inband - 1u;
                        wb_p2_defer(a, s, i0 + ii, j);
                    }
                }
                // append: prefix and total of the lanes' counts by bit-sliced ballots
                const int n = __popc(passed);
                int prefix = 0, total = 0;
#pragma unroll
                for (int bit = 0; bit < 5; ++bit) {
                    const unsigned long long m = __ballot((n >> bit) & 1);
                    prefix += __popcll(m & lt) << bit;
                    total += __popcll(m) << bit;
                }
                if (total == 0) continue;
                if (total > WB_P2_STAGE) {
                    // more than the buffer holds (a dense corner of a streamer): straight to the list
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&a.npairs[s], total);
                    base = __builtin_amdgcn_readfirstlane(base);
                    int slot = base + prefix;
                    while (passed) {
                        const int ii = __ffs((int)passed) - 1;
                        passed &= passed - 1u;
                        if (slot < a.cap_pairs) pairs[slot] = ((unsigned long long)(i0 + ii) << 32) | (unsigned long long)j;
                        ++slot;
                    }
                    continue;
                }
                if (fill + total > WB_P2_STAGE) flush();
                int slot = fill + prefix;
                while (passed) {
                    const int ii = __ffs((int)passed) - 1;
                    passed &= passed - 1u;
                    stage[slot++] = ((unsigned long long)(i0 + ii) << 32) | (unsigned long long)j;
                }
                fill += total;
                WB_PSTAMP(a.ctl, 4);
            }
        }
    }
    if (fill) flush();
    WB_PSTAMP(a.ctl, 5);
    }
    WB_PSTAMP_FLUSH(a.ctl);
    // one global atomic per wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) evaluated += __shfl_xor(evaluated, d, 64);
    if (lane == 0 && evaluated) atomicAdd((unsigned long long*)&a.ctl->evaluated_pairs, (unsigned long long)evaluated);
}
#endif
