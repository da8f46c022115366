// raster2.hip -- pipeline 2 of the hot path (edge list -> RGBA8 framebuffer in HBM), CDNA4 / gfx950.
//
// Same arithmetic as raster_kernels.hip (Cairo's "tor" scan conversion; the row routines below are
// the ones of that file with another output format), different data flow:
//
//   k2_front   per edge: scan-converter constants (A.5 make_edge); per (path, tile-row): the band entry; counters cleared
//   k2_rows    one wavefront per (path, <= 64 pixel rows), lane = row: active edges, FULL / SUB decision, roles -- and then the
//              row's CELLS {column, covered height, uncovered area}, i.e. Cairo's cell list itself (A.5 render_edge /
//              add_subspan), densely packed per wavefront (one allocation per wavefront from eight bump allocators), plus the
//              class byte of every (tile, path) pair.  A row whose edge order needs Cairo's list history (coincident edges), or
//              with more active edges than the registers hold, goes to a queue.
//   k2_rows_slow / k2_rows_huge   the queued rows, one wavefront (workgroup) each, same output
//   k2_tiles   persistent wavefronts, each walks 64x8-pixel strips: class bytes -> surviving band entries (everything under the
//              last opaque full cover is culled from the class bytes alone) -> the cells of the partial paths as one coalesced
//              stream -> LDS accumulators -> wave64 prefix sum -> alpha -> shade -> blend in registers -> one store per pixel.
//              No edge arithmetic, no 64-bit or floating-point instruction on the solid-colour path.
//
// Every kernel takes an array of frame descriptors and blockIdx.y picks the frame: a batch of frames is one launch per kernel.
#include "raster_kernels.hip"

namespace swfr {

// Everything the kernels need to know about one frame (device memory; blockIdx.y indexes an array of these).
struct Frame2 {
    // the scene (read-only)
    const swfr_edge* raw; const DevPath* paths; const swfr_style* styles;
    const ChunkInfo* chunks; const BandSlot* band_slots; const uint32_t* band_off; const uint32_t* order;
    Sources src;
    // per frame in flight (kernel-written)
    DevEdge* edges; BandEntry2* band_list; uint8_t* cls; RowInfo2* rows; Cell* cells; SlowRow* slow; SlowRow* huge; uint32_t* counters;
    uint32_t* fb;
    uint32_t n_edges, n_paths, n_chunks, n_slots, n_bands, n_strips, cell_slice, slow_cap;
    int32_t width, height, tiles_x;
    uint32_t band_index, band_count, fast_limit, any_shader, dbg;
    uint32_t pad[3];
};

#define CLS_OPAQUE 32u                // the path is an opaque solid blended with the lerp rule: a full cover of it hides what lies below

// ---------------------------------------------------------------------------------------------
// cells
// ---------------------------------------------------------------------------------------------
// one cell, clipped to the converter's column range [xminp, xmaxp): at / after the right bound it is never emitted (the slot is
// still written, as a zero), left of the left bound only its height counts, folded into the first column  (oracle: cell_add)
__device__ __forceinline__ void put_cell(Cell* __restrict__ dst, int col, int ch, int ua, int xminp, int xmaxp) {
    if (col >= xmaxp) { col = xmaxp - 1; ch = 0; ua = 0; }
    else if (col < xminp) { col = xminp; ua = 0; }
    Cell c; c.col = (int16_t)col; c.ch = (int16_t)ch; c.ua = ua;
    *dst = c;
}
// number of cell slots a FULL-row edge with end-point quotients q1 (row top), q2 (row bottom) gets
__device__ __forceinline__ int full_span(int32_t q1, int32_t q2) {
    const int a = q1 >> 8, b = q2 >> 8;
    const int n = (a > b ? a - b : b - a) + 1;
    return n > MAX_CELLS_PER_EDGE_ROW ? MAX_CELLS_PER_EDGE_ROW : n;
}
// Cells of a FULL-row edge (A.5 render_edge) from its exact x at the row top (q1 + r1/dy) and bottom (q2 + r2/dy); writes exactly
// full_span(q1, q2) cells.  An edge that spans more columns than that has at most 17 cells with a non-zero height (the heights add
// up to the row's fifteen sample rows): the zero ones are skipped.
__device__ __forceinline__ void full_cells(int32_t q1, int64_t r1, int32_t q2, int64_t r2, int64_t edy, int sign, int xminp, int xmaxp, Cell* __restrict__ dst) {
    int ix1 = q1 >> 8, f1 = q1 & 255, ix2 = q2 >> 8, f2 = q2 & 255;
    if (ix1 == ix2) { put_cell(dst, ix1, sign * 15, sign * (f1 + f2) * 15, xminp, xmaxp); return; }
    if (ix2 < ix1) { int t = ix1; ix1 = ix2; ix2 = t; t = f1; f1 = f2; f2 = t; int32_t tq = q1; q1 = q2; q2 = tq; int64_t tr = r1; r1 = r2; r2 = tr; }
    const int span = ix2 - ix1 + 1;
    const int64_t dx = (int64_t)(q2 - q1) * edy + (r2 - r1);
    const int64_t t0 = ((int64_t)((ix1 + 1) * 256 - q1) * edy - r1) * 15;
    int64_t yq, yr, fq = 0, fr = 0;
    floor_div(t0, dx, yq, yr);
    if (span > 2) floor_div(15ll * 256 * edy, dx, fq, fr);
    int y_prev = (int)yq;
    if (span <= MAX_CELLS_PER_EDGE_ROW) {
        put_cell(dst, ix1, sign * y_prev, sign * y_prev * (256 + f1), xminp, xmaxp);
#pragma unroll 1
        for (int k = 1; k < span - 1; ++k) {
            yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; }
            const int h = (int)yq - y_prev;
            put_cell(dst + k, ix1 + k, sign * h, sign * h * 256, xminp, xmaxp);
            y_prev = (int)yq;
        }
        put_cell(dst + span - 1, ix2, sign * (15 - y_prev), sign * (15 - y_prev) * f2, xminp, xmaxp);
        return;
    }
    int n = 0;
    if (y_prev) put_cell(dst + n++, ix1, sign * y_prev, sign * y_prev * (256 + f1), xminp, xmaxp);
#pragma unroll 1
    for (int c = ix1 + 1; c < ix2; ++c) {
        yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; }
        const int h = (int)yq - y_prev;
        if (h && n < MAX_CELLS_PER_EDGE_ROW - 1) put_cell(dst + n++, c, sign * h, sign * h * 256, xminp, xmaxp);
        y_prev = (int)yq;
    }
    if (15 - y_prev) put_cell(dst + n++, ix2, sign * (15 - y_prev), sign * (15 - y_prev) * f2, xminp, xmaxp);
    while (n < MAX_CELLS_PER_EDGE_ROW) put_cell(dst + n++, xminp, 0, 0, xminp, xmaxp);
}
// one end of a sample-row span at cell position x (24.8, already rounded to the sample grid): A.5 add_subspan
__device__ __forceinline__ void sub_cell(Cell* __restrict__ dst, int x, int sgn, int xminp, int xmaxp) {
    put_cell(dst, x >> 8, sgn, sgn * 2 * (x & 255), xminp, xmaxp);
}
// `n` cells for the calling wavefront (one lane allocates; every lane gets the base): eight bump allocators, each with its own
// slice of the frame's arena, picked by workgroup number.  ~0u when the slice is full (the frame then fails loudly).
__device__ __forceinline__ uint32_t alloc_cells(uint32_t* __restrict__ counters, uint32_t n, uint32_t slice, int lane) {
    uint32_t base = 0;
    if (lane == 0 && n) {
        const uint32_t head = blockIdx.x % C2_HEADS;
        const uint32_t old = atomicAdd(&counters[C2_HEAD + head], n);
        base = head * slice + old;
        if ((uint64_t)old + n > slice) { atomicOr(&counters[C2_ERROR], E2_CELL_ARENA); base = ~0u; }
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
}

// ---------------------------------------------------------------------------------------------
// k2_front
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ BandEntry2 make_band_entry2(const DevPath& P, uint32_t p, uint32_t band, const swfr_style* __restrict__ styles) {
    BandEntry2 e;
    e.x_min = (int16_t)P.x_min; e.x_max = (int16_t)P.x_max; e.y_min = (int16_t)P.y_min; e.y_max = (int16_t)P.y_max;
    e.style = P.style; e.first_edge = P.first_edge; e.n_edges = P.n_edges; e.path = p;
    const uint32_t kind = styles[P.style].kind, pixel = styles[P.style].pixel;
    uint32_t fl = 0;
    if (P.kind == SWFR_PATH_BOXES) fl |= BE_BOXES;
    if (P.lerp) fl |= BE_LERP;
    if (kind == SWFR_STYLE_SOLID) fl |= BE_SOLID;
    if (kind == SWFR_STYLE_SOLID && P.lerp && (pixel >> 24) == 0xffu) fl |= BE_OPAQUE_COVER;
    e.flags = fl | (band << 8); e.solid = pixel;
    return e;
}
// blocks [0, n_setup): one thread per edge; the rest: one thread per (path, tile-row) pair -- its band entry and, for a boxes
// path (which has no rows for k2_rows to classify), the class bytes of its tiles
__global__ __launch_bounds__(256) void k2_front(const Frame2* __restrict__ frames, uint32_t n_setup_max) {
    const Frame2& F = frames[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x < C2_WORDS) F.counters[threadIdx.x] = 0;
    if (blockIdx.x < n_setup_max) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= F.n_edges) return;
        const swfr_edge e = F.raw[i];
        F.edges[i] = make_dev_edge(e, F.paths[e.reserved]);
        return;
    }
    const uint32_t g = (blockIdx.x - n_setup_max) * 256 + threadIdx.x;
    if (g >= F.n_slots) return;
    const BandSlot bs = F.band_slots[g];
    const DevPath P = F.paths[bs.path];
    const BandEntry2 e = make_band_entry2(P, bs.path, bs.band, F.styles);
    F.band_list[bs.slot] = e;
    if (P.kind == SWFR_PATH_BOXES) {
        const uint32_t b0 = F.band_off[bs.band], n_b = F.band_off[bs.band + 1] - b0;
        uint8_t* out = F.cls + (size_t)F.tiles_x * b0 + (bs.slot - b0);
        const int ty0 = (int)bs.band * TILE_H, tile_y1 = min(ty0 + TILE_H, F.height);
        const uint32_t opq = (e.flags & BE_OPAQUE_COVER) ? CLS_OPAQUE : 0u;
        for (int tc = P.x_min / TILE_W; tc <= (P.x_max - 1) / TILE_W; ++tc) {
            const int tx0 = tc * TILE_W, tile_x1 = min(tx0 + TILE_W, F.width);
            uint32_t f = CLS_BOX | CLS_NONEMPTY | CLS_NOTFULL;
            if (P.n_edges == 1) {                     // one box that contains the whole tile: full cover
                const swfr_edge bx = F.raw[P.first_edge];
                if (bx.x1 <= tx0 * 256 && bx.x2 >= tile_x1 * 256 && bx.y1 <= ty0 * 256 && bx.y2 >= tile_y1 * 256) f = CLS_NONEMPTY | opq;
            }
            out[(size_t)tc * n_b] = (uint8_t)f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k2_rows
// ---------------------------------------------------------------------------------------------
struct SubStage {
    Cell cells[4][ROWS_FAST_N * 15];   // the cells of the (up to four) SUB rows of one pass
    uint32_t cnt[4];
};

// fast_rows of raster_kernels.hip with two changes.  (1) A row in which two edges on different lines coincide at the first sample
// row (their order is a matter of Cairo's list history) is not decided here: `defer_out`.  (2) The sample lanes of a SUB row do
// not only set role bits, they produce the row's cells: staged in LDS per pass of four rows, then allocated and copied out by the
// whole wavefront, and the rows' RowInfo2 written (index `ri` per row lane; ~0u: the path has no band entry, nothing is kept).
template <class EPTR>
__device__ __forceinline__ void rows2_fast(EPTR E, uint32_t n_list, const DevPath& P, int r, bool live, int fast_limit, FastLds& F, SubStage& S, int lane,
                                           uint32_t& mode_out, int& n_out_edges, bool& overflow_out, bool& defer_out,
                                           int32_t (&roles)[ROWS_FAST_N], int32_t (&cols)[ROWS_FAST_N], int (&el)[ROWS_FAST_N],
                                           int32_t (&Q1)[ROWS_FAST_N], int64_t (&R1)[ROWS_FAST_N], int32_t (&Q2)[ROWS_FAST_N], int64_t (&R2)[ROWS_FAST_N],
                                           int& nmax_out, uint32_t ri, const Frame2& FR) {
    const int s0 = r * 15;
    const unsigned mask = P.fill_rule ? 1u : ~0u;
    int n = 0;
    bool mid_row = false, overflow = false;
    int cs[ROWS_FAST_N], ce[ROWS_FAST_N], cp[ROWS_FAST_N], dr[ROWS_FAST_N], nw[ROWS_FAST_N];
#pragma unroll
    for (int s = 0; s < ROWS_FAST_N; ++s) { cs[s] = ce[s] = cp[s] = dr[s] = nw[s] = 0; el[s] = 0; roles[s] = 0; cols[s] = 0; Q1[s] = Q2[s] = 0; R1[s] = R2[s] = 0; }
    // ---- gather: which edges are active in this row (sample rows [ytop, ybot) against the row's fifteen); no arithmetic yet
    if (live) {
        for (uint32_t k = 0; k < n_list; ++k) {
            const int ytop = E[k].ytop, ybot = E[k].ybot;
            if (ybot <= s0 || ytop >= s0 + 15) continue;
            if (n >= fast_limit) { overflow = true; break; }
            mid_row |= (ytop > s0) | (ybot < s0 + 15);
#pragma unroll
            for (int s = 0; s < ROWS_FAST_N; ++s) if (s == n) el[s] = (int)k;
            ++n;
        }
    }
    if (overflow) n = 0;
    // wave-uniform bound on the active edges of any row of this wave: the unrolled slot loops stop there
    const int nmax = __ballot(n > 6) ? 8 : __ballot(n > 4) ? 6 : __ballot(n > 2) ? 4 : 2;
    nmax_out = nmax;
    // ---- rows that can still be FULL: x of every active edge at the first sample row of this pixel row and of the next
    if (n > 0 && !mid_row) {
#pragma unroll
        for (int s = 0; s < ROWS_FAST_N; ++s) {
            if (s >= nmax) continue;                          // wave-uniform
            if (s >= n) continue;
            const DevEdge e = E[el[s]];
            int32_t qa = e.x1, qb = e.x1; int64_t ra = 0, rb = 0;
            int c0 = e.x1, c1 = e.x1, cpv = e.x1;
            if (e.dy) {
                edge_x_at(e, s0, qa, ra);
                edge_x_at(e, s0 + 15, qb, rb);
                c0 = cell_of(qa, ra, e.dy);
                c1 = cell_of(qb, rb, e.dy);
                cpv = c0;
                if (e.ytop < s0) {                        // cell one sample row earlier
                    int32_t q = qa - (int32_t)e.dq; int64_t rm = ra - e.dr;
                    if (rm < 0) { --q; rm += e.dy; } else if (rm >= e.dy) { ++q; rm -= e.dy; }
                    cpv = cell_of(q, rm, e.dy);
                }
                const int32_t hq = (int32_t)(e.dq / 2); const int64_t hr = e.dr / 2;   // half a sample row back: row top / bottom
                qa -= hq; ra -= hr; if (ra < 0) { --qa; ra += e.dy; } else if (ra >= e.dy) { ++qa; ra -= e.dy; }
                qb -= hq; rb -= hr; if (rb < 0) { --qb; rb += e.dy; } else if (rb >= e.dy) { ++qb; rb -= e.dy; }
            }
            cs[s] = c0; ce[s] = c1; cp[s] = cpv; dr[s] = e.dir; nw[s] = (e.ytop == s0) ? 1 : 0;
            Q1[s] = qa; R1[s] = ra; Q2[s] = qb; R2[s] = rb;
        }
    }
    uint32_t mode = ROW_EMPTY;
    bool is_sub = false, defer = false;
    if (n > 0) {
        bool full = !mid_row;
        bool deep = false;
        int wb[ROWS_FAST_N];
        unsigned firstg = 0, lastg = 0;
#pragma unroll
        for (int j = 0; j < ROWS_FAST_N; ++j) wb[j] = 0;
        if (full) {
#pragma unroll
            for (int j = 0; j < ROWS_FAST_N; ++j) {
                if (j >= nmax) continue;                  // wave-uniform
                int w = 0; bool fg = true, lg = true;
#pragma unroll
                for (int i = 0; i < ROWS_FAST_N; ++i) {
                    if (i == j || i >= nmax) continue;
                    const bool valid = i < n && j < n;
                    // does edge i sort before edge j?  Only the cell decides here; rows with coincident cells are settled below
                    const bool tie = cs[i] == cs[j];
                    const bool before = cs[i] < cs[j] || (tie && i < j);
                    deep |= valid && tie;
                    if (valid && before) { w += dr[i]; if (ce[i] > ce[j]) full = false; if (tie) fg = false; }
                    if (valid && !before && tie) lg = false;
                }
                wb[j] = w;
                if (fg) firstg |= 1u << j;
                if (lg) lastg |= 1u << j;
            }
        }
        // coincident cells: edges on one and the same line (a shape edge with fill0 == fill1 is there twice) can go in either
        // order -- index order was used above; any other tie needs the history of Cairo's edge list: the slow-row kernel's job
        if (__ballot(deep && !mid_row) != 0ull) {
#pragma unroll
            for (int s = 0; s < ROWS_FAST_N; ++s) { F.roles[s][lane] = cs[s]; F.eid[s][lane] = (uint16_t)el[s]; }
            if (deep && !mid_row) {
                bool real = false;
                for (int i = 0; i < n && !real; ++i)
                    for (int j = i + 1; j < n && !real; ++j)
                        if (F.roles[i][lane] == F.roles[j][lane]) { const DevEdge ea = E[F.eid[i][lane]], eb = E[F.eid[j][lane]]; real = !same_line(ea, eb); }
                defer = real;
            }
        }
        if (defer) mode = ROW_DEFER;
        else if (full) {
            mode = ROW_FULL;
#pragma unroll
            for (int j = 0; j < ROWS_FAST_N; ++j) {
                if (j >= n) continue;
                const bool in_b = ((unsigned)wb[j] & mask) != 0, in_a = ((unsigned)(wb[j] + dr[j]) & mask) != 0;
                uint32_t role = 0;
                if (!in_b && ((firstg >> j) & 1u)) role = REC_FULL | 1u;          // left edge of a span
                else if (!in_a && ((lastg >> j) & 1u)) role = REC_FULL | 2u;      // right edge
                if (role) {
                    const int a = Q1[j] >> 8, b = Q2[j] >> 8;
                    cols[j] = (int32_t)(clamp_col(min(a, b)) | (clamp_col(max(a, b)) << 16));
                }
                roles[j] = (int32_t)role;
            }
        } else {
            mode = ROW_SUB;
            is_sub = true;
#pragma unroll
            for (int s = 0; s < ROWS_FAST_N; ++s) {
                if (s >= nmax) continue;                      // wave-uniform: the sample lanes stop at nmax as well
                F.eid[s][lane] = (uint16_t)el[s]; F.roles[s][lane] = 0; F.clo[s][lane] = 65535; F.chi[s][lane] = 0;
            }
        }
    }
    if (lane < 4) S.cnt[lane] = 0;
    __syncthreads();                                          // F.* written by the row owners, read by the sample lanes
    // ---- the wave's SUB rows, 4 rows x 15 sample rows per pass: roles for the classification, cells for the tile pass
    unsigned long long pending = __ballot(is_sub);
    const int g = lane / 15, sub = lane - g * 15;
    const int n_all = n;
    while (pending) {
        unsigned long long m = pending;
        int R = -1;
        for (int t = 0; t <= g && t < 4; ++t) { if (!m) { R = -1; break; } R = __ffsll((long long)m) - 1; m &= m - 1; }
        if (g >= 4) R = -1;
        int rows_in_pass = 0;
        for (int t = 0; t < 4 && pending; ++t) { pending &= pending - 1; ++rows_in_pass; }
        // cross-lane reads must run with every lane active: ds_bpermute returns 0 for a disabled source lane
        const int Rsrc = R >= 0 ? R : 0;
        const int nR = __shfl(n_all, Rsrc);
        const int rR = __shfl(r, Rsrc);
        const uint32_t riR = (uint32_t)__shfl((int)ri, Rsrc);
        if (R >= 0) {
            const int ss = rR * 15 + sub;
            int cc[ROWS_FAST_N], dd[ROWS_FAST_N];
            unsigned act = 0;
#pragma unroll
            for (int s = 0; s < ROWS_FAST_N; ++s) {
                cc[s] = 0; dd[s] = 0;
                if (s >= nmax) continue;
                if (s < nR) {
                    const DevEdge e = E[F.eid[s][R]];
                    if (e.ytop <= ss && ss < e.ybot) {
                        act |= 1u << s;
                        dd[s] = e.dir;
                        if (e.dy) { int32_t q; int64_t rm; edge_x_at(e, ss, q, rm); cc[s] = cell_of(q, rm, e.dy); } else cc[s] = e.x1;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < ROWS_FAST_N; ++j) {
                if (j >= nmax) continue;
                if (!((act >> j) & 1u)) continue;
                int wbj = 0, gsum = dd[j]; bool rep = true;
#pragma unroll
                for (int i = 0; i < ROWS_FAST_N; ++i) {
                    if (i >= nmax) continue;
                    if (i == j || !((act >> i) & 1u)) continue;
                    if (cc[i] < cc[j]) wbj += dd[i];
                    else if (cc[i] == cc[j]) { gsum += dd[i]; if (i < j) rep = false; }
                }
                if (!rep) continue;                      // one edge per group of equal cells carries the role
                const bool in_b = ((unsigned)wbj & mask) != 0, in_a = ((unsigned)(wbj + gsum) & mask) != 0;
                if (in_a != in_b) {
                    atomicOr(&F.roles[j][R], (in_a ? 1 : 2) << (2 * sub));
                    const int col = (int)clamp_col(cc[j] >> 8);
                    atomicMin(&F.clo[j][R], col);
                    atomicMax(&F.chi[j][R], col);
                    if (riR != ~0u) {
                        const uint32_t at = atomicAdd(&S.cnt[g], 1u);
                        sub_cell(&S.cells[g][at], cc[j], in_a ? 1 : -1, P.x_min, P.x_max);
                    }
                }
            }
        }
        __syncthreads();                                      // the pass's cells are staged
        // ---- allocate, copy out (coalesced), row headers
        {
            const uint32_t c0 = S.cnt[0], c1 = S.cnt[1], c2 = S.cnt[2], c3 = S.cnt[3];
            const uint32_t total = c0 + c1 + c2 + c3;
            const uint32_t base = alloc_cells(FR.counters, total, FR.cell_slice, lane);
            if (base != ~0u) {
                for (uint32_t t = (uint32_t)lane; t < total; t += 64) {
                    const int gg = t < c0 ? 0 : (t < c0 + c1 ? 1 : (t < c0 + c1 + c2 ? 2 : 3));
                    const uint32_t pre = gg == 0 ? 0u : (gg == 1 ? c0 : (gg == 2 ? c0 + c1 : c0 + c1 + c2));
                    FR.cells[base + t] = S.cells[gg][t - pre];
                }
            }
            if (sub == 0 && R >= 0 && riR != ~0u) {          // the first sample lane of each of the pass's rows writes its header
                const uint32_t pre = g == 0 ? 0u : (g == 1 ? c0 : (g == 2 ? c0 + c1 : c0 + c1 + c2));
                RowInfo2 h; h.off = base == ~0u ? 0u : base + pre; h.n = base == ~0u ? (uint16_t)0 : (uint16_t)S.cnt[g]; h.mode = (uint16_t)ROW_SUB;
                FR.rows[riR] = h;
            }
        }
        __syncthreads();                                      // staging read: reset the counters for the next pass
        if (lane < 4) S.cnt[lane] = 0;
        __syncthreads();
        (void)rows_in_pass;
    }
    __syncthreads();                                          // role bits OR-ed in by the sample lanes
    if (is_sub) {
#pragma unroll
        for (int s = 0; s < ROWS_FAST_N; ++s) {
            if (s >= nmax) continue;
            roles[s] = F.roles[s][lane]; cols[s] = (int32_t)((uint32_t)F.clo[s][lane] | ((uint32_t)F.chi[s][lane] << 16));
        }
    }
    mode_out = mode; n_out_edges = n; overflow_out = overflow; defer_out = defer;
}


__device__ __forceinline__ void rows2_chunk_body(const Frame2& FR, uint32_t block) {
    __shared__ FastLds F;
    __shared__ SubStage S;
    __shared__ DevEdge staged[ROWS_STAGE];
    const int lane = threadIdx.x;
    const ChunkInfo ck = FR.chunks[block];                               // wave-uniform: path and edge reads are scalar
    const uint32_t lo = ck.path;
    const DevPath P = FR.paths[lo];
    const int r = (int)ck.first_row + lane;
    const int chunk_rows = (int)ck.rows;                                 // 16, 32 or 64: whole tile-rows, starting on a tile-row boundary
    const uint32_t band_index = FR.band_index, band_count = FR.band_count;
    // the band entry of this lane's tile-row: where its row headers and class bytes go
    const int g16 = lane >> 4;
    const int band = (int)ck.first_row / TILE_H + g16;
    const int band_lo = P.y_min / TILE_H, band_hi = (P.y_max - 1) / TILE_H;
    const bool band_ok = ck.slot0 != ~0u && g16 < chunk_rows / TILE_H && band >= band_lo && band <= band_hi && P.kind == SWFR_PATH_TOR;
    BandSlot cls_bs = {0u, 0u, 0u, 0u};
    uint32_t cls_b0 = 0, cls_b1 = 0;
    if (band_ok) {
        cls_bs = FR.band_slots[ck.slot0 + (uint32_t)g16];
        cls_b0 = FR.band_off[band];
        cls_b1 = FR.band_off[band + 1];
    }
    const uint32_t ri = band_ok ? cls_bs.slot * TILE_H + (uint32_t)(lane & (TILE_H - 1)) : ~0u;
    const bool in_path = P.kind == SWFR_PATH_TOR && lane < chunk_rows && r >= P.y_min && r < P.y_max;
    bool live = in_path;
    if (live && band_count > 1 && (uint32_t)((r / TILE_H) % band_count) != band_index) live = false;
    int fast_limit = (int)FR.fast_limit;
    if (P.n_edges > 65535u) fast_limit = 0;                             // 16-bit local edge indices in the fast path
    // ---- stage the edges that can be active in this chunk's rows (path order kept)
    const int lo_s = (int)ck.first_row * 15, hi_s = lo_s + chunk_rows * 15;
    uint32_t n_list = 0;
    bool use_lds = true;
    for (uint32_t eb = 0; eb < P.n_edges; eb += 64) {
        const uint32_t k = eb + (uint32_t)lane;
        const DevEdge ek = FR.edges[P.first_edge + min(k, P.n_edges - 1u)];
        const bool hit = k < P.n_edges && ek.ytop < hi_s && ek.ybot > lo_s;
        const unsigned long long hb = __ballot(hit);
        const uint32_t at = n_list + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (hit && at < ROWS_STAGE) staged[at] = ek;
        n_list += (uint32_t)__popcll(hb);
        if (n_list > ROWS_STAGE) { use_lds = false; break; }
    }
    __syncthreads();
    uint32_t mode; int n, nmax = ROWS_FAST_N; bool overflow, defer;
    int32_t roles[ROWS_FAST_N], cols[ROWS_FAST_N]; int el[ROWS_FAST_N];
    int32_t Q1[ROWS_FAST_N], Q2[ROWS_FAST_N]; int64_t R1[ROWS_FAST_N], R2[ROWS_FAST_N];
    if (use_lds) rows2_fast((const DevEdge*)staged, n_list, P, r, live, fast_limit, F, S, lane, mode, n, overflow, defer, roles, cols, el, Q1, R1, Q2, R2, nmax, ri, FR);
    else rows2_fast(FR.edges + P.first_edge, P.n_edges, P, r, live, fast_limit, F, S, lane, mode, n, overflow, defer, roles, cols, el, Q1, R1, Q2, R2, nmax, ri, FR);
    const bool slow = live && (overflow || defer);
    // ---- FULL rows: cells of every boundary edge, densely packed behind one allocation of the wavefront
    int n_cells = 0;
    const bool emit = mode == ROW_FULL && ri != ~0u && !slow;
    if (emit) {
#pragma unroll
        for (int s = 0; s < ROWS_FAST_N; ++s) {
            if (s >= nmax) continue;                                     // wave-uniform
            if (s < n && roles[s] != 0) n_cells += full_span(Q1[s], Q2[s]);
        }
    }
    const uint32_t incl = (uint32_t)wave_scan_incl(n_cells);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t base = alloc_cells(FR.counters, total, FR.cell_slice, lane);
    if (ri != ~0u && lane < chunk_rows && mode != ROW_SUB) {            // (the SUB rows' headers were written with their cells)
        RowInfo2 h; h.off = 0; h.n = 0; h.mode = (uint16_t)(slow ? ROW_DEFER : mode);
        if (emit && base != ~0u) {
            uint32_t off = base + incl - (uint32_t)n_cells;
            h.off = off; h.n = (uint16_t)n_cells;
#pragma unroll
            for (int s = 0; s < ROWS_FAST_N; ++s) {
                if (s >= nmax) continue;
                if (s < n && roles[s] != 0) {
                    const int64_t edy = use_lds ? staged[el[s]].dy : FR.edges[P.first_edge + el[s]].dy;
                    full_cells(Q1[s], R1[s], Q2[s], R2[s], edy, ((uint32_t)roles[s] & 1u) ? +1 : -1, P.x_min, P.x_max, &FR.cells[off]);
                    off += (uint32_t)full_span(Q1[s], Q2[s]);
                }
            }
        }
        FR.rows[ri] = h;
    }
    // ---- rows left to the slow-row kernel
    {
        const bool q = slow && ri != ~0u;
        const unsigned long long qm = __ballot(q);
        if (qm) {
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&FR.counters[C2_SLOW], (uint32_t)__popcll(qm));
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            if (q) {
                const uint32_t at = qb + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull));
                if (at < FR.slow_cap) { SlowRow sr; sr.path = lo; sr.row = r; sr.ri = ri; sr.pad = defer ? 1u : 0u; FR.slow[at] = sr; }
                else atomicOr(&FR.counters[C2_ERROR], E2_SLOW_QUEUE);
            }
        }
    }
    // ---- classification of this chunk's (tile, path) pairs from the row summaries still in registers: lanes 16g..16g+15 are the
    //      pixel rows of tile-row g.  Only the columns of the path's rectangle are written (the rest of the class matrix was
    //      cleared when the scene was uploaded and nothing ever writes there).
    if (ck.slot0 != ~0u && P.kind == SWFR_PATH_TOR) {
        const int width = FR.width, height = FR.height;
        uint8_t* out = FR.cls;
        uint32_t n_b = 0;
        if (band_ok) {
            n_b = cls_b1 - cls_b0;
            out = FR.cls + (size_t)FR.tiles_x * cls_b0 + (cls_bs.slot - cls_b0);
        }
        const swfr_style& st = FR.styles[P.style];
        const uint32_t opq = (st.kind == SWFR_STYLE_SOLID && P.lerp && (st.pixel >> 24) == 0xffu) ? CLS_OPAQUE : 0u;
        const int tc0 = P.x_min / TILE_W, tc1 = (P.x_max - 1) / TILE_W;
        const int y = r;
        const bool in_frame = y < height && band_ok, in_rows = in_frame && in_path;
        for (int tc = tc0; tc <= tc1; ++tc) {                 // wave-uniform
            const int tx0 = tc * TILE_W, tile_x1 = min(tx0 + TILE_W, width);
            uint32_t f = 0;
            if (in_frame) {
                if (!in_rows) f = CLS_NOTFULL;
                else if (slow) f = CLS_PARTIAL | CLS_NOTFULL | CLS_NONEMPTY;      // not known yet: the general route is always right
                else {
                    int carry = 0;
                    bool inter = false;
#pragma unroll
                    for (int s2 = 0; s2 < ROWS_FAST_N; ++s2) {
                        if (s2 >= nmax) continue;                        // wave-uniform
                        if (s2 >= n || roles[s2] == 0) continue;
                        const int clo = (int)((uint32_t)cols[s2] & 0xffffu), chi = (int)((uint32_t)cols[s2] >> 16);
                        if (chi < tx0 && chi < 65535) carry += record_height((uint32_t)roles[s2]);
                        else if (clo >= tx0 + TILE_W && clo < 65535) { /* right of the tile */ }
                        else inter = true;
                    }
                    const bool inside_x = P.x_min <= tx0 && P.x_max >= tile_x1;
                    const uint32_t a = (uint32_t)((carry * 512 * 17 + 256) >> 9) & 255u;
                    if (inter) f = CLS_PARTIAL | CLS_NOTFULL | CLS_NONEMPTY;
                    else if (a == 0) f = CLS_NOTFULL | CLS_HOLE;
                    else if (a == 255 && inside_x) f = CLS_NONEMPTY;
                    else f = CLS_PARTIAL | CLS_NOTFULL | CLS_NONEMPTY;
                }
            }
            // OR over the tile-row's sixteen lanes (one DPP row): four rotations; every lane is active here
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x128, 0xf, 0xf, false);   // row_ror:8
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x124, 0xf, 0xf, false);   // row_ror:4
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x122, 0xf, 0xf, false);   // row_ror:2
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x121, 0xf, 0xf, false);   // row_ror:1
            if ((f & (CLS_HOLE | CLS_NONEMPTY)) == (CLS_HOLE | CLS_NONEMPTY)) f |= CLS_PARTIAL;
            f &= ~CLS_HOLE;
            if ((f & (CLS_PARTIAL | CLS_NOTFULL | CLS_NONEMPTY)) == CLS_NONEMPTY) f |= opq;        // a full cover that hides what lies below
            if ((lane & 15) == 0 && band_ok) out[(size_t)tc * n_b] = (uint8_t)f;
        }
    }
}

__global__ __launch_bounds__(64) void k2_rows(const Frame2* __restrict__ frames) {
    const Frame2& FR = frames[blockIdx.y];
    if (blockIdx.x >= FR.n_chunks) return;
    rows2_chunk_body(FR, blockIdx.x);
}


// ---------------------------------------------------------------------------------------------
// k2_tiles
// ---------------------------------------------------------------------------------------------
#define T2_LIST 64                     // band entries of a tile kept per round (lane = list position)
#define T2_PRE 2                       // rounds of 64 cells of a batch fetched ahead into registers

// Persistent wavefronts: each walks strips w = blockIdx.x, blockIdx.x + gridDim.x, ... of the launch order (heaviest first when
// the scene has one).  lane = pixel column; the strip's eight rows of pixels live in registers until the single store.
template <bool SHADERS>
__device__ __forceinline__ void tiles2_body(const Frame2& FR) {
    __shared__ __attribute__((aligned(16))) int acc[STRIP_H][ACC_STRIDE];   // also the queue of the compacted blend (8-byte pairs)
    __shared__ __attribute__((aligned(16))) uint32_t ent[T2_LIST][8];       // BandEntry2 as dwords
    __shared__ uint32_t sel[T2_LIST];                                       // list position -> band list index | class << 24
    __shared__ uint32_t seg_off[64], seg_start[64 + 1];                     // per (batch path, strip row): first cell, exclusive prefix of counts
    __shared__ int plist[PBATCH];

    const int lane = threadIdx.x;
    const int width = FR.width, height = FR.height, tiles_x = FR.tiles_x;
    const swfr_style* __restrict__ styles = FR.styles;
    const Sources bitmaps = FR.src;
    for (int i = lane; i < STRIP_H * ACC_STRIDE; i += 64) (&acc[0][0])[i] = 0;
    __syncthreads();

    for (uint32_t w = blockIdx.x; w < FR.n_strips; w += gridDim.x) {
        const uint32_t wg = FR.order ? FR.order[w] : w;
        const int tile = (int)(wg / STRIPS_PER_TILE), strip = (int)(wg % STRIPS_PER_TILE);
        const int tcol = tile % tiles_x;
        int trow = tile / tiles_x;
        if (FR.band_count > 1) trow = trow * (int)FR.band_count + (int)FR.band_index;
        const int tx0 = tcol * TILE_W, ty0 = trow * TILE_H + strip * STRIP_H;
        if (ty0 >= height) continue;
        const int cx = tx0 + lane;
        uint32_t px[STRIP_H];
#pragma unroll
        for (int rr = 0; rr < STRIP_H; ++rr) px[rr] = 0u;

        const uint32_t band_begin = FR.band_off[trow], band_end = FR.band_off[trow + 1];
        const uint32_t n_b = band_end - band_begin;
        const uint8_t* mycls = FR.cls + (size_t)tiles_x * band_begin + (size_t)tcol * n_b;   // this tile's class byte per band entry
        uint32_t next = 0;
        while (next < n_b) {
            // ---- bin: band entries with a non-empty class for this tile, painter's order kept (wave-local compaction)
            int ln = 0;
            while (next < n_b && ln < T2_LIST) {
                const uint32_t bi = next + lane;
                const uint32_t f = bi < n_b ? (uint32_t)mycls[bi] : 0u;
                bool hit = (f & CLS_NONEMPTY) != 0;
                unsigned long long b = __ballot(hit);
                const int room = T2_LIST - ln;
                int cnt = __popcll(b);
                if (cnt > room) {                                 // keep the first `room` hits, rescan the rest next round
                    int keep = room; unsigned long long m = b, kept = 0ull; uint32_t last = 0;
                    while (keep--) { const int bit = __ffsll((long long)m) - 1; kept |= 1ull << bit; m &= m - 1; last = (uint32_t)bit; }
                    b = kept; hit = hit && ((kept >> lane) & 1ull); cnt = room;
                    next += last + 1;
                } else next += 64;
                if (hit) sel[ln + __popcll(b & ((1ull << lane) - 1ull))] = bi | (f << 24);
                ln += cnt;
            }
            __syncthreads();                                      // sel written by other lanes
            // ---- occlusion, from the class bytes alone: everything below the last opaque full cover is invisible in this tile
            int start = 0;
            {
                const uint32_t f = lane < ln ? sel[lane] >> 24 : 0u;
                const unsigned long long b = __ballot((f & (CLS_PARTIAL | CLS_NOTFULL | CLS_BOX | CLS_OPAQUE)) == CLS_OPAQUE);
                if (b) start = 63 - __clzll((long long)b);
            }
            // ---- the surviving entries, one per lane: 32 bytes as two 16-byte loads
            if (lane >= start && lane < ln) {
                const uint4* src = reinterpret_cast<const uint4*>(&FR.band_list[band_begin + (sel[lane] & 0xffffffu)]);
                const uint4 q0 = src[0], q1 = src[1];
                *reinterpret_cast<uint4*>(&ent[lane][0]) = q0;
                *reinterpret_cast<uint4*>(&ent[lane][4]) = q1;
            }
            __syncthreads();

            // ---- painter's order walk
            int batch_n = 0, batch_i = 0;
            int total = 0;
            Cell pre[T2_PRE];                                     // cells lane, lane + 64 of the batch's flat sequence
            uint32_t pre_seg[T2_PRE];                             // ... and the (path, row) segment each belongs to
            for (int li = start; li < ln; ++li) {
                // per-entry fields are wave-uniform: readfirstlane moves them (and everything computed from them) to the scalar unit
                const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)sel[li]) >> 24;
                const uint32_t xw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][0]), yw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][1]);
                const int e_xmin = (int)(int16_t)(xw & 0xffffu), e_xmax = (int)(int16_t)(xw >> 16);
                const int e_ymin = (int)(int16_t)(yw & 0xffffu), e_ymax = (int)(int16_t)(yw >> 16);
                const uint32_t eflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][2]), solid = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][3]);
                const uint32_t style = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][4]);
                const int row_lo = max(e_ymin, ty0) - ty0, row_hi = min(min(e_ymax, ty0 + STRIP_H), height) - ty0;
                if (row_hi <= row_lo) continue;                    // the path misses this strip of the tile
                if (f & CLS_BOX) {
                    // ---- rectilinear (A.6): exact area of disjoint boxes, alpha = (c>>8) - (c>>16)
                    const uint32_t e_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][5]), e_nedges = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][6]);
#pragma unroll
                    for (int rr = 0; rr < STRIP_H; ++rr) {
                        if (rr < row_lo || rr >= row_hi) continue;         // wave-uniform
                        const int cy = ty0 + rr;
                        uint32_t cov = 0u;
                        for (uint32_t k = 0; k < e_nedges; ++k) {
                            const swfr_edge bx = FR.raw[e_first + k];
                            const int wx = min(bx.x2, (cx + 1) * 256) - max(bx.x1, cx * 256);
                            const int wy = min(bx.y2, (cy + 1) * 256) - max(bx.y1, cy * 256);
                            if (wx > 0 && wy > 0) cov += (uint32_t)(wx * wy);
                        }
                        const uint32_t a = ((cov >> 8) - (cov >> 16)) & 255u;
                        if (a) px[rr] = blend_pixel_t<SHADERS>(px[rr], a, eflags, solid, styles, style, bitmaps, cx, cy);
                    }
                } else if (f & CLS_PARTIAL) {
                    // ---- tor (A.5): the path's cells of this strip's rows
                    if (batch_i == batch_n) {
                        // the next PBATCH partial tor paths of the list, this one first (lane = list position): their row headers in
                        // one round trip, lane = (path of the batch, row of the strip), and the first cells of the flat sequence
                        bool isp = lane >= li && lane < ln && ((sel[lane] >> 24) & (CLS_PARTIAL | CLS_BOX)) == CLS_PARTIAL;
                        if (isp) {
                            const uint32_t lyw = ent[lane][1];
                            const int l_ymin = (int)(int16_t)(lyw & 0xffffu), l_ymax = (int)(int16_t)(lyw >> 16);
                            isp = min(min(l_ymax, ty0 + STRIP_H), height) > max(l_ymin, ty0);
                        }
                        const unsigned long long pm = __ballot(isp);
                        const int rank = __popcll(pm & ((1ull << lane) - 1ull));
                        if (isp && rank < PBATCH) plist[rank] = lane;
                        batch_n = min((int)__popcll(pm), PBATCH); batch_i = 0;
                        __syncthreads();                                   // plist visible
                        uint32_t my_cnt = 0, off = 0;
                        {
                            const int bp = lane / STRIP_H, row = lane % STRIP_H;
                            if (bp < batch_n && ty0 + row < height) {
                                const uint32_t bidx = band_begin + (sel[plist[bp]] & 0xffffffu);
                                const RowInfo2 ri = FR.rows[(size_t)bidx * TILE_H + (uint32_t)(strip * STRIP_H + row)];
                                off = ri.off; my_cnt = ri.n;
                                if ((uint64_t)off + my_cnt > (uint64_t)FR.cell_slice * C2_HEADS) { atomicOr(&FR.counters[C2_ERROR], E2_CELL_RANGE); my_cnt = 0; }
                            }
                        }
                        const int incl = wave_scan_incl((int)my_cnt);       // every lane active
                        total = __builtin_amdgcn_readlane(incl, 63);
                        seg_off[lane] = off;
                        seg_start[lane] = (uint32_t)(incl - (int)my_cnt);
                        seg_start[64] = (uint32_t)total;
                        __syncthreads();                                   // seg_off / seg_start visible to every lane
#pragma unroll
                        for (int u = 0; u < T2_PRE; ++u) {
                            const uint32_t g = (uint32_t)(u * 64 + lane);
                            pre[u].col = 0; pre[u].ch = 0; pre[u].ua = 0; pre_seg[u] = 0;
                            if (u * 64 < total) {                          // wave-uniform
                                int lo = 0, hi = 64;                        // last segment with seg_start <= g
                                while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (seg_start[mid] <= g) lo = mid; else hi = mid; }
                                pre_seg[u] = (uint32_t)lo;
                                if (g < (uint32_t)total) pre[u] = FR.cells[seg_off[lo] + (g - seg_start[lo])];
                            }
                        }
                    }
                    const int bp = batch_i++;
                    const int g0 = (int)__builtin_amdgcn_readfirstlane((int)seg_start[bp * STRIP_H]);
                    const int g1 = (int)__builtin_amdgcn_readfirstlane((int)seg_start[(bp + 1) * STRIP_H]);   // this path's cells [g0, g1) of the batch sequence
                    // ---- accumulate: cells left of the tile fold into the row's carry, cells right of it do not matter
                    for (int gb = g0 & ~63; gb < g1; gb += 64) {           // wave-uniform
                        const int g = gb + lane;
                        Cell c; uint32_t sg;
                        if (gb < T2_PRE * 64) { c = pre[0]; sg = pre_seg[0];
#pragma unroll
                            for (int u = 1; u < T2_PRE; ++u) if (gb == u * 64) { c = pre[u]; sg = pre_seg[u]; } }
                        else {
                            c.col = 0; c.ch = 0; c.ua = 0; sg = 0;
                            if (g >= g0 && g < g1) {
                                int lo = bp * STRIP_H, hi = (bp + 1) * STRIP_H;
                                while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (seg_start[mid] <= (uint32_t)g) lo = mid; else hi = mid; }
                                sg = (uint32_t)lo;
                                c = FR.cells[seg_off[lo] + ((uint32_t)g - seg_start[lo])];
                            }
                        }
                        if (g >= g0 && g < g1) {
                            int* arow = acc[sg % STRIP_H];
                            const int i = (int)c.col - tx0;
                            if (i < 0) atomicAdd(&arow[ACC_CARRY], (int)c.ch);
                            else if (i < TILE_W) atomicAdd(&arow[i], (int)c.ch * (1 << 20) + c.ua);
                        }
                    }
                    __syncthreads();                                       // acc complete
                    int (*A)[ACC_STRIDE] = acc;
                    // ---- prefix sum, alpha, blend; clears as it reads.  All eight rows in one straight-line block so that their LDS
                    //      round trips and DPP scan chains interleave; a row nothing was accumulated into scans zeros
                    {
                        int v[STRIP_H], carry[STRIP_H];
#pragma unroll
                        for (int u = 0; u < STRIP_H; ++u) { v[u] = A[u][lane]; carry[u] = A[u][ACC_CARRY]; }
#pragma unroll
                        for (int u = 0; u < STRIP_H; ++u) { A[u][lane] = 0; if (lane == 0) A[u][ACC_CARRY] = 0; }
                        uint32_t al[STRIP_H];
#pragma unroll
                        for (int u = 0; u < STRIP_H; ++u) {
                            const int ua = (v[u] << 12) >> 12;             // low 20 bits, sign-extended
                            int ch = (v[u] - ua) >> 20;
                            if (lane == 0) ch += carry[u];
                            const int scan = wave_scan_incl(ch);
                            const int area = scan * 512 - ua;
                            al[u] = (uint32_t)((((area << 4) + area) + 256) >> 9) & 255u;   // area * 17
                            if (cx < e_xmin || cx >= e_xmax || u < row_lo || u >= row_hi) al[u] = 0;
                        }
                        bool blended = false;
                        if (!SHADERS && (eflags & BE_LERP)) {
                            // Solid colour, SOURCE-lerp: coverage 255 takes the colour, 0 keeps the pixel, and only the few edge pixels
                            // need the two rounded products: queued -- {coverage, pixel} through the (now empty) accumulator -- and
                            // blended with lanes = queued pixels, one pass for the strip's eight rows
                            unsigned long long pmask[STRIP_H];
                            int qbase[STRIP_H], nq = 0;
#pragma unroll
                            for (int u = 0; u < STRIP_H; ++u) {
                                pmask[u] = __ballot(al[u] - 1u < 254u);
                                qbase[u] = nq;
                                nq += (int)__popcll(pmask[u]);
                                px[u] = al[u] == 255u ? solid : px[u];
                            }
                            if (nq <= BLEND_QUEUE) {                       // wave-uniform; more edge pixels: the per-row path below
                                uint2* q = reinterpret_cast<uint2*>(&A[0][0]);
#pragma unroll
                                for (int u = 0; u < STRIP_H; ++u) {
                                    const int qi = qbase[u] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pmask[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask[u], 0u));
                                    if ((pmask[u] >> lane) & 1ull) q[qi] = make_uint2(al[u], px[u]);
                                }
                                __syncthreads();
                                for (int b = lane; b < nq; b += 64) {
                                    const uint2 e = q[b];
                                    q[b].x = lerp_pixel(solid, e.x, e.y);
                                }
                                __syncthreads();
#pragma unroll
                                for (int u = 0; u < STRIP_H; ++u) {
                                    const int qi = qbase[u] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pmask[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask[u], 0u));
                                    if ((pmask[u] >> lane) & 1ull) px[u] = q[qi].x;
                                }
                                __syncthreads();
                                for (int b = lane; b < nq; b += 64) q[b] = make_uint2(0u, 0u);   // the accumulator is handed back empty
                                blended = true;
                            }
                        }
                        if (!blended) {
#pragma unroll
                            for (int u = 0; u < STRIP_H; ++u) {
                                if (SHADERS) { if (al[u]) px[u] = blend_pixel_t<SHADERS>(px[u], al[u], eflags, solid, styles, style, bitmaps, cx, ty0 + u); }
                                else { const uint32_t b = blend_pixel_t<SHADERS>(px[u], al[u], eflags, solid, styles, style, bitmaps, cx, ty0 + u); px[u] = al[u] ? b : px[u]; }
                            }
                        }
                    }
                    __syncthreads();                                       // acc cleared before the next path accumulates
                } else {
                    // full cover: every in-frame pixel of the path's rows in this tile has coverage 255
#pragma unroll
                    for (int rr = 0; rr < STRIP_H; ++rr)
                        if (rr >= row_lo && rr < row_hi) px[rr] = blend_pixel_t<SHADERS>(px[rr], 255u, eflags, solid, styles, style, bitmaps, cx, ty0 + rr);
                }
            }
            __syncthreads();                                               // ent / sel are rewritten by the next round
        }
        // ---- one store per pixel: premultiplied R,G,B,A bytes; the wave writes 256 contiguous bytes per row
        if (cx < width) {
            uint32_t* rowp = FR.fb + (size_t)ty0 * (size_t)width + cx;
#pragma unroll
            for (int rr = 0; rr < STRIP_H; ++rr) {
                if (ty0 + rr < height) {
                    const uint32_t p = px[rr];
                    *rowp = (p & 0xff00ff00u) | ((p >> 16) & 0xffu) | ((p & 0xffu) << 16);
                }
                rowp += width;
            }
        }
    }
}

__global__ __launch_bounds__(64) void k2_tiles_solid(const Frame2* __restrict__ frames) { tiles2_body<false>(frames[blockIdx.y]); }
__global__ __launch_bounds__(64) void k2_tiles_shaded(const Frame2* __restrict__ frames) { tiles2_body<true>(frames[blockIdx.y]); }

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch2_front(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t max_edges, uint32_t max_slots) {
    uint32_t n_setup = (max_edges + 255) / 256;
    const uint32_t n_b = (max_slots + 255) / 256;
    if (n_setup + n_b == 0) n_setup = 1;          // counters are still cleared
    hipLaunchKernelGGL(k2_front, dim3(n_setup + n_b, n_frames), dim3(256), 0, st, frames, n_setup);
}
void launch2_rows(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t max_chunks) {
    if (!max_chunks) return;
    hipLaunchKernelGGL(k2_rows, dim3(max_chunks, n_frames), dim3(64), 0, st, frames);
}
void launch2_tiles(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t max_strips, uint32_t grid_cap, bool any_shader) {
    if (!max_strips) return;
    const uint32_t g = max_strips < grid_cap ? max_strips : grid_cap;
    if (any_shader) hipLaunchKernelGGL(k2_tiles_shaded, dim3(g, n_frames), dim3(64), 0, st, frames);
    else hipLaunchKernelGGL(k2_tiles_solid, dim3(g, n_frames), dim3(64), 0, st, frames);
}

}  // namespace swfr
