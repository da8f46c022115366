// raster_kernels.hip -- CDNA4 (gfx950) kernels of the hot path: edge list -> RGBA8 framebuffer in HBM.
//
// Pipeline per frame (all on one stream, no host round trips):
//   k_setup     one thread per edge: Cairo "tor" edge constants (sub-row span, slope quotient/remainder)
//               [SURVEY.md A.5 make_edge]
//   k_bands     one workgroup per tile-row: painter-ordered list of the paths whose rows touch the band
//   k_rows      one workgroup per (path, 64 pixel rows), one lane per row: gathers the row's active edges,
//               decides the row mode (analytic FULL row vs 15x sub-sampled SUB row), sorts by cell and runs
//               the winding prefix to give every edge its role (span start / span end / interior).
//               Output: compact per-row records {edge, roles, column range}.   [A.5 can_do_full_row/full_row/sub_row]
//   k_rows_big  same routine with a 4x larger per-row capacity for the rows that overflowed k_rows
//   k_tiles     one 256-thread workgroup per 64x16-pixel tile: bins the band's paths to the tile, classifies
//               each (empty / fully covered / partial), culls everything under the last opaque full cover,
//               then walks the rest in painter's order: covered-height / uncovered-area per cell in LDS
//               (atomics, double buffered), wave64 DPP prefix sum along x, coverage -> 8-bit alpha, shade
//               (solid / gradient / bitmap), blend into tile-resident pixels held in registers; one coalesced
//               store per pixel.                                            [A.5 render_edge/blit, A.6, A.7]
// Integer arithmetic is exact (int64 products, double-estimated quotients with integer fix-up), so results are
// bit-identical to the CPU scan converter for solid fills.
//
// No MFMA here: there is no dense contraction on this path; the roof is HBM store bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.hpp"

// SWFR_OPAQUE(x): hides a per-lane value from the optimiser (keeps address arithmetic out of a kernel's prologue).
// (SWFR_EMU: tools/emu compiles this file as plain C++ for the lock-step emulator, a development aid.)
#ifdef SWFR_EMU
#define SWFR_OPAQUE(x) asm volatile("" : "+r"(x))
#else
#define SWFR_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

namespace swfr {

// ---------------------------------------------------------------------------------------------
// exact helpers
// ---------------------------------------------------------------------------------------------
// floor(a / b) for b > 0 with remainder in [0, b); quotient magnitude < 2^31 in all call sites.
__device__ __forceinline__ void floor_div(int64_t a, int64_t b, int64_t& q, int64_t& r) {
    q = (int64_t)floor((double)a / (double)b);
    r = a - q * b;
    while (r < 0) { --q; r += b; }
    while (r >= b) { ++q; r -= b; }
}
// Same with a precomputed reciprocal of b (|a / b| < 2^31, so the estimate is off by at most one before the fix-up).
__device__ __forceinline__ void floor_div_inv(int64_t a, int64_t b, double inv_b, int64_t& q, int64_t& r) {
    q = (int64_t)(int32_t)floor((double)a * inv_b);
    r = a - q * b;
    while (r < 0) { --q; r += b; }
    while (r >= b) { ++q; r -= b; }
}
// C (truncating) division, b > 0.
__device__ __forceinline__ void trunc_div(int64_t a, int64_t b, int64_t& q, int64_t& r) {
    floor_div(a, b, q, r);
    if (a < 0 && r != 0) { ++q; r -= b; }
}
// x of the edge at the centre of sub-row s: quo + rem/dy, rem in [0,dy)  (closed form of A.5 stepping)
__device__ __forceinline__ void edge_x_at(const DevEdge& e, int s, int32_t& quo, int64_t& rem) {
    if (e.dy == 0) { quo = e.x1; rem = 0; return; }
    const int64_t a = ((int64_t)(2 * s + 1) << 8) - 30 * (int64_t)e.y1;
    int64_t q, r;
    floor_div_inv(a * e.ex, e.dy, e.inv_dy, q, r);
    quo = e.x1 + (int32_t)q;
    rem = r;
}
__device__ __forceinline__ int cell_of(int32_t quo, int64_t rem, int64_t dy) { return quo + (rem >= dy / 2 ? 1 : 0); }
__device__ __forceinline__ void step_x(int32_t& quo, int64_t& rem, const DevEdge& e) {
    quo += (int32_t)e.dq; rem += e.dr;
    if (rem < 0) { --quo; rem += e.dy; } else if (rem >= e.dy) { ++quo; rem -= e.dy; }
}
// FULL-row end points of an edge over pixel row s0/15, stepped back from the sub-row centre to the row top
__device__ __forceinline__ void full_row_ends(const DevEdge& e, int s0, int32_t& q1, int64_t& r1, int32_t& q2, int64_t& r2) {
    edge_x_at(e, s0, q1, r1);
    edge_x_at(e, s0 + 15, q2, r2);
    if (e.dy) {
        const int32_t hq = (int32_t)(e.dq / 2); const int64_t hr = e.dr / 2;
        q1 -= hq; r1 -= hr; if (r1 < 0) { --q1; r1 += e.dy; } else if (r1 >= e.dy) { ++q1; r1 -= e.dy; }
        q2 -= hq; r2 -= hr; if (r2 < 0) { --q2; r2 += e.dy; } else if (r2 >= e.dy) { ++q2; r2 -= e.dy; }
    }
}
__device__ __forceinline__ uint32_t clamp_col(int c) { return (uint32_t)min(max(c, 0), 65535); }
// net covered height a record adds to everything right of it
__device__ __forceinline__ int record_height(uint32_t roles) {
    if (roles & REC_CELLS) return (int)(int8_t)(roles >> 8);
    if (roles & REC_FULL) return (roles & 1u) ? 15 : -15;
    return __popc(roles & 0x15555555u) - __popc(roles & 0x2aaaaaaau);
}
// self-contained record of edge e for pixel row s0/15 with the given roles / column range
__device__ __forceinline__ Rec make_record(const DevEdge& e, uint32_t eid, int s0, uint32_t roles, uint32_t cols) {
    Rec rc;
    rc.roles = roles; rc.cols = cols; rc.eid = eid; rc.dy = e.dy; rc.span = 0;
    if (roles & REC_FULL) {
        full_row_ends(e, s0, rc.q1, rc.r1, rc.q2, rc.r2);
    } else {
        const int first = max(e.ytop, s0), last = min(e.ybot, s0 + 15);
        edge_x_at(e, first, rc.q1, rc.r1);
        rc.q2 = (int32_t)e.dq; rc.r2 = e.dr;
        rc.span = (uint32_t)(first - s0) | ((uint32_t)(last - s0) << 8);
    }
    return rc;
}

__device__ __forceinline__ uint32_t pack_cell(int col_rel, int ch, int ua) {
    return (uint32_t)(col_rel & 255) | ((uint32_t)(ch & 255) << 8) | ((uint32_t)(ua & 0xffff) << 16);
}
// Cells of a FULL-row edge (A.5 render_edge) as a cell record, from the edge's exact x at the row top (q1 + r1/dy) and
// bottom (q2 + r2/dy), written straight into the record's slot in global memory (the cell index is a run-time value: a record
// built in a local struct first would live in scratch memory); false -- nothing written -- when the edge spans more than
// REC_MAX_CELLS columns
__device__ __forceinline__ bool full_cells_ends(int32_t q1, int64_t r1, int32_t q2, int64_t r2, int64_t edy, int sign, Rec* __restrict__ dst) {
    int ix1 = q1 >> 8, f1 = q1 & 255, ix2 = q2 >> 8, f2 = q2 & 255;
    if (ix2 < ix1) { int t = ix1; ix1 = ix2; ix2 = t; t = f1; f1 = f2; f2 = t; int32_t tq = q1; q1 = q2; q2 = tq; int64_t tr = r1; r1 = r2; r2 = tr; }
    const int n = ix2 - ix1 + 1;
    if (n > REC_MAX_CELLS || ix1 < 0 || ix2 > 65534) return false;
    const uint32_t cols = clamp_col(ix1) | (clamp_col(ix2) << 16);
    const uint32_t roles = REC_CELLS | (uint32_t)n | ((uint32_t)((sign * 15) & 255) << 8);
    uint32_t* cells = reinterpret_cast<uint32_t*>(&dst->q1);
    if (n == 1) {                                           // header and the only cell in one 16-byte store
        *reinterpret_cast<uint4*>(dst) = make_uint4(roles, cols, pack_cell(0, sign * 15, sign * (f1 + f2) * 15), 0u);
        return true;
    }
    const int64_t dx = (int64_t)(q2 - q1) * edy + (r2 - r1);
    const int64_t t0 = ((int64_t)((ix1 + 1) * 256 - q1) * edy - r1) * 15;
    const int64_t F = 15ll * 256 * edy;
    int64_t yq, yr, fq = 0, fr = 0;
    floor_div(t0, dx, yq, yr);
    if (n > 2) floor_div(F, dx, fq, fr);
    // first two cells with the header (every record has at least two here), the others one by one
    const int h0 = (int)yq;
    uint32_t c1;
    int y_prev = h0;
    if (n == 2) c1 = pack_cell(1, sign * (15 - y_prev), sign * (15 - y_prev) * f2);
    else {
        yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; }
        const int h = (int)yq - y_prev;
        c1 = pack_cell(1, sign * h, sign * h * 256);
        y_prev = (int)yq;
    }
    *reinterpret_cast<uint4*>(dst) = make_uint4(roles, cols, pack_cell(0, sign * h0, sign * h0 * (256 + f1)), c1);
    if (n > 2) {
#pragma unroll 1
        for (int k = 2; k < n - 1; ++k) {
            yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; }
            const int h = (int)yq - y_prev;
            cells[k] = pack_cell(k, sign * h, sign * h * 256);
            y_prev = (int)yq;
        }
        cells[n - 1] = pack_cell(n - 1, sign * (15 - y_prev), sign * (15 - y_prev) * f2);
    }
    return true;
}

// wave64 inclusive prefix sum with DPP row shifts + row broadcasts (no LDS traffic)
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// ---------------------------------------------------------------------------------------------
// k_setup
// ---------------------------------------------------------------------------------------------
// per-edge constants of a tor path's edge (A.5 make_edge): sample-row span clamped to the path, slope quotient / remainder, 1/dy
__device__ __forceinline__ DevEdge make_dev_edge(const swfr_edge& e, const DevPath& p) {
    DevEdge d;
    d.x1 = e.x1; d.y1 = e.y1; d.dir = e.dir; d.pad = 0; d.inv_dy = 0.0;
    if (p.kind != SWFR_PATH_TOR) {          // boxes are consumed raw by k_tiles
        d.ytop = d.ybot = 0; d.dy = 0; d.ex = 0; d.dq = d.dr = 0;
        return d;
    }
    int ytop = (int)((15ll * e.top + 128) >> 8), ybot = (int)((15ll * e.bottom + 128) >> 8);
    ytop = max(ytop, p.y_min * 15);
    ybot = min(ybot, p.y_max * 15);
    if (ybot <= ytop) { ytop = ybot = 0; }  // never active
    d.ytop = ytop; d.ybot = ybot;
    if (e.x1 == e.x2) {
        d.dy = 0; d.ex = 0; d.dq = d.dr = 0;
    } else {
        d.ex = (int64_t)(e.x2 - e.x1) * 256;
        d.dy = (int64_t)(e.y2 - e.y1) * 15 * 512;
        d.inv_dy = 1.0 / (double)d.dy;
        trunc_div(d.ex * 512, d.dy, d.dq, d.dr);
    }
    return d;
}
__device__ __forceinline__ void setup_block(uint32_t block, const swfr_edge* __restrict__ in, const DevPath* __restrict__ paths,
                                            DevEdge* __restrict__ out, uint32_t n_edges) {
    const uint32_t i = block * 256 + threadIdx.x;
    if (i >= n_edges) return;
    const swfr_edge e = in[i];
    out[i] = make_dev_edge(e, paths[e.reserved]);
}

// ---------------------------------------------------------------------------------------------
// k_bands: per tile-row, the paths whose pixel rows intersect it, in painter's order
// ---------------------------------------------------------------------------------------------
// everything a tile needs to bin, classify and cull a path without touching paths[] / styles[]
__device__ __forceinline__ BandEntry make_band_entry(const DevPath& P, uint32_t p, uint32_t band, uint32_t row_base_p, const swfr_style* __restrict__ styles) {
    BandEntry e;
    e.path = p;
    e.x_min = (int16_t)P.x_min; e.x_max = (int16_t)P.x_max; e.y_min = (int16_t)P.y_min; e.y_max = (int16_t)P.y_max;
    e.row_base = row_base_p;
    e.style = P.style; e.first_edge = P.first_edge; e.n_edges = P.n_edges;
    const uint32_t kind = styles[P.style].kind, pixel = styles[P.style].pixel;
    uint32_t fl = 0;
    if (P.kind == SWFR_PATH_BOXES) fl |= BE_BOXES;
    if (P.lerp) fl |= BE_LERP;
    if (kind == SWFR_STYLE_SOLID) fl |= BE_SOLID;
    if (kind == SWFR_STYLE_SOLID && P.lerp && (pixel >> 24) == 0xffu) fl |= BE_OPAQUE_COVER;
    e.flags = fl | (band << 8); e.solid = pixel;             // the tile-row index rides in the upper bits
    return e;
}
__device__ __forceinline__ void bands_block(uint32_t block, const DevPath* __restrict__ paths, const BandSlot* __restrict__ slots,
                                            uint32_t n_slots, const uint32_t* __restrict__ row_base,
                                            const swfr_style* __restrict__ styles, BandEntry* __restrict__ band_list) {
    const uint32_t g = block * 256 + threadIdx.x;
    if (g >= n_slots) return;
    const BandSlot bs = slots[g];
    const uint32_t p = bs.path;
    const BandEntry e = make_band_entry(paths[p], p, bs.band, row_base[p], styles);
    band_list[bs.slot] = e;
}

// k_front: one launch for the per-frame front end.  Blocks [0, n_setup) convert edges (one thread per edge), the rest
// write the band entries (one thread per (path, tile-row) pair, slots assigned by the host); block 0 clears the counters.
__global__ __launch_bounds__(256) void k_front(const swfr_edge* __restrict__ in, const DevPath* __restrict__ paths, DevEdge* __restrict__ out,
                                               uint32_t n_edges, uint32_t n_setup, const BandSlot* __restrict__ slots, uint32_t n_slots,
                                               const uint32_t* __restrict__ row_base, const swfr_style* __restrict__ styles,
                                               BandEntry* __restrict__ band_list, uint32_t* __restrict__ counters) {
    if (blockIdx.x == 0 && threadIdx.x < CNT_WORDS) counters[threadIdx.x] = 0;
    if (blockIdx.x < n_setup) setup_block(blockIdx.x, in, paths, out, n_edges);
    else bands_block(blockIdx.x - n_setup, paths, slots, n_slots, row_base, styles, band_list);
}

// ---------------------------------------------------------------------------------------------
// k_rows
// ---------------------------------------------------------------------------------------------
// classification of a (tile, path) pair
#define CLS_PARTIAL 1u                // some row needs the general accumulate + scan path
#define CLS_NOTFULL 2u                // some in-frame row of the tile is not uniformly alpha 255
#define CLS_NONEMPTY 4u               // some row has coverage
#define CLS_HOLE 16u                  // (inside the classifiers only) a row of the path has no coverage in this tile: with CLS_NONEMPTY
                                      // from another row the tile is partial even if no single pixel is -- e.g. the last row of a path
                                      // whose bottom lies less than a sample row below a pixel boundary
#define CLS_BOX 8u                    // rectilinear path evaluated per pixel from its boxes

#define ROWS_FAST_N 8            // active edges per row handled in registers by k_rows
#define ROWS_BIG_MAXA 64         // capacity of the generic (LDS list) routine in k_rows_big
#define ROWS_STAGE 64            // paths with at most this many edges are staged into LDS

// All edges of one path, whichever form the kernel has them in (k_front's DevEdge array, or the raw edges when the row pass
// computes the constants itself).
struct PathEdges {
    const DevEdge* dev; const swfr_edge* raw; const DevPath* P; bool from_raw;
    uint32_t* diag;          // pipeline 2: counters; the replay's capacity limits are counted there (the frame then fails loudly)
    __device__ __forceinline__ void limit_hit(uint32_t which) const { if (diag) atomicOr(&diag[which], 1u); }
    // pipeline 2: the row headers the row kernel has already written say how an earlier row of this path was converted
    const RowInfo2* rows2; const BandSlot* band_slots; uint32_t bs0;      // bs0: band_slots index of the path's first tile-row
    uint32_t* retry;         // set (and a placeholder returned) when an earlier row's header is not there yet: the row is queued again
    __device__ __forceinline__ uint32_t known_mode(int rho) const {
        if (!rows2 || rho < P->y_min || rho >= P->y_max) return ROW_DEFER;
        const BandSlot bs = band_slots[bs0 + (uint32_t)(rho / TILE_H - P->y_min / TILE_H)];
        return rows2[(size_t)bs.slot * TILE_H + (uint32_t)(rho & (TILE_H - 1))].mode;
    }
    __device__ __forceinline__ DevEdge operator()(uint32_t k) const { return from_raw ? make_dev_edge(raw[P->first_edge + k], *P) : dev[P->first_edge + k]; }
    __device__ __forceinline__ uint32_t size() const { return P->n_edges; }
    // the sample-row span alone (the part of make_dev_edge that needs no division)
    __device__ __forceinline__ void span(uint32_t k, int& ytop, int& ybot) const {
        if (!from_raw) { ytop = dev[P->first_edge + k].ytop; ybot = dev[P->first_edge + k].ybot; return; }
        const swfr_edge& e = raw[P->first_edge + k];
        ytop = max((int)((15ll * e.top + 128) >> 8), P->y_min * 15);
        ybot = min((int)((15ll * e.bottom + 128) >> 8), P->y_max * 15);
        if (ybot <= ytop) ytop = ybot = 0;
    }
};
// Two edges on one and the same line (a shape edge with fill0 == fill1 is decoded twice, once per direction: decode-swf-shape.ts:364-369):
// their x agrees at every sample row, so they add and remove the same cells whichever comes first in Cairo's list -- their mutual
// order needs no history.  (Only their order against a third edge that ties with them does.)
__device__ __forceinline__ bool same_line(const DevEdge& a, const DevEdge& b) {
    return a.x1 == b.x1 && a.y1 == b.y1 && a.ex == b.ex && a.dy == b.dy;
}
// ---- the order Cairo gives edges that become active at the same sample row m: the row's bucket holds them in path order, and
//      sort_edges -- pairs, then merges of runs of 2, 4, ... with merge_sorted_edges, whose two loops consume the lists in
//      alternating runs ("<=" on both sides: on a tie the list being consumed keeps going) -- sorts them by cell.  Restated for up
//      to sixteen such edges in registers (lists are packed 4-bit slot numbers, cells are looked up by select chains); more than
//      sixteen: path order.
#define NEW_SORT_MAX 16
__device__ __forceinline__ int sel_cell(const int (&v)[NEW_SORT_MAX], int i) {
    int r = v[0];
#pragma unroll
    for (int t = 1; t < NEW_SORT_MAX; ++t) r = (i == t) ? v[t] : r;
    return r;
}
__device__ __forceinline__ uint64_t merge_runs(uint64_t A, int na, uint64_t B, int nb, const int (&cell)[NEW_SORT_MAX]) {
    if (nb == 0) return A;
    if (na == 0) return B;
    uint64_t out = 0; int no = 0, ia = 0, ib = 0;
    auto a_slot = [&](int i) { return (int)((A >> (4 * i)) & 15ull); };
    auto b_slot = [&](int i) { return (int)((B >> (4 * i)) & 15ull); };
    bool phase_a = sel_cell(cell, a_slot(0)) <= sel_cell(cell, b_slot(0));
    for (int guard = 0; guard < 2 * NEW_SORT_MAX + 2; ++guard) {
        if (phase_a) {
            const int x = sel_cell(cell, b_slot(ib));
            while (ia < na && sel_cell(cell, a_slot(ia)) <= x) { out |= (uint64_t)a_slot(ia) << (4 * no); ++no; ++ia; }
            if (ia == na) { while (ib < nb) { out |= (uint64_t)b_slot(ib) << (4 * no); ++no; ++ib; } break; }
        }
        {
            const int x = sel_cell(cell, a_slot(ia));
            while (ib < nb && sel_cell(cell, b_slot(ib)) <= x) { out |= (uint64_t)b_slot(ib) << (4 * no); ++no; ++ib; }
            if (ib == nb) { while (ia < na) { out |= (uint64_t)a_slot(ia) << (4 * no); ++no; ++ia; } break; }
        }
        phase_a = true;
    }
    return out;
}
// does path edge ka come before kb in that order?  (both become active at sample row m)
__device__ __forceinline__ bool new_order_before(const PathEdges& PE, uint32_t ka, uint32_t kb, int m, bool path_order) {
    if (PE.rows2) return PE.dev[PE.P->first_edge + ka].pad < PE.dev[PE.P->first_edge + kb].pad;   // pipeline 2: k2_start_ranks has replayed the sort
    int cell[NEW_SORT_MAX];
#pragma unroll
    for (int t = 0; t < NEW_SORT_MAX; ++t) cell[t] = 0;
    int cnt = 0, sa = -1, sb = -1;
    const uint32_t ne = PE.size();
    for (uint32_t k = 0; k < ne; ++k) {
        int yt, yb;
        PE.span(k, yt, yb);
        if (yt != m || yb <= m) continue;
        if (cnt >= NEW_SORT_MAX) { PE.limit_hit(C2_TIE_SORT_OVERFLOW); return path_order; }
        const DevEdge e = PE(k);
        int c = e.x1;
        if (e.dy) { int32_t q; int64_t r; edge_x_at(e, m, q, r); c = cell_of(q, r, e.dy); }
#pragma unroll
        for (int t = 0; t < NEW_SORT_MAX; ++t) if (t == cnt) cell[t] = c;
        if (k == ka) sa = cnt;
        if (k == kb) sb = cnt;
        ++cnt;
    }
    if (sa < 0 || sb < 0) return path_order;
    // sort_edges on slots 0..cnt-1: pairs, then runs of 2 + 2, 4 + 4, 8 + 8
    uint64_t run[NEW_SORT_MAX / 2]; int rn[NEW_SORT_MAX / 2];
#pragma unroll
    for (int p2 = 0; p2 < NEW_SORT_MAX / 2; ++p2) {
        const int x = 2 * p2, y = 2 * p2 + 1;
        if (y < cnt) { const bool keep = sel_cell(cell, x) <= sel_cell(cell, y); run[p2] = keep ? (uint64_t)(x | (y << 4)) : (uint64_t)(y | (x << 4)); rn[p2] = 2; }
        else if (x < cnt) { run[p2] = (uint64_t)x; rn[p2] = 1; }
        else { run[p2] = 0; rn[p2] = 0; }
    }
#pragma unroll
    for (int width = 1; width < NEW_SORT_MAX / 2; width *= 2) {
#pragma unroll
        for (int p2 = 0; p2 < NEW_SORT_MAX / 2; p2 += 2 * width) {
            run[p2] = merge_runs(run[p2], rn[p2], run[p2 + width], rn[p2 + width], cell);
            rn[p2] += rn[p2 + width];
        }
    }
    const uint64_t all = run[0];
    int pa = 0, pb = 0;
    for (int i = 0; i < cnt; ++i) { const int slot = (int)((all >> (4 * i)) & 15ull); if (slot == sa) pa = i; if (slot == sb) pb = i; }
    return pa < pb;
}

// Two edges that tie at sample row m, where one of them arrives while the other is already active: the active one stays in front
// unless another edge arriving at m sorts between the active edge's predecessor and the tie (merge_sorted_edges consumes its lists
// in alternating runs; active edges that tie with it are left out of the predecessor search).  True when a goes first.
__device__ __forceinline__ bool arrival_order(const PathEdges& PE, const DevEdge& a, const DevEdge& b, uint32_t ka, uint32_t kb) {
    const bool a_active = a.ytop < b.ytop;
    const DevEdge& act = a_active ? a : b;
    const uint32_t k_act = a_active ? ka : kb, k_new = a_active ? kb : ka;
    const int m = max(a.ytop, b.ytop);
    int c = act.x1;
    if (act.dy) { int32_t q; int64_t r; edge_x_at(act, m, q, r); c = cell_of(q, r, act.dy); }
    const uint32_t ne = PE.size();
    int L = INT_MIN;
    for (uint32_t k = 0; k < ne; ++k) {
        if (k == k_act) continue;
        const DevEdge e = PE(k);
        if (!(e.ytop < m && e.ybot > m)) continue;
        int ce = e.x1;
        if (e.dy) { int32_t q; int64_t r; edge_x_at(e, m, q, r); ce = cell_of(q, r, e.dy); }
        if (ce < c) L = max(L, ce);
    }
    bool new_first = false;
    for (uint32_t k = 0; k < ne && !new_first; ++k) {
        if (k == k_new) continue;
        const DevEdge e = PE(k);
        if (!(e.ytop == m && e.ybot > m)) continue;
        int ce = e.x1;
        if (e.dy) { int32_t q; int64_t r; edge_x_at(e, m, q, r); ce = cell_of(q, r, e.dy); }
        new_first = ce >= L && ce < c;
    }
    return a_active ? !new_first : new_first;
}
// Was pixel row rho of the path converted sample row by sample row (Cairo then re-sorts its edge list at every sample row), or
// analytically (the list is looked at only at the row's first sample row)?  Sampled iff an edge becomes active after the first
// sample row, an active edge ends before the last, or two edges swap places over the row.  Edges that tie at the row's first
// sample row swap when the one in front ends up behind: their order is known when at least one of them became active at that
// sample row (the sort / merge rules above); two older edges are ordered by their own history, one level deep (DEPTH), and
// taken as not swapping beyond that.  The pair test is skipped
// when (active edges of the row) x (edges of the path) exceeds 2^21 -- thousands of edges in one row.
template <int DEPTH>
__device__ __forceinline__ bool tied_order_at(const PathEdges& PE, const DevEdge& a, const DevEdge& b, uint32_t ka, uint32_t kb, int s0, bool path_order);
template <int DEPTH>
__device__ __forceinline__ bool row_was_sampled(const PathEdges& PE, int rho) {
    {   // the row kernel has decided that row already unless it, too, was left to the slow kernel
        const uint32_t m = PE.known_mode(rho);
        if (m == ROW_SUB) return true;
        if (m == ROW_FULL || m == ROW_EMPTY) return false;
        if (m == ROW_DEFER && PE.retry && rho >= PE.P->y_min && rho < PE.P->y_max) { *PE.retry = 1u; return false; }   // that row is queued, too: next pass
    }
    const int s = rho * 15;
    const uint32_t ne = PE.size();
    uint32_t n_active = 0;
    for (uint32_t k = 0; k < ne; ++k) {
        int yt, yb;
        PE.span(k, yt, yb);
        if (yb <= s || yt >= s + 15) continue;
        if (yt > s || yb < s + 15) return true;
        ++n_active;
    }
    if ((uint64_t)n_active * ne > (1u << 21)) { PE.limit_hit(C2_TIE_PAIRTEST_SKIPPED); return false; }      // the pair test below reads n_active * ne spans
    for (uint32_t u = 0; u < ne; ++u) {
        int yt, yb;
        PE.span(u, yt, yb);
        if (yb <= s || yt >= s + 15) continue;
        const DevEdge eu = PE(u);
        int u0 = eu.x1, u1 = eu.x1;
        if (eu.dy) { int32_t q; int64_t r; edge_x_at(eu, s, q, r); u0 = cell_of(q, r, eu.dy); edge_x_at(eu, s + 15, q, r); u1 = cell_of(q, r, eu.dy); }
        for (uint32_t v = u + 1; v < ne; ++v) {
            PE.span(v, yt, yb);
            if (yb <= s || yt >= s + 15) continue;
            const DevEdge ev = PE(v);
            int v0 = ev.x1, v1 = ev.x1;
            if (ev.dy) { int32_t q; int64_t r; edge_x_at(ev, s, q, r); v0 = cell_of(q, r, ev.dy); edge_x_at(ev, s + 15, q, r); v1 = cell_of(q, r, ev.dy); }
            if ((u0 < v0 && u1 > v1) || (u0 > v0 && u1 < v1)) return true;
            if (u0 == v0 && u1 != v1) {
                bool u_first;
                if (eu.ytop == s || ev.ytop == s) u_first = eu.ytop == ev.ytop ? new_order_before(PE, u, v, s, true) : arrival_order(PE, eu, ev, u, v);
                else if constexpr (DEPTH > 0) u_first = tied_order_at<DEPTH - 1>(PE, eu, ev, u, v, s, true);   // two older edges: their history
                else { if (!same_line(eu, ev)) PE.limit_hit(C2_TIE_DEPTH); continue; }
                if (u_first ? u1 > v1 : v1 > u1) return true;
            }
        }
    }
    return false;
}
// Order of two active edges a, b whose cells coincide at the first sample row s0 of a pixel row (near-parallel edges leaving a
// common vertex: round joins and caps produce them).  Cairo's list is re-sorted whenever it is looked at and a cell order is
// violated, and left alone on ties: a sorts first iff it had the smaller cell the last time the list was looked at while the two
// differed -- every sample row of a sampled pixel row, the first sample row only of an analytically converted one -- and if they
// never differed since the later one became active, the one that became active earlier, else path order.
template <int DEPTH>
__device__ __forceinline__ bool tied_order_at(const PathEdges& PE, const DevEdge& a, const DevEdge& b, uint32_t ka, uint32_t kb, int s0, bool path_order) {
    const int lo = max(a.ytop, b.ytop);
    auto differ = [&](int s, bool& a_first) {
        int ca = a.x1, cb = b.x1;
        if (a.dy) { int32_t q; int64_t r; edge_x_at(a, s, q, r); ca = cell_of(q, r, a.dy); }
        if (b.dy) { int32_t q; int64_t r; edge_x_at(b, s, q, r); cb = cell_of(q, r, b.dy); }
        a_first = ca < cb;
        return ca != cb;
    };
    bool af = path_order;
    for (int rho = s0 / 15 - 1; rho * 15 + 14 >= lo; --rho) {
        const int rs = rho * 15;
        if (row_was_sampled<DEPTH>(PE, rho)) {
            for (int s = rs + 14; s >= max(rs, lo); --s) if (differ(s, af)) return af;
        } else if (rs >= lo && differ(rs, af)) return af;
    }
    if (a.ytop != b.ytop) return arrival_order(PE, a, b, ka, kb);
    return new_order_before(PE, ka, kb, a.ytop, path_order);
}
// one level of history behind the history: whether an earlier row was sampled may itself hinge on a tie of two older edges
__device__ __forceinline__ bool tied_order(const PathEdges& PE, const DevEdge& a, const DevEdge& b, uint32_t ka, uint32_t kb, int s0, bool path_order) {
    return tied_order_at<1>(PE, a, b, ka, kb, s0, path_order);
}

#ifdef SWFR_PHASES                 // -DSWFR_PHASES: clocks per phase of a k_rows wavefront, summed into counters[8..15] (diagnostic builds only)
#define RPHASE(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); rph[i] += (uint32_t)(now_ - rph_t); rph_t = now_; } while (0)
#else
#define RPHASE(i) do { } while (0)
#endif

struct FastLds {
    uint16_t eid[ROWS_FAST_N][64];      // per row (lane): local indices of its active edges
    int32_t roles[ROWS_FAST_N][64];     // SUB rows: role bits, OR-ed in by the 15 sub-row lanes
    int32_t clo[ROWS_FAST_N][64], chi[ROWS_FAST_N][64];
    uint16_t flag[ROWS_FAST_N][64];     // slow_full_row: in = new-in-row | (dir + 1) << 1, out |= first << 3 | last << 4 | (winding before + 32) << 5
};

// The FULL test of one row (lane) in loop form, for rows with coincident active edges (see tied_order): keys are read from the
// lane's own column of F, the per-edge results go back into F.flag.  Same decisions as the unrolled test in fast_rows otherwise.
__device__ __forceinline__ bool slow_full_row(FastLds& F, const PathEdges& PE, const DevEdge* E, const uint16_t* sid, const uint16_t* shi, int lane, int n, int s0) {
    auto path_index = [&](int local) { return sid ? ((uint32_t)sid[local] | ((uint32_t)shi[local] << 16)) : (uint32_t)local; };
    bool full = true;
    // a new edge that ties with an active one goes first when another new edge sorts between the active edge's predecessor and the
    // tie (Cairo merges the sorted new edges into the active list in alternating runs): bit k of nfmask says so for active edge k
    unsigned nfmask = 0;
    for (int k = 0; k < n; ++k) {
        const int ck = F.roles[k][lane], pk = F.chi[k][lane];
        if ((int)F.flag[k][lane] & 1) continue;
        int L = INT_MIN; bool tied_before = false, any_new = false;
        for (int a = 0; a < n; ++a) {
            if (a == k || ((int)F.flag[a][lane] & 1)) continue;
            const int ca = F.roles[a][lane], pa = F.chi[a][lane];
            if (ca < ck) L = max(L, ca);
            else if (ca == ck && (pa < pk || (pa == pk && a < k))) tied_before = true;
        }
        for (int b = 0; b < n; ++b) {
            if (!((int)F.flag[b][lane] & 1)) continue;
            const int cb = F.roles[b][lane];
            any_new |= cb >= L && cb < ck;
        }
        if (!tied_before && any_new) nfmask |= 1u << k;
    }
    for (int j = 0; j < n; ++j) {
        const int cj = F.roles[j][lane], ej = F.clo[j][lane], pj = F.chi[j][lane], fj = (int)F.flag[j][lane], nwj = fj & 1;
        int w = 0; bool fg = true, lg = true;
        for (int i = 0; i < n; ++i) {
            if (i == j) continue;
            const int ci = F.roles[i][lane], ei = F.clo[i][lane], pi = F.chi[i][lane], fi = (int)F.flag[i][lane], nwi = fi & 1, di = ((fi >> 1) & 3) - 1;
            const bool tie = ci == cj, tie2 = nwi == nwj;
            bool first = i < j;
            if (tie && nwi == nwj) {
                const DevEdge ea = E[F.eid[i][lane]], eb = E[F.eid[j][lane]];
                if (same_line(ea, eb)) first = i < j;
                else if (nwi == 0) first = tied_order(PE, ea, eb, path_index(F.eid[i][lane]), path_index(F.eid[j][lane]), s0, i < j);
                else first = new_order_before(PE, path_index(F.eid[i][lane]), path_index(F.eid[j][lane]), s0, i < j);
            }
            const bool t3 = first;
            const bool t_mixed = nwi == 0 ? !((nfmask >> i) & 1u) : ((nfmask >> j) & 1u) != 0;
            const bool before = ci < cj || (tie && (tie2 ? t3 : t_mixed));
            if (before) { w += di; if (ei > ej) full = false; if (tie) fg = false; }
            else if (tie) lg = false;
        }
        F.flag[j][lane] = (uint16_t)((fj & 7) | (fg ? 8 : 0) | (lg ? 16 : 0) | ((w + 32) << 5));
    }
    return full;
}

// Register-resident row routine for rows with at most ROWS_FAST_N active edges.
//   phase A (lane = row): gather the active edges, FULL/SUB decision by pairwise order tests, FULL roles from
//                         winding prefix sums (no sort: every test is a sum over "edge i sorts before edge j")
//   phase B (lane = (SUB row, sub-row)): four SUB rows x 15 sample rows per pass; closed-form x per sample,
//                         open/close role per group of equal cells, OR-ed into the row's role words in LDS
template <class EPTR>
__device__ __forceinline__ void fast_rows(EPTR E, uint32_t n_list, const DevPath& P, int r, bool live, int fast_limit, FastLds& F, int lane,
                                          uint32_t& mode_out, int& n_out_edges, bool& overflow_out,
                                          int32_t (&roles)[ROWS_FAST_N], int32_t (&cols)[ROWS_FAST_N], int (&el)[ROWS_FAST_N],
                                          int32_t (&Q1)[ROWS_FAST_N], int64_t (&R1)[ROWS_FAST_N], int32_t (&Q2)[ROWS_FAST_N], int64_t (&R2)[ROWS_FAST_N],
                                          uint32_t* rph, unsigned long long& rph_t, int& nmax_out, const PathEdges& PE,
                                          const uint16_t* sid, const uint16_t* shi) {
    (void)rph; (void)rph_t;
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
    RPHASE(2);
    // wave-uniform bound on the active edges of any row of this wave: the unrolled slot loops stop there
    const int nmax = __ballot(n > 6) ? 8 : __ballot(n > 4) ? 6 : __ballot(n > 2) ? 4 : 2;
    nmax_out = nmax;
    // ---- rows that can still be FULL: x of every active edge at the first sample row of this pixel row and of the next
    //      (one reciprocal multiply + integer fix-up each); kept as the exact row-top / row-bottom end points for the records
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
                if (e.ytop < s0) {                        // cell one sample row earlier (tie-break of the sorted list)
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
    if (overflow) n = 0;
    RPHASE(3);
    uint32_t mode = ROW_EMPTY;
    bool is_sub = false;
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
                if (j >= nmax) continue;                  // wave-uniform: no row of this wave has that many edges
                int w = 0; bool fg = true, lg = true;
#pragma unroll
                for (int i = 0; i < ROWS_FAST_N; ++i) {
                    if (i == j || i >= nmax) continue;
                    const bool valid = i < n && j < n;
                    // does edge i sort before edge j?  (cell, active-before-new, previous cell, path order)
                    const bool tie = cs[i] == cs[j];
                    const bool tie2 = nw[i] == nw[j];
                    const bool cpeq = cp[i] == cp[j];
                    const bool t3 = nw[i] == 0 ? (cp[i] < cp[j] || (cpeq && i < j)) : (i < j);
                    const bool before = cs[i] < cs[j] || (tie && (nw[i] < nw[j] || (tie2 && t3)));
                    deep |= valid && tie;                         // coincident edges: their order is settled below (rare)
                    if (valid && before) { w += dr[i]; if (ce[i] > ce[j]) full = false; if (tie) fg = false; }
                    if (valid && !before && tie) lg = false;
                }
                wb[j] = w;
                if (fg) firstg |= 1u << j;
                if (lg) lastg |= 1u << j;
            }
        }
        // rows in which two active edges coincide at this sample row and the one before: the whole test again, in loop form,
        // with the order of such pairs taken from where they last differed (tied_order); the keys travel through the lane's own
        // LDS column.  Wave-uniform and rare: near-parallel edges leaving a common vertex (round joins and caps produce them).
        if (__ballot(deep && !mid_row) != 0ull) {
#pragma unroll
            for (int s = 0; s < ROWS_FAST_N; ++s) {
                F.roles[s][lane] = cs[s]; F.clo[s][lane] = ce[s]; F.chi[s][lane] = cp[s]; F.eid[s][lane] = (uint16_t)el[s];
                F.flag[s][lane] = (uint16_t)(nw[s] | ((dr[s] + 1) << 1));
            }
            if (deep && !mid_row) {
                // ties between two edges on one and the same line need no history (see same_line): when every tie of the row is
                // of that kind the unrolled test above already has the answer
                bool real = false;
                for (int i = 0; i < n && !real; ++i)
                    for (int j = i + 1; j < n && !real; ++j)
                        if (F.roles[i][lane] == F.roles[j][lane]) { const DevEdge ea = E[F.eid[i][lane]], eb = E[F.eid[j][lane]]; real = !same_line(ea, eb); }
                deep = real;
            }
            if (deep && !mid_row) {
                full = slow_full_row(F, PE, (const DevEdge*)E, sid, shi, lane, n, s0);
                firstg = lastg = 0;
#pragma unroll
                for (int j = 0; j < ROWS_FAST_N; ++j) {
                    const int f = (int)F.flag[j][lane];
                    wb[j] = ((f >> 5) & 63) - 32;
                    if (f & 8) firstg |= 1u << j;
                    if (f & 16) lastg |= 1u << j;
                }
            }
        }
        if (full) {
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
    __syncthreads();                                          // F.* written by the row owners, read by the sample lanes
    RPHASE(4);
    // ---- phase B: the wave's SUB rows, 4 rows x 15 sub-rows per pass
    unsigned long long pending = __ballot(is_sub);
    const int g = lane / 15, sub = lane - g * 15;
    const int n_all = n;
    while (pending) {
        // the g-th pending row of this pass
        unsigned long long m = pending;
        int R = -1;
        for (int t = 0; t <= g && t < 4; ++t) { if (!m) { R = -1; break; } R = __ffsll((long long)m) - 1; m &= m - 1; }
        if (g >= 4) R = -1;
        // consume up to four rows
        for (int t = 0; t < 4 && pending; ++t) pending &= pending - 1;
        // cross-lane reads must run with every lane active: ds_bpermute returns 0 for a disabled source lane
        const int Rsrc = R >= 0 ? R : 0;
        const int nR = __shfl(n_all, Rsrc);
        const int rR = __shfl(r, Rsrc);
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
                }
            }
        }
    }
    __syncthreads();                                          // role bits OR-ed in by the sample lanes
    if (is_sub) {
#pragma unroll
        for (int s = 0; s < ROWS_FAST_N; ++s) {
            if (s >= nmax) continue;                          // (roles of the slots beyond stay 0)
            roles[s] = F.roles[s][lane]; cols[s] = (int32_t)((uint32_t)F.clo[s][lane] | ((uint32_t)F.chi[s][lane] << 16));
        }
    }
    RPHASE(5);
    mode_out = mode; n_out_edges = n; overflow_out = overflow;
}

__device__ __forceinline__ void rows_chunk_body(uint32_t block, const DevEdge* __restrict__ edges, const DevPath* __restrict__ paths,
                                                const uint32_t* __restrict__ row_base, const ChunkInfo* __restrict__ chunks,
                                                uint32_t n_paths, RowInfo* __restrict__ rows, Rec* __restrict__ records,
                                                uint32_t band_index, uint32_t band_count, int fast_limit, int cell_mode,
                                                const BandSlot* __restrict__ band_slots, const uint32_t* __restrict__ band_off,
                                                uint8_t* __restrict__ cls_t, int width, int height, int fused,
                                                const swfr_edge* __restrict__ raw, const swfr_style* __restrict__ styles,
                                                BandEntry* __restrict__ band_list, uint32_t* __restrict__ counters) {
    (void)counters;
    uint32_t rph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long rph_t = 0;
#ifdef SWFR_PHASES
    rph_t = __builtin_amdgcn_s_memtime();
#endif
    __shared__ FastLds F;
    __shared__ DevEdge staged[ROWS_STAGE];
    __shared__ uint16_t staged_id[ROWS_STAGE], staged_hi[ROWS_STAGE];   // path-relative index of a staged edge (diagnostic eid)
    const int lane = threadIdx.x;
    // workgroup -> (path, 64 rows): one wave-uniform descriptor load, so path and edge reads are scalar
    const ChunkInfo ck = chunks[block];
    const uint32_t lo = ck.path;
    (void)n_paths;
    const DevPath P = paths[lo];
    const int r = (int)ck.first_row + lane;
    const int chunk_rows = (int)ck.rows;                                // 64, or fewer for scenes of a few tall paths
    // the classification at the end needs where this chunk's tile-rows keep their class bytes: asked for now, so that the answer
    // is not queued behind the record stores (loads and stores return in order on this part)
    BandSlot cls_bs = {0u, 0u, 0u, 0u};
    uint32_t cls_b0 = 0, cls_b1 = 0;
    if ((fused & 1) && ck.slot0 != ~0u && (lane >> 4) < chunk_rows / TILE_H && ((int)ck.first_row / TILE_H + (lane >> 4)) * TILE_H < height) {
        const int band = (int)ck.first_row / TILE_H + (lane >> 4);
        cls_bs = band_slots[ck.slot0 + (uint32_t)(lane >> 4)];
        cls_b0 = band_off[band];
        cls_b1 = band_off[band + 1];
    }
    const bool in_path = P.kind == SWFR_PATH_TOR && lane < chunk_rows && r >= P.y_min && r < P.y_max;   // chunks start on tile-row boundaries
    bool live = in_path;
    if (live && band_count > 1 && (uint32_t)((r / TILE_H) % band_count) != band_index) live = false;
    const uint32_t t = row_base[lo] + (uint32_t)(r - P.y_min);          // row task index (valid when in_path)
    RPHASE(0);
    if (P.n_edges > 65535u) fast_limit = 0;                             // 16-bit local edge indices in the fast path
    // ---- stage the edges that can be active in this chunk's 64 rows (path order kept): the row loops then run over that
    //      short list in LDS instead of over every edge of the path; only if more than ROWS_STAGE overlap do they read L2
    const int lo_s = (int)ck.first_row * 15, hi_s = lo_s + chunk_rows * 15;
    uint32_t n_list = 0;
    bool use_lds = true;
    for (uint32_t eb = 0; eb < P.n_edges; eb += 64) {
        const uint32_t k = eb + (uint32_t)lane;
        DevEdge ek;
        bool hit = false;
        if (k < P.n_edges) {
            // (fused front end: the edge constants are computed here from the raw edge instead of being read from k_front's array)
            ek = (fused & 2) ? make_dev_edge(raw[P.first_edge + k], P) : edges[P.first_edge + k];
            hit = ek.ytop < hi_s && ek.ybot > lo_s;
        }
        const unsigned long long hb = __ballot(hit);
        const uint32_t at = n_list + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (hit && at < ROWS_STAGE) { staged[at] = ek; staged_id[at] = (uint16_t)(k & 0xffffu); staged_hi[at] = (uint16_t)(k >> 16); }
        n_list += (uint32_t)__popcll(hb);
        if (n_list > ROWS_STAGE) { use_lds = false; break; }
    }
    __syncthreads();
    RPHASE(1);
    const PathEdges PE = {edges, raw, &P, (fused & 2) != 0, nullptr, nullptr, nullptr, 0u, nullptr};
    uint32_t mode; int n, nmax = ROWS_FAST_N; bool overflow;
    int32_t roles[ROWS_FAST_N], cols[ROWS_FAST_N]; int el[ROWS_FAST_N];
    int32_t Q1[ROWS_FAST_N], Q2[ROWS_FAST_N]; int64_t R1[ROWS_FAST_N], R2[ROWS_FAST_N];
    if (use_lds) fast_rows((const DevEdge*)staged, n_list, P, r, live, fast_limit, F, lane, mode, n, overflow, roles, cols, el, Q1, R1, Q2, R2, rph, rph_t, nmax, PE, (const uint16_t*)staged_id, (const uint16_t*)staged_hi);
    else fast_rows(edges + P.first_edge, P.n_edges, P, r, live, fast_limit, F, lane, mode, n, overflow, roles, cols, el, Q1, R1, Q2, R2, rph, rph_t, nmax, PE, (const uint16_t*)nullptr, (const uint16_t*)nullptr);
    uint32_t n_out = 0;
#pragma unroll
    for (int s = 0; s < ROWS_FAST_N; ++s) n_out += (s < nmax && s < n && roles[s] != 0) ? 1u : 0u;
    // ---- record slots: the chunk owns [rec_base, rec_base + bound); lanes take consecutive pieces (no atomics)
    const uint32_t incl = (uint32_t)wave_scan_incl((int)n_out);
    const uint32_t base = ck.rec_base;
    if (in_path && !overflow) {
        uint32_t off = base + incl - n_out;
        RowInfo ri; ri.rec_off = off; ri.n_rec = (uint16_t)n_out; ri.mode = (uint16_t)mode;
        rows[t] = ri;
#pragma unroll
        for (int s = 0; s < ROWS_FAST_N; ++s) {
            if (s >= nmax) continue;                                     // wave-uniform: no row of this wave has that many edges
            if (s < n && roles[s] != 0) {
                const uint32_t eidx = use_lds ? ((uint32_t)staged_id[el[s]] | ((uint32_t)staged_hi[el[s]] << 16)) : (uint32_t)el[s];
                if ((uint32_t)roles[s] & REC_FULL) {     // end points are already known: cells, or the generic FULL record
                    const int64_t edy = use_lds ? staged[el[s]].dy : edges[P.first_edge + el[s]].dy;
                    bool as_cells = false;
                    if (cell_mode & 1) as_cells = full_cells_ends(Q1[s], R1[s], Q2[s], R2[s], edy, ((uint32_t)roles[s] & 1u) ? +1 : -1, &records[off]);
                    if (!as_cells) {
                        Rec rc;
                        rc.roles = (uint32_t)roles[s]; rc.cols = (uint32_t)cols[s]; rc.eid = P.first_edge + eidx; rc.dy = edy; rc.span = 0;
                        rc.q1 = Q1[s]; rc.r1 = R1[s]; rc.q2 = Q2[s]; rc.r2 = R2[s];
                        records[off] = rc;
                    }
                } else {
                    const DevEdge e = use_lds ? staged[el[s]] : edges[P.first_edge + el[s]];
                    records[off] = make_record(e, P.first_edge + eidx, r * 15, (uint32_t)roles[s], (uint32_t)cols[s]);
                }
                ++off;
            }
        }
    }
    RPHASE(6);
    // ---- classification of this chunk's (tile, path) pairs (what k_class does, from the record headers still in registers):
    //      a chunk holds whole tile-rows of its path, lanes 16g..16g+15 are the pixel rows of tile-row g
    if ((fused & 2) && ck.slot0 != ~0u && lane < chunk_rows / TILE_H) {
        // fused front end: this chunk writes the band entries of its tile-rows (what k_front's band blocks do)
        const int band = (int)ck.first_row / TILE_H + lane;
        if (band >= P.y_min / TILE_H && band <= (P.y_max - 1) / TILE_H) {
            const BandSlot bs = band_slots[ck.slot0 + (uint32_t)lane];
            band_list[bs.slot] = make_band_entry(P, lo, (uint32_t)band, row_base[lo], styles);
        }
    }
    if ((fused & 1) && ck.slot0 != ~0u && P.kind == SWFR_PATH_TOR) {
        const int tiles_x = (width + TILE_W - 1) / TILE_W;
        const int g = lane >> 4;
        const int band = (int)ck.first_row / TILE_H + g;
        const int band_lo = P.y_min / TILE_H, band_hi = (P.y_max - 1) / TILE_H;
        const bool band_ok = g < chunk_rows / TILE_H && band >= band_lo && band <= band_hi;
        uint8_t* out = cls_t;                                 // + tile column * n_b
        uint32_t n_b = 0;
        if (band_ok) {
            n_b = cls_b1 - cls_b0;
            out = cls_t + (size_t)tiles_x * cls_b0 + (cls_bs.slot - cls_b0);
        }
        const int tc0 = P.x_min / TILE_W, tc1 = (P.x_max - 1) / TILE_W;
        // columns the path's rectangle does not reach: empty (sixteen lanes share a tile-row and stride over the columns)
        if (band_ok)
            for (int tc = lane & 15; tc < tiles_x; tc += 16)
                if (tc < tc0 || tc > tc1) out[(size_t)tc * n_b] = 0;
        const int y = r;
        const bool in_frame = y < height && band_ok, in_rows = in_frame && in_path;
        for (int tc = tc0; tc <= tc1; ++tc) {                 // wave-uniform
            const int tx0 = tc * TILE_W, tile_x1 = min(tx0 + TILE_W, width);
            uint32_t f = 0;
            if (in_frame) {
                if (!in_rows) f = CLS_NOTFULL;
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
            // OR over the tile-row's sixteen lanes (one DPP row): four rotations, no LDS round trips; every lane is active here
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x128, 0xf, 0xf, false);   // row_ror:8
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x124, 0xf, 0xf, false);   // row_ror:4
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x122, 0xf, 0xf, false);   // row_ror:2
            f |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x121, 0xf, 0xf, false);   // row_ror:1
            if ((f & (CLS_HOLE | CLS_NONEMPTY)) == (CLS_HOLE | CLS_NONEMPTY)) f |= CLS_PARTIAL;
            f &= ~CLS_HOLE;
            if ((lane & 15) == 0 && band_ok) out[(size_t)tc * n_b] = (uint8_t)f;
        }
    }
    RPHASE(7);
#ifdef SWFR_PHASES
    if (lane == 0 && (block & 63u) == 0u) {                                      // a sample of the wavefronts: contended atomics are slow
        for (int i = 0; i < 8; ++i) atomicAdd(&counters[8 + i], rph[i] >> 4);   // units of 16 clocks
        atomicAdd(&counters[16], 1u);
    }
#endif
}

// k_rows for scenes made of a few tall paths (the host then cuts the paths into 8-row chunks): lane = (row of the chunk, slot
// of that row's active-edge list), so the per-edge evaluation is one step instead of a loop over slots and the "edge i sorts
// before edge j" sums are eight shuffles inside the row's 8-lane group.  Same decisions and records as k_rows.
template <class EPTR>
__device__ __forceinline__ void rows_by_slot(EPTR E, uint32_t n_list, const uint32_t* staged_k, const DevPath& P, const ChunkInfo& ck,
                                             const uint32_t* __restrict__ row_base, RowInfo* __restrict__ rows, Rec* __restrict__ records,
                                             uint32_t band_index, uint32_t band_count, int fast_limit, int cell_mode, int lane, const PathEdges& PE) {
    const int g = lane >> 3, slot = lane & 7, gbase = lane & ~7;
    const int r = (int)ck.first_row + g, s0 = r * 15;
    const bool in_path = P.kind == SWFR_PATH_TOR && g < (int)ck.rows && r < P.y_max;
    bool live = in_path;
    if (live && band_count > 1 && (uint32_t)((r / TILE_H) % band_count) != band_index) live = false;
    const uint32_t t = row_base[ck.path] + (uint32_t)(r - P.y_min);
    const unsigned mask = P.fill_rule ? 1u : ~0u;
    // ---- gather: the eight lanes of a row walk the list together; lane `slot` keeps the slot-th active edge
    int n = 0, my_k = 0;
    bool mid_row = false, overflow = false;
    if (live) {
        for (uint32_t k = 0; k < n_list; ++k) {
            const int ytop = E[k].ytop, ybot = E[k].ybot;
            if (ybot <= s0 || ytop >= s0 + 15) continue;
            if (n >= fast_limit) { overflow = true; break; }
            mid_row |= (ytop > s0) | (ybot < s0 + 15);
            if (n == slot) my_k = (int)k;
            ++n;
        }
    }
    if (overflow) n = 0;
    const bool mine = slot < n;
    const DevEdge e = E[n_list ? (uint32_t)my_k : 0u];
    const bool slanted = e.dy != 0;
    const bool cand = n > 0 && !mid_row;
    int c0 = e.x1, c1 = e.x1, cpv = e.x1;
    int32_t q1 = e.x1, q2 = e.x1; int64_t r1 = 0, r2 = 0;
    if (mine && cand && slanted) {
        int32_t qa, qb; int64_t ra, rb;
        edge_x_at(e, s0, qa, ra);
        edge_x_at(e, s0 + 15, qb, rb);
        c0 = cell_of(qa, ra, e.dy);
        c1 = cell_of(qb, rb, e.dy);
        cpv = c0;
        if (e.ytop < s0) {
            int32_t q = qa - (int32_t)e.dq; int64_t rm = ra - e.dr;
            if (rm < 0) { --q; rm += e.dy; } else if (rm >= e.dy) { ++q; rm -= e.dy; }
            cpv = cell_of(q, rm, e.dy);
        }
        const int32_t hq = (int32_t)(e.dq / 2); const int64_t hr = e.dr / 2;
        qa -= hq; ra -= hr; if (ra < 0) { --qa; ra += e.dy; } else if (ra >= e.dy) { ++qa; ra -= e.dy; }
        qb -= hq; rb -= hr; if (rb < 0) { --qb; rb += e.dy; } else if (rb >= e.dy) { ++qb; rb -= e.dy; }
        q1 = qa; r1 = ra; q2 = qb; r2 = rb;
    }
    const int nw = (e.ytop == s0) ? 1 : 0, dr = e.dir;
    // ---- FULL test: keys of the other slots of this row by shuffles (every lane takes part: bpermute reads 0 from idle lanes)
    // (a new edge that ties with an active one: Cairo merges the sorted new edges into the active list in alternating runs, and
    //  the new edge goes first when another new edge sorts between the active edge's predecessor and the tie -- `nf` below is that
    //  predicate of an active edge; it stays false, "active first", unless such a tie exists in the wave)
    int w = 0; bool fg = true, lg = true, ok = true, mixed = false;
    bool nf = false;
    for (int pass = 0; pass < 2; ++pass) {                      // second pass only when a new edge ties with an active one (rare)
        w = 0; fg = lg = ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int src = gbase | i;
            const int ci = __shfl(c0, src), ei = __shfl(c1, src), pi = __shfl(cpv, src), ni = __shfl(nw, src), di = __shfl(dr, src);
            const int nfi = __shfl((int)nf, src);
            // two edges already active whose cells coincide here and one sample row earlier (near-parallel edges leaving a common
            // vertex): their list order is the cell order of the last sample row where they differed, else the order of insertion
            const bool deep = mine && cand && i < n && i != slot && ci == c0 && ni == nw;
            bool deep_first = i < slot;
            if (__ballot(deep) != 0ull) {                       // wave-uniform, rare
                const int ki = __shfl(my_k, src);
                if (deep) {
                    const uint32_t pk_i = staged_k ? staged_k[ki] : (uint32_t)ki, pk_me = staged_k ? staged_k[my_k] : (uint32_t)my_k;
                    const DevEdge eo = E[ki];
                    if (same_line(eo, e)) deep_first = i < slot;
                    else if (nw == 0) deep_first = tied_order(PE, eo, e, pk_i, pk_me, s0, i < slot);
                    else deep_first = new_order_before(PE, pk_i, pk_me, s0, i < slot);
                }
            }
            if (!(mine && cand) || i >= n || i == slot) continue;
            const bool tie = ci == c0, tie2 = ni == nw;
            const bool t3 = deep_first;
            const bool t_mixed = ni == 0 ? !nfi : nf;           // active / new tie: the active one first unless its `nf` holds
            const bool before = ci < c0 || (tie && (tie2 ? t3 : t_mixed));
            mixed |= tie && !tie2;
            if (before) { w += di; if (ei > c1) ok = false; if (tie) fg = false; }
            else if (tie) lg = false;
        }
        if (pass == 1 || __ballot(mixed) == 0ull) break;
        // nf of this lane's edge, if it is active: no active edge ties with it and sorts before it, and some new edge has its
        // cell in [L, c0) where L is the largest cell of the active edges left of it
        int L = INT_MIN; bool tied_before = false, any_new = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int src = gbase | i;
            const int ci = __shfl(c0, src), pi = __shfl(cpv, src), ni = __shfl(nw, src);
            if (i >= n || i == slot || ni != 0) continue;
            if (ci < c0) L = max(L, ci);
            else if (ci == c0 && (pi < cpv || (pi == cpv && i < slot))) tied_before = true;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int src = gbase | i;
            const int ci = __shfl(c0, src), ni = __shfl(nw, src);
            if (i >= n || i == slot || ni != 1) continue;
            any_new |= ci >= L && ci < c0;
        }
        nf = mine && cand && nw == 0 && !tied_before && any_new;
    }
    const unsigned long long bad = __ballot(mine && cand && !ok);
    const bool full = cand && ((bad >> gbase) & 0xffull) == 0ull;
    uint32_t role = 0, cols = 0;
    if (full && mine) {
        const bool in_b = ((unsigned)w & mask) != 0, in_a = ((unsigned)(w + dr) & mask) != 0;
        if (!in_b && fg) role = REC_FULL | 1u;
        else if (!in_a && lg) role = REC_FULL | 2u;
        if (role) { const int a = q1 >> 8, b = q2 >> 8; cols = clamp_col(min(a, b)) | (clamp_col(max(a, b)) << 16); }
    }
    // ---- SUB rows: fifteen sample rows, ranked inside the row's lane group
    const bool is_sub = n > 0 && !full;
    if (__ballot(is_sub) != 0ull) {
        int clo = 65535, chi = 0;
        for (int sub = 0; sub < 15; ++sub) {
            const int ss = s0 + sub;
            const bool act = mine && is_sub && e.ytop <= ss && ss < e.ybot;
            int cc = e.x1;
            if (act && slanted) { int32_t q; int64_t rm; edge_x_at(e, ss, q, rm); cc = cell_of(q, rm, e.dy); }
            const int dd = act ? e.dir : 0;
            const unsigned long long am = __ballot(act);
            int wb = 0, gsum = dd; bool rep = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int src = gbase | i;
                const int ci = __shfl(cc, src), di = __shfl(dd, src);
                if (!act || i == slot || !((am >> src) & 1ull)) continue;
                if (ci < cc) wb += di;
                else if (ci == cc) { gsum += di; if (i < slot) rep = false; }
            }
            if (act && rep) {
                const bool in_b = ((unsigned)wb & mask) != 0, in_a = ((unsigned)(wb + gsum) & mask) != 0;
                if (in_a != in_b) {
                    role |= (uint32_t)(in_a ? 1 : 2) << (2 * sub);
                    const int col = (int)clamp_col(cc >> 8);
                    clo = min(clo, col); chi = max(chi, col);
                }
            }
        }
        if (is_sub) cols = (uint32_t)clo | ((uint32_t)chi << 16);
    }
    // ---- records in (row, slot) order: the chunk owns consecutive slots from rec_base
    const bool has = mine && role != 0 && in_path && !overflow;
    const unsigned long long hm = __ballot(has);
    if (has) {
        const uint32_t off = ck.rec_base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        const uint32_t eidx = staged_k ? staged_k[my_k] : (uint32_t)my_k;
        if (role & REC_FULL) {
            bool as_cells = false;
            if (cell_mode & 1) as_cells = full_cells_ends(q1, r1, q2, r2, e.dy, (role & 1u) ? +1 : -1, &records[off]);
            if (!as_cells) {
                Rec rc;
                rc.roles = role; rc.cols = cols; rc.eid = P.first_edge + eidx; rc.dy = e.dy; rc.span = 0;
                rc.q1 = q1; rc.r1 = r1; rc.q2 = q2; rc.r2 = r2;
                records[off] = rc;
            }
        } else records[off] = make_record(e, P.first_edge + eidx, s0, role, cols);
    }
    if (slot == 0 && in_path && !overflow) {
        RowInfo ri;
        ri.rec_off = ck.rec_base + (uint32_t)__popcll(hm & ((1ull << gbase) - 1ull));
        ri.n_rec = (uint16_t)__popcll((hm >> gbase) & 0xffull);
        ri.mode = (uint16_t)(n == 0 ? ROW_EMPTY : (full ? ROW_FULL : ROW_SUB));
        rows[t] = ri;
    }
}

__device__ __forceinline__ void rows_rs_body(uint32_t block, const DevEdge* __restrict__ edges, const DevPath* __restrict__ paths,
                                             const uint32_t* __restrict__ row_base, const ChunkInfo* __restrict__ chunks,
                                             RowInfo* __restrict__ rows, Rec* __restrict__ records,
                                             uint32_t band_index, uint32_t band_count, int fast_limit, int cell_mode) {
    __shared__ DevEdge staged[ROWS_STAGE];
    __shared__ uint32_t staged_k[ROWS_STAGE];
    const int lane = threadIdx.x;
    const ChunkInfo ck = chunks[block];
    const DevPath P = paths[ck.path];
    if (P.n_edges > 65535u) fast_limit = 0;                  // same rule as k_rows: the host lists those rows for k_rows_big
    const int lo_s = (int)ck.first_row * 15, hi_s = lo_s + (int)ck.rows * 15;
    uint32_t n_list = 0;
    bool use_lds = true;
    for (uint32_t eb = 0; eb < P.n_edges; eb += 64) {
        const uint32_t k = eb + (uint32_t)lane;
        // (read unconditionally from a clamped index: a struct assigned under a branch ends up in scratch memory)
        const DevEdge ek = edges[P.first_edge + min(k, P.n_edges - 1u)];
        const bool hit = k < P.n_edges && ek.ytop < hi_s && ek.ybot > lo_s;
        const unsigned long long hb = __ballot(hit);
        const uint32_t at = n_list + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        if (hit && at < ROWS_STAGE) { staged[at] = ek; staged_k[at] = k; }
        n_list += (uint32_t)__popcll(hb);
        if (n_list > ROWS_STAGE) { use_lds = false; break; }
    }
    __syncthreads();
    const PathEdges PE = {edges, nullptr, &P, false, nullptr, nullptr, nullptr, 0u, nullptr};
    if (use_lds) rows_by_slot((const DevEdge*)staged, n_list, (const uint32_t*)staged_k, P, ck, row_base, rows, records, band_index, band_count, fast_limit, cell_mode, lane, PE);
    else rows_by_slot(edges + P.first_edge, P.n_edges, (const uint32_t*)nullptr, P, ck, row_base, rows, records, band_index, band_count, fast_limit, cell_mode, lane, PE);
}

// Rows with more than ROWS_FAST_N active edges of one path (the host lists them at upload, with their record slots):
// one wavefront per row, lane = active edge (up to 64).  Same decisions as fast_rows, but the "edge i sorts before edge j"
// sums run over lanes with v_readlane broadcasts instead of over register slots, and the fifteen sample rows are a loop.
__device__ __forceinline__ void big_row_body(uint32_t block, const DevEdge* __restrict__ edges, const DevPath* __restrict__ paths,
                                             const uint32_t* __restrict__ row_base, const BigRow* __restrict__ big_rows, uint32_t n_big,
                                             RowInfo* __restrict__ rows, Rec* __restrict__ records, uint32_t* __restrict__ counters,
                                             int cell_mode) {
    __shared__ uint32_t active[ROWS_BIG_MAXA];
    const int lane = threadIdx.x;
    if (block >= n_big) return;
    const BigRow br = big_rows[block];
    const DevPath P = paths[br.path];
    const int r = br.row, s0 = r * 15;
    const uint32_t t = row_base[br.path] + (uint32_t)(r - P.y_min);
    const unsigned mask = P.fill_rule ? 1u : ~0u;
    const DevEdge* E = edges + P.first_edge;
    const PathEdges PE = {edges, nullptr, &P, false, nullptr, nullptr, nullptr, 0u, nullptr};
    // ---- gather: compact the indices of the active edges, 64 candidates per pass (path order is kept)
    int n = 0;
    bool too_many = false;
    for (uint32_t base = 0; base < P.n_edges; base += 64) {
        const uint32_t k = base + (uint32_t)lane;
        bool act = false;
        if (k < P.n_edges) { const int ytop = E[k].ytop, ybot = E[k].ybot; act = !(ybot <= s0 || ytop >= s0 + 15); }
        const unsigned long long b = __ballot(act);
        const int at = n + __popcll(b & ((1ull << lane) - 1ull));
        if (act && at < ROWS_BIG_MAXA) active[at] = k;
        n += __popcll(b);
        if (n > ROWS_BIG_MAXA) { too_many = true; break; }
    }
    __syncthreads();
    RowInfo ri; ri.rec_off = br.rec_base; ri.n_rec = 0; ri.mode = ROW_EMPTY;
    if (too_many) {
        if (lane == 0) { atomicOr(&counters[CNT_ERROR], 1u); rows[t] = ri; }
        return;
    }
    const bool mine = lane < n;
    const uint32_t k_mine = mine ? active[lane] : active[0];
    const DevEdge e = E[n ? k_mine : 0];                      // lanes past the list compute on a valid edge and are ignored
    const bool slanted = e.dy != 0;
    const bool mid_row = __ballot(mine && ((e.ytop > s0) | (e.ybot < s0 + 15))) != 0ull;
    uint32_t role = 0, cols = 0;
    int32_t q1 = e.x1, q2 = e.x1; int64_t r1 = 0, r2 = 0;
    bool full = !mid_row && n > 0;
    if (full) {
        // ---- FULL candidate: keys at the first sample row of this pixel row and of the next; exact row-top / bottom end points
        int c0 = e.x1, c1 = e.x1, cpv = e.x1;
        if (slanted) {
            int32_t qa, qb; int64_t ra, rb;
            edge_x_at(e, s0, qa, ra);
            edge_x_at(e, s0 + 15, qb, rb);
            c0 = cell_of(qa, ra, e.dy);
            c1 = cell_of(qb, rb, e.dy);
            cpv = c0;
            if (e.ytop < s0) {
                int32_t q = qa - (int32_t)e.dq; int64_t rm = ra - e.dr;
                if (rm < 0) { --q; rm += e.dy; } else if (rm >= e.dy) { ++q; rm -= e.dy; }
                cpv = cell_of(q, rm, e.dy);
            }
            const int32_t hq = (int32_t)(e.dq / 2); const int64_t hr = e.dr / 2;
            qa -= hq; ra -= hr; if (ra < 0) { --qa; ra += e.dy; } else if (ra >= e.dy) { ++qa; ra -= e.dy; }
            qb -= hq; rb -= hr; if (rb < 0) { --qb; rb += e.dy; } else if (rb >= e.dy) { ++qb; rb -= e.dy; }
            q1 = qa; r1 = ra; q2 = qb; r2 = rb;
        }
        const int nw = (e.ytop == s0) ? 1 : 0, dr = e.dir;
        int w = 0; bool fg = true, lg = true, ok = true, mixed = false;
        unsigned long long nfb = 0ull;                       // bit k: active edge k lets a tying new edge go first (see rows_by_slot)
        for (int pass = 0; pass < 2; ++pass) {
            w = 0; fg = lg = ok = true;
            for (int i = 0; i < n; ++i) {                     // wave-uniform loop: lane i's keys broadcast to every lane
                const int ci = __builtin_amdgcn_readlane(c0, i), ei = __builtin_amdgcn_readlane(c1, i), pi = __builtin_amdgcn_readlane(cpv, i);
                const int ni = __builtin_amdgcn_readlane(nw, i), di = __builtin_amdgcn_readlane(dr, i);
                if (i == lane) continue;
                // does edge i sort before this lane's edge?  (cell, active / new, previous cell, path order)
                const bool tie = ci == c0, tie2 = ni == nw;
                bool deep_first = i < lane;
                if (mine && tie && ni == nw) {             // coincident edges: see tied_order (rare)
                    const DevEdge eo = E[active[i]];
                    if (same_line(eo, e)) deep_first = i < lane;
                    else if (nw == 0) deep_first = tied_order(PE, eo, e, active[i], k_mine, s0, i < lane);
                    else deep_first = new_order_before(PE, active[i], k_mine, s0, i < lane);
                }
                const bool t3 = deep_first;
                const bool t_mixed = ni == 0 ? !((nfb >> i) & 1ull) : ((nfb >> lane) & 1ull) != 0ull;
                const bool before = ci < c0 || (tie && (tie2 ? t3 : t_mixed));
                mixed |= mine && tie && !tie2;
                if (before) { w += di; if (ei > c1) ok = false; if (tie) fg = false; }
                else if (tie) lg = false;
            }
            if (pass == 1 || __ballot(mixed) == 0ull) break;
            int L = INT_MIN; bool tied_before = false, any_new = false;
            for (int i = 0; i < n; ++i) {
                const int ci = __builtin_amdgcn_readlane(c0, i), pi = __builtin_amdgcn_readlane(cpv, i), ni = __builtin_amdgcn_readlane(nw, i);
                if (i == lane || ni != 0) continue;
                if (ci < c0) L = max(L, ci);
                else if (ci == c0 && (pi < cpv || (pi == cpv && i < lane))) tied_before = true;
            }
            for (int i = 0; i < n; ++i) {
                const int ci = __builtin_amdgcn_readlane(c0, i), ni = __builtin_amdgcn_readlane(nw, i);
                if (i == lane || ni != 1) continue;
                any_new |= ci >= L && ci < c0;
            }
            nfb = __ballot(mine && nw == 0 && !tied_before && any_new);
        }
        full = __ballot(mine && !ok) == 0ull;
        if (full && mine) {
            const bool in_b = ((unsigned)w & mask) != 0, in_a = ((unsigned)(w + dr) & mask) != 0;
            if (!in_b && fg) role = REC_FULL | 1u;            // left edge of a span
            else if (!in_a && lg) role = REC_FULL | 2u;       // right edge
            if (role) { const int a = q1 >> 8, b = q2 >> 8; cols = clamp_col(min(a, b)) | (clamp_col(max(a, b)) << 16); }
        }
    }
    uint32_t mode = n == 0 ? ROW_EMPTY : (full ? ROW_FULL : ROW_SUB);
    if (n > 0 && !full) {
        // ---- SUB row: fifteen sample rows.  With at most 16 (32) active edges the wavefront works on four (two) sample rows at
        //      a time: lane = (sample row group g, edge je); the cells of a sample row are ranked inside its lane group by shuffles
        const int W = n <= 16 ? 16 : (n <= 32 ? 32 : 64), G = 64 / W;      // wave-uniform
        const int g = lane / W, je = lane - g * W;
        const bool mine_s = je < n;
        const DevEdge es = E[active[mine_s ? je : 0]];
        const bool slanted_s = es.dy != 0;
        int clo = 65535, chi = 0;
        for (int p = 0; p * G < 15; ++p) {
            const int sub = p * G + g, ss = s0 + sub;
            const bool act = mine_s && sub < 15 && es.ytop <= ss && ss < es.ybot;
            int cc = es.x1;
            if (act && slanted_s) { int32_t q; int64_t rm; edge_x_at(es, ss, q, rm); cc = cell_of(q, rm, es.dy); }
            const int dd = act ? es.dir : 0;                            // +-1 on an active edge, 0 otherwise
            int wb = 0, gsum = dd; bool rep = true;
            for (int i = 0; i < n; ++i) {                               // wave-uniform; every lane takes part in the shuffles
                const int src = g * W + i;
                const int ci = __shfl(cc, src), di = __shfl(dd, src);
                if (!act || i == je || di == 0) continue;
                if (ci < cc) wb += di;
                else if (ci == cc) { gsum += di; if (i < je) rep = false; }
            }
            if (act && rep) {
                const bool in_b = ((unsigned)wb & mask) != 0, in_a = ((unsigned)(wb + gsum) & mask) != 0;
                if (in_a != in_b) {
                    role |= (uint32_t)(in_a ? 1 : 2) << (2 * sub);
                    const int col = (int)clamp_col(cc >> 8);
                    clo = min(clo, col); chi = max(chi, col);
                }
            }
        }
        // the sample rows of an edge were spread over the lane groups: OR / min / max them back together (lane je of group 0 == lane je)
        for (int off = W; off < 64; off <<= 1) {
            role |= (uint32_t)__shfl_xor((int)role, off);
            clo = min(clo, __shfl_xor(clo, off));
            chi = max(chi, __shfl_xor(chi, off));
        }
        cols = (uint32_t)clo | ((uint32_t)chi << 16);
    }
    // ---- records: one per edge that carries a role, in path order
    const bool has = mine && role != 0;
    const unsigned long long hm = __ballot(has);
    if (has) {
        const uint32_t off = br.rec_base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        if (role & REC_FULL) {
            bool as_cells = false;
            if (cell_mode & 1) as_cells = full_cells_ends(q1, r1, q2, r2, e.dy, (role & 1u) ? +1 : -1, &records[off]);
            if (!as_cells) {
                Rec rc;
                rc.roles = role; rc.cols = cols; rc.eid = P.first_edge + k_mine; rc.dy = e.dy; rc.span = 0;
                rc.q1 = q1; rc.r1 = r1; rc.q2 = q2; rc.r2 = r2;
                records[off] = rc;
            }
        } else records[off] = make_record(e, P.first_edge + k_mine, s0, role, cols);
    }
    if (lane == 0) { ri.n_rec = (uint16_t)__popcll(hm); ri.mode = (uint16_t)mode; rows[t] = ri; }
}

// Rows with more than ROWS_BIG_MAXA (64) and up to ROWS_HUGE_MAXA (2048) active edges of one path (a line of text outlines
// filled with one style, hatching): one 256-thread workgroup per row; thread t owns the active edges t, t + 256, ... (path
// order), the sort keys of all of them sit in LDS and every owned edge is ranked against them.  Same decisions as big_row_body;
// launched only when the host listed such rows.
#define ROWS_HUGE_MAXA 2048
#define ROWS_HUGE_EPT (ROWS_HUGE_MAXA / 256)
// x of edge e at the top and bottom of pixel row s0/15 (exact end points of a FULL record) and the sort keys of the row
__device__ __forceinline__ void huge_full_keys(const DevEdge& e, int s0, int& c0, int& c1, int& cpv, int32_t& q1, int64_t& r1, int32_t& q2, int64_t& r2) {
    c0 = c1 = cpv = e.x1; q1 = q2 = e.x1; r1 = r2 = 0;
    if (!e.dy) return;
    int32_t qa, qb; int64_t ra, rb;
    edge_x_at(e, s0, qa, ra);
    edge_x_at(e, s0 + 15, qb, rb);
    c0 = cell_of(qa, ra, e.dy);
    c1 = cell_of(qb, rb, e.dy);
    cpv = c0;
    if (e.ytop < s0) {
        int32_t q = qa - (int32_t)e.dq; int64_t rm = ra - e.dr;
        if (rm < 0) { --q; rm += e.dy; } else if (rm >= e.dy) { ++q; rm -= e.dy; }
        cpv = cell_of(q, rm, e.dy);
    }
    const int32_t hq = (int32_t)(e.dq / 2); const int64_t hr = e.dr / 2;
    qa -= hq; ra -= hr; if (ra < 0) { --qa; ra += e.dy; } else if (ra >= e.dy) { ++qa; ra -= e.dy; }
    qb -= hq; rb -= hr; if (rb < 0) { --qb; rb += e.dy; } else if (rb >= e.dy) { ++qb; rb -= e.dy; }
    q1 = qa; r1 = ra; q2 = qb; r2 = rb;
}
__global__ __launch_bounds__(256) void k_rows_huge(const DevEdge* __restrict__ edges, const DevPath* __restrict__ paths,
                                                   const uint32_t* __restrict__ row_base, const BigRow* __restrict__ huge_rows, uint32_t n_huge,
                                                   RowInfo* __restrict__ rows, Rec* __restrict__ records, uint32_t* __restrict__ counters,
                                                   int cell_mode) {
    __shared__ uint32_t active[ROWS_HUGE_MAXA];
    // FULL test: cell at the row top / bottom / one sample row earlier and (new-in-this-row | direction);
    // sample rows: k_a = cell, k_d = direction (0 = inactive there)
    __shared__ int k_a[ROWS_HUGE_MAXA], k_b[ROWS_HUGE_MAXA], k_c[ROWS_HUGE_MAXA], k_d[ROWS_HUGE_MAXA];
    __shared__ uint32_t wave_cnt[4];
    __shared__ int flags;                                            // bit 0: some edge starts / ends inside the row, bit 1: FULL test failed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x >= n_huge) return;
    const BigRow br = huge_rows[blockIdx.x];
    const DevPath P = paths[br.path];
    const int r = br.row, s0 = r * 15;
    const uint32_t t = row_base[br.path] + (uint32_t)(r - P.y_min);
    const unsigned mask = P.fill_rule ? 1u : ~0u;
    const DevEdge* E = edges + P.first_edge;
    const PathEdges PE = {edges, nullptr, &P, false, nullptr, nullptr, nullptr, 0u, nullptr};
    if (tid == 0) flags = 0;
    // ---- gather the active edges in path order: ballot per wavefront, wavefront totals through LDS
    int n = 0;
    for (uint32_t base = 0; base < P.n_edges; base += 256) {
        const uint32_t k = base + (uint32_t)tid;
        bool act = false;
        if (k < P.n_edges) { const int ytop = E[k].ytop, ybot = E[k].ybot; act = !(ybot <= s0 || ytop >= s0 + 15); }
        const unsigned long long b = __ballot(act);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        int at = n;
        for (int w = 0; w < wave; ++w) at += (int)wave_cnt[w];
        at += __popcll(b & ((1ull << lane) - 1ull));
        if (act && at < ROWS_HUGE_MAXA) active[at] = k;
        n += (int)(wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
        __syncthreads();
    }
    RowInfo ri; ri.rec_off = br.rec_base; ri.n_rec = 0; ri.mode = ROW_EMPTY;
    if (n > ROWS_HUGE_MAXA) {                                        // workgroup-uniform
        if (tid == 0) { atomicOr(&counters[CNT_ERROR], 1u); rows[t] = ri; }
        return;
    }
    const int nb = (n + 255) >> 8;                                   // owned edges per thread (workgroup-uniform)
    uint32_t role[ROWS_HUGE_EPT], cols[ROWS_HUGE_EPT];
#pragma unroll
    for (int m = 0; m < ROWS_HUGE_EPT; ++m) { role[m] = 0; cols[m] = 0; }
    for (int j = tid; j < n; j += 256) {
        const DevEdge e = E[active[j]];
        if ((e.ytop > s0) | (e.ybot < s0 + 15)) atomicOr(&flags, 1);
    }
    __syncthreads();
    bool full = (flags & 1) == 0;
    if (full) {
        for (int j = tid; j < n; j += 256) {
            const DevEdge e = E[active[j]];
            int c0, c1, cpv; int32_t q1, q2; int64_t r1, r2;
            huge_full_keys(e, s0, c0, c1, cpv, q1, r1, q2, r2);
            k_a[j] = c0; k_b[j] = c1; k_c[j] = cpv; k_d[j] = ((e.ytop == s0) ? 4 : 0) | (e.dir + 1);
        }
        __syncthreads();
        // (k_d bit 3, set between the two passes: this active edge lets a tying new edge go first -- see rows_by_slot; the second
        //  pass runs only when a new edge ties with an active one, flags bit 2)
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
                const int j = m * 256 + tid;
                role[m] = 0;
                if (m >= nb || j >= n) continue;
                const int c0 = k_a[j], c1 = k_b[j], cpv = k_c[j], nw = (k_d[j] >> 2) & 1, dr = (k_d[j] & 3) - 1, nfj = (k_d[j] >> 3) & 1;
                int w = 0; bool fg = true, lg = true, ok = true, mixed = false;
                for (int i = 0; i < n; ++i) {                        // LDS broadcast reads
                    if (i == j) continue;
                    const int ci = k_a[i], ei = k_b[i], pi = k_c[i], ni = (k_d[i] >> 2) & 1, di = (k_d[i] & 3) - 1, nfi = (k_d[i] >> 3) & 1;
                    const bool tie = ci == c0, tie2 = ni == nw;
                    bool deep_first = i < j;
                    if (tie && ni == nw) {              // coincident edges: see tied_order (rare)
                        const DevEdge ea = E[active[i]], eb = E[active[j]];
                        if (same_line(ea, eb)) deep_first = i < j;
                        else if (nw == 0) deep_first = tied_order(PE, ea, eb, active[i], active[j], s0, i < j);
                        else deep_first = new_order_before(PE, active[i], active[j], s0, i < j);
                    }
                    const bool t3 = deep_first;
                    const bool t_mixed = ni == 0 ? !nfi : nfj != 0;
                    const bool before = ci < c0 || (tie && (tie2 ? t3 : t_mixed));
                    mixed |= tie && !tie2;
                    if (before) { w += di; if (ei > c1) ok = false; if (tie) fg = false; }
                    else if (tie) lg = false;
                }
                if (!ok) atomicOr(&flags, 2);
                if (mixed) atomicOr(&flags, 4);
                const bool in_b = ((unsigned)w & mask) != 0, in_a = ((unsigned)(w + dr) & mask) != 0;
                if (!in_b && fg) role[m] = REC_FULL | 1u;
                else if (!in_a && lg) role[m] = REC_FULL | 2u;
            }
            __syncthreads();
            if (pass == 1 || !(flags & 4)) break;                   // workgroup-uniform
            __syncthreads();
            if (tid == 0) flags &= ~2;                               // the order test is repeated with the final order
            unsigned nfbits = 0;
#pragma unroll
            for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
                const int j = m * 256 + tid;
                if (m >= nb || j >= n || ((k_d[j] >> 2) & 1)) continue;
                const int c0 = k_a[j], cpv = k_c[j];
                int L = INT_MIN; bool tied_before = false, any_new = false;
                for (int i = 0; i < n; ++i) {
                    if (i == j || ((k_d[i] >> 2) & 1)) continue;
                    const int ci = k_a[i], pi = k_c[i];
                    if (ci < c0) L = max(L, ci);
                    else if (ci == c0 && (pi < cpv || (pi == cpv && i < j))) tied_before = true;
                }
                for (int i = 0; i < n; ++i) {
                    if (!((k_d[i] >> 2) & 1)) continue;
                    const int ci = k_a[i];
                    any_new |= ci >= L && ci < c0;
                }
                if (!tied_before && any_new) nfbits |= 1u << m;
            }
            __syncthreads();                                         // every thread has read the keys it needs
#pragma unroll
            for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
                const int j = m * 256 + tid;
                if (m < nb && j < n && ((nfbits >> m) & 1u)) k_d[j] |= 8;
            }
            __syncthreads();
        }
        __syncthreads();
        full = (flags & 2) == 0;
    }
    const uint32_t mode = full ? ROW_FULL : ROW_SUB;
    if (!full) {
        int clo[ROWS_HUGE_EPT], chi[ROWS_HUGE_EPT];
#pragma unroll
        for (int m = 0; m < ROWS_HUGE_EPT; ++m) { role[m] = 0; clo[m] = 65535; chi[m] = 0; }
        for (int sub = 0; sub < 15; ++sub) {
            const int ss = s0 + sub;
            for (int j = tid; j < n; j += 256) {
                const DevEdge e = E[active[j]];
                const bool act = e.ytop <= ss && ss < e.ybot;
                int cc = e.x1;
                if (act && e.dy) { int32_t q; int64_t rm; edge_x_at(e, ss, q, rm); cc = cell_of(q, rm, e.dy); }
                k_a[j] = cc; k_d[j] = act ? e.dir : 0;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
                const int j = m * 256 + tid;
                if (m >= nb || j >= n) continue;
                const int dd = k_d[j], cc = k_a[j];
                if (dd == 0) continue;
                int wb = 0, gsum = dd; bool rep = true;
                for (int i = 0; i < n; ++i) {
                    const int di = k_d[i];
                    if (i == j || di == 0) continue;             // dir is +-1 for an active edge
                    const int ci = k_a[i];
                    if (ci < cc) wb += di;
                    else if (ci == cc) { gsum += di; if (i < j) rep = false; }
                }
                if (rep) {
                    const bool in_b = ((unsigned)wb & mask) != 0, in_a = ((unsigned)(wb + gsum) & mask) != 0;
                    if (in_a != in_b) {
                        role[m] |= (uint32_t)(in_a ? 1 : 2) << (2 * sub);
                        const int col = (int)clamp_col(cc >> 8);
                        clo[m] = min(clo[m], col); chi[m] = max(chi[m], col);
                    }
                }
            }
            __syncthreads();                                         // k_a / k_d are rewritten for the next sample row
        }
#pragma unroll
        for (int m = 0; m < ROWS_HUGE_EPT; ++m) cols[m] = (uint32_t)clo[m] | ((uint32_t)chi[m] << 16);
    }
    // ---- records in path order: rank among the edges that carry a role (blocks of 256 owned edges in turn)
    uint32_t emitted = 0;
#pragma unroll
    for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
        if (m >= nb) continue;                                       // workgroup-uniform
        const int j = m * 256 + tid;
        const bool has = j < n && role[m] != 0;
        const unsigned long long hm = __ballot(has);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(hm);
        __syncthreads();
        if (has) {
            uint32_t off = br.rec_base + emitted + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) off += wave_cnt[w];
            const uint32_t k_mine = active[j];
            const DevEdge e = E[k_mine];
            if (role[m] & REC_FULL) {
                int c0, c1, cpv; int32_t q1, q2; int64_t r1, r2;
                huge_full_keys(e, s0, c0, c1, cpv, q1, r1, q2, r2);
                const int a = q1 >> 8, b = q2 >> 8;
                const uint32_t fcols = clamp_col(min(a, b)) | (clamp_col(max(a, b)) << 16);
                bool as_cells = false;
                if (cell_mode & 1) as_cells = full_cells_ends(q1, r1, q2, r2, e.dy, (role[m] & 1u) ? +1 : -1, &records[off]);
                if (!as_cells) {
                    Rec rc;
                    rc.roles = role[m]; rc.cols = fcols; rc.eid = P.first_edge + k_mine; rc.dy = e.dy; rc.span = 0;
                    rc.q1 = q1; rc.r1 = r1; rc.q2 = q2; rc.r2 = r2;
                    records[off] = rc;
                }
            } else records[off] = make_record(e, P.first_edge + k_mine, s0, role[m], cols[m]);
        }
        emitted += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();                                             // wave_cnt is rewritten by the next block
    }
    if (tid == 0) { ri.n_rec = (uint16_t)emitted; ri.mode = (uint16_t)mode; rows[t] = ri; }
}

// The row pass is one launch: the first n_big workgroups take the crowded rows (the longest wavefronts start first), the rest
// take the chunks -- lane = row for crowded scenes (k_rows), lane = (row, slot) for scenes of a few tall paths (k_rows_rs).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void k_rows(const DevEdge* __restrict__ edges, const DevPath* __restrict__ paths,
                                             const uint32_t* __restrict__ row_base, const ChunkInfo* __restrict__ chunks,
                                             uint32_t n_paths, RowInfo* __restrict__ rows, Rec* __restrict__ records,
                                             uint32_t band_index, uint32_t band_count, int fast_limit, int cell_mode,
                                             const BigRow* __restrict__ big_rows, uint32_t n_big, uint32_t* __restrict__ counters,
                                             const BandSlot* __restrict__ band_slots, const uint32_t* __restrict__ band_off,
                                             uint8_t* __restrict__ cls_t, int width, int height, int fused,
                                             const swfr_edge* __restrict__ raw, const swfr_style* __restrict__ styles,
                                             BandEntry* __restrict__ band_list) {
    if (blockIdx.x < n_big) big_row_body(blockIdx.x, edges, paths, row_base, big_rows, n_big, rows, records, counters, cell_mode);
    else rows_chunk_body(blockIdx.x - n_big, edges, paths, row_base, chunks, n_paths, rows, records, band_index, band_count, fast_limit, cell_mode,
                         band_slots, band_off, cls_t, width, height, fused, raw, styles, band_list, counters);
}
__global__ __launch_bounds__(64) void k_rows_rs(const DevEdge* __restrict__ edges, const DevPath* __restrict__ paths,
                                                const uint32_t* __restrict__ row_base, const ChunkInfo* __restrict__ chunks,
                                                RowInfo* __restrict__ rows, Rec* __restrict__ records,
                                                uint32_t band_index, uint32_t band_count, int fast_limit, int cell_mode,
                                                const BigRow* __restrict__ big_rows, uint32_t n_big, uint32_t* __restrict__ counters) {
    if (blockIdx.x < n_big) big_row_body(blockIdx.x, edges, paths, row_base, big_rows, n_big, rows, records, counters, cell_mode);
    else rows_rs_body(blockIdx.x - n_big, edges, paths, row_base, chunks, rows, records, band_index, band_count, fast_limit, cell_mode);
}

// ---------------------------------------------------------------------------------------------
// k_tiles helpers: blending (A.7), shading
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mul8x2_7f(uint32_t a, uint32_t b) {
    uint32_t t = (a & 0xff00ffu) * b + 0x7f007fu;
    return ((t + ((t >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
}
__device__ __forceinline__ uint32_t add8x2_sat(uint32_t a, uint32_t b) {
    uint32_t t = a + b;
    t |= 0x1000100u - ((t >> 8) & 0xff00ffu);
    return t & 0xff00ffu;
}
// Cairo's span lerp for SOURCE with 8-bit coverage (0x7f rounding)
__device__ __forceinline__ uint32_t lerp_pixel(uint32_t src, uint32_t a, uint32_t dst) {
    // Cairo adds the two products with a saturating add; they cannot exceed 255 per channel (a + (255 - a) = 255 and both
    // products round down from src*a/255 + 1/2), so a plain add gives the same bits
    const uint32_t ia = 255u - a;
    return (mul8x2_7f(src, a) + mul8x2_7f(dst, ia)) | ((mul8x2_7f(src >> 8, a) + mul8x2_7f(dst >> 8, ia)) << 8);
}
// pixman UN8x4_MUL_UN8 (0x80 rounding) and OVER
__device__ __forceinline__ uint32_t mul_un8(uint32_t x, uint32_t a) {
    uint32_t rb = (x & 0xff00ffu) * a + 0x800080u;
    rb = ((rb + ((rb >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
    uint32_t ag = ((x >> 8) & 0xff00ffu) * a + 0x800080u;
    ag = ((ag + ((ag >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
    return rb | (ag << 8);
}
__device__ __forceinline__ uint32_t over_pixel(uint32_t src, uint32_t dst) {
    const uint32_t m = mul_un8(dst, 255u - (src >> 24));
    const uint32_t rb = add8x2_sat(m & 0xff00ffu, src & 0xff00ffu);
    const uint32_t ag = add8x2_sat((m >> 8) & 0xff00ffu, (src >> 8) & 0xff00ffu);
    return rb | (ag << 8);
}

__device__ uint32_t gradient_color(const swfr_style& s, double t) {
    const int n = (int)s.n_stops;
    if (n == 0) return 0;
    double r, g, b, a;
    if (t <= (double)s.stop_offset[0]) { r = s.stop_rgba[0][0]; g = s.stop_rgba[0][1]; b = s.stop_rgba[0][2]; a = s.stop_rgba[0][3]; }
    else if (t >= (double)s.stop_offset[n - 1]) { r = s.stop_rgba[n - 1][0]; g = s.stop_rgba[n - 1][1]; b = s.stop_rgba[n - 1][2]; a = s.stop_rgba[n - 1][3]; }
    else {
        int i = 0;
        while (i + 1 < n && (double)s.stop_offset[i + 1] <= t) ++i;
        const double t0 = s.stop_offset[i], t1 = s.stop_offset[i + 1], span = t1 - t0, f = span > 0 ? (t - t0) / span : 0;
        r = s.stop_rgba[i][0] + ((double)s.stop_rgba[i + 1][0] - s.stop_rgba[i][0]) * f;
        g = s.stop_rgba[i][1] + ((double)s.stop_rgba[i + 1][1] - s.stop_rgba[i][1]) * f;
        b = s.stop_rgba[i][2] + ((double)s.stop_rgba[i + 1][2] - s.stop_rgba[i][2]) * f;
        a = s.stop_rgba[i][3] + ((double)s.stop_rgba[i + 1][3] - s.stop_rgba[i][3]) * f;
    }
    const uint32_t A = (uint32_t)(a * 255.0 + 0.5), R = (uint32_t)(r * a * 255.0 + 0.5);
    const uint32_t G = (uint32_t)(g * a * 255.0 + 0.5), B = (uint32_t)(b * a * 255.0 + 0.5);
    return (A << 24) | (R << 16) | (G << 8) | B;
}

// premultiplied ARGB source colour at pixel centre (px+0.5, py+0.5): radial gradients and bitmaps follow pixman operation by
// operation (bit-exact); linear gradients -- an extension, the reference throws -- are a float64 model (within +-1 LSB of Cairo)
// pixman-gradient-walker.c: the interval of position x (a position equal to a stop belongs to the interval on its right), the
// single-precision ramp of that interval, premultiplied in floats, rounded by + .5 and truncation
__device__ __forceinline__ uint32_t radial_walker_pixel(const DevGradient& G, long long x) {
    int k = 0;
    while (k < G.n_intervals - 1 && !(x < (long long)G.x[k + 1])) ++k;
    const float* w = G.ramp[k];
    const float y = (float)x * (1.0f / 65536.0f);
    const float fa = 255.f * (w[0] * y + w[1]);
    const float fr = fa * (w[2] * y + w[3]), fg = fa * (w[4] * y + w[5]), fb = fa * (w[6] * y + w[7]);
    return (((uint32_t)(fa + .5f) << 24) & 0xff000000u) | (((uint32_t)(fr + .5f) << 16) & 0x00ff0000u) |
           (((uint32_t)(fg + .5f) << 8) & 0x0000ff00u) | ((uint32_t)(fb + .5f) & 0x000000ffu);
}
// pixman-radial-gradient.c radial_get_scanline / radial_compute_color, extend PAD: B and C are exact 64-bit integers of the
// pixel's 16.16 sample position (stepping them along a scanline, as pixman does, is the same arithmetic), the root in doubles
__device__ uint32_t shade_radial(const DevGradient& G, int px, int py) {
    if (!G.n_intervals) return 0u;
    const long long vx = G.base_x + (long long)px * G.m00 + (long long)py * G.m01 - G.c1x;
    const long long vy = G.base_y + (long long)px * G.m10 + (long long)py * G.m11 - G.c1y;
    const long long bi = vx * G.dx + vy * G.dy + (long long)G.c1r * G.dr;
    const long long ci = vx * vx + vy * vy - (long long)G.c1r * G.c1r;
    const double a = G.a, b = (double)bi, c = (double)ci, dr = (double)G.dr;
    if (a == 0) {
        if (b == 0) return 0u;
        const double t = 65536 / 2 * c / b;
        if (t * dr >= G.mindr) return radial_walker_pixel(G, (long long)t);
        return 0u;
    }
    const double discr = b * b + a * -c;
    if (discr >= 0) {
        const double sq = __dsqrt_rn(discr), t0 = (b + sq) * G.inva, t1 = (b - sq) * G.inva;
        if (t0 * dr >= G.mindr) return radial_walker_pixel(G, (long long)t0);
        else if (t1 * dr >= G.mindr) return radial_walker_pixel(G, (long long)t1);
    }
    return 0u;
}

__device__ __noinline__ uint32_t shade(const swfr_style& s, uint32_t style_index, const Sources bitmaps, int px, int py) {
    if (s.kind == SWFR_STYLE_RADIAL) {
        const int gi = bitmaps.filters[style_index].pad;
        if (gi > 0) return shade_radial(bitmaps.gradients[gi - 1], px, py);
    }
    double x = px + 0.5, y = py + 0.5;
    const double ux = s.inv[0] * x + s.inv[2] * y + s.inv[4];
    const double uy = s.inv[1] * x + s.inv[3] * y + s.inv[5];
    if (s.kind == SWFR_STYLE_RADIAL) {
        const double cdx = s.c1x - s.c0x, cdy = s.c1y - s.c0y, dr = s.r1 - s.r0;
        const double pdx = ux - s.c0x, pdy = uy - s.c0y;
        const double A = cdx * cdx + cdy * cdy - dr * dr;
        const double B = pdx * cdx + pdy * cdy + s.r0 * dr;
        const double C = pdx * pdx + pdy * pdy - s.r0 * s.r0;
        double t;
        if (A == 0) { if (B == 0) return 0; t = 0.5 * C / B; if (s.r0 + t * dr < 0) return 0; }
        else {
            const double disc = B * B - A * C;
            if (disc < 0) return 0;
            const double sq = sqrt(disc), t0 = (B + sq) / A, t1 = (B - sq) / A;
            if (s.r0 + t0 * dr >= 0) t = t0; else if (s.r0 + t1 * dr >= 0) t = t1; else return 0;
        }
        t = fmin(fmax(t, 0.0), 1.0);
        return gradient_color(s, t);
    }
    if (s.kind == SWFR_STYLE_LINEAR) {
        const double dx = s.c1x - s.c0x, dy = s.c1y - s.c0y, l = dx * dx + dy * dy;
        double t = l == 0 ? 0 : ((ux - s.c0x) * dx + (uy - s.c0y) * dy) / l;
        t = fmin(fmax(t, 0.0), 1.0);
        return gradient_color(s, t);
    }
    const DevBitmap bm = bitmaps.bitmaps[s.bitmap];
    const DevFilter flt = bitmaps.filters[style_index];
    // pixman's own 16.16 sample position of this pixel's centre
    const long long fxp = flt.base_x + (long long)px * flt.m00 + (long long)py * flt.m01;
    const long long fyp = flt.base_y + (long long)px * flt.m10 + (long long)py * flt.m11;
    if (flt.on) {
        // CAIRO_FILTER_GOOD below scale 0.75: pixman's separable convolution (integer tables and accumulation)
        long long x = fxp, y = fyp;
        const int xsh = 16 - flt.xbits, ysh = 16 - flt.ybits;
        const long long x_off = (((long long)flt.cw << 16) - 65536) >> 1, y_off = (((long long)flt.ch << 16) - 65536) >> 1;
        x = ((x >> xsh) << xsh) + ((1 << xsh) >> 1);          // the middle of the closest phase
        y = ((y >> ysh) << ysh) + ((1 << ysh) >> 1);
        const int phx = (int)((x & 0xffff) >> xsh), phy = (int)((y & 0xffff) >> ysh);
        const int32_t* yp = bitmaps.fparams + flt.y_off + phy * flt.ch;
        const int32_t* xp0 = bitmaps.fparams + flt.x_off + phx * flt.cw;
        const int x1 = (int)((x - 1 - x_off) >> 16), y1 = (int)((y - 1 - y_off) >> 16);
        long long sr = 0, sg = 0, sb = 0, sa = 0;
        for (int i = 0; i < flt.ch; ++i) {
            const long long fy = yp[i];
            if (!fy) continue;
            int ry = y1 + i;
            if (s.extend == 1) ry = ((ry % (int)bm.height) + (int)bm.height) % (int)bm.height;
            for (int j = 0; j < flt.cw; ++j) {
                const int32_t fx = xp0[j];
                if (!fx) continue;
                int rx = x1 + j;
                uint32_t pixel;
                if (s.extend == 1) { rx = ((rx % (int)bm.width) + (int)bm.width) % (int)bm.width; pixel = bm.pixels[(size_t)ry * bm.width + rx]; }
                else pixel = (rx < 0 || ry < 0 || rx >= (int)bm.width || ry >= (int)bm.height) ? 0u : bm.pixels[(size_t)ry * bm.width + rx];
                const int f = (int)((fy * fx + 0x8000) >> 16);
                sr += (int)((pixel >> 16) & 255u) * f; sg += (int)((pixel >> 8) & 255u) * f; sb += (int)(pixel & 255u) * f; sa += (int)(pixel >> 24) * f;
            }
        }
        sa = (sa + 0x8000) >> 16; sr = (sr + 0x8000) >> 16; sg = (sg + 0x8000) >> 16; sb = (sb + 0x8000) >> 16;
        sa = min(max(sa, 0ll), 255ll); sr = min(max(sr, 0ll), 255ll); sg = min(max(sg, 0ll), 255ll); sb = min(max(sb, 0ll), 255ll);
        return ((uint32_t)sa << 24) | ((uint32_t)sr << 16) | ((uint32_t)sg << 8) | (uint32_t)sb;
    }
    // bilinear with 7-bit weights (what CAIRO_FILTER_GOOD becomes for scales > .75)
    const long long bxp = fxp - 0x8000, byp = fyp - 0x8000;
    const int x0 = (int)(bxp >> 16), y0 = (int)(byp >> 16);
    const int wx = (int)((bxp >> 9) & 0x7f), wy = (int)((byp >> 9) & 0x7f);
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        if (s.extend == 1) {
            xx = ((xx % (int)bm.width) + (int)bm.width) % (int)bm.width;
            yy = ((yy % (int)bm.height) + (int)bm.height) % (int)bm.height;
            c[k] = bm.pixels[(size_t)yy * bm.width + xx];
        } else {
            c[k] = (xx < 0 || yy < 0 || xx >= (int)bm.width || yy >= (int)bm.height) ? 0u : bm.pixels[(size_t)yy * bm.width + xx];
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int sh = 0; sh < 32; sh += 8) {
        const uint32_t v00 = (c[0] >> sh) & 255, v10 = (c[1] >> sh) & 255, v01 = (c[2] >> sh) & 255, v11 = (c[3] >> sh) & 255;
        const uint32_t acc = v00 * (128 - wx) * (128 - wy) + v10 * wx * (128 - wy) + v01 * (128 - wx) * wy + v11 * wx * wy;
        out |= ((acc >> 14) & 255) << sh;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// k_tiles
// ---------------------------------------------------------------------------------------------
#define ACC_STRIDE 66                 // 64 cells + carry slot + touched flag
#define ACC_CARRY 64
#define ACC_TOUCH 65
#define LIST_CAP 256


struct TileCtx {
    int tx0, xminp, xmaxp;
};


// accumulate one cell contribution (covered height dch, uncovered area dua) of row `acc`
__device__ __forceinline__ void cell_add(int* acc, const TileCtx& c, int i, int dch, int dua) {
    if (i >= c.xmaxp) return;                            // at/after the converter's right bound: never emitted
    if (i < c.xminp) { i = c.xminp; dua = 0; }           // left of it: height only, folded into the first column
    if (i < c.tx0) { atomicAdd(&acc[ACC_CARRY], dch); return; }
    if (i >= c.tx0 + TILE_W) return;
    atomicAdd(&acc[i - c.tx0], dch * (1 << 20) + dua);
}

// FULL-row edge (A.5 render_edge): analytic trapezoid coverage of one edge over one pixel row
__device__ void full_edge(const Rec& rec, int sign, int* acc, const TileCtx& c) {
    int32_t q1 = rec.q1, q2 = rec.q2; int64_t r1 = rec.r1, r2 = rec.r2;
    const int64_t edy = rec.dy;
    int ix1 = q1 >> 8, f1 = q1 & 255, ix2 = q2 >> 8, f2 = q2 & 255;
    if (ix1 == ix2) { cell_add(acc, c, ix1, sign * 15, sign * (f1 + f2) * 15); return; }
    if (ix2 < ix1) { int t = ix1; ix1 = ix2; ix2 = t; t = f1; f1 = f2; f2 = t; int32_t tq = q1; q1 = q2; q2 = tq; int64_t tr = r1; r1 = r2; r2 = tr; }
    const int lo = max(c.tx0, c.xminp);                  // first column whose own area matters to this tile
    const int hi = min(c.tx0 + TILE_W, c.xmaxp);         // one past the last such column
    if (ix1 >= hi) return;                               // entirely to the right: invisible here
    const int64_t dx = (int64_t)(q2 - q1) * edy + (r2 - r1);
    const int64_t t0 = ((int64_t)((ix1 + 1) * 256 - q1) * edy - r1) * 15;
    const int64_t F = 15ll * 256 * edy;
    // Y(col) = covered sub-rows accumulated over columns ix1..col (ix1 <= col < ix2), exact floor
    const int first = max(ix1, lo);
    int y_prev = 0;
    if (first > ix1) {
        // columns ix1..first-1 lie left of the tile (or of the converter): only their summed height counts
        if (first - 1 >= ix2) { cell_add(acc, c, lo - 1, sign * 15, 0); return; }
        int64_t q, r;
        floor_div(t0 + (int64_t)(first - 1 - ix1) * F, dx, q, r);
        y_prev = (int)q;
        cell_add(acc, c, lo - 1, sign * y_prev, 0);
    }
    int64_t yq = 0, yr = 0, fq = 0, fr = 0;
    if (first < ix2) {
        floor_div(t0 + (int64_t)(first - ix1) * F, dx, yq, yr);
        floor_div(F, dx, fq, fr);
    }
    for (int col = first; col < hi && col <= ix2; ++col) {
        if (col == ix2) { cell_add(acc, c, col, sign * (15 - y_prev), sign * (15 - y_prev) * f2); break; }
        if (col > first) { yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; } }
        const int h = (int)yq - y_prev;
        cell_add(acc, c, col, sign * h, sign * h * (col == ix1 ? 256 + f1 : 256));
        y_prev = (int)yq;
    }
}


// ---------------------------------------------------------------------------------------------
// k_class: one wavefront per band entry (path x tile-row); lane = tile column of the path's pixel rectangle.
// Classifies every (tile, path) pair from the record headers alone (uniform loads, no edge arithmetic):
// CLS_PARTIAL (a boundary passes through the tile), full cover, or empty.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_class(const BandEntry* __restrict__ band_list, uint32_t n_entries,
                                              const uint32_t* __restrict__ band_off, uint32_t n_bands,
                                              const swfr_edge* __restrict__ raw_edges, const RowInfo* __restrict__ rows,
                                              const Rec* __restrict__ records, uint8_t* __restrict__ cls_t, int width, int height,
                                              int tiles_x, uint32_t band_index, uint32_t band_count) {
    const uint32_t ei = blockIdx.x;
    if (ei >= n_entries) return;
    const int lane = threadIdx.x;
    const BandEntry e = band_list[ei];
    const uint32_t band = e.flags >> 8;
    if (band >= n_bands) return;
    if (band_count > 1 && band % band_count != band_index) return;      // another rank's tile-row
    const uint32_t b0 = band_off[band], n_b = band_off[band + 1] - b0, e_local = ei - b0;
    uint8_t* out = cls_t + (size_t)tiles_x * b0 + e_local;              // + tile column * n_b
    const int ty0 = (int)band * TILE_H, tile_y1 = min(ty0 + TILE_H, height);
    const int tc0 = (int)e.x_min / TILE_W, tc1 = ((int)e.x_max - 1) / TILE_W;
    for (int tc = lane; tc < tiles_x; tc += 64)                          // columns the path's rectangle does not reach: empty
        if (tc < tc0 || tc > tc1) out[(size_t)tc * n_b] = 0;
    if (e.flags & BE_BOXES) {
        for (int tc = tc0 + lane; tc <= tc1; tc += 64) {
            const int tx0 = tc * TILE_W, tile_x1 = min(tx0 + TILE_W, width);
            uint32_t flags = CLS_BOX | CLS_NONEMPTY | CLS_NOTFULL;
            if (e.n_edges == 1) {                     // one box that contains the whole tile: full cover
                const swfr_edge bx = raw_edges[e.first_edge];
                if (bx.x1 <= tx0 * 256 && bx.x2 >= tile_x1 * 256 && bx.y1 <= ty0 * 256 && bx.y2 >= tile_y1 * 256) flags = CLS_NONEMPTY;
            }
            out[(size_t)tc * n_b] = (uint8_t)flags;
        }
        return;
    }
    // lane = (tile column within a group of four, pixel row of the band): the 16 rows are read once, in parallel
    const int row = lane & 15, sub = lane >> 4;
    const int y = ty0 + row;
    const bool in_frame = y < tile_y1, in_rows = in_frame && y >= e.y_min && y < e.y_max;
    uint32_t n_rec = 0, rec_off = 0;
    if (in_rows) { const RowInfo ri = rows[e.row_base + (uint32_t)(y - e.y_min)]; n_rec = ri.n_rec; rec_off = ri.rec_off; }
    constexpr int HN = 6;                             // record headers kept in registers; longer rows re-read the rest
    uint32_t hroles[HN], hcols[HN];
#pragma unroll
    for (int k = 0; k < HN; ++k) {
        hroles[k] = 0; hcols[k] = 0;
        if ((uint32_t)k < n_rec) { const Rec* rp = &records[rec_off + k]; hroles[k] = rp->roles; hcols[k] = rp->cols; }
    }
    for (int tcb = tc0; tcb <= tc1; tcb += 4) {
        const int tc = tcb + sub;
        const int tx0 = tc * TILE_W, tile_x1 = min(tx0 + TILE_W, width);
        uint32_t f = 0;
        if (in_frame && tc <= tc1) {
            if (!in_rows) f = CLS_NOTFULL;
            else {
                int carry = 0;
                bool inter = false;
#pragma unroll
                for (int k = 0; k < HN; ++k) {
                    if ((uint32_t)k >= n_rec) continue;
                    const int clo = (int)(hcols[k] & 0xffffu), chi = (int)(hcols[k] >> 16);
                    if (chi < tx0 && chi < 65535) carry += record_height(hroles[k]);
                    else if (clo >= tx0 + TILE_W && clo < 65535) { /* right of the tile */ }
                    else inter = true;
                }
                for (uint32_t k = HN; k < n_rec; ++k) {
                    const Rec* rp = &records[rec_off + k];
                    const uint32_t rroles = rp->roles, rcols = rp->cols;
                    const int clo = (int)(rcols & 0xffffu), chi = (int)(rcols >> 16);
                    if (chi < tx0 && chi < 65535) carry += record_height(rroles);
                    else if (clo >= tx0 + TILE_W && clo < 65535) { /* right of the tile */ }
                    else inter = true;
                }
                const bool inside_x = e.x_min <= tx0 && e.x_max >= tile_x1;
                const uint32_t a = (uint32_t)((carry * 512 * 17 + 256) >> 9) & 255u;
                if (inter) f = CLS_PARTIAL | CLS_NOTFULL | CLS_NONEMPTY;
                else if (a == 0) f = CLS_NOTFULL | CLS_HOLE;
                else if (a == 255 && inside_x) f = CLS_NONEMPTY;
                else f = CLS_PARTIAL | CLS_NOTFULL | CLS_NONEMPTY;       // uniform partial alpha or column masking
            }
        }
        // OR over the 16 rows of the tile (lanes of one 16-lane group); every lane active
        f |= (uint32_t)__shfl_xor((int)f, 8);
        f |= (uint32_t)__shfl_xor((int)f, 4);
        f |= (uint32_t)__shfl_xor((int)f, 2);
        f |= (uint32_t)__shfl_xor((int)f, 1);
        if ((f & (CLS_HOLE | CLS_NONEMPTY)) == (CLS_HOLE | CLS_NONEMPTY)) f |= CLS_PARTIAL;
        f &= ~CLS_HOLE;
        if (row == 0 && tc <= tc1) out[(size_t)tc * n_b] = (uint8_t)f;
    }
}

#define TLIST 32                       // tile list entries per round
#define REC_STAGE 32                   // records staged in LDS per round
#define P2B 8                          // rows scanned + blended per straight-line step
#ifndef BLEND_QUEUE
#define BLEND_QUEUE 256                // edge pixels of one (path, strip) blended in compacted form; the accumulator holds 264 pairs
#endif
#define PBATCH (64 / STRIP_H)           // partial paths whose row headers and records are fetched in one round trip each

template <bool SHADERS>
__device__ __forceinline__ uint32_t blend_pixel_t(uint32_t dst, uint32_t a, uint32_t eflags, uint32_t solid, const swfr_style* __restrict__ styles,
                                                  uint32_t style, const Sources bitmaps, int cx, int cy) {
    if (!SHADERS || (eflags & BE_SOLID)) {
        if (eflags & BE_LERP) return a == 255u ? solid : lerp_pixel(solid, a, dst);
        return over_pixel(a == 255u ? solid : mul_un8(solid, a), dst);
    }
    const uint32_t s = mul_un8(shade(styles[style], style, bitmaps, cx, cy), a);
    return (eflags & BE_LERP) ? s : over_pixel(s, dst);
}
#define blend_pixel blend_pixel_t<SHADERS>

// one staged record -> covered height / uncovered area per cell of its row (LDS atomics into `acc`)
// returns true for a SUB-row record that reaches into the tile: its fifteen sample rows are spread over lanes by the caller
__device__ __forceinline__ bool accumulate_record(const uint32_t* sw, int* acc, const TileCtx& c) {
    // the whole 48-byte record in three 16-byte LDS reads (staged records start on 16-byte boundaries)
    const uint4* s4 = reinterpret_cast<const uint4*>(sw);
    const uint4 w0 = s4[0], w1 = s4[1], w2 = s4[2];
    const uint32_t roles = w0.x, rcols = w0.y;
    const int clo = (int)(rcols & 0xffffu), chi = (int)(rcols >> 16);
    if (clo >= c.tx0 + TILE_W && clo < 65535) return false;         // entirely right of the tile
    if (chi < c.tx0 && chi < 65535) {                               // entirely left: only its net height reaches us
        cell_add(acc, c, chi, record_height(roles), 0);
        return false;
    }
    if (roles & REC_CELLS) {                                        // precomputed by k_rows: no arithmetic left
        const int n = (int)(roles & 15u);
        const uint32_t cw[REC_MAX_CELLS] = {w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int k = 0; k < REC_MAX_CELLS; ++k) {
            if (k < n) cell_add(acc, c, clo + (int)(cw[k] & 255u), (int)(int8_t)(cw[k] >> 8), (int)(int16_t)(cw[k] >> 16));
        }
        return false;
    }
    if (!(roles & REC_FULL)) return true;
    Rec rec;
    rec.roles = roles; rec.cols = rcols; rec.q1 = (int32_t)w0.z; rec.q2 = (int32_t)w0.w;
    rec.r1 = (int64_t)((uint64_t)w1.x | ((uint64_t)w1.y << 32));
    rec.r2 = (int64_t)((uint64_t)w1.z | ((uint64_t)w1.w << 32));
    rec.dy = (int64_t)((uint64_t)w2.x | ((uint64_t)w2.y << 32));
    rec.span = w2.z; rec.eid = w2.w;
    full_edge(rec, (rec.roles & 1u) ? +1 : -1, acc, c);
    return false;
}

// One sample row of a SUB-row record (lane = sample): x at sample ss in closed form from the record's first sample and slope
__device__ __forceinline__ void accumulate_sub_sample(const uint32_t* sw, int ss, int* acc, const TileCtx& c) {
    const uint4* s4 = reinterpret_cast<const uint4*>(sw);
    const uint4 w0 = s4[0], w1 = s4[1], w2 = s4[2];
    const uint32_t roles = w0.x, span = w2.z;
    const int first = (int)(span & 255u), last = (int)(span >> 8);
    if (ss < first || ss >= last) return;
    const uint32_t role = (roles >> (2 * ss)) & 3u;
    if (!role) return;
    const int64_t r1 = (int64_t)((uint64_t)w1.x | ((uint64_t)w1.y << 32)), r2 = (int64_t)((uint64_t)w1.z | ((uint64_t)w1.w << 32));
    const int64_t dy = (int64_t)((uint64_t)w2.x | ((uint64_t)w2.y << 32));
    int cell = (int32_t)w0.z;
    if (dy) {
        const int k = ss - first;
        int64_t dq, rm;
        floor_div(r1 + (int64_t)k * r2, dy, dq, rm);          // |quotient| <= 15
        cell = cell_of((int32_t)w0.z + k * (int32_t)w0.w + (int32_t)dq, rm, dy);
    }
    const int sgn = role == 1 ? 1 : -1;
    cell_add(acc, c, cell >> 8, sgn, sgn * 2 * (cell & 255));
}

// One wavefront per 64x16 tile: lane = pixel column, the tile's pixels live in LDS.  No workgroup barriers:
// LDS traffic of a wave is ordered.  Global memory is read in coalesced pieces only: the tile's class bytes
// (contiguous per tile), the few band entries that survive, 16 row headers per path (one line) and the
// path's records as a dword stream staged through LDS.
// SHADERS = false is the solid-colour specialisation (no call into the f64 gradient/bitmap shader, fewer VGPRs)
template <bool SHADERS>
__device__ __forceinline__ void tiles_body(const swfr_edge* __restrict__ raw_edges,
                                              const uint32_t* __restrict__ band_off, const BandEntry* __restrict__ band_list,
                                              const uint8_t* __restrict__ cls_t, const RowInfo* __restrict__ rows,
                                              const Rec* __restrict__ records, const swfr_style* __restrict__ styles,
                                              const Sources bitmaps, uint32_t* __restrict__ fb,
                                              int width, int height, int tiles_x, uint32_t band_index, uint32_t band_count, int dbg,
                                              uint32_t* __restrict__ counters, uint32_t n_rows_total, uint32_t n_rec_cap,
                                              const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) int acc[STRIP_H][ACC_STRIDE];   // also the queue of the compacted blend (8-byte pairs)
    __shared__ int plist[PBATCH];
    __shared__ __attribute__((aligned(16))) uint32_t ent[TLIST][12];   // BandEntry as 9 dwords in a 48-byte slot (16-byte LDS writes)
    __shared__ uint32_t cls[TLIST];
    __shared__ __attribute__((aligned(16))) uint32_t stage[REC_STAGE * 12];   // records as dwords
    __shared__ uint32_t rec_src[REC_STAGE];
    __shared__ uint32_t row_off[64], row_start[64 + 1];   // per (batch path, strip row): first record, exclusive prefix of counts
    __shared__ uint8_t rec_row[REC_STAGE];

    const int lane = threadIdx.x;
    // blockIdx -> tile: consecutive workgroups walk along x inside one tile-row, so the 8 tiles that are
    // co-scheduled round-robin over the 8 XCDs read the same band entries / path records
    // (`order`, when given, is the host's launch order: strips with the most edges first, so that the heaviest
    //  wavefronts do not start last)
    const uint32_t wg = order ? order[blockIdx.x] : blockIdx.x;
    const int tile = (int)(wg / STRIPS_PER_TILE), strip = (int)(wg % STRIPS_PER_TILE);
    const int tcol = tile % tiles_x;
    int trow = tile / tiles_x;
    if (band_count > 1) trow = trow * (int)band_count + (int)band_index;
    const int tx0 = tcol * TILE_W, ty0 = trow * TILE_H + strip * STRIP_H;   // ty0: first pixel row of this wave's strip
    if (ty0 >= height) return;
    const int cx = tx0 + lane;
    const unsigned long long t_start = dbg == 8 ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t dbg_pairs = 0, dbg_recs = 0;
#ifdef SWFR_PHASES                         // build with -DSWFR_PHASES for tools/strip_times.py: clocks per phase (costs VGPRs)
    const unsigned long long w_start = __builtin_amdgcn_s_memrealtime();
    uint32_t ph_bin = 0, ph_batch = 0, ph_stage = 0, ph_acc = 0, ph_p2 = 0;
    unsigned long long ph_t = t_start;
#define PHASE(var) do { if (dbg == 8) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += (uint32_t)(now_ - ph_t); ph_t = now_; } } while (0)
#else
#define PHASE(var) do { } while (0)
#endif

    const int lane_t = lane / 12, lane_w = lane - lane_t * 12;   // record / word of dword `lane` in a stream of 12-dword records
    uint32_t px[STRIP_H];                                 // this lane's column of the strip, in registers (row loops are unrolled)
#pragma unroll
    for (int rr = 0; rr < STRIP_H; ++rr) px[rr] = 0u;
    for (int i = lane; i < STRIP_H * ACC_STRIDE; i += 64) (&acc[0][0])[i] = 0;

    const uint32_t band_begin = band_off[trow], band_end = band_off[trow + 1];
    const uint32_t n_b = band_end - band_begin;
    const uint8_t* mycls = cls_t + (size_t)tiles_x * band_begin + (size_t)tcol * n_b;   // this tile's class byte per band entry
    uint32_t next = 0;
    while (next < n_b) {
        // ---- bin: band entries with a non-empty class for this tile, painter's order kept (wave-local compaction)
        int ln = 0;
        while (next < n_b && ln < TLIST) {
            const uint32_t bi = next + lane;
            const uint32_t f = bi < n_b ? (uint32_t)mycls[bi] : 0u;
            bool hit = (f & CLS_NONEMPTY) != 0;
            unsigned long long b = __ballot(hit);
            const int room = TLIST - ln;
            int cnt = __popcll(b);
            if (cnt > room) {                                 // keep the first `room` hits, rescan the rest next round
                int keep = room; unsigned long long m = b, kept = 0ull; uint32_t last = 0;
                while (keep--) { const int bit = __ffsll((long long)m) - 1; kept |= 1ull << bit; m &= m - 1; last = (uint32_t)bit; }
                b = kept; hit = hit && ((kept >> lane) & 1ull); cnt = room;
                next += last + 1;
            } else next += 64;
            if (hit) {
                const int at = ln + __popcll(b & ((1ull << lane) - 1ull));
                const uint32_t* src = reinterpret_cast<const uint32_t*>(&band_list[band_begin + bi]);
                // 36-byte entry: two 16-byte global loads + one dword (the entries are only 4-byte aligned: dword-aligned vector type)
                struct __attribute__((aligned(4))) u32x4_a4 { uint32_t x, y, z, w; };
                const u32x4_a4 q0 = *reinterpret_cast<const u32x4_a4*>(src), q1 = *reinterpret_cast<const u32x4_a4*>(src + 4);
                const uint32_t q2 = src[8];
                *reinterpret_cast<uint4*>(&ent[at][0]) = make_uint4(q0.x, q0.y, q0.z, q0.w);
                *reinterpret_cast<uint4*>(&ent[at][4]) = make_uint4(q1.x, q1.y, q1.z, q1.w);
                ent[at][8] = q2;
                cls[at] = f;
            }
            ln += cnt;
        }
        if (dbg == 1) ln = 0;
        __syncthreads();                                      // ent/cls written by other lanes (one-wave workgroup: cheap)

        // ---- occlusion: everything below the last opaque, lerp-blended full cover is invisible in this tile
        int start = 0;
        {
            bool cover = false;
            if (lane < ln) {
                const uint32_t f = cls[lane];
                cover = (f & (CLS_PARTIAL | CLS_NOTFULL | CLS_BOX)) == 0 && (ent[lane][7] & BE_OPAQUE_COVER);
            }
            const unsigned long long b = __ballot(cover);
            if (b) start = 63 - __clzll((long long)b);
        }
        if (dbg == 2) start = ln;
        PHASE(ph_bin);
        if (dbg == 9 && lane == 0) {                          // statistics (SWFR_TILES_DEBUG=9)
            uint32_t np = 0, nf = 0;
            for (int li = start; li < ln; ++li) { const uint32_t f = cls[li]; if (f & (CLS_PARTIAL | CLS_BOX)) ++np; else ++nf; }
            atomicAdd(&counters[CNT_PAIRS], (uint32_t)ln); atomicAdd(&counters[CNT_PARTIAL], np); atomicAdd(&counters[CNT_FULL], nf);
            atomicAdd(&counters[CNT_CULLED], (uint32_t)start);
        }

        // ---- painter's order walk.  The row headers of up to PBATCH partial tor paths are fetched in one round trip and
        //      their records form one sequence that is staged through LDS in windows of REC_STAGE (usually a single one);
        //      each path is then accumulated and blended in order from the staged records.
        int batch_n = 0, batch_i = 0;                          // paths in the batch, next one to consume
        int total = 0, wbase = 0, wn = 0;                      // records of the batch; staged window [wbase, wbase + wn)
        for (int li = start; li < ln; ++li) {
            // per-entry fields are wave-uniform: readfirstlane moves them (and everything computed from them) to the scalar unit
            const uint32_t f = __builtin_amdgcn_readfirstlane(cls[li]);
            const uint32_t xw = __builtin_amdgcn_readfirstlane(ent[li][1]), yw = __builtin_amdgcn_readfirstlane(ent[li][2]);
            const int e_xmin = (int)(int16_t)(xw & 0xffffu), e_xmax = (int)(int16_t)(xw >> 16);
            const int e_ymin = (int)(int16_t)(yw & 0xffffu), e_ymax = (int)(int16_t)(yw >> 16);
            const uint32_t style = __builtin_amdgcn_readfirstlane(ent[li][4]), e_first = __builtin_amdgcn_readfirstlane(ent[li][5]),
                           e_nedges = __builtin_amdgcn_readfirstlane(ent[li][6]);
            const uint32_t eflags = __builtin_amdgcn_readfirstlane(ent[li][7]), solid = __builtin_amdgcn_readfirstlane(ent[li][8]);
            const int row_lo = max(e_ymin, ty0) - ty0, row_hi = min(min(e_ymax, ty0 + STRIP_H), height) - ty0;
            if (row_hi <= row_lo) continue;                    // the path misses this strip of the tile
            if (f & CLS_BOX) {
                // ---- rectilinear (A.6): exact area of disjoint boxes, alpha = (c>>8) - (c>>16)
#pragma unroll
                for (int rr = 0; rr < STRIP_H; ++rr) {
                    if (rr < row_lo || rr >= row_hi) continue;         // wave-uniform
                    const int cy = ty0 + rr;
                    uint32_t cov = 0u;
                    for (uint32_t k = 0; k < e_nedges; ++k) {
                        const swfr_edge bx = raw_edges[e_first + k];
                        const int wx = min(bx.x2, (cx + 1) * 256) - max(bx.x1, cx * 256);
                        const int wy = min(bx.y2, (cy + 1) * 256) - max(bx.y1, cy * 256);
                        if (wx > 0 && wy > 0) cov += (uint32_t)(wx * wy);
                    }
                    const uint32_t a = ((cov >> 8) - (cov >> 16)) & 255u;
                    if (a) px[rr] = blend_pixel(px[rr], a, eflags, solid, styles, style, bitmaps, cx, cy);
                }
            } else if (f & CLS_PARTIAL) {
                if (dbg == 3) continue;
                // ---- tor (A.5)
                PHASE(ph_bin);
                if (batch_i == batch_n) {
                    // the next PBATCH partial tor paths of the list, this one first (lane = list position)
                    //   (same visibility test as the walk: entries that miss this strip's rows are skipped there)
                    bool isp = lane >= li && lane < ln && (cls[lane] & (CLS_PARTIAL | CLS_BOX)) == CLS_PARTIAL;
                    if (isp) {
                        const uint32_t lyw = ent[lane][2];
                        const int l_ymin = (int)(int16_t)(lyw & 0xffffu), l_ymax = (int)(int16_t)(lyw >> 16);
                        isp = min(min(l_ymax, ty0 + STRIP_H), height) > max(l_ymin, ty0);
                    }
                    const unsigned long long pm = __ballot(isp);
                    const int rank = __popcll(pm & ((1ull << lane) - 1ull));
                    if (isp && rank < PBATCH) plist[rank] = lane;
                    batch_n = min((int)__popcll(pm), PBATCH); batch_i = 0;
                    __syncthreads();                                   // plist visible
                    // row headers: lane = (path of the batch, row of the strip)
                    uint32_t my_cnt = 0, off = 0;
                    {
                        int lv = lane;
                        SWFR_OPAQUE(lv);                       // as below: keep this address arithmetic out of the prologue
                        const int bp = lv / STRIP_H, row = lv % STRIP_H;
                        if (bp < batch_n) {
                            const int mine = plist[bp];
                            const uint32_t myw = ent[mine][2];
                            const int p_ymin = (int)(int16_t)(myw & 0xffffu), p_ymax = (int)(int16_t)(myw >> 16);
                            const int y = ty0 + row;
                            if (y >= p_ymin && y < p_ymax && y < height) {
                                const uint32_t ridx = ent[mine][3] + (uint32_t)(y - p_ymin);
                                if (ridx < n_rows_total) {
                                    const RowInfo ri = rows[ridx];
                                    off = ri.rec_off; my_cnt = ri.n_rec;
                                    if ((uint64_t)off + my_cnt > n_rec_cap) { atomicOr(&counters[CNT_ERROR], 4u); my_cnt = 0; }
                                } else atomicOr(&counters[CNT_ERROR], 2u);  // defensive: never read outside the row table
                            }
                        }
                    }
                    const int incl = wave_scan_incl((int)my_cnt);       // every lane active
                    total = __builtin_amdgcn_readlane(incl, 63);       // wave-uniform, on the scalar unit
                    row_off[lane] = off;
                    row_start[lane] = (uint32_t)(incl - (int)my_cnt);
                    row_start[64] = (uint32_t)total;                    // same value from every lane
                    wbase = 0; wn = 0;
                    dbg_recs += (uint32_t)total;
                    if (dbg == 11) { row_start[lane] = 0; row_start[64] = 0; total = 0; }
                    __syncthreads();                                   // row_off / row_start visible to every lane
                    PHASE(ph_batch);
                }
                const int bp = batch_i++;
                ++dbg_pairs;
                TileCtx c; c.tx0 = tx0; c.xminp = e_xmin; c.xmaxp = e_xmax;
                int g0 = (int)__builtin_amdgcn_readfirstlane(row_start[bp * STRIP_H]);
                const int g1 = (int)__builtin_amdgcn_readfirstlane(row_start[(bp + 1) * STRIP_H]);   // this path's records [g0, g1) of the batch sequence
                const uint32_t* rdw = reinterpret_cast<const uint32_t*>(records);
                while (g0 < g1) {
                    if (g0 >= wbase + wn) {
                        // ---- stage the window that starts at this path's next record
                        wbase = g0; wn = min(REC_STAGE, total - wbase);
                        // which (path, row) does each staged record belong to, and where does it live
                        for (int t = lane; t < wn; t += 64) {
                            const uint32_t g = (uint32_t)(wbase + t);
                            int lo = 0, hi = 64;                        // last virtual row with row_start <= g
                            while (lo + 1 < hi) { const int mid = (lo + hi) >> 1; if (row_start[mid] <= g) lo = mid; else hi = mid; }
                            rec_row[t] = (uint8_t)lo;
                            uint32_t src = row_off[lo] + (g - row_start[lo]);
                            if (src >= n_rec_cap) { atomicOr(&counters[CNT_ERROR], 8u); src = 0; }   // defensive
                            rec_src[t] = src;
                        }
                        __syncthreads();
                        // coalesced dword stream of the records into LDS, six independent loads in flight per lane
                        if (dbg != 12)
                        for (int d0 = 0; d0 < wn * 12; d0 += 64 * 6) {
                            uint32_t tmp[6];
                            // dword d = d0 + 64 u + lane is word w of staged record t: (t, w) advance by (5, 4) per u, no divisions
                            // (the lane's position is made opaque here: otherwise the compiler hoists the twelve per-lane addresses of
                            //  this loop to the kernel's prologue and spills them -- scratch stores in every wavefront, used by few)
                            int lt = lane_t, lw = lane_w;
                            SWFR_OPAQUE(lt); SWFR_OPAQUE(lw);
                            int t = d0 / 12 + lt, w = lw;                  // d0 is a multiple of 384 = 32 records
#pragma unroll
                            for (int u = 0; u < 6; ++u) {
                                const int d = d0 + u * 64 + lane;
                                tmp[u] = 0u;
                                if (d < wn * 12) tmp[u] = rdw[(size_t)rec_src[t] * 12 + w];
                                w += 4; t += 5;
                                if (w >= 12) { w -= 12; ++t; }
                            }
#pragma unroll
                            for (int u = 0; u < 6; ++u) {
                                const int d = d0 + u * 64 + lane;
                                if (d < wn * 12) stage[d] = tmp[u];
                            }
                        }
                        __syncthreads();
                        PHASE(ph_stage);
                    }
                    const int hi_g = min(g1, wbase + wn);
                    // lanes = this path's staged records
                    if (dbg != 12 && dbg != 13) {
                        bool sub_pending = false;
                        const int t = g0 - wbase + lane;               // a window holds at most REC_STAGE <= 64 records: one pass
                        if (t < hi_g - wbase) {
                            const uint32_t* sw = &stage[t * 12];
                            const int r = rec_row[t] % STRIP_H;
                            sub_pending = accumulate_record(sw, acc[r], c);
                        }
                        // SUB-row records: four at a time, lane = (record, sample row)
                        unsigned long long pend = __ballot(sub_pending);
                        while (pend) {
                            const int g = lane / 15, ss = lane - g * 15;
                            unsigned long long m = pend;
                            int src = -1;
                            for (int q = 0; q < 4; ++q) {
                                const int bit = m ? __ffsll((long long)m) - 1 : -1;
                                if (q == g) src = bit;
                                m &= m - 1;
                            }
                            pend = m;
                            if (g < 4 && src >= 0) {
                                const int t = g0 - wbase + src;
                                accumulate_sub_sample(&stage[t * 12], ss, acc[rec_row[t] % STRIP_H], c);
                            }
                        }
                    }
                    g0 = hi_g;
                    __syncthreads();                                   // acc complete; the window may be restaged
                    PHASE(ph_acc);
                }
                int (*A)[ACC_STRIDE] = acc;
                // ---- prefix sum, alpha, blend; clears as it reads.  Four rows per step so their LDS round trips overlap
#pragma unroll
                for (int r4 = 0; r4 < STRIP_H; r4 += P2B) {
                    if (r4 + P2B <= row_lo || r4 >= row_hi) continue;    // wave-uniform
                    if (dbg == 4) continue;
                    // straight-line over the four rows (no per-row branches) so that their LDS round trips and DPP scan
                    // chains interleave; a row nothing was accumulated into scans zeros and leaves its pixels unchanged
                    int v[P2B], carry[P2B];
#pragma unroll
                    for (int u = 0; u < P2B; ++u) {
                        const int rr = r4 + u;
                        v[u] = A[rr][lane];
                        carry[u] = A[rr][ACC_CARRY];
                    }
#pragma unroll
                    for (int u = 0; u < P2B; ++u) {
                        const int rr = r4 + u;
                        A[rr][lane] = 0;
                        if (lane == 0) A[rr][ACC_CARRY] = 0;
                    }
                    uint32_t al[P2B];
#pragma unroll
                    for (int u = 0; u < P2B; ++u) {
                        const int ua = (v[u] << 12) >> 12;             // low 20 bits, sign-extended
                        int ch = (v[u] - ua) >> 20;
                        if (lane == 0) ch += carry[u];
                        const int scan = wave_scan_incl(ch);
                        const int area = scan * 512 - ua;
                        al[u] = (uint32_t)((((area << 4) + area) + 256) >> 9) & 255u;   // area * 17 without a 64-bit multiply-add
                        if (cx < e_xmin || cx >= e_xmax) al[u] = 0;
                    }
                    bool blended = false;
                    if (!SHADERS && P2B == STRIP_H && (eflags & BE_LERP) && dbg != 14) {
                        // Solid colour, SOURCE-lerp: a pixel with coverage 255 takes the colour, one with 0 keeps its own, and only
                        // the few edge pixels need the two rounded products.  Those are queued -- {coverage, pixel} through the
                        // (now empty) accumulator -- and blended with lanes = queued pixels: one pass for the strip's eight rows
                        // instead of one masked pass per row.
                        unsigned long long pmask[P2B];
                        int qbase[P2B], nq = 0;
#pragma unroll
                        for (int u = 0; u < P2B; ++u) {
                            pmask[u] = __ballot(al[u] - 1u < 254u);
                            qbase[u] = nq;
                            nq += (int)__popcll(pmask[u]);
                            px[u] = al[u] == 255u ? solid : px[u];
                        }
                        if (nq <= (dbg == 15 ? 12 : BLEND_QUEUE)) {            // wave-uniform; more edge pixels: the per-row path below
                            uint2* q = reinterpret_cast<uint2*>(&A[0][0]);
                            // (the slot of a queued pixel is recomputed when its result is read back: eight live registers less)
#pragma unroll
                            for (int u = 0; u < P2B; ++u) {
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
                            for (int u = 0; u < P2B; ++u) {
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
                    for (int u = 0; u < P2B; ++u) {
                        const int rr = r4 + u;
                        if (SHADERS) { if (al[u]) px[rr] = blend_pixel(px[rr], al[u], eflags, solid, styles, style, bitmaps, cx, ty0 + rr); }
                        else { const uint32_t b = blend_pixel(px[rr], al[u], eflags, solid, styles, style, bitmaps, cx, ty0 + rr); px[rr] = al[u] ? b : px[rr]; }
                    }
                    }
                }
                __syncthreads();                                   // acc cleared before the next path accumulates
                PHASE(ph_p2);
            } else {
                // full cover: every in-frame pixel of the tile has coverage 255
#pragma unroll
                for (int rr = 0; rr < STRIP_H; ++rr)
                    if (rr >= row_lo && rr < row_hi) px[rr] = blend_pixel(px[rr], 255u, eflags, solid, styles, style, bitmaps, cx, ty0 + rr);
            }
        }
    }

    if (dbg == 8) {                                           // diagnostics (tools/strip_times.py): first pixels of the strip's first row
        uint32_t d[16] = {};
        d[0] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start); d[1] = dbg_pairs; d[2] = dbg_recs;
#ifdef SWFR_PHASES
        d[3] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - w_start); d[4] = (uint32_t)w_start; d[5] = (uint32_t)(w_start >> 32);
        d[8] = ph_bin; d[9] = ph_batch; d[10] = ph_stage; d[11] = ph_acc; d[12] = ph_p2;
#endif
#pragma unroll
        for (int k = 0; k < 16; ++k) if (lane == k) px[0] = d[k];
    }
    // ---- one store per pixel: premultiplied R,G,B,A bytes; the wave writes 256 contiguous bytes per row
    if (cx < width) {
        uint32_t* rowp = fb + (size_t)ty0 * (size_t)width + cx;   // one 64-bit multiply per strip, then pointer steps
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

#undef blend_pixel
#undef PHASE

// Two instances: solid-colour scenes (no shader call; the register budget is capped so that six wavefronts fit a SIMD --
// measured best on S1, +5 % with two spilled registers; five or seven are slower) and scenes with gradient / bitmap
// styles (uncapped: the f64 shader would spill).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void k_tiles_solid(
    const swfr_edge* __restrict__ raw_edges, const uint32_t* __restrict__ band_off, const BandEntry* __restrict__ band_list,
    const uint8_t* __restrict__ cls_t, const RowInfo* __restrict__ rows, const Rec* __restrict__ records,
    const swfr_style* __restrict__ styles, const Sources bitmaps, uint32_t* __restrict__ fb, int width, int height, int tiles_x,
    uint32_t band_index, uint32_t band_count, int dbg, uint32_t* __restrict__ counters, uint32_t n_rows_total, uint32_t n_rec_cap,
    const uint32_t* __restrict__ order) {
    tiles_body<false>(raw_edges, band_off, band_list, cls_t, rows, records, styles, bitmaps, fb, width, height, tiles_x, band_index, band_count, dbg,
                      counters, n_rows_total, n_rec_cap, order);
}
__global__ __launch_bounds__(64) void k_tiles_shaded(
    const swfr_edge* __restrict__ raw_edges, const uint32_t* __restrict__ band_off, const BandEntry* __restrict__ band_list,
    const uint8_t* __restrict__ cls_t, const RowInfo* __restrict__ rows, const Rec* __restrict__ records,
    const swfr_style* __restrict__ styles, const Sources bitmaps, uint32_t* __restrict__ fb, int width, int height, int tiles_x,
    uint32_t band_index, uint32_t band_count, int dbg, uint32_t* __restrict__ counters, uint32_t n_rows_total, uint32_t n_rec_cap,
    const uint32_t* __restrict__ order) {
    tiles_body<true>(raw_edges, band_off, band_list, cls_t, rows, records, styles, bitmaps, fb, width, height, tiles_x, band_index, band_count, dbg,
                     counters, n_rows_total, n_rec_cap, order);
}

// ---------------------------------------------------------------------------------------------
// auxiliary kernels
// ---------------------------------------------------------------------------------------------
// un-premultiply (node-canvas getImageData / PNG encode): c' = (c*255 + a/2) / a, a == 0 -> 0
__global__ __launch_bounds__(256) void k_unpremultiply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = in[i];
    const uint32_t a = p >> 24;
    if (a == 0) { out[i] = 0; return; }
    const uint32_t r = ((p & 255u) * 255u + a / 2) / a, g = (((p >> 8) & 255u) * 255u + a / 2) / a, b = (((p >> 16) & 255u) * 255u + a / 2) / a;
    out[i] = (a << 24) | (b << 16) | (g << 8) | r;
}

// pack this rank's tile-rows (t % band_count == band_index) into a dense slab for the RCCL gather
__global__ __launch_bounds__(256) void k_pack_band(const uint32_t* __restrict__ fb, uint32_t* __restrict__ slab, int width, int height,
                                                   uint32_t band_index, uint32_t band_count, uint32_t n_tile_rows_local) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_tile_row = (size_t)width * TILE_H;
    if (i >= per_tile_row * n_tile_rows_local) return;
    const uint32_t lt = (uint32_t)(i / per_tile_row);
    const size_t within = i % per_tile_row;
    const size_t y = (size_t)(lt * band_count + band_index) * TILE_H + within / width;
    slab[i] = y < (size_t)height ? fb[y * width + within % width] : 0u;
}

// ---------------------------------------------------------------------------------------------
// launchers (called from renderer.cpp, which is compiled as plain C++ by the same hipcc)
// ---------------------------------------------------------------------------------------------
void launch_front(hipStream_t st, const swfr_edge* in, const DevPath* paths, DevEdge* out, uint32_t n_edges, const BandSlot* slots,
                  uint32_t n_slots, const uint32_t* row_base, const swfr_style* styles, BandEntry* band_list, uint32_t* counters) {
    uint32_t n_setup = (n_edges + 255) / 256;
    const uint32_t n_b = (n_slots + 255) / 256;
    if (n_setup + n_b == 0) n_setup = 1;          // counters are still cleared
    hipLaunchKernelGGL(k_front, dim3(n_setup + n_b), dim3(256), 0, st, in, paths, out, n_edges, n_setup, slots, n_slots, row_base, styles,
                       band_list, counters);
}
void launch_rows(hipStream_t st, const DevEdge* edges, const DevPath* paths, const uint32_t* row_base, const ChunkInfo* chunk_base,
                 uint32_t n_paths, RowInfo* rows, Rec* records, uint32_t* counters, const BigRow* big_rows, uint32_t n_big, uint32_t n_chunks,
                 uint32_t band_index, uint32_t band_count, int fast_limit, int cell_mode, uint32_t chunk_rows,
                 const BandSlot* band_slots, const uint32_t* band_off, uint8_t* cls_t, int width, int height, int fused,
                 const swfr_edge* raw, const swfr_style* styles, BandEntry* band_list, const BigRow* huge_rows, uint32_t n_huge) {
    if (!n_chunks) return;
    if (n_huge)
        hipLaunchKernelGGL(k_rows_huge, dim3(n_huge), dim3(256), 0, st, edges, paths, row_base, huge_rows, n_huge, rows, records, counters, cell_mode);
    fast_limit = fast_limit < 0 ? 0 : (fast_limit > ROWS_FAST_N ? ROWS_FAST_N : fast_limit);
    if (chunk_rows <= 8)
        hipLaunchKernelGGL(k_rows_rs, dim3(n_chunks + n_big), dim3(64), 0, st, edges, paths, row_base, chunk_base, rows, records, band_index, band_count,
                           fast_limit, cell_mode, big_rows, n_big, counters);
    else
        hipLaunchKernelGGL(k_rows, dim3(n_chunks + n_big), dim3(64), 0, st, edges, paths, row_base, chunk_base, n_paths, rows, records,
                           band_index, band_count, fast_limit, cell_mode, big_rows, n_big, counters, band_slots, band_off, cls_t, width, height, fused,
                           raw, styles, band_list);
}
void launch_class(hipStream_t st, const BandEntry* band_list, uint32_t n_entries, const uint32_t* band_off, uint32_t n_bands,
                  const swfr_edge* raw, const RowInfo* rows, const Rec* records, uint8_t* cls_t, int width, int height,
                  uint32_t band_index, uint32_t band_count) {
    if (!n_entries) return;
    const int tiles_x = (width + TILE_W - 1) / TILE_W;
    hipLaunchKernelGGL(k_class, dim3(n_entries), dim3(64), 0, st, band_list, n_entries, band_off, n_bands, raw, rows, records, cls_t,
                       width, height, tiles_x, band_index, band_count);
}
void launch_tiles(hipStream_t st, const swfr_edge* raw, const uint32_t* band_off, const BandEntry* band_list, const uint8_t* cls_mat,
                  const RowInfo* rows, const Rec* records, const swfr_style* styles, const Sources bitmaps, uint32_t* fb, int width, int height,
                  uint32_t band_index, uint32_t band_count, int dbg, uint32_t* counters, uint32_t n_rows_total, uint32_t n_rec_cap,
                  bool any_shader, const uint32_t* order) {
    const int tiles_x = (width + TILE_W - 1) / TILE_W, tile_rows = (height + TILE_H - 1) / TILE_H;
    uint32_t local_rows = tile_rows;
    if (band_count > 1) local_rows = (tile_rows > (int)band_index) ? (tile_rows - band_index + band_count - 1) / band_count : 0;
    if (!local_rows) return;
    if (any_shader)
        hipLaunchKernelGGL(k_tiles_shaded, dim3(tiles_x * local_rows * STRIPS_PER_TILE), dim3(64), 0, st, raw, band_off, band_list, cls_mat, rows,
                           records, styles, bitmaps, fb, width, height, tiles_x, band_index, band_count, dbg, counters, n_rows_total, n_rec_cap, order);
    else
        hipLaunchKernelGGL(k_tiles_solid, dim3(tiles_x * local_rows * STRIPS_PER_TILE), dim3(64), 0, st, raw, band_off, band_list, cls_mat, rows,
                           records, styles, bitmaps, fb, width, height, tiles_x, band_index, band_count, dbg, counters, n_rows_total, n_rec_cap, order);
}
void launch_unpremultiply(hipStream_t st, const uint32_t* in, uint32_t* out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_unpremultiply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}
void launch_pack_band(hipStream_t st, const uint32_t* fb, uint32_t* slab, int width, int height, uint32_t band_index,
                      uint32_t band_count, uint32_t local_rows) {
    const size_t n = (size_t)width * TILE_H * local_rows;
    if (!n) return;
    hipLaunchKernelGGL(k_pack_band, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fb, slab, width, height, band_index,
                       band_count, local_rows);
}

}  // namespace swfr
