// raster_common.hip -- device functions shared by the kernels of raster2.hip (which includes this file; it is not a translation
// unit of its own): the exact integer helpers of the scan converter (floor division with remainder, an edge's x at a sample
// row, cell positions), make_dev_edge (SURVEY.md A.5 make_edge), the wave64 DPP prefix sum, the replay of the order Cairo's edge
// list gives coincident edges (same_line .. tied_order, row_was_sampled), the Cairo / pixman blends (0x7f lerp, 0x80 over) and
// shaders (gradients in double precision, bitmaps by bilinear or separable-convolution sampling at pixman's 16.16 positions), and
// the two small framebuffer kernels (un-premultiply for read-back, band slab packing).
// Integer arithmetic is exact (int64 products, double-estimated quotients with integer fix-up), so results are bit-identical to
// the CPU scan converter.  No MFMA: there is no dense contraction on this path; the roof is HBM bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#ifdef SWFR_EMU
#include <cstdio>
#include <cstdlib>
#endif

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
__device__ __forceinline__ uint32_t clamp_col(int c) { return (uint32_t)min(max(c, 0), 65535); }
// net covered height a record adds to everything right of it
__device__ __forceinline__ int record_height(uint32_t roles) {
    if (roles & REC_CELLS) return (int)(int8_t)(roles >> 8);
    if (roles & REC_FULL) return (roles & 1u) ? 15 : -15;
    return __popc(roles & 0x15555555u) - __popc(roles & 0x2aaaaaaau);
}

// 24-bit signed multiply at the full VALU rate (v_mul_i32_i24; `a * b` on 32-bit operands is v_mul_lo_u32: a quarter of the rate)
__device__ __forceinline__ int mul_i24(int a, int b) {
#ifdef SWFR_EMU
    return a * b;
#else
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
// (a << SH) + b as ONE full-rate instruction (the compiler canonicalises shift-adds by a constant into a multiply, i.e. v_mul_lo_u32)
template <int SH>
__device__ __forceinline__ int lshl_add(int a, int b) {
#ifdef SWFR_EMU
    return (int)(((uint32_t)a << SH) + (uint32_t)b);
#else
    int r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "i"(SH), "v"(b));
    return r;
#endif
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
    d.inv_dx = 0.0; d.fr = 0; d.r15 = 0; d.fq = 0; d.q15 = 0;
    if (p.kind != SWFR_PATH_TOR) {          // boxes are consumed raw by k2_tiles
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
        const int64_t adx = (d.ex < 0 ? -d.ex : d.ex) * 7680;
        d.inv_dx = 1.0 / (double)adx;
        int64_t fq; floor_div_inv(3840 * d.dy, adx, d.inv_dx, fq, d.fr);     // (estimates from the reciprocals, exact after the integer fix-up)
        d.fq = (int32_t)fq;                                   // (dy < 2^37, |ex| >= 256: below 2^28)
        if (e.y2 - e.y1 >= 200) {                             // only an edge that crosses a whole pixel row (>= 14/15 px tall) is ever stepped by a row
            int64_t q15; floor_div_inv(d.ex * 7680, d.dy, d.inv_dy, q15, d.r15);
            d.q15 = (int32_t)q15;                             // (|ex| <= 2^32, dy >= 200 * 7680: below 2^25)
        }
    }
    return d;
}

// ---- the 32-bit form of the same edge (FastEdge, device_types.hpp) for the fast row routine
__device__ __forceinline__ FastEdge make_fast_edge(const swfr_edge& e, const DevPath& p) {
    FastEdge f;
    f.x1 = e.x1; f.a0 = 256 - 30 * e.y1; f.DX = 0; f.D = 30; f.invD = 1.0 / 30.0; f.q15 = f.r15 = f.hq = f.hr = f.dqf = f.drf = 0;
    {   // active sample rows, clamped to the path's rows (as make_dev_edge)
        int ytop = (int)((15ll * e.top + 128) >> 8), ybot = (int)((15ll * e.bottom + 128) >> 8);
        ytop = max(ytop, p.y_min * 15);
        ybot = min(ybot, p.y_max * 15);
        if (ybot <= ytop) { ytop = ybot = 0; }
        f.ytop = ytop; f.ybot = ybot;
    }
    f.dir = e.dir; f.fq = 0; f.invW = 0.0; f.fr = 0.0;
    if (p.kind != SWFR_PATH_TOR || e.y2 <= e.y1) { f.ytop = f.ybot = 0; return f; }
    const int64_t DX = (int64_t)e.x2 - e.x1, D = 30ll * ((int64_t)e.y2 - e.y1);
    f.D = (int32_t)D; f.invD = 1.0 / (double)D;
    if (DX == 0) return f;                                   // vertical: x = x1 everywhere (every step is 0)
    f.DX = (int32_t)DX;
    int64_t q, r;
    floor_div_inv(512 * DX, D, f.invD, q, r);                 // per sample row (estimates from the reciprocal, exact after the integer fix-up)
    f.dqf = (int32_t)q; f.drf = (int32_t)r;
    // Cairo halves its TRUNCATED step (quotient and remainder each truncated towards zero; the remainder 512 DX - q D is even)
    int64_t tq = q, tr = r;
    if (DX < 0 && r != 0) { tq = q + 1; tr = r - D; }
    int64_t hq = tq / 2, hr = tr / 2;
    if (hr < 0) { --hq; hr += D; }
    f.hq = (int32_t)hq; f.hr = (int32_t)hr;
    if (e.y2 - e.y1 >= 200) {                                 // only an edge that crosses a whole pixel row is ever stepped by one
        floor_div_inv(7680 * DX, D, f.invD, q, r);
        f.q15 = (int32_t)q; f.r15 = (int32_t)r;
    }
    const int64_t W = 512 * (DX < 0 ? -DX : DX);
    f.invW = 1.0 / (double)W;
    floor_div_inv(256 * D, W, f.invW, q, r);
    f.fq = (int32_t)q; f.fr = (double)r;
    return f;
}
// x of the edge at the centre of sample row s, relative to x1: quo + rem / D with rem in [0, D).  A * DX < 2^53 is exact in double
// precision, the quotient estimate is off by at most one, and the remainder of the estimate (an fma: exact) fits 32 bits.
__device__ __forceinline__ void fast_x_at(int32_t a0, int32_t DX, int32_t D, double invD, int s, int32_t& quo, int32_t& rem) {
    const double n = (double)(512 * s + a0) * (double)DX;
    const double qf = floor(n * invD);
    const double rf = fma(-qf, (double)D, n);
    int32_t q = (int32_t)qf, r = (int32_t)rf;
    if (r < 0) { --q; r += D; }
    if (r >= D) { ++q; r -= D; }
    quo = q; rem = r;
}
// the FastEdge array of a frame sits behind its DevEdge array (the host reserves 2 n + 1 DevEdge records: 80 <= 96 bytes each)
__device__ __forceinline__ FastEdge* fast_edges_of(DevEdge* edges, uint32_t n_edges) { return reinterpret_cast<FastEdge*>(edges + n_edges); }

// ---------------------------------------------------------------------------------------------
// row kernels: classification bits, capacities, the order of coincident edges
// ---------------------------------------------------------------------------------------------
// classification of a (tile, path) pair
#define CLS_PARTIAL 1u                // some row needs the general accumulate + scan path
#define CLS_NOTFULL 2u                // some in-frame row of the tile is not uniformly alpha 255
#define CLS_NONEMPTY 4u               // some row has coverage
#define CLS_HOLE 16u                  // (inside the classifiers only) a row of the path has no coverage in this tile: with CLS_NONEMPTY
                                      // from another row the tile is partial even if no single pixel is -- e.g. the last row of a path
                                      // whose bottom lies less than a sample row below a pixel boundary
#define CLS_BOX 8u                    // rectilinear path evaluated per pixel from its boxes

#define ROWS_FAST_N 8            // active edges per row handled in registers by k2_rows
#define ROWS_FAST_WIDE 16         // ... by its second instance (k2_rows_wide: scenes with a path of more than ROWS_STAGE edges)
#define ROWS_BIG_MAXA 64         // capacity of the generic (LDS list) routine of k2_rows_slow
#ifndef ROWS_STAGE
#define ROWS_STAGE 32            // chunks with at most this many edges of their path in reach are staged into LDS (3 KB of 96-byte records:
#endif                           // with 4.5 KB -- 48 of them -- two wavefronts fewer fit a CU and the pipelined frame rate drops by a tenth)
#define ROWS_STAGE_WIDE 64       // ... of the row kernel's second instance, for scenes with a path of more than ROWS_STAGE edges

// All edges of one path, whichever form the kernel has them in (k_front's DevEdge array, or the raw edges when the row pass
// computes the constants itself).
struct PathEdges {
    const DevEdge* dev; const swfr_edge* raw; const DevPath* P; bool from_raw;
    uint32_t* diag;          // counters; the replay's capacity limits are counted there (the frame then fails loudly)
    __device__ __forceinline__ void limit_hit(uint32_t which) const { if (diag) atomicOr(&diag[which], 1u); }
    // the row headers the row kernel has already written say how an earlier row of this path was converted
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
    if (PE.rows2) return PE.dev[PE.P->first_edge + ka].pad < PE.dev[PE.P->first_edge + kb].pad;   // k2_start_ranks has replayed the sort
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
// sample row (the sort / merge rules above); two older edges are ordered by their own history, DEPTH levels deep (two: tied_order);
// beyond that, and when (active edges of the row) x (edges of the path) exceeds 2^21 -- thousands of edges in one row -- the limit is
// counted (PathEdges::limit_hit) and the frame refused.
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
// two levels of history behind the history: whether an earlier row was sampled may itself hinge on a tie of two older edges
__device__ __forceinline__ bool tied_order(const PathEdges& PE, const DevEdge& a, const DevEdge& b, uint32_t ka, uint32_t kb, int s0, bool path_order) {
    return tied_order_at<2>(PE, a, b, ka, kb, s0, path_order);
}

// Rows with more than ROWS_BIG_MAXA (64) and up to ROWS_HUGE_MAXA (8192) active edges of one path (a line of text outlines
// filled with one style, hatching): one 1024-thread workgroup per row; thread t owns the active edges t, t + 1024, ... (path
// order), the sort keys of all of them sit in LDS and every owned edge is ranked against them.  Same decisions as big_row_body;
// launched only when the host listed such rows.
#define ROWS_HUGE_MAXA 8192
#define HUGE_THREADS 1024
#define ROWS_HUGE_EPT (ROWS_HUGE_MAXA / HUGE_THREADS)
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

// ---------------------------------------------------------------------------------------------
// k2_tiles helpers: blending (A.7), shading
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
        x = (x & ~((1ll << xsh) - 1)) + ((1 << xsh) >> 1);          // the middle of the closest phase
        y = (y & ~((1ll << ysh) - 1)) + ((1 << ysh) >> 1);
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
