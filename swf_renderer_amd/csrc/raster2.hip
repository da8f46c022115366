// raster2.hip -- the kernels of the hot path (raw edge list -> RGBA8 framebuffer in HBM), CDNA4 / gfx950.  The build's only device
// translation unit; raster_common.hip (included below) holds the exact-arithmetic helpers, the replay of Cairo's edge-list order
// and the shaders the kernels share.  Arithmetic: Cairo 1.16 "tor" scan conversion + pixman sampling.
//
// One frame = four launches on one stream (DESIGN.md section 3 has the data layout and the bytes):
//   k2_bin     binning on the device from the raw edge list: scan-converter constants per edge (A.5 make_edge), the row chunks of
//              every path, the band lists (paths per tile-row, painter's order), box class bytes, and -- its last workgroup -- the
//              tile pass's launch list: per XCD class (tile-row % 8) the strips heaviest first by the previous frame's costs
//   k2_rows    one wavefront per (path, <= 64 pixel rows), lane = row: active edges, FULL / SUB decision, roles -- and then the
//              row's CELLS {column, covered height, uncovered area}, i.e. Cairo's cell list itself (A.5 render_edge /
//              add_subspan) in a fixed region per chunk (no allocator), plus the class byte of every (tile, path) pair and the
//              strips' costs.  A row whose edge order needs Cairo's list history (coincident edges on different lines), or with
//              more than eight active edges, goes to a queue.
//   k2_start_ranks / k2_rows_slow / k2_rows_huge   the queued rows (launched only while a resident scene may have any): the order
//              Cairo's bucket sort gives edges that start together, then one wavefront (<= 64 active edges) or one 256-thread
//              workgroup (<= 8192) per row, up to SLOW_PASSES passes (a row whose history runs through another queued row waits)
//   k2_tiles   one wavefront per 64x8-pixel strip of the launch list: class bytes -> surviving band entries (everything under the
//              last opaque full cover is culled from the class bytes alone) -> the cells of the partial paths as one coalesced
//              stream -> LDS accumulators -> wave64 prefix sum -> alpha -> shade -> blend in registers -> one store per pixel.
//              No edge arithmetic, no 64-bit or floating-point instruction on the solid-colour path.  Three instances: solid
//              colours only, + bitmaps, + gradients; a scene gets the lightest that covers its styles.
//
// Every kernel takes an array of frame descriptors (Frame2, device memory, read through scalar loads) and blockIdx.y picks the
// frame: a batch of frames is one launch per kernel (swfr_render_batch).
#include "raster_common.hip"

namespace swfr {

// Barrier for hand-offs through LDS only.  __syncthreads() also waits for every outstanding global store (s_waitcnt vmcnt(0)), which
// costs a full memory round trip wherever a wavefront has just written cells or pixels; this one orders LDS traffic alone.
__device__ __forceinline__ void lds_barrier() {
#ifdef SWFR_EMU
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

// Kernels get their frame's descriptor as the blockIdx.y-th element of an array in device memory and read its fields through the
// constant address space: wave-uniform scalar loads, issued where a field is first needed, like those of kernel arguments (through
// a plain pointer the compiler uses vector loads and every field costs a vector register).
#ifdef SWFR_EMU
typedef const Frame2* FramePtr;
#define FRAME_PTR(frames, i) ((frames) + (i))
#else
typedef const Frame2 __attribute__((address_space(4)))* FramePtr;
#define FRAME_PTR(frames, i) ((FramePtr)((frames) + (i)))
#endif

#ifdef SWFR_TSTATS                 // -DSWFR_TSTATS (diagnostic builds, tools/tile_stats.py): per-strip work counts of k2_tiles through the trace buffer
#define TRACE_WGS 32768
__device__ uint32_t swfr_trace_buf[3][TRACE_WGS][8];
#define TRACE_DECL uint32_t tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TRACE(i) do { } while (0)
#define TRACE_NOWAIT(i) do { } while (0)
#define STAT(i, x) do { tr_[i] += (uint32_t)(x); } while (0)
#define TRACE_OUT(k, slot) do { if ((k) == 2 && (threadIdx.x & 63) == 0 && (slot) < TRACE_WGS) { tr_[7] = 1; for (int i_ = 0; i_ < 8; ++i_) swfr_trace_buf[k][slot][i_] = tr_[i_]; for (int i_ = 0; i_ < 8; ++i_) tr_[i_] = 0; } } while (0)
#elif defined(SWFR_TRACE)         // -DSWFR_TRACE (diagnostic builds, tools/trace_wg.py): 100 MHz wall-clock stamps per workgroup of the three kernels
#define TRACE_WGS 32768
__device__ uint32_t swfr_trace_buf[3][TRACE_WGS][8];
#define TRACE_DECL uint32_t tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TRACE(i) do { __builtin_amdgcn_s_waitcnt(0); tr_[i] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
#define TRACE_NOWAIT(i) do { tr_[i] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
#define TRACE_OUT(k, slot) do { if ((threadIdx.x & 63) == 0 && (slot) < TRACE_WGS) { for (int i_ = 0; i_ < 8; ++i_) swfr_trace_buf[k][slot][i_] = tr_[i_]; } } while (0)
#else
#define TRACE_DECL do { } while (0)
#define TRACE(i) do { } while (0)
#define TRACE_NOWAIT(i) do { } while (0)
#define TRACE_OUT(k, slot) do { } while (0)
#endif
#ifndef STAT
#define STAT(i, x) do { } while (0)
#endif
// the handle's share of the frame's tile-rows (multi-GPU): local tile-row l is frame tile-row band_first + l * band_stride
__device__ __forceinline__ uint32_t local_band_rows(FramePtr FR) { return FR->n_strips / (STRIPS_PER_TILE * (uint32_t)FR->tiles_x); }
__device__ __forceinline__ bool owns_band(FramePtr FR, int band, uint32_t& local) {
    const int d = band - (int)FR->band_first;
    local = (uint32_t)d / FR->band_stride;
    return d >= 0 && (uint32_t)d % FR->band_stride == 0u && local < local_band_rows(FR);
}

#define CLS_OPAQUE 32u                // the path is an opaque solid blended with the lerp rule: a full cover of it hides what lies below

// a (path, strip) pair's class, noted in the strip's StripTop record (device_types.hpp): pos1 = the entry's position in its tile-row's
// band list + 1; device-scope atomics are resolved behind the XCDs' L2s, so workgroups on different XCDs agree
__device__ __forceinline__ void strip_top_note(FramePtr FR, uint32_t strip_id, uint32_t pos1, uint32_t f, uint32_t pixel) {
    if (!(f & CLS_NONEMPTY)) return;
    StripTop* t = FR->strip_top + strip_id;
    atomicMax(&t->any, pos1);
    if ((f & 0x3fu) == (CLS_NONEMPTY | CLS_OPAQUE)) atomicMax(&t->cover, ((unsigned long long)pos1 << 32) | (unsigned long long)pixel);
}

// ---------------------------------------------------------------------------------------------
// cells
// ---------------------------------------------------------------------------------------------
// one cell, clipped to the converter's column range [xminp, xmaxp): at / after the right bound it is never emitted (the slot is
// still written, as a zero), left of the left bound only its height counts, folded into the first column  (oracle: cell_add)
__device__ __forceinline__ void put_cell(Cell* __restrict__ dst, int col, int ch, int ua, int xminp, int xmaxp) {
    if (col >= xmaxp) { col = xmaxp - 1; ch = 0; ua = 0; }
    else if (col < xminp) { col = xminp; ua = 0; }
    *dst = make_cell(col - xminp, ch, ua);
}
// the same for a cell whose area is `area_per_h` times its height: V = h * (CELL_H + area_per_h), one multiplication
__device__ __forceinline__ void put_cell_h(Cell* __restrict__ dst, int col, int h, int area_per_h, int xminp, int xmaxp) {
    int v = mul_i24(h, CELL_H + area_per_h);                   // (|h| <= 15, the factor < 2^15: the full-rate 24-bit multiply)
    if (col >= xmaxp) { col = xmaxp - 1; v = 0; }
    else if (col < xminp) { col = xminp; v = h * CELL_H; }
    *dst = make_cell_v(col - xminp, v);
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
// inv_dx, fq0, fr0: the edge's constants (DevEdge): 1 / dx and floor_div(15 * 256 * edy, dx) -- dx, the row's x extent in units of
// 1 / edy, is 7680 * |ex| in every row the edge crosses completely
__device__ __forceinline__ void full_cells(int32_t q1, int64_t r1, int32_t q2, int64_t r2, int64_t edy, double inv_dx, int32_t fq0, int64_t fr0, int sign, int xminp, int xmaxp, Cell* __restrict__ dst) {
    int ix1 = q1 >> 8, f1 = q1 & 255, ix2 = q2 >> 8, f2 = q2 & 255;
    if (ix1 == ix2) { put_cell_h(dst, ix1, sign * 15, f1 + f2, xminp, xmaxp); return; }
    if (ix2 < ix1) { int t = ix1; ix1 = ix2; ix2 = t; t = f1; f1 = f2; f2 = t; int32_t tq = q1; q1 = q2; q2 = tq; int64_t tr = r1; r1 = r2; r2 = tr; }
    const int span = ix2 - ix1 + 1;
    const int64_t dx = (int64_t)(q2 - q1) * edy + (r2 - r1);
    const int64_t t0 = ((int64_t)((ix1 + 1) * 256 - q1) * edy - r1) * 15;
    int64_t yq, yr;
    floor_div_inv(t0, dx, inv_dx, yq, yr);
    const int64_t fq = fq0, fr = fr0;
#ifdef SWFR_EMU
    {   // (emulator builds check the per-edge constants against the row's own numbers)
        int64_t cq, cr2; floor_div(15ll * 256 * edy, dx, cq, cr2);
        int64_t dq2, dr2; floor_div(t0, dx, dq2, dr2);
        if (cq != fq || cr2 != fr || dq2 != yq || dr2 != yr || inv_dx != 1.0 / (double)dx) { std::fprintf(stderr, "full_cells: per-edge row constants disagree\n"); std::abort(); }
    }
#endif
    int y_prev = (int)yq;
    if (span <= MAX_CELLS_PER_EDGE_ROW) {
        put_cell_h(dst, ix1, sign * y_prev, 256 + f1, xminp, xmaxp);
#pragma unroll 1
        for (int k = 1; k < span - 1; ++k) {
            yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; }
            const int h = (int)yq - y_prev;
            put_cell_h(dst + k, ix1 + k, sign * h, 256, xminp, xmaxp);
            y_prev = (int)yq;
        }
        put_cell_h(dst + span - 1, ix2, sign * (15 - y_prev), f2, xminp, xmaxp);
        return;
    }
    int n = 0;
    if (y_prev) put_cell_h(dst + n++, ix1, sign * y_prev, 256 + f1, xminp, xmaxp);
#pragma unroll 1
    for (int c = ix1 + 1; c < ix2; ++c) {
        yq += fq; yr += fr; if (yr >= dx) { ++yq; yr -= dx; }
        const int h = (int)yq - y_prev;
        if (h && n < MAX_CELLS_PER_EDGE_ROW - 1) put_cell_h(dst + n++, c, sign * h, 256, xminp, xmaxp);
        y_prev = (int)yq;
    }
    if (15 - y_prev) put_cell_h(dst + n++, ix2, sign * (15 - y_prev), f2, xminp, xmaxp);
    while (n < MAX_CELLS_PER_EDGE_ROW) put_cell(dst + n++, xminp, 0, 0, xminp, xmaxp);
}
// one end of a sample-row span at cell position x (24.8, already rounded to the sample grid): A.5 add_subspan
__device__ __forceinline__ void sub_cell(Cell* __restrict__ dst, int x, int sgn, int xminp, int xmaxp) {
    put_cell(dst, x >> 8, sgn, sgn * 2 * (x & 255), xminp, xmaxp);
}
// `n` cells for the calling wavefront of a slow-row kernel (one lane allocates; every lane gets the base): a bump allocator over the
// part of the frame's arena behind the chunk wavefronts' region.  ~0u when it is full (the frame then fails loudly).
__device__ __forceinline__ uint32_t alloc_cells(FramePtr FR, uint32_t n, int lane) {
    uint32_t base = 0;
    if (lane == 0 && n) {
        const uint32_t old = atomicAdd(&FR->counters[C2_HEAD], n);
        base = FR->cell_main + old;
        if ((uint64_t)base + n > FR->cell_slice) { atomicOr(&FR->counters[C2_ERROR], E2_CELL_ARENA); base = ~0u; }
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
}

// ---------------------------------------------------------------------------------------------
// binning (per frame, on the device, from the raw edge list)
//   k2_bin    one workgroup per tile-row: the paths that touch it, in painter's order (ordered compaction over the path list) ->
//             band entries, (path, tile-row) -> entry table, class bytes of boxes paths; one thread per edge: the scan converter's
//             constants (A.5 make_edge); one thread per path: its row chunks; counters cleared
//   k2_rows   (below) also adds, per (tile, path) pair with a boundary in it, the boundary rows to the cost of the tile's strips
//             ... and the last workgroup of k2_bin sorts the strips by the costs of the previous frame rendered with the same buffers
//             (counting sort): the launch list of the tile pass, heaviest strips first
// The host contributes the LAYOUT of the tables only -- prefix sums over the paths' rectangles (chunks, band slots) and row spans
// (cells), and over the tile-rows (band list offsets): O(paths) additions, no per-row, per-edge-row or per-tile work.
// ---------------------------------------------------------------------------------------------
// style i of the frame (whole records or their {kind, pixel} heads: Frame2::style_stride)
__device__ __forceinline__ const swfr_style& style_at(FramePtr FR, uint32_t i) {
    return *reinterpret_cast<const swfr_style*>(reinterpret_cast<const char*>(FR->styles) + (size_t)i * FR->style_stride);
}
__device__ __forceinline__ BandEntry2 make_band_entry2(const DevPath& P, uint32_t p, uint32_t band, FramePtr FR) {
    BandEntry2 e;
    e.x_min = (int16_t)P.x_min; e.x_max = (int16_t)P.x_max; e.y_min = (int16_t)P.y_min; e.y_max = (int16_t)P.y_max;
    e.style = P.style; e.first_edge = P.first_edge; e.n_edges = P.n_edges; e.path = p;
    const uint32_t kind = style_at(FR, P.style).kind, pixel = style_at(FR, P.style).pixel;
    uint32_t fl = 0;
    if (P.kind == SWFR_PATH_BOXES) fl |= BE_BOXES;
    if (P.lerp) fl |= BE_LERP;
    if (kind == SWFR_STYLE_SOLID) fl |= BE_SOLID;
    if (kind == SWFR_STYLE_SOLID && P.lerp && (pixel >> 24) == 0xffu) fl |= BE_OPAQUE_COVER;
    e.flags = fl | (band << 8); e.solid = pixel;
    return e;
}
__device__ __forceinline__ bool path_has_area(const DevPath& P) { return P.y_max > P.y_min && P.x_max > P.x_min; }
// first pixel row of a tor path's first chunk (chunks are whole tile-rows) and how many chunks it has
__device__ __forceinline__ uint32_t path_chunk_count(const DevPath& P, uint32_t chunk_rows, uint32_t& a0) {
    a0 = (uint32_t)P.y_min / TILE_H * TILE_H;
    if (P.kind != SWFR_PATH_TOR || P.y_max <= P.y_min) return 0u;
    return ((uint32_t)P.y_max - a0 + chunk_rows - 1) / chunk_rows;
}

__device__ __forceinline__ void order_body(FramePtr F, uint32_t xcd_class);
constexpr uint32_t BIN_THREADS = 1024;
constexpr uint32_t BAND_PER_THREAD = 16;                                  // paths a thread tests per round
constexpr uint32_t BAND_LIST = 4096;                                      // hits of a round staged at a time
__device__ __forceinline__ void bin_body(FramePtr F, uint32_t slow_kernels) {
    __shared__ uint32_t wave_cnt[BIN_THREADS / 64];
    __shared__ uint32_t band_hits[BAND_LIST];                // a tile-row's hits of one round, in painter's order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && tid < C2_WORDS) F->counters[tid] = 0;
    if (blockIdx.x >= gridDim.x - XCDS) { order_body(F, blockIdx.x - (gridDim.x - XCDS)); return; }   // the last eight workgroups: the tile pass's launch list
    if (blockIdx.x < F->n_bands) {
        // ---- the paths that touch tile-row `band`, in painter's order
        const int band = (int)blockIdx.x;
        const uint32_t b0 = F->band_off[band], n_b = F->band_off[band + 1] - b0;
        // Two steps per round of BAND_PER_THREAD * 1024 paths (one round for a scene of up to 16 384 paths).  Step 1: every thread tests
        // sixteen consecutive paths against the host's table of tile-row spans (four bytes a path, 64 contiguous bytes a thread) and
        // the hits are compacted in painter's order into an LDS list: per-thread counts, one wavefront scan, sixteen wavefront totals.
        // Step 2: one thread per HIT fetches the path and its style and writes the band entry -- every hit's chain of dependent loads
        // runs side by side (round 4, before: four paths per thread and three rounds of the whole chain for S2's 10 000 paths, 22 us).
        uint32_t n = 0;
        const uint32_t n_paths = F->n_paths;
        for (uint32_t base = 0; base < n_paths && n < n_b; base += BAND_PER_THREAD * BIN_THREADS) {    // (workgroup-uniform: stops when the list is complete)
            const uint32_t p0 = base + (uint32_t)tid * BAND_PER_THREAD;
            uint32_t mask = 0;
            if (p0 < n_paths) {                                          // (the table has sixteen "no tile-row" entries behind the last path)
                const uint4* t4 = reinterpret_cast<const uint4*>(F->path_bands + p0);
                const uint4 q[4] = {t4[0], t4[1], t4[2], t4[3]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) if ((v[i] & 0xffffu) <= (uint32_t)band && (uint32_t)band <= (v[i] >> 16)) mask |= 1u << (4 * k + i);
                }
            }
            const uint32_t cnt = (uint32_t)__popc(mask);
            const uint32_t incl = (uint32_t)wave_scan_incl((int)cnt);
            if (lane == 63) wave_cnt[wave] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < (int)(BIN_THREADS / 64); ++w) { const uint32_t t = wave_cnt[w]; if (w < wave) before += t; total += t; }
            const uint32_t first = before + incl - cnt;                  // this thread's first hit among the round's
            for (uint32_t lo = 0; lo < total; lo += BAND_LIST) {         // (workgroup-uniform; one window unless a tile-row has more than 4 096 hits in a round)
                uint32_t m = mask, idx = first;
                while (m) {
                    const uint32_t bit = (uint32_t)__ffs((int)m) - 1u; m &= m - 1u;
                    if (idx >= lo && idx < lo + BAND_LIST) band_hits[idx - lo] = p0 + bit;
                    ++idx;
                }
                __syncthreads();
                const uint32_t in_window = min(total - lo, BAND_LIST);
                uint32_t local_trow = 0;
                const bool own = owns_band(F, band, local_trow);             // (workgroup-uniform: the tile-row)
                const int ty0 = band * TILE_H, tile_y1 = min(ty0 + TILE_H, F->height);
                for (uint32_t ib = 0; ib < in_window; ib += BIN_THREADS) {   // (workgroup-uniform trip count: the box step below is a wavefront's)
                    const uint32_t i = ib + (uint32_t)tid;
                    uint32_t p = 0, at = 0;
                    bool valid = i < in_window;
                    if (valid) { p = band_hits[i]; at = n + lo + i; valid = at < n_b; }
                    int x_min = 0, x_max = 0;
                    uint32_t e_flags = 0, e_solid = 0, p_first = 0, p_nedges = 0;
                    bool box = false;
                    if (valid) {
                        const DevPath P = F->paths[p];
                        const uint32_t slot = b0 + at;
                        const BandEntry2 e = make_band_entry2(P, p, (uint32_t)band, F);
                        F->band_list[slot] = e;
                        BandSlot bs; bs.path = p; bs.slot = slot; bs.band = (uint32_t)band; bs.pad = 0;
                        F->band_slots[F->path_slots[p] + (uint32_t)(band - P.y_min / TILE_H)] = bs;
                        box = P.kind == SWFR_PATH_BOXES;
                        x_min = P.x_min; x_max = P.x_max; e_flags = e.flags; e_solid = e.solid; p_first = P.first_edge; p_nedges = P.n_edges;
                    }
                    // box paths: the class bytes and strip records of the hit's tile columns, shared by the wavefront -- (tile column,
                    // strip) pairs over the lanes (one thread looping over a full-frame rectangle's 120 was this kernel's long pole
                    // on a textured 4K frame: 27-40 us)
                    for (unsigned long long bm = __ballot(box); bm; bm &= bm - 1ull) {          // wave-uniform
                        const int src = __ffsll((long long)bm) - 1;
                        const uint32_t b_at = (uint32_t)__shfl((int)at, src), b_flags = (uint32_t)__shfl((int)e_flags, src), b_solid = (uint32_t)__shfl((int)e_solid, src);
                        const uint32_t b_first = (uint32_t)__shfl((int)p_first, src), b_nedges = (uint32_t)__shfl((int)p_nedges, src);
                        const int b_xmin = __shfl(x_min, src), b_xmax = __shfl(x_max, src);
                        uint8_t* out = F->cls + (size_t)STRIPS_PER_TILE * F->tiles_x * b0 + b_at;    // [tile column][strip][entry] inside the tile-row
                        const uint32_t opq = (b_flags & BE_OPAQUE_COVER) ? CLS_OPAQUE : 0u;
                        const swfr_edge bx = F->raw[b_first];                  // (only looked at when the path is a single box)
                        const bool one_box = b_nedges == 1 && bx.y1 <= ty0 * 256 && bx.y2 >= tile_y1 * 256;
                        const int tc_a = b_xmin / TILE_W, tc_b = (b_xmax - 1) / TILE_W;
                        const int items = (tc_b - tc_a + 1) * STRIPS_PER_TILE;
                        for (int k = lane; k < items; k += 64) {
                            const int tc = tc_a + k / STRIPS_PER_TILE, sp = k % STRIPS_PER_TILE;
                            const int tx0 = tc * TILE_W, tile_x1 = min(tx0 + TILE_W, F->width);
                            uint32_t f = CLS_BOX | CLS_NONEMPTY | CLS_NOTFULL;
                            if (one_box && bx.x1 <= tx0 * 256 && bx.x2 >= tile_x1 * 256) f = CLS_NONEMPTY | opq;     // the box contains the whole tile: full cover
                            out[(size_t)(tc * STRIPS_PER_TILE + sp) * n_b] = (uint8_t)f;
                            if (own && ty0 + sp * STRIP_H < F->height) strip_top_note(F, (local_trow * (uint32_t)F->tiles_x + (uint32_t)tc) * STRIPS_PER_TILE + (uint32_t)sp, b_at + 1u, f, b_solid);
                        }
                    }
                }
                __syncthreads();                                         // (the list is rewritten by the next window / round)
            }
            n += total;
            __syncthreads();                                             // (the wavefront totals are rewritten by the next round)
        }
        if (n != n_b && tid == 0) atomicOr(&F->counters[C2_ERROR], E2_ROW_TABLE);      // the host counted the same rectangles: cannot happen
        return;
    }
    // the workgroups behind the tile-rows': first one thread per edge, then -- in workgroups of their own, so that the two run side by
    // side (round 4: the workgroup that did both was the kernel's long pole) -- one thread per path
    // (batched launches: a frame with fewer tile-rows, edges or paths leaves blocks idle)
    const uint32_t rel = blockIdx.x - F->n_bands, edge_blocks = (F->n_edges + BIN_THREADS - 1) / BIN_THREADS;
    const bool path_block = rel >= edge_blocks;
    const uint32_t i = path_block ? ~0u : rel * BIN_THREADS + (uint32_t)tid;
    // ---- one thread per edge: scan converter constants
    if (i < F->n_edges) {
        const swfr_edge e = F->raw[i];
        const DevPath EP = F->paths[e.reserved];
        // (the 64-bit form is read by the queued-row kernels only: neither computed nor written for a scene known to have no queued rows)
        if (slow_kernels) F->edges[i] = make_dev_edge(e, EP);
        const FastEdge fe = make_fast_edge(e, EP);
        fast_edges_of(F->edges, F->n_edges)[i] = fe;
#ifdef SWFR_EMU
        const DevEdge de = make_dev_edge(e, EP);
        if (de.ytop != fe.ytop || de.ybot != fe.ybot) { std::fprintf(stderr, "FastEdge span disagrees with DevEdge\n"); std::abort(); }
        // (emulator builds: the 32-bit form gives Cairo's numbers -- quotient equal, remainder 1/256 of the 64-bit one)
        if (de.ybot > de.ytop && de.dy) {
            const int probe[3] = {de.ytop, (de.ytop + de.ybot) / 2, de.ybot};
            for (int t = 0; t < 3; ++t) {
                int32_t q0, q1, r1; int64_t r0;
                edge_x_at(de, probe[t], q0, r0);
                fast_x_at(fe.a0, fe.DX, fe.D, fe.invD, probe[t], q1, r1);
                if (q0 != fe.x1 + q1 || r0 != 256ll * r1) { std::fprintf(stderr, "fast_x_at disagrees with edge_x_at\n"); std::abort(); }
            }
            int64_t tq = fe.dqf, tr = fe.drf;
            if (fe.DX < 0 && tr != 0) { tq += 1; tr -= fe.D; }
            if (tq != de.dq || 256 * tr != de.dr || (e.y2 - e.y1 >= 200 && (fe.q15 != de.q15 || 256ll * fe.r15 != de.r15)) || fe.fq != de.fq) { std::fprintf(stderr, "FastEdge constants disagree with DevEdge\n"); std::abort(); }
        }
#endif
    }
    // ---- one thread per path: its chunk descriptors.  A path of many chunks (a frame of a few tall paths cut into 8-row chunks: a 1080-row
    //      path has 135) is written by its whole WAVEFRONT, a chunk per lane -- one thread looping over them was the kernel's long pole there
    const uint32_t ip = path_block ? (rel - edge_blocks) * BIN_THREADS + (uint32_t)tid : ~0u;
    if (!path_block) return;                                             // (wave-uniform: blockIdx decides)
    uint32_t nc = 0, a0 = 0, c0 = 0, inc0 = 0, slots0 = ~0u, first_edge = 0, n_path_edges = 0, band0 = 0;
    if (ip < F->n_paths) {
        const DevPath P = F->paths[ip];
        F->path_flag[ip] = 0;
        nc = path_chunk_count(P, F->chunk_rows, a0);
        c0 = F->path_chunks[ip]; inc0 = F->path_inc[ip];
        slots0 = path_has_area(P) ? F->path_slots[ip] : ~0u;
        first_edge = P.first_edge; n_path_edges = P.n_edges; band0 = (uint32_t)P.y_min / TILE_H;
    }
    auto write_chunk = [&](uint32_t path, uint32_t c, uint32_t a0_, uint32_t c0_, uint32_t inc0_, uint32_t slots0_, uint32_t fe_, uint32_t ne_, uint32_t band0_) {
        ChunkInfo ck;
        ck.path = path; ck.first_row = a0_ + c * F->chunk_rows; ck.rec_base = inc0_; ck.rows = F->chunk_rows;
        ck.slot0 = slots0_ != ~0u ? slots0_ + (ck.first_row / TILE_H - band0_) : ~0u;
        ck.first_edge = fe_; ck.n_edges = ne_; ck.pad = 0;
        if (c0_ + c < F->chunk_cap) F->chunks[c0_ + c] = ck;
    };
    constexpr uint32_t OWN_CHUNKS = 8;                                   // up to this many a thread writes itself
    if (nc <= OWN_CHUNKS)
        for (uint32_t c = 0; c < nc; ++c) write_chunk(ip, c, a0, c0, inc0, slots0, first_edge, n_path_edges, band0);
    for (unsigned long long big = __ballot(nc > OWN_CHUNKS); big; big &= big - 1ull) {    // wave-uniform
        const int src = __ffsll((long long)big) - 1;
        const uint32_t b_nc = (uint32_t)__shfl((int)nc, src), b_path = (uint32_t)__shfl((int)ip, src), b_a0 = (uint32_t)__shfl((int)a0, src);
        const uint32_t b_c0 = (uint32_t)__shfl((int)c0, src), b_inc0 = (uint32_t)__shfl((int)inc0, src), b_slots0 = (uint32_t)__shfl((int)slots0, src);
        const uint32_t b_fe = (uint32_t)__shfl((int)first_edge, src), b_ne = (uint32_t)__shfl((int)n_path_edges, src), b_band0 = (uint32_t)__shfl((int)band0, src);
        for (uint32_t c = (uint32_t)lane; c < b_nc; c += 64u) write_chunk(b_path, c, b_a0, b_c0, b_inc0, b_slots0, b_fe, b_ne, b_band0);
    }
}
// The launch list of the tile pass.  Slot of a strip: XCDS * (its rank among the strips of its class) + class, class = local
// tile-row % XCDS -- the hardware deals a launch's workgroups round-robin over the XCDs, so all strips of a tile-row run on one XCD
// and share its L2.  Ranks: heaviest first by the cost k2_rows added up during the PREVIOUS frame rendered with these buffers (a
// scheduling hint only -- a scene's first frame runs in row-major order); the costs are cleared for this frame's k2_rows.
// One 1024-thread workgroup per XCD class (the last XCDS workgroups of the k2_bin launch).  Round 4: a stable partition into eight
// cost ranges -- one round of loads, per-wavefront counts (eight byte counters in two words, one DPP scan each) in LDS, one scan by one
// wavefront, two barriers -- instead of
// a 128-bucket counting sort with returning LDS atomics (these workgroups were k2_bin's long pole: 9-11 us beside the others' 3-4).
constexpr uint32_t ORDER_RANGES = 8;
__device__ __forceinline__ uint32_t order_range(uint32_t cost) {          // 0 = heaviest
    return cost >= 48u ? 0u : cost >= 32u ? 1u : cost >= 24u ? 2u : cost >= 16u ? 3u : cost >= 8u ? 4u : cost >= 4u ? 5u : cost >= 1u ? 6u : 7u;
}
constexpr uint32_t ORDER_U = 16;                                           // strips per thread and round: 16 384 strips of a class per round
__device__ __forceinline__ void order_body(FramePtr F, uint32_t x) {
    __shared__ uint2 rowinfo[256];                         // per tile-row of the class: {first band list entry, entries}
    __shared__ uint32_t cnt[ORDER_RANGES][ORDER_U][BIN_THREADS / 64];      // strips per (cost range, round of 1024, wavefront); then their exclusive prefix
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_strips = F->n_strips, bc = F->band_stride, bi = F->band_first;
    const uint32_t per_row = STRIPS_PER_TILE * (uint32_t)F->tiles_x;
    const uint32_t n_local = n_strips / per_row;
    const uint32_t rows_x = (n_local + XCDS - 1 - x) / XCDS, max_rows = (n_local + XCDS - 1) / XCDS;     // tile-rows x, x + 8, ... of this class
    const uint32_t n_mine = rows_x * per_row;
    const bool by_cost = F->strip_order != 0u;
    for (uint32_t j = (uint32_t)tid; j < rows_x && j < 256u; j += BIN_THREADS) {
        const uint32_t trow = (x + j * XCDS) * bc + bi, b0 = F->band_off[trow];
        rowinfo[j] = make_uint2(b0, F->band_off[trow + 1] - b0);
    }
    // a class one tile-row short of the largest: its last slots stay without a strip
    if (rows_x < max_rows)
        for (uint32_t j = (uint32_t)tid; j < per_row; j += BIN_THREADS) {
            StripDesc sd; sd.wg = ~0u; sd.band_begin = 0; sd.n_b = 0; sd.pad = 0;
            F->strips[(size_t)(n_mine + j) * XCDS + x] = sd;
        }
    __syncthreads();                                       // rowinfo is complete
    // strip i of the class: tile-row x + 8 * (i / per_row), position i % per_row in it.  A thread's strips are 1024 apart: one division
    // per thread and round, then (row, position) advance by the (workgroup-uniform) quotient and remainder of 1024 / per_row
    // (three divisions per strip made these workgroups k2_bin's long pole on S2: 18.6 us of the kernel's 19.4).
    const uint32_t dq = BIN_THREADS / per_row, dr = BIN_THREADS - dq * per_row;
    auto advance = [&](uint32_t& j, uint32_t& pos) { j += dq; pos += dr; if (pos >= per_row) { pos -= per_row; ++j; } };
    auto strip_at = [&](uint32_t j, uint32_t pos) { return (x + j * XCDS) * per_row + pos; };
    for (uint32_t base = 0; base < n_mine; base += ORDER_U * BIN_THREADS) {                // (one round unless a class has more than 16 384 strips)
        const uint32_t n_round = min(n_mine - base, ORDER_U * BIN_THREADS), U = (n_round + BIN_THREADS - 1) / BIN_THREADS;   // workgroup-uniform
        unsigned long long ranges = 0;                     // this thread's strips' cost ranges, three bits each
        uint32_t behind[ORDER_U];                          // strips of the same range before this one in its wavefront
        const uint32_t i0 = base + (uint32_t)tid, j0 = i0 / per_row, pos0 = i0 - j0 * per_row;
        if (by_cost) {
            // every cost is read once (all of a thread's loads in flight together) and cleared
            uint32_t c[ORDER_U];
            uint32_t j = j0, pos = pos0;
#pragma unroll
            for (uint32_t u = 0; u < ORDER_U; ++u) {
                c[u] = 0;
                if (u >= U) continue;                      // workgroup-uniform
                const uint32_t i = base + u * BIN_THREADS + (uint32_t)tid;
                if (i < n_mine) c[u] = F->strip_cost[strip_at(j, pos)];
                advance(j, pos);
            }
            j = j0; pos = pos0;
#pragma unroll
            for (uint32_t u = 0; u < ORDER_U; ++u) {
                behind[u] = 0;
                if (u >= U) continue;
                const uint32_t i = base + u * BIN_THREADS + (uint32_t)tid;
                const bool valid = i < n_mine;
                if (valid && c[u]) F->strip_cost[strip_at(j, pos)] = 0;
                advance(j, pos);
                const uint32_t k = order_range(c[u]);
                ranges |= (unsigned long long)k << (3 * u);
                // the wavefront's strips per range, and this strip's place among those of its range: eight 8-bit counters (a
                // wavefront has at most 64 strips of one range) packed in two words, one inclusive scan of each
                const uint32_t sh = 8u * (k & 3u), one = valid ? 1u << sh : 0u;
                const uint32_t lo = (uint32_t)wave_scan_incl((int)(k < 4u ? one : 0u)), hi = (uint32_t)wave_scan_incl((int)(k < 4u ? 0u : one));
                behind[u] = (((k < 4u ? lo : hi) >> sh) & 0xffu) - 1u;                   // (unused when !valid)
                const uint32_t tot_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63), tot_hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
                if (lane < (int)ORDER_RANGES) cnt[lane][u][wave] = ((lane < 4 ? tot_lo : tot_hi) >> (8 * (lane & 3))) & 0xffu;
            }
            __syncthreads();
            if (tid < 64) {                                // exclusive prefix in (range, round, wavefront) order by one wavefront
                uint32_t* flat = &cnt[0][0][0];
                uint32_t carry = 0;
                for (uint32_t q = 0; q < ORDER_RANGES; ++q)
                    for (uint32_t e0 = 0; e0 < U * (BIN_THREADS / 64); e0 += 64) {       // (the rounds in use: U * 16 counts per range, contiguous)
                        const uint32_t e = e0 + (uint32_t)tid, lim = U * (BIN_THREADS / 64);
                        const uint32_t v = e < lim ? flat[q * ORDER_U * (BIN_THREADS / 64) + e] : 0u;
                        const uint32_t incl = (uint32_t)wave_scan_incl((int)v);
                        if (e < lim) flat[q * ORDER_U * (BIN_THREADS / 64) + e] = carry + incl - v;
                        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    }
            }
            __syncthreads();
        } else {
#pragma unroll
            for (uint32_t u = 0; u < ORDER_U; ++u) behind[u] = 0;
        }
        uint32_t j = j0, pos = pos0;
#pragma unroll
        for (uint32_t u = 0; u < ORDER_U; ++u) {
            if (u >= U) continue;
            const uint32_t i = base + u * BIN_THREADS + (uint32_t)tid;
            if (i < n_mine) {
                uint32_t rank = i;                                           // row-major inside the class
                if (by_cost) rank = base + cnt[(uint32_t)(ranges >> (3 * u)) & 7u][u][wave] + behind[u];
                uint2 ri;
                if (j < 256u) ri = rowinfo[j]; else { const uint32_t trow = (x + j * XCDS) * bc + bi, b0 = F->band_off[trow]; ri = make_uint2(b0, F->band_off[trow + 1] - b0); }
                // (the strip's tile column and local tile-row travel with it: k2_tiles divides nothing)
                StripDesc sd; sd.wg = strip_at(j, pos); sd.band_begin = ri.x; sd.n_b = ri.y; sd.pad = (pos / STRIPS_PER_TILE) | ((x + j * XCDS) << 16);
                F->strips[(size_t)rank * XCDS + x] = sd;
            }
            advance(j, pos);
        }
        __syncthreads();                                   // (cnt is rewritten by the next round)
    }
}
// At most 80 SGPRs: the hardware hands a wavefront its SGPRs in blocks of 16 plus 16 (MI355X_MICROARCH.md, "Occupancy API one block/CU
// high"), so 81-96 leave room for 7 wavefronts per SIMD -- ONE 1024-thread workgroup per CU instead of two, and a frame with more
// than 256 of them (S2: 386) ran k2_bin in two rounds (r04y: 44.6 us; its edge and path workgroups entered when the tile-rows' left).
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) void k2_bin_b(const Frame2* __restrict__ frames, uint32_t slow_kernels) { TRACE_DECL; TRACE_NOWAIT(0); bin_body(FRAME_PTR(frames, blockIdx.y), slow_kernels); TRACE(7); TRACE_OUT(0, blockIdx.x); }

// ---------------------------------------------------------------------------------------------
// k2_rows
// ---------------------------------------------------------------------------------------------
// a SUB cell: the span end at cell position x (24.8) opens (sgn > 0) or closes a span
__device__ __forceinline__ uint32_t pack_sub_cell(int x, int sgn, int xminp, int xmaxp) {
    int col = x >> 8, f = x & 255, live = 1;
    if (col >= xmaxp) { col = xmaxp - 1; f = 0; live = 0; }
    else if (col < xminp) { col = xminp; f = 0; }
    const int v = live * CELL_H + 2 * f, neg = sgn < 0 ? -1 : 0;       // (negated without a multiply)
    return make_cell_v(col - xminp, (v ^ neg) - neg).w;                // (the staged word is the cell itself)
}

#include "rows3.hip"

#ifndef R2_WAVES
#define R2_WAVES 5                 // 92 VGPRs, no scratch, 5.4 KB of LDS: five wavefronts per SIMD (end of round 4: a slot's role lives in two bits of one
                                   // register and its staged edge is looked up in LDS -- 108 -> 92 registers; with 24 SPILLED registers five wavefronts were
                                   // already 1.5 % faster per S1 frame than four, and cost 10 MB of scratch traffic; six lose: tools/lib_sweep.sh)
#endif
#define R2_ATTR __attribute__((amdgpu_waves_per_eu(R2_WAVES)))
#ifndef R2_WIDE_WAVES
#define R2_WIDE_WAVES 3            // the sixteen-slot instance: 168 VGPRs, 11 KB of LDS (two until the roles were packed: 192 VGPRs)
#endif
__global__ __launch_bounds__(64) R2_ATTR void k2_rows_b(const Frame2* __restrict__ frames) {
    FramePtr FR = FRAME_PTR(frames, blockIdx.y);
    if (blockIdx.x >= FR->n_chunks) return;
    rows3_chunk_body<ROWS_STAGE, ROWS_FAST_N>(FR, blockIdx.x);      // (the trace build's stamps are taken inside)
}
// (the instance for scenes with a path of more than ROWS_STAGE edges: 64 staged edges and SIXTEEN edge slots per row -- the rows of a
//  stroke outline or of a shape with a hole inside a hole -- at two to three wavefronts per SIMD)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(R2_WIDE_WAVES))) void k2_rows_wide_b(const Frame2* __restrict__ frames) {
    FramePtr FR = FRAME_PTR(frames, blockIdx.y);
    if (blockIdx.x >= FR->n_chunks) return;
    rows3_chunk_body<ROWS_STAGE_WIDE, ROWS_FAST_WIDE>(FR, blockIdx.x);
}


// ---------------------------------------------------------------------------------------------
// The order Cairo gives the edges that become active at one sample row (oracle: sort_edges / merge_sorted_edges -- pairs, then
// merges of runs of 2, 4, ... where a merge keeps consuming the list it is on through ties), for any number of them: slot
// numbers 0 .. cnt-1 (path order) in `a` / `b` (ping-pong), keys in `cell`.  merge_step merges the runs [s, s + width) and
// [s + width, s + 2 width) of `src` into `dst`; the caller loops over widths (one lane for a wavefront's row, one thread per
// merge for a workgroup's).  new_order_before of raster_common.hip reads the ranks this produces (DevEdge::pad).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sort_pairs(uint16_t* __restrict__ dst, const int* __restrict__ cell, int p, int cnt) {
    const int x = 2 * p, y = 2 * p + 1;
    if (y < cnt) { const bool keep = cell[x] <= cell[y]; dst[x] = (uint16_t)(keep ? x : y); dst[y] = (uint16_t)(keep ? y : x); }
    else if (x < cnt) dst[x] = (uint16_t)x;
}
__device__ __forceinline__ void merge_step(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, const int* __restrict__ cell, int s, int width, int cnt) {
    const int a1 = min(s + width, cnt), b1 = min(s + 2 * width, cnt);
    int ia = s, ib = a1, o = s;
    if (ib >= b1) { while (ia < a1) dst[o++] = src[ia++]; return; }
    bool phase_a = cell[src[ia]] <= cell[src[ib]];
    for (;;) {
        if (phase_a) {
            const int x = cell[src[ib]];
            while (ia < a1 && cell[src[ia]] <= x) dst[o++] = src[ia++];
            if (ia == a1) { while (ib < b1) dst[o++] = src[ib++]; return; }
        }
        const int x = cell[src[ia]];
        while (ib < b1 && cell[src[ib]] <= x) dst[o++] = src[ib++];
        if (ib == b1) { while (ia < a1) dst[o++] = src[ia++]; return; }
        phase_a = true;
    }
}

// k2_start_ranks: for every path with queued rows, the position of each edge among the edges that start at the same sample row in
// the order Cairo's sort of that bucket gives them (DevEdge::pad): what new_order_before needs, for any number of edges.  One
// 256-thread workgroup per path; an edge alone at its sample row keeps 0, a pair is ordered directly, larger groups are replayed
// one after the other with the merge sort above.
#define START_MAX 8192                 // edges of one path that may start at one sample row (more: the frame fails loudly); 104 KB of LDS
__device__ __forceinline__ void rank_tmp(DevEdge* E, int i, int rank) { E[i].pad = rank; }
#define START_LDS 1024                 // paths with at most this many edges keep their start rows and marks in LDS while they are ranked
__device__ __forceinline__ void start_ranks_body(FramePtr FR, uint32_t p) {
    __shared__ int sort_cell[START_MAX];
    __shared__ uint16_t sort_a[START_MAX], sort_b[START_MAX];
    __shared__ uint32_t member[START_MAX];
    __shared__ int yt_l[START_LDS], mark_l[START_LDS];    // first sample row of an edge (INT_MAX: never active); its rank, -1 = member of a group still to be sorted
    __shared__ uint32_t wave_cnt[4];
    __shared__ int next_y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DevPath P = FR->paths[p];
    if (P.kind != SWFR_PATH_TOR) return;
    DevEdge* E = FR->edges + P.first_edge;
    const int ne = (int)P.n_edges;
    const bool staged = ne <= START_LDS;                    // (workgroup-uniform) otherwise every look-up below goes to the edge array
    auto start_row = [&](int i) { if (staged) return yt_l[i]; const int yt = E[i].ytop; return E[i].ybot > yt ? yt : INT_MAX; };
    auto mark = [&](int i) { return staged ? mark_l[i] : E[i].pad; };
    if (staged) {
        for (int i = tid; i < ne; i += 256) { const int yt = E[i].ytop; yt_l[i] = E[i].ybot > yt ? yt : INT_MAX; }
        __syncthreads();
    }
    // edges alone at their sample row keep rank 0, pairs are ordered directly; members of larger groups are marked (-1)
    for (int i = tid; i < ne; i += 256) {
        const int yt = start_row(i);
        int cnt = 0, partner = -1;
        if (yt != INT_MAX)
            for (int j = 0; j < ne; ++j)
                if (j != i && start_row(j) == yt) { ++cnt; partner = j; }
        int rank = 0;
        if (cnt == 1) {
            const DevEdge a = E[i], b = E[partner];
            int ca = a.x1, cb = b.x1;
            if (a.dy) { int32_t q; int64_t rm; edge_x_at(a, yt, q, rm); ca = cell_of(q, rm, a.dy); }
            if (b.dy) { int32_t q; int64_t rm; edge_x_at(b, yt, q, rm); cb = cell_of(q, rm, b.dy); }
            rank = i < partner ? (ca <= cb ? 0 : 1) : (cb <= ca ? 1 : 0);      // sort_edges on a pair: the first stays first unless its cell is larger
        } else if (cnt > 1) rank = -1;
        rank_tmp(E, i, rank);
        if (staged) mark_l[i] = rank;
    }
    __syncthreads();
    // the larger groups, in ascending order of their sample row; `done_y`: everything up to it has been handled
    int done_y = INT_MIN;
    for (;;) {
        if (tid == 0) next_y = INT_MAX;
        __syncthreads();
        for (int i = tid; i < ne; i += 256) {
            const int yt = start_row(i);
            if (yt != INT_MAX && mark(i) == -1 && yt > done_y) atomicMin(&next_y, yt);
        }
        __syncthreads();
        const int y = next_y;
        __syncthreads();
        if (y == INT_MAX) break;                             // workgroup-uniform
        // its members in path order
        int n = 0;
        for (int base = 0; base < ne; base += 256) {
            const int i = base + tid;
            const bool is = i < ne && start_row(i) == y && mark(i) == -1;
            const unsigned long long b = __ballot(is);
            if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            int at = n;
            for (int w = 0; w < wave; ++w) at += (int)wave_cnt[w];
            at += __popcll(b & ((1ull << lane) - 1ull));
            if (is && at < START_MAX) {
                member[at] = (uint32_t)i;
                const DevEdge e = E[i];
                int c = e.x1;
                if (e.dy) { int32_t q; int64_t rm; edge_x_at(e, y, q, rm); c = cell_of(q, rm, e.dy); }
                sort_cell[at] = c;
            }
            n += (int)(wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
            __syncthreads();
        }
        __syncthreads();
        if (n > START_MAX) { if (tid == 0) atomicOr(&FR->counters[C2_ERROR], E2_START_GROUP); return; }     // workgroup-uniform
        for (int q = tid; q < (n + 1) / 2; q += 256) sort_pairs(sort_a, sort_cell, q, n);
        __syncthreads();
        uint16_t *src = sort_a, *dst = sort_b;
        for (int width = 2; width < n; width *= 2) {
            for (int s2 = tid * 2 * width; s2 < n; s2 += 256 * 2 * width) merge_step(src, dst, sort_cell, s2, width, n);
            __syncthreads();
            uint16_t* t = src; src = dst; dst = t;
        }
        for (int q = tid; q < n; q += 256) { const uint32_t m = member[src[q]]; E[m].pad = q; if (staged) mark_l[m] = q; }
        __syncthreads();
        done_y = y;
    }
}
__device__ __forceinline__ void start_ranks_loop(FramePtr FR) {
    const uint32_t nq = min(FR->counters[C2_PATHQ], FR->n_paths);
    for (uint32_t i = blockIdx.x; i < nq; i += gridDim.x) {
        start_ranks_body(FR, FR->path_queue[i]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k2_start_ranks_b(const Frame2* __restrict__ frames) { start_ranks_loop(FRAME_PTR(frames, blockIdx.y)); }

// ---------------------------------------------------------------------------------------------
// k2_rows_slow: the queued rows, one wavefront each, lane = active edge (up to 64; more: the huge queue).  big_row_body of
// raster_common.hip -- including the replay of Cairo's list order for coincident edges (tied_order) -- with cells as output.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t slow_count_index(uint32_t pass) { return pass == 0 ? (uint32_t)C2_SLOW : C2_SLOWQ + pass; }
__device__ __forceinline__ uint32_t huge_count_index(uint32_t pass) { return pass == 0 ? (uint32_t)C2_HUGE : C2_HUGEQ + pass; }
__device__ __forceinline__ void slow_row_body(FramePtr FR, const SlowRow sr, uint32_t pass) {
    __shared__ uint32_t active[ROWS_BIG_MAXA];
    __shared__ uint32_t retry;
    const int lane = threadIdx.x;
    const DevPath P = FR->paths[sr.path];
    const int r = sr.row, s0 = r * 15;
    const unsigned mask = P.fill_rule ? 1u : ~0u;
    const DevEdge* E = FR->edges + P.first_edge;
    const PathEdges PE = {FR->edges, nullptr, &P, false, FR->counters, FR->rows, FR->band_slots, sr.pad & 0x7fffffffu, pass + 1 < SLOW_PASSES ? &retry : nullptr};
    if (threadIdx.x == 0) retry = 0;
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
    lds_barrier();
    if (too_many) {                                            // a workgroup of k2_rows_huge takes it
        if (lane == 0) {
            const uint32_t at = atomicAdd(&FR->counters[huge_count_index(pass)], 1u);
            if (at < FR->slow_cap) FR->huge[(pass & 1u) * FR->slow_cap + at] = sr; else atomicOr(&FR->counters[C2_ERROR], E2_SLOW_QUEUE);
        }
        return;
    }
    const bool mine = lane < n;
    const uint32_t k_mine = mine ? active[lane] : active[0];
    const DevEdge e = E[n ? k_mine : 0];                      // lanes past the list compute on a valid edge and are ignored
    const bool slanted = e.dy != 0;
    const bool mid_row = __ballot(mine && ((e.ytop > s0) | (e.ybot < s0 + 15))) != 0ull;
    uint32_t role = 0;
    int32_t q1 = e.x1, q2 = e.x1; int64_t r1 = 0, r2 = 0;
    bool full = !mid_row && n > 0;
    if (full) {
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
        const int new_rank = e.pad;                          // among the edges that start at the same sample row (k2_start_ranks)
        int w = 0; bool fg = true, lg = true, ok = true, mixed = false;
        unsigned long long nfb = 0ull;                       // bit k: active edge k lets a tying new edge go first (see rows_by_slot)
        for (int pass = 0; pass < 2; ++pass) {
            w = 0; fg = lg = ok = true;
            for (int i = 0; i < n; ++i) {                     // wave-uniform loop: lane i's keys broadcast to every lane
                const int ci = __builtin_amdgcn_readlane(c0, i), ei = __builtin_amdgcn_readlane(c1, i), pi = __builtin_amdgcn_readlane(cpv, i);
                const int ni = __builtin_amdgcn_readlane(nw, i), di = __builtin_amdgcn_readlane(dr, i), ri_new = __builtin_amdgcn_readlane(new_rank, i);
                (void)pi;
                if (i == lane) continue;
                const bool tie = ci == c0, tie2 = ni == nw;
                bool deep_first = i < lane;
                if (mine && tie && ni == nw) {             // coincident edges: see tied_order
                    const DevEdge eo = E[active[i]];
                    if (same_line(eo, e)) deep_first = i < lane;
                    else if (nw == 0) deep_first = tied_order(PE, eo, e, active[i], k_mine, s0, i < lane);
                    else deep_first = ri_new < new_rank;
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
        }
    }
    const uint32_t mode = n == 0 ? ROW_EMPTY : (full ? ROW_FULL : ROW_SUB);
    if (n > 0 && !full) {
        // ---- SUB row: fifteen sample rows; lane = (sample row group g, edge je), cells ranked inside the group by shuffles
        role = 0;
        const int W = n <= 16 ? 16 : (n <= 32 ? 32 : 64), G = 64 / W;      // wave-uniform
        const int g = lane / W, je = lane - g * W;
        const bool mine_s = je < n;
        const DevEdge es = E[active[mine_s ? je : 0]];
        const bool slanted_s = es.dy != 0;
        for (int p = 0; p * G < 15; ++p) {
            const int sub = p * G + g, ss = s0 + sub;
            const bool act = mine_s && sub < 15 && es.ytop <= ss && ss < es.ybot;
            int cc = es.x1;
            if (act && slanted_s) { int32_t q; int64_t rm; edge_x_at(es, ss, q, rm); cc = cell_of(q, rm, es.dy); }
            const int dd = act ? es.dir : 0;
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
                if (in_a != in_b) role |= (uint32_t)(in_a ? 1 : 2) << (2 * sub);
            }
        }
        for (int off = W; off < 64; off <<= 1) role |= (uint32_t)__shfl_xor((int)role, off);
    }
    lds_barrier();
    if (retry) {                                               // (wave-uniform) an earlier row it depends on is still queued: next pass
        if (lane == 0) {
            const uint32_t at = atomicAdd(&FR->counters[slow_count_index(pass + 1)], 1u);
            FR->slow[((pass + 1) & 1u) * FR->slow_cap + at] = sr;       // (never more rows than this pass had)
        }
        return;
    }
    // ---- cells: one allocation for the row
    const bool has = mine && role != 0;
    int n_cells = 0;
    if (has) n_cells = (role & REC_FULL) ? full_span(q1, q2) : __popc((role | (role >> 1)) & 0x15555555u);
    const uint32_t incl = (uint32_t)wave_scan_incl(n_cells);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t base = alloc_cells(FR, total, lane);
    if (has && base != ~0u) {
        Cell* dst = &FR->cells[base + incl - (uint32_t)n_cells];
        if (role & REC_FULL) full_cells(q1, r1, q2, r2, e.dy, e.inv_dx, e.fq, e.fr, (role & 1u) ? +1 : -1, P.x_min, P.x_max, dst);
        else {
            for (int sub = 0; sub < 15; ++sub) {
                const uint32_t f = (role >> (2 * sub)) & 3u;
                if (!f) continue;
                int cc = e.x1;
                if (slanted) { int32_t q; int64_t rm; edge_x_at(e, s0 + sub, q, rm); cc = cell_of(q, rm, e.dy); }
                sub_cell(dst++, cc, f == 1u ? 1 : -1, P.x_min, P.x_max);
            }
        }
    }
    if (lane == 0) {
        RowInfo2 h; h.off = base == ~0u ? 0u : base; h.n = base == ~0u ? (uint16_t)0 : (uint16_t)total; h.mode = (uint16_t)mode;
        if (total > 65535u) { atomicOr(&FR->counters[C2_ERROR], E2_CELL_RANGE); h.n = 0; }
        FR->rows[sr.ri] = h;
        if (sr.pad >> 31) atomicAdd(&FR->counters[C2_TIE_ROWS], 1u);
    }
}
__device__ __forceinline__ void slow_rows_loop(FramePtr FR, uint32_t pass) {
    const uint32_t n_slow = min(FR->counters[slow_count_index(pass)], FR->slow_cap);
    for (uint32_t i = blockIdx.x; i < n_slow; i += gridDim.x) {
        slow_row_body(FR, FR->slow[(pass & 1u) * FR->slow_cap + i], pass);
        __syncthreads();                                       // `active` is rewritten by the next row
    }
}
__global__ __launch_bounds__(64) void k2_rows_slow_b(const Frame2* __restrict__ frames, uint32_t pass) { slow_rows_loop(FRAME_PTR(frames, blockIdx.y), pass); }

// k2_rows_huge: rows with 65 .. 8192 active edges of one path, one 1024-thread workgroup each (136 KB of LDS: one workgroup per CU):
// thread t owns the active edges t, t + 1024, ... and ranks each against the row's sort keys in LDS.
__device__ __forceinline__ void huge_row_body(FramePtr FR, const SlowRow sr, uint32_t pass) {
    __shared__ uint32_t retry;
    __shared__ uint32_t active[ROWS_HUGE_MAXA];
    __shared__ int k_a[ROWS_HUGE_MAXA], k_b[ROWS_HUGE_MAXA], k_c[ROWS_HUGE_MAXA];
    __shared__ int8_t k_d[ROWS_HUGE_MAXA];                           // new / direction / "lets a tying new edge go first" bits, or the direction alone
    __shared__ uint32_t wave_cnt[HUGE_THREADS / 64];
    __shared__ int flags;                                            // bit 0: some edge starts / ends inside the row, bit 1: FULL test failed
    __shared__ uint32_t cell_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DevPath P = FR->paths[sr.path];
    const int r = sr.row, s0 = r * 15;
    const unsigned mask = P.fill_rule ? 1u : ~0u;
    const DevEdge* E = FR->edges + P.first_edge;
    const PathEdges PE = {FR->edges, nullptr, &P, false, FR->counters, FR->rows, FR->band_slots, sr.pad & 0x7fffffffu, pass + 1 < SLOW_PASSES ? &retry : nullptr};
    if (tid == 0) { flags = 0; retry = 0; }
    int n = 0;
    for (uint32_t base = 0; base < P.n_edges; base += HUGE_THREADS) {
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
        for (int w = 0; w < HUGE_THREADS / 64; ++w) n += (int)wave_cnt[w];
        __syncthreads();
    }
    RowInfo2 ri; ri.off = 0; ri.n = 0; ri.mode = ROW_EMPTY;
    if (n > ROWS_HUGE_MAXA) {                                        // workgroup-uniform
        if (tid == 0) { atomicOr(&FR->counters[C2_ERROR], E2_ACTIVE_EDGES); FR->rows[sr.ri] = ri; }
        return;
    }
    const int nb = (n + HUGE_THREADS - 1) / HUGE_THREADS;                                   // owned edges per thread (workgroup-uniform)
    uint32_t role[ROWS_HUGE_EPT];
#pragma unroll
    for (int m = 0; m < ROWS_HUGE_EPT; ++m) role[m] = 0;
    for (int j = tid; j < n; j += HUGE_THREADS) {
        const DevEdge e = E[active[j]];
        if ((e.ytop > s0) | (e.ybot < s0 + 15)) atomicOr(&flags, 1);
    }
    __syncthreads();
    bool full = (flags & 1) == 0;
    if (full) {
        for (int j = tid; j < n; j += HUGE_THREADS) {
            const DevEdge e = E[active[j]];
            int c0, c1, cpv; int32_t q1, q2; int64_t r1, r2;
            huge_full_keys(e, s0, c0, c1, cpv, q1, r1, q2, r2);
            k_a[j] = c0; k_b[j] = c1; k_c[j] = cpv; k_d[j] = ((e.ytop == s0) ? 4 : 0) | (e.dir + 1);
        }
        __syncthreads();
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
                const int j = m * HUGE_THREADS + tid;
                role[m] = 0;
                if (m >= nb || j >= n) continue;
                const int c0 = k_a[j], c1 = k_b[j], nw = (k_d[j] >> 2) & 1, dr = (k_d[j] & 3) - 1, nfj = (k_d[j] >> 3) & 1;
                int w = 0; bool fg = true, lg = true, ok = true, mixed = false;
                for (int i = 0; i < n; ++i) {                        // LDS broadcast reads
                    if (i == j) continue;
                    const int ci = k_a[i], ei = k_b[i], ni = (k_d[i] >> 2) & 1, di = (k_d[i] & 3) - 1, nfi = (k_d[i] >> 3) & 1;
                    const bool tie = ci == c0, tie2 = ni == nw;
                    bool deep_first = i < j;
                    if (tie && ni == nw) {              // coincident edges: see tied_order (rare)
                        const DevEdge ea = E[active[i]], eb = E[active[j]];
                        if (same_line(ea, eb)) deep_first = i < j;
                        else if (nw == 0) deep_first = tied_order(PE, ea, eb, active[i], active[j], s0, i < j);
                        else deep_first = ea.pad < eb.pad;              // start ranks (k2_start_ranks)
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
                const int j = m * HUGE_THREADS + tid;
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
                const int j = m * HUGE_THREADS + tid;
                if (m < nb && j < n && ((nfbits >> m) & 1u)) k_d[j] |= 8;
            }
            __syncthreads();
        }
        __syncthreads();
        full = (flags & 2) == 0;
    }
    const uint32_t mode = full ? ROW_FULL : ROW_SUB;
    if (!full) {
#pragma unroll
        for (int m = 0; m < ROWS_HUGE_EPT; ++m) role[m] = 0;
        for (int sub = 0; sub < 15; ++sub) {
            const int ss = s0 + sub;
            for (int j = tid; j < n; j += HUGE_THREADS) {
                const DevEdge e = E[active[j]];
                const bool act = e.ytop <= ss && ss < e.ybot;
                int cc = e.x1;
                if (act && e.dy) { int32_t q; int64_t rm; edge_x_at(e, ss, q, rm); cc = cell_of(q, rm, e.dy); }
                k_a[j] = cc; k_d[j] = act ? e.dir : 0;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
                const int j = m * HUGE_THREADS + tid;
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
                    if (in_a != in_b) role[m] |= (uint32_t)(in_a ? 1 : 2) << (2 * sub);
                }
            }
            __syncthreads();                                         // k_a / k_d are rewritten for the next sample row
        }
    }
    __syncthreads();
    if (retry) {                                                     // (workgroup-uniform) depends on a row that is still queued: next pass
        if (tid == 0) {
            const uint32_t at = atomicAdd(&FR->counters[huge_count_index(pass + 1)], 1u);
            if (at < FR->slow_cap) FR->huge[((pass + 1) & 1u) * FR->slow_cap + at] = sr; else atomicOr(&FR->counters[C2_ERROR], E2_SLOW_QUEUE);
        }
        return;
    }
    // ---- cells: count per thread, one allocation for the row, every thread writes its edges' cells behind its prefix
    uint32_t mine_cells = 0;
#pragma unroll
    for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
        const int j = m * HUGE_THREADS + tid;
        if (m >= nb || j >= n || role[m] == 0) continue;
        if (role[m] & REC_FULL) {
            const DevEdge e = E[active[j]];
            int c0, c1, cpv; int32_t q1, q2; int64_t r1, r2;
            huge_full_keys(e, s0, c0, c1, cpv, q1, r1, q2, r2);
            mine_cells += (uint32_t)full_span(q1, q2);
        } else mine_cells += (uint32_t)__popc((role[m] | (role[m] >> 1)) & 0x15555555u);
    }
    const uint32_t incl = (uint32_t)wave_scan_incl((int)mine_cells);
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine_cells;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    uint32_t total = 0;
    for (int w = 0; w < HUGE_THREADS / 64; ++w) total += wave_cnt[w];
    if (tid == 0) {
        uint32_t base = 0;
        if (total) {
            const uint32_t old = atomicAdd(&FR->counters[C2_HEAD], total);
            base = FR->cell_main + old;
            if ((uint64_t)base + total > FR->cell_slice || total > 65535u) { atomicOr(&FR->counters[C2_ERROR], total > 65535u ? E2_CELL_RANGE : E2_CELL_ARENA); base = ~0u; }
        }
        cell_base = base;
    }
    __syncthreads();
    const uint32_t base = cell_base;
    if (base != ~0u) {
        Cell* dst = &FR->cells[base + before];
#pragma unroll
        for (int m = 0; m < ROWS_HUGE_EPT; ++m) {
            const int j = m * HUGE_THREADS + tid;
            if (m >= nb || j >= n || role[m] == 0) continue;
            const DevEdge e = E[active[j]];
            if (role[m] & REC_FULL) {
                int c0, c1, cpv; int32_t q1, q2; int64_t r1, r2;
                huge_full_keys(e, s0, c0, c1, cpv, q1, r1, q2, r2);
                full_cells(q1, r1, q2, r2, e.dy, e.inv_dx, e.fq, e.fr, (role[m] & 1u) ? +1 : -1, P.x_min, P.x_max, dst);
                dst += full_span(q1, q2);
            } else {
                for (int sub = 0; sub < 15; ++sub) {
                    const uint32_t f = (role[m] >> (2 * sub)) & 3u;
                    if (!f) continue;
                    int cc = e.x1;
                    if (e.dy) { int32_t q; int64_t rm; edge_x_at(e, s0 + sub, q, rm); cc = cell_of(q, rm, e.dy); }
                    sub_cell(dst++, cc, f == 1u ? 1 : -1, P.x_min, P.x_max);
                }
            }
        }
    }
    if (tid == 0) { ri.off = base == ~0u ? 0u : base; ri.n = base == ~0u ? (uint16_t)0 : (uint16_t)total; ri.mode = (uint16_t)mode; FR->rows[sr.ri] = ri; }
}
__device__ __forceinline__ void huge_rows_loop(FramePtr FR, uint32_t pass) {
    const uint32_t n_huge = min(FR->counters[huge_count_index(pass)], FR->slow_cap);
    for (uint32_t i = blockIdx.x; i < n_huge; i += gridDim.x) {
        huge_row_body(FR, FR->huge[(pass & 1u) * FR->slow_cap + i], pass);
        __syncthreads();
    }
}
__global__ __launch_bounds__(HUGE_THREADS) void k2_rows_huge_b(const Frame2* __restrict__ frames, uint32_t pass) { huge_rows_loop(FRAME_PTR(frames, blockIdx.y), pass); }

// ---------------------------------------------------------------------------------------------
// shading: three instances of the tile kernel -- solid colours only; + bitmap fills (integer arithmetic only: pixman's
// 16.16 sample positions, bilinear or separable convolution -- the branch of raster_common.hip's shade() for bitmaps, inlined);
// + gradients (that file's shade(): double precision, a call).  A scene gets the lightest instance that covers its styles.
// ---------------------------------------------------------------------------------------------
// the four texels of a bilinear sample (7-bit weights: what CAIRO_FILTER_GOOD becomes for scales > .75): where they are -- `p`, already
// clamped / wrapped into the bitmap, so the loads can be issued unconditionally and several pixels' loads together -- which of them lie
// outside a non-repeating bitmap, and the weights
struct BilinearTap {
    uint32_t o[2][2];                    // [x][y]: texel index in the bitmap (32-bit offsets from the bitmap's base: half the registers of pointers)
    uint32_t wf;                         // weights and the outside flags in one register: wx | wy << 7 | oxa << 14 | oxb << 15 | oya << 16 | oyb << 17
};
// x mod n in [0, n) for n > 0: a reciprocal estimate with an integer fix-up wherever the quotient is exact in single precision
// (|x| < 2^22: every 16.16 position pixman can represent), the integer remainder -- some forty instructions -- only beyond
__device__ __forceinline__ int wrap_mod(int x, int n, float inv_n) {
    if ((unsigned)(x + (1 << 22)) < (1u << 23)) {
        const int q = (int)floorf((float)x * inv_n);
        int r = x - mul_i24(q, n);
        if (r < 0) r += n;
        if (r >= n) r -= n;
        return r;
    }
    const int r = x % n;
    return r < 0 ? r + n : r;
}
// bxp, byp: pixman's 16.16 sample position of the pixel centre, half a texel back (64-bit sums; the caller steps them along a row:
// whole multiples of the matrix entries, so every position is the exact sum)
__device__ __forceinline__ void bilinear_taps_at(const DevFilter& flt, long long bxp, long long byp, BilinearTap& t) {
    const bool repeat = flt.extend == 1;
    const int bw = (int)flt.width, bh = (int)flt.height;
    const int x0 = (int)(bxp >> 16), y0 = (int)(byp >> 16);
    uint32_t wf = ((uint32_t)bxp >> 9 & 0x7fu) | (((uint32_t)byp >> 9 & 0x7fu) << 7);
    int xa = x0, xb = x0 + 1, ya = y0, yb = y0 + 1;
    if (repeat) {
        xa = wrap_mod(xa, bw, 1.0f / (float)bw); xb = xa + 1 == bw ? 0 : xa + 1;
        ya = wrap_mod(ya, bh, 1.0f / (float)bh); yb = ya + 1 == bh ? 0 : ya + 1;
    } else {
        wf |= (xa < 0 || xa >= bw ? 1u << 14 : 0u) | (xb < 0 || xb >= bw ? 1u << 15 : 0u) | (ya < 0 || ya >= bh ? 1u << 16 : 0u) | (yb < 0 || yb >= bh ? 1u << 17 : 0u);
        xa = min(max(xa, 0), bw - 1); xb = min(max(xb, 0), bw - 1); ya = min(max(ya, 0), bh - 1); yb = min(max(yb, 0), bh - 1);
    }
    const uint32_t rowa = (uint32_t)ya * flt.width, rowb = (uint32_t)yb * flt.width;     // (a bitmap has fewer than 2^32 texels)
    t.o[0][0] = rowa + (uint32_t)xa; t.o[1][0] = rowa + (uint32_t)xb; t.o[0][1] = rowb + (uint32_t)xa; t.o[1][1] = rowb + (uint32_t)xb;
    t.wf = wf;
}
__device__ __forceinline__ void bilinear_taps(const DevFilter& flt, int px, int py, BilinearTap& t) {
    bilinear_taps_at(flt, flt.base_x + (long long)px * flt.m00 + (long long)py * flt.m01 - 0x8000,
                     flt.base_y + (long long)px * flt.m10 + (long long)py * flt.m11 - 0x8000, t);
}
__device__ __forceinline__ uint32_t bilinear_mix(const BilinearTap& t, const uint32_t* q00, const uint32_t* q10, const uint32_t* q01, const uint32_t* q11) {
    uint32_t c00 = *q00, c10 = *q10, c01 = *q01, c11 = *q11;
    const uint32_t wx = t.wf & 0x7fu, wy = (t.wf >> 7) & 0x7fu;
    const bool oxa = (t.wf >> 14) & 1u, oxb = (t.wf >> 15) & 1u, oya = (t.wf >> 16) & 1u, oyb = (t.wf >> 17) & 1u;
    if (oxa || oya) c00 = 0;
    if (oxb || oya) c10 = 0;
    if (oxa || oyb) c01 = 0;
    if (oxb || oyb) c11 = 0;
    const uint32_t w00 = (128 - wx) * (128 - wy), w10 = wx * (128 - wy), w01 = (128 - wx) * wy, w11 = wx * wy;
    uint32_t out = 0;
#pragma unroll
    for (int sh = 0; sh < 32; sh += 8) {
        const uint32_t acc = ((c00 >> sh) & 255u) * w00 + ((c10 >> sh) & 255u) * w10 + ((c01 >> sh) & 255u) * w01 + ((c11 >> sh) & 255u) * w11;
        out |= ((acc >> 14) & 255u) << sh;
    }
    return out;
}
// the same for a sample whose four texels all lie inside the bitmap (no outside flags to apply)
__device__ __forceinline__ uint32_t bilinear_mix_inside(uint32_t wf, uint32_t c00, uint32_t c10, uint32_t c01, uint32_t c11) {
    const uint32_t wx = wf & 0x7fu, wy = (wf >> 7) & 0x7fu;
    const uint32_t w11 = __umul24(wx, wy), w10 = (wx << 7) - w11, w01 = (wy << 7) - w11, w00 = 16384u - (wx << 7) - w01;   // (128 - wx)(128 - wy) etc.
    uint32_t out = 0;
#pragma unroll
    for (int sh = 0; sh < 32; sh += 8) {
        const uint32_t acc = __umul24((c00 >> sh) & 255u, w00) + __umul24((c10 >> sh) & 255u, w10) + __umul24((c01 >> sh) & 255u, w01) + __umul24((c11 >> sh) & 255u, w11);
        out |= (acc >> 14) << sh;                                  // (the weights add up to 2^14: acc >> 14 <= 255)
    }
    return out;
}
__device__ __forceinline__ uint32_t shade_bitmap(uint32_t style_index, const Sources& src, int px, int py) {
    const DevFilter& flt = src.filters[style_index];
    const struct { const uint32_t* pixels; uint32_t width, height; } bm = {flt.pixels, flt.width, flt.height};
    // pixman's own 16.16 sample position of this pixel's centre
    const long long fxp = flt.base_x + (long long)px * flt.m00 + (long long)py * flt.m01;
    const long long fyp = flt.base_y + (long long)px * flt.m10 + (long long)py * flt.m11;
    const bool repeat = flt.extend == 1;
    const int bw = (int)bm.width, bh = (int)bm.height;
    if (flt.on) {
        // CAIRO_FILTER_GOOD below scale 0.75: pixman's separable convolution (integer tables and accumulation)
        long long x = fxp, y = fyp;
        const int xsh = 16 - flt.xbits, ysh = 16 - flt.ybits;
        const long long x_off = (((long long)flt.cw << 16) - 65536) >> 1, y_off = (((long long)flt.ch << 16) - 65536) >> 1;
        x = (x & ~((1ll << xsh) - 1)) + ((1 << xsh) >> 1);          // the middle of the closest phase
        y = (y & ~((1ll << ysh) - 1)) + ((1 << ysh) >> 1);
        const int phx = (int)((x & 0xffff) >> xsh), phy = (int)((y & 0xffff) >> ysh);
        const int32_t* yp = src.fparams + flt.y_off + phy * flt.ch;
        const int32_t* xp0 = src.fparams + flt.x_off + phx * flt.cw;
        const int x1 = (int)((x - 1 - x_off) >> 16), y1 = (int)((y - 1 - y_off) >> 16);
        long long sr = 0, sg = 0, sb = 0, sa = 0;
        for (int i = 0; i < flt.ch; ++i) {
            const long long fy = yp[i];
            if (!fy) continue;
            int ry = y1 + i;
            if (repeat) ry = ((ry % bh) + bh) % bh;
            for (int j = 0; j < flt.cw; ++j) {
                const int32_t fx = xp0[j];
                if (!fx) continue;
                int rx = x1 + j;
                uint32_t pixel;
                if (repeat) { rx = ((rx % bw) + bw) % bw; pixel = bm.pixels[(size_t)ry * bm.width + rx]; }
                else pixel = (rx < 0 || ry < 0 || rx >= bw || ry >= bh) ? 0u : bm.pixels[(size_t)ry * bm.width + rx];
                const int f = (int)((fy * fx + 0x8000) >> 16);
                sr += (int)((pixel >> 16) & 255u) * f; sg += (int)((pixel >> 8) & 255u) * f; sb += (int)(pixel & 255u) * f; sa += (int)(pixel >> 24) * f;
            }
        }
        sa = (sa + 0x8000) >> 16; sr = (sr + 0x8000) >> 16; sg = (sg + 0x8000) >> 16; sb = (sb + 0x8000) >> 16;
        sa = min(max(sa, 0ll), 255ll); sr = min(max(sr, 0ll), 255ll); sg = min(max(sg, 0ll), 255ll); sb = min(max(sb, 0ll), 255ll);
        return ((uint32_t)sa << 24) | ((uint32_t)sr << 16) | ((uint32_t)sg << 8) | (uint32_t)sb;
    }
    BilinearTap t;
    bilinear_taps(flt, px, py, t);
    return bilinear_mix(t, flt.pixels + t.o[0][0], flt.pixels + t.o[1][0], flt.pixels + t.o[0][1], flt.pixels + t.o[1][1]);
}
// SHADERS: 0 solid colours only, 1 + bitmaps, 2 + gradients
template <int SHADERS>
__device__ __forceinline__ uint32_t blend2(uint32_t dst, uint32_t a, uint32_t eflags, uint32_t solid, const swfr_style* __restrict__ styles,
                                           uint32_t style, const Sources& src, int cx, int cy) {
    if (SHADERS == 0 || (eflags & BE_SOLID)) {
        if (eflags & BE_LERP) return a == 255u ? solid : lerp_pixel(solid, a, dst);
        return over_pixel(a == 255u ? solid : mul_un8(solid, a), dst);
    }
    uint32_t c;
    if (SHADERS == 1) c = shade_bitmap(style, src, cx, cy);
    else c = src.filters[style].kind == SWFR_STYLE_BITMAP ? shade_bitmap(style, src, cx, cy) : shade(styles[style], style, src, cx, cy);
    const uint32_t s = mul_un8(c, a);
    return (eflags & BE_LERP) ? s : over_pixel(s, dst);
}

// ---------------------------------------------------------------------------------------------
// k2_tiles
// ---------------------------------------------------------------------------------------------
// One wavefront per 64x8-pixel strip of the launch list.  Lane layout: lane = 16 * g + cg owns the pixels of columns 4 cg .. 4 cg + 3
// in the rows g and g + 4 of the strip (slot j = 4 h + i: row g + 4 h, column 4 cg + i), so
//   * a DPP row (16 lanes) is one pixel row: the prefix sums of FOUR pixel rows are one row_shr sequence (no row broadcasts),
//   * the accumulators of a lane's four columns are one 16-byte LDS read, its four pixels of a row one 16-byte store.
// Dependent memory round trips per strip: strip descriptor -> class bytes -> {band entries, row headers} -> cells -> stores.
#define T3_LIST 16                     // non-empty band entries of a strip staged per round (painter's order)
#define T3_ACC_STRIDE 68               // ints per accumulator row: 64 columns, padded so that every row starts 16-byte aligned
#define T3_UNITS 32                    // units (a lane's four pixels of a row) of one (path, strip) blended in compacted form per round
#define T3_PAIR_FROM 8192u            // strips per frame from which a frame's wavefronts paint two strips each
#define T3_CLS_PRE 2                   // x 64 class bytes of a strip fetched up front

// 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 issue at full rate, v_mul_lo_u32 at a quarter): two 8-bit channels in the
// 0x00ff00ff layout times an 8-bit factor fit
__device__ __forceinline__ uint32_t mul8x2_7f_24(uint32_t a, uint32_t b) {
    uint32_t t = __umul24(a & 0xff00ffu, b) + 0x7f007fu;
    return ((t + ((t >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
}
__device__ __forceinline__ uint32_t mul_un8_24(uint32_t x, uint32_t a) {
    uint32_t rb = __umul24(x & 0xff00ffu, a) + 0x800080u;
    rb = ((rb + ((rb >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
    uint32_t ag = __umul24((x >> 8) & 0xff00ffu, a) + 0x800080u;
    ag = ((ag + ((ag >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
    return rb | (ag << 8);
}
// what the compacted blend does with a queued pixel of a solid-colour path: Cairo's SOURCE lerp (0x7f rounding: lerp_pixel) or
// pixman's OVER (0x80 rounding: over_pixel of the colour times the coverage); coverage 0 keeps the pixel
__device__ __forceinline__ uint32_t solid_blend(uint32_t solid, uint32_t a, uint32_t dst, bool lerp) {
    uint32_t out;
    if (lerp) {
        const uint32_t ia = 255u - a;
        out = (mul8x2_7f_24(solid, a) + mul8x2_7f_24(dst, ia)) | ((mul8x2_7f_24(solid >> 8, a) + mul8x2_7f_24(dst >> 8, ia)) << 8);
    } else {
        const uint32_t sc = a == 255u ? solid : mul_un8_24(solid, a);
        const uint32_t m = mul_un8_24(dst, 255u - (sc >> 24));
        out = add8x2_sat(m & 0xff00ffu, sc & 0xff00ffu) | (add8x2_sat((m >> 8) & 0xff00ffu, (sc >> 8) & 0xff00ffu) << 8);
    }
    return a == 0u ? dst : out;
}

// The strip's eight pixels of one lane blended with coverages al[] (slot j: pixel (cx0 + (j & 3), cy0 + 4 * (j >> 2))).
// Solid colours: coverage 255 of an opaque colour is a select, coverage 0 keeps the pixel, and the pixels that need the rounded
// products -- the few on an edge -- are compacted through LDS and blended with lanes = queued pixels: the unit of the compaction
// is a lane's four pixels of one row (two 16-byte LDS writes per unit that has an edge pixel, one ballot + mbcnt per row half).
// Bitmaps: the texel loads of a lane's four pixels of a row are issued together.  Gradients: pixel by pixel (f64, a call).
template <int SHADERS>
__device__ __forceinline__ void blend8(uint32_t (&px)[8], const uint32_t (&al)[8], uint32_t eflags, uint32_t solid, const swfr_style* __restrict__ styles,
                                       uint32_t style, const Sources& src, int cx0, int cy0, uint32_t* __restrict__ bq, int lane) {
    if (SHADERS == 0 || (eflags & BE_SOLID)) {
        const bool lerp = (eflags & BE_LERP) != 0;
        const bool sel255 = lerp || (solid >> 24) == 0xffu;       // coverage 255 puts the colour itself (wave-uniform)
        bool need[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // some pixel of the unit has a coverage in 1..254 (1..255 when the colour is translucent): al - 1 wraps 0 to the top
            const uint32_t t0 = al[4 * h] - 1u, t1 = al[4 * h + 1] - 1u, t2 = al[4 * h + 2] - 1u, t3 = al[4 * h + 3] - 1u;
            need[h] = min(min(t0, t1), min(t2, t3)) < (sel255 ? 254u : 255u);
        }
        if (sel255) {
#pragma unroll
            for (int j = 0; j < 8; ++j) px[j] = al[j] == 255u ? solid : px[j];
        }
        const unsigned long long m0 = __ballot(need[0]), m1 = __ballot(need[1]);
        if ((m0 | m1) == 0ull) return;                           // wave-uniform: no edge pixel in this strip
        const int n0 = (int)__popcll(m0), nu = n0 + (int)__popcll(m1);          // units queued
        // queue: unit u holds {coverage[4]} at bq[8 u .. 8 u + 3] and {pixel[4]} at bq[8 u + 4 .. 8 u + 7]
        const int u0 = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
        const int u1 = n0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
        for (int ub = 0; ub < nu; ub += T3_UNITS) {                // wave-uniform; one round unless more than T3_UNITS units have an edge pixel
            if (need[0] && u0 >= ub && u0 < ub + T3_UNITS) {
                uint4* q = reinterpret_cast<uint4*>(bq + 8 * (u0 - ub));
                q[0] = make_uint4(al[0], al[1], al[2], al[3]); q[1] = make_uint4(px[0], px[1], px[2], px[3]);
            }
            if (need[1] && u1 >= ub && u1 < ub + T3_UNITS) {
                uint4* q = reinterpret_cast<uint4*>(bq + 8 * (u1 - ub));
                q[0] = make_uint4(al[4], al[5], al[6], al[7]); q[1] = make_uint4(px[4], px[5], px[6], px[7]);
            }
            lds_barrier();
            const int ne = 4 * min(nu - ub, T3_UNITS);             // queued pixels of this round
            for (int b = lane; b < ne; b += 64) {
                uint32_t* e = bq + 8 * (b >> 2) + (b & 3);
                e[4] = solid_blend(solid, e[0], e[4], lerp);
            }
            lds_barrier();
            if (need[0] && u0 >= ub && u0 < ub + T3_UNITS) {
                const uint4 r = reinterpret_cast<const uint4*>(bq + 8 * (u0 - ub))[1];
                px[0] = r.x; px[1] = r.y; px[2] = r.z; px[3] = r.w;
            }
            if (need[1] && u1 >= ub && u1 < ub + T3_UNITS) {
                const uint4 r = reinterpret_cast<const uint4*>(bq + 8 * (u1 - ub))[1];
                px[4] = r.x; px[5] = r.y; px[6] = r.z; px[7] = r.w;
            }
            lds_barrier();                                        // (the queue is rewritten by the next round / the next path)
        }
        return;
    }
    const DevFilter& flt = src.filters[style];
    if (flt.kind == SWFR_STYLE_BITMAP && !flt.on) {
        // Bilinear bitmap: the texels are fetched TRANSPOSED -- in round i a lane samples column 16 i + cg of its pixel row, so that
        // neighbouring lanes read neighbouring texels (a lane's own four pixels are four columns apart from its neighbour's: a
        // quarter of every cache line per load) -- and the colours go back to the lanes that own the pixels through LDS.
        const int cgl = lane & 15, tx0 = cx0 - 4 * cgl;
        uint32_t* tq = bq + 64 * (lane >> 4);                        // this pixel row's 64 colours (bq: 4 rows x 64)
        const uint32_t* __restrict__ texels = flt.pixels;            // (wave-uniform base: scalar address + 32-bit lane offsets)
        // pixman's 16.16 sample position of the lane's first sample (column tx0 + cgl, row cy0), evaluated once; the other seven are
        // wave-uniform steps away (16 columns, 4 rows): 64-bit additions instead of four 32 x 32 -> 64 multiplies per sample (the
        // quarter-rate multiplies were a quarter of this kernel's vector issue time) -- the same integers, modulo 2^64
        const long long bx00 = flt.base_x + (long long)(tx0 + cgl) * flt.m00 + (long long)cy0 * flt.m01 - 0x8000;
        const long long by00 = flt.base_y + (long long)(tx0 + cgl) * flt.m10 + (long long)cy0 * flt.m11 - 0x8000;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!(__ballot((al[4 * h] | al[4 * h + 1] | al[4 * h + 2] | al[4 * h + 3]) != 0u))) continue;       // wave-uniform: nothing to paint in these rows
            // Every sample of the wavefront strictly inside the bitmap -- the usual strip -- needs no clamping, no wrap and no outside
            // flags: the four texels are o, o + 1, o + width, o + width + 1 (ONE lane offset, two scalar bases, two immediate offsets).
            // Anything else takes the general routine.
            const int bw1 = (int)flt.width - 1, bh1 = (int)flt.height - 1;
            bool inside = flt.width < (1u << 23) && flt.height < (1u << 23);
            uint32_t o[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long bxp = bx00 + (long long)(16 * i) * flt.m00 + (long long)(4 * h) * flt.m01;
                const long long byp = by00 + (long long)(16 * i) * flt.m10 + (long long)(4 * h) * flt.m11;
                const int x0 = (int)(bxp >> 16), y0 = (int)(byp >> 16);
                inside = inside && (unsigned)x0 < (unsigned)bw1 && (unsigned)y0 < (unsigned)bh1;
                o[i] = __umul24((uint32_t)y0, flt.width) + (uint32_t)x0;          // (meaningless unless inside)
                wf[i] = ((uint32_t)bxp >> 9 & 0x7fu) | (((uint32_t)byp >> 9 & 0x7fu) << 7);
            }
            if (__ballot(!inside) == 0ull) {                       // wave-uniform
                const uint32_t* __restrict__ row_b = texels + flt.width;
                uint32_t c[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { c[i][0] = texels[o[i]]; c[i][1] = (texels + 1)[o[i]]; c[i][2] = row_b[o[i]]; c[i][3] = (row_b + 1)[o[i]]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) tq[16 * i + cgl] = bilinear_mix_inside(wf[i], c[i][0], c[i][1], c[i][2], c[i][3]);
            } else {
#pragma unroll 1
                for (int i = 0; i < 4; ++i) {
                    BilinearTap t;
                    bilinear_taps(flt, tx0 + 16 * i + cgl, cy0 + 4 * h, t);
                    tq[16 * i + cgl] = bilinear_mix(t, texels + t.o[0][0], texels + t.o[1][0], texels + t.o[0][1], texels + t.o[1][1]);
                }
            }
            lds_barrier();
            const uint4 mine = *reinterpret_cast<const uint4*>(tq + 4 * cgl);
            lds_barrier();                                        // (the next half / the next path rewrites the rows)
            const uint32_t col[4] = {mine.x, mine.y, mine.z, mine.w};
            // the whole wavefront at coverage 255 with opaque texels (the inside of a bitmap-filled shape; SWF bitmaps are opaque away
            // from a non-repeating bitmap's rim): MUL_UN8(c, 255) = c and OVER of an opaque source is the source -- no arithmetic
            const bool plain = (al[4 * h] & al[4 * h + 1] & al[4 * h + 2] & al[4 * h + 3]) == 255u && (mine.x & mine.y & mine.z & mine.w) >> 24 == 255u;
            if (__ballot(!plain) == 0ull) {
#pragma unroll
                for (int i = 0; i < 4; ++i) px[4 * h + i] = col[i];
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 4 * h + i;
                const uint32_t sc = mul_un8(col[i], al[j]);
                const uint32_t b = (eflags & BE_LERP) ? sc : over_pixel(sc, px[j]);
                px[j] = al[j] ? b : px[j];
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (al[j]) px[j] = blend2<SHADERS>(px[j], al[j], eflags, solid, styles, style, src, cx0 + (j & 3), cy0 + 4 * (j >> 2));
}

template <int SHADERS>
__device__ __forceinline__ void tiles3_body(FramePtr FR, uint32_t* fb_to) {
    __shared__ __attribute__((aligned(16))) int acc[STRIP_H][T3_ACC_STRIDE];     // per pixel: covered height << 20 | uncovered area (20 bits, signed)
    __shared__ __attribute__((aligned(16))) uint32_t ent[T3_LIST][12];         // BandEntry2 as dwords, [8] = its class byte for this strip
    __shared__ __attribute__((aligned(16))) uint32_t hdr[T3_LIST][2 * STRIP_H]; // the strip's eight RowInfo2 of a partial tor entry
    __shared__ __attribute__((aligned(16))) uint32_t bq[8 * T3_UNITS];         // the compacted blend's queue: per unit {coverage[4], pixel[4]}

    const int lane = threadIdx.x;
    const int g = lane >> 4, cg = lane & 15;                   // pixel rows g, g + 4; columns 4 cg .. 4 cg + 3
    const int cr = lane >> 3, ck = lane & 7;                   // cell fetch: lane = (row of the strip, k-th cell of the row)
    TRACE_DECL;
    TRACE_NOWAIT(0);
    const int width = FR->width, height = FR->height, tiles_x = FR->tiles_x;
    const swfr_style* __restrict__ styles = FR->styles;
    const Sources bitmaps = {FR->src.bitmaps, FR->src.filters, FR->src.fparams, FR->src.gradients};
    bool acc_clean = false;                                // the accumulators are zeroed before the first partial path needs them (many strips have none)

    // The wavefront's strips: slots blockIdx.x, + gridDim.x, ... of the launch list.  Two strips' records are in flight while a strip is
    // painted (a persistent launch -- fewer wavefronts than strips -- hides three of a strip's dependent round trips this way): the
    // descriptor of the strip after next, and the next strip's StripTop and first class bytes.  VECTOR loads, a few lanes each, moved
    // to the scalar unit with v_readlane when their turn comes: vector loads return in order, so the waits of the strip being painted
    // do not wait for them (scalar loads return out of order: every lgkmcnt wait would).
    const uint32_t n_slots = FR->n_strip_slots, G = gridDim.x;
    auto fetch_desc = [&](uint32_t slot) -> uint32_t {                   // lanes 0..3: the StripDesc of a slot ({~0, ..} beyond the list)
        uint32_t v = ~0u;
        if (slot < n_slots && lane < 4) v = reinterpret_cast<const uint32_t*>(FR->strips + slot)[lane];
        return v;
    };
    uint32_t n_wg = ~0u, n_band_begin = 0u, n_nb = 0u;                    // the next strip (wave-uniform) ...
    int n_tcol = 0, n_ty0 = 0;
    uint32_t n_top = 0u, n_cls[T3_CLS_PRE];                              // ... and, on their way: its StripTop (lanes 0..3), its first class bytes
    auto prepare = [&](uint32_t dv) {
        n_wg = (uint32_t)__builtin_amdgcn_readlane((int)dv, 0); n_band_begin = (uint32_t)__builtin_amdgcn_readlane((int)dv, 1); n_nb = (uint32_t)__builtin_amdgcn_readlane((int)dv, 2);
        n_top = 0u; n_tcol = 0; n_ty0 = height;
#pragma unroll
        for (int u = 0; u < T3_CLS_PRE; ++u) n_cls[u] = 0u;
        if (n_wg == ~0u) return;                                         // a padding slot of the launch list, or the list's end
        const uint32_t where = (uint32_t)__builtin_amdgcn_readlane((int)dv, 3);        // tile column | local tile-row << 16 (k2_bin's launch list)
        const int strip = (int)(n_wg % STRIPS_PER_TILE);
        n_tcol = (int)(where & 0xffffu);
        const int trow = (int)FR->band_first + (int)(where >> 16) * (int)FR->band_stride;
        n_ty0 = trow * TILE_H + strip * STRIP_H;
        // what the row pass knows about the strip as a whole (StripTop), and the strip's class byte per band entry
        if (lane < 4) n_top = reinterpret_cast<const uint32_t*>(FR->strip_top + n_wg)[lane];
        const uint8_t* cl = FR->cls + (size_t)STRIPS_PER_TILE * tiles_x * n_band_begin + (size_t)(n_tcol * STRIPS_PER_TILE + strip) * n_nb;
#pragma unroll
        for (int u = 0; u < T3_CLS_PRE; ++u) if ((uint32_t)(u * 64 + lane) < n_nb) n_cls[u] = (uint32_t)cl[u * 64 + lane];
    };
    uint32_t dv2;
    {
        const uint32_t dv1 = fetch_desc(blockIdx.x);
        dv2 = fetch_desc(blockIdx.x + G);
        prepare(dv1);
    }
    for (uint32_t w = blockIdx.x; w < n_slots; w += G) {
        TRACE(1);                                                        // descriptor fields in
        const uint32_t wg = n_wg, band_begin = n_band_begin, n_b = n_nb;
        const int tcol = n_tcol, ty0 = n_ty0;
        StripTop top;
        top.any = (uint32_t)__builtin_amdgcn_readlane((int)n_top, 0); top.pad = 0u;
        top.cover = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)n_top, 3) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)n_top, 2);
        uint32_t cpre[T3_CLS_PRE];
#pragma unroll
        for (int u = 0; u < T3_CLS_PRE; ++u) cpre[u] = n_cls[u];
        prepare(dv2);
        dv2 = fetch_desc(w + 2u * G);
        if (wg == ~0u) continue;                                         // a padding slot of the launch list
        TRACE(2);                                                        // strip descriptor in
        const int strip = (int)(wg % STRIPS_PER_TILE);
        const int tx0 = tcol * TILE_W;
        if (ty0 >= height) continue;
        const int cx0 = tx0 + 4 * cg, cy0 = ty0 + g;
        uint32_t px[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) px[j] = 0u;

        const uint8_t* mycls = FR->cls + (size_t)STRIPS_PER_TILE * tiles_x * band_begin + (size_t)(tcol * STRIPS_PER_TILE + strip) * n_b;   // this strip's class byte per band entry
        auto cls_chunk = [&](uint32_t c0) -> uint32_t {               // the class bytes of entries c0 .. c0 + 63 (c0 a multiple of 64, wave-uniform)
            if (c0 < 64u * T3_CLS_PRE) {
                uint32_t f = cpre[0];
#pragma unroll
                for (int u = 1; u < T3_CLS_PRE; ++u) if (c0 == 64u * (uint32_t)u) f = cpre[u];
                return f;
            }
            return c0 + lane < n_b ? (uint32_t)mycls[c0 + lane] : 0u;
        };
        // ---- occlusion: everything below the topmost opaque full cover is invisible in this strip -- and when nothing paints above it
        //      (or nothing paints at all) the strip is that colour: no walk.  The record is cleared for the next frame once read.
        const uint32_t cover_pos = (uint32_t)(top.cover >> 32);
        if (lane == 0 && top.any != 0u) {                                  // (depends on the loaded value: never overtakes the read)
#ifdef SWFR_EMU
            StripTop z; z.any = 0u; z.pad = 0u; z.cover = 0ull; FR->strip_top[wg] = z;
#else
            // (the zero is made here: as a loop invariant the compiler kept four zeroed registers alive across the whole strip loop --
            //  in the bitmap instance it spilled them at kernel entry, 16 bytes of scratch per lane)
            uint32_t z = 0u;
            asm volatile("" : "+v"(z));
            *reinterpret_cast<uint4*>(&FR->strip_top[wg]) = make_uint4(z, z, z, z);
#endif
        }
        const bool uniform = top.any == cover_pos;                        // (wave-uniform)
        if (cover_pos) {
            // the topmost opaque full cover paints every pixel of the strip with the record's colour: the walk starts BEHIND it
            // (its band entry is never fetched); with nothing above it the strip is done
            const uint32_t colour = (uint32_t)top.cover;
#pragma unroll
            for (int j = 0; j < 8; ++j) px[j] = colour;
        }
        const uint32_t start = cover_pos;                                // (first entry above the cover; 0: the whole list)
        TRACE(3);                                                        // class bytes in
        const uint32_t walk_end = min(n_b, top.any);                     // (nothing paints above the record's topmost painting entry)
        for (uint32_t c0 = uniform ? walk_end : (start & ~63u); c0 < walk_end; c0 += 64u) {
            const uint32_t f = cls_chunk(c0);
            const uint32_t bi = c0 + (uint32_t)lane;
            const bool hit = (f & CLS_NONEMPTY) != 0u && bi >= start;
            unsigned long long todo = __ballot(hit);
            STAT(0, n_b); STAT(1, __popcll(todo));
            while (todo) {                                               // wave-uniform
                // ---- stage the next T3_LIST non-empty entries (the lane that holds an entry's class byte fetches the entry -- two
                //      16-byte loads -- and, for a partial tor path, the headers of this strip's eight rows -- four more)
                const int rank = (int)__popcll(todo & ((1ull << lane) - 1ull));
                const bool take = ((todo >> lane) & 1ull) != 0ull && rank < T3_LIST;
                const unsigned long long taken = __ballot(take);
                const int n_st = (int)__popcll(taken);
                todo &= ~taken;
                if (take) {
                    const uint32_t bidx = band_begin + bi;
                    const uint4* s4 = reinterpret_cast<const uint4*>(&FR->band_list[bidx]);
                    const uint4 q0 = s4[0], q1 = s4[1];
                    uint4 h0 = make_uint4(0u, 0u, 0u, 0u), h1 = h0, h2 = h0, h3 = h0;
                    const bool part = (f & (CLS_PARTIAL | CLS_BOX)) == CLS_PARTIAL;
                    if (part) {
                        const uint4* r4 = reinterpret_cast<const uint4*>(&FR->rows[(size_t)bidx * TILE_H + (uint32_t)(strip * STRIP_H)]);
                        h0 = r4[0]; h1 = r4[1]; h2 = r4[2]; h3 = r4[3];
                    }
                    uint4* e4 = reinterpret_cast<uint4*>(&ent[rank][0]);
                    e4[0] = q0; e4[1] = q1; ent[rank][8] = f;
                    if (part) { uint4* d4 = reinterpret_cast<uint4*>(&hdr[rank][0]); d4[0] = h0; d4[1] = h1; d4[2] = h2; d4[3] = h3; }
                }
                lds_barrier();
                TRACE(4);                                               // entries + row headers in
                // ---- painter's order walk
#ifdef ABL_T_NOWALK
                if (n_st < 0)
#endif
                for (int li = 0; li < n_st; ++li) {
                    // per-entry fields are wave-uniform: readfirstlane moves them (and everything computed from them) to the scalar unit
                    const uint4 ea = *reinterpret_cast<const uint4*>(&ent[li][0]);
                    const uint4 eb = *reinterpret_cast<const uint4*>(&ent[li][4]);
                    const uint32_t fe = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent[li][8]);
                    const uint32_t xw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ea.x), yw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ea.y);
                    const int e_xmin = (int)(int16_t)(xw & 0xffffu), e_xmax = (int)(int16_t)(xw >> 16);
                    const int e_ymin = (int)(int16_t)(yw & 0xffffu), e_ymax = (int)(int16_t)(yw >> 16);
                    const uint32_t eflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)ea.z), solid = (uint32_t)__builtin_amdgcn_readfirstlane((int)ea.w);
                    const uint32_t style = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb.x);
                    const int row_lo = max(e_ymin, ty0) - ty0, row_hi = min(min(e_ymax, ty0 + STRIP_H), height) - ty0;
                    if (row_hi <= row_lo) continue;                    // the path misses this strip of the tile
                    STAT(2, 1);
#ifdef ABL_T_NOPARTIAL
                    if ((fe & (CLS_PARTIAL | CLS_BOX)) == CLS_PARTIAL) continue;
#endif
                    uint32_t al[8];
                    if (fe & CLS_BOX) {
                        // ---- rectilinear (A.6): exact area of disjoint boxes, alpha = (c>>8) - (c>>16)
                        const uint32_t e_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb.y), e_nedges = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb.z);
                        // (one pixel row of the lane at a time: eight running sums at once cost the shaded instances a wavefront per SIMD)
#pragma unroll 1
                        for (int h = 0; h < 2; ++h) {
                            uint32_t cov[4] = {0u, 0u, 0u, 0u};
                            const int cy = cy0 + 4 * h;
                            for (uint32_t k = 0; k < e_nedges; ++k) {
                                const swfr_edge bx = FR->raw[e_first + k];
                                const int wy = min(bx.y2, (cy + 1) * 256) - max(bx.y1, cy * 256);
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int cx = cx0 + i;
                                    const int wx = min(bx.x2, (cx + 1) * 256) - max(bx.x1, cx * 256);
                                    if (wx > 0 && wy > 0) cov[i] += (uint32_t)(wx * wy);
                                }
                            }
                            const int rr = g + 4 * h;
                            const bool in = rr >= row_lo && rr < row_hi;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const uint32_t a = in ? (((cov[i] >> 8) - (cov[i] >> 16)) & 255u) : 0u;
                                if (h == 0) al[i] = a; else al[4 + i] = a;
                            }
                        }
                    } else if (fe & CLS_PARTIAL) {
                        // ---- tor (A.5): the cells of this strip's eight rows of the path, lane = (row, k-th cell)
                        STAT(3, 1);
                        if (!acc_clean) {                                    // (wave-uniform)
                            *reinterpret_cast<int4*>(&acc[g][4 * cg]) = make_int4(0, 0, 0, 0);     // (the padding columns are never touched)
                            *reinterpret_cast<int4*>(&acc[g + 4][4 * cg]) = make_int4(0, 0, 0, 0);
                            acc_clean = true;                               // every path leaves them empty behind itself
                            lds_barrier();
                        }
                        uint32_t off = 0; int n_c = 0;
                        if (ty0 + cr < height) {
                            const uint2 hq = *reinterpret_cast<const uint2*>(&hdr[li][2 * cr]);
                            off = hq.x; n_c = (int)(hq.y & 0xffffu);
                            if (off > FR->cell_slice - min((uint32_t)n_c, FR->cell_slice)) { atomicOr(&FR->counters[C2_ERROR], E2_CELL_RANGE); n_c = 0; }   // off + n_c > cell_slice
                        }
                        const Cell* __restrict__ cp = FR->cells + off;
                        // the first sixteen cells of every row in one round trip, the rest (long shallow edges) eight per round
                        Cell c0c, c1c; c0c.w = 0; c1c.w = 0;
                        if (ck < n_c) c0c = cp[ck];
                        if (ck + 8 < n_c) c1c = cp[ck + 8];
                        STAT(4, n_c);
                        int* arow = acc[cr];
                        const int xrel = e_xmin - tx0;
                        auto add_cell = [&](Cell c, bool valid) {
                            const int i = cell_col(c) + xrel;                 // (columns are stored relative to the path's x_min)
                            const int v = (int)(c.w << 13) >> 13;             // covered height * 16384 + uncovered area
                            const int ua = (v << 18) >> 18;
                            const int hgt = (v - ua) << 6;                    // height << 20
                            // a cell left of the tile only adds its height to everything right of it: to column 0, without an area
                            if (valid && i < TILE_W) atomicAdd(&arow[max(i, 0)], i < 0 ? hgt : hgt + ua);
                        };
#ifndef ABL_T_NOACC
                        add_cell(c0c, ck < n_c);
                        add_cell(c1c, ck + 8 < n_c);
#endif
                        for (int kb = 16; __ballot(kb < n_c) != 0ull; kb += 8) {   // wave-uniform
                            Cell cc; cc.w = 0;
                            if (kb + ck < n_c) cc = cp[kb + ck];
                            add_cell(cc, kb + ck < n_c);
                        }
                        TRACE(5);                                           // cells in
                        lds_barrier();                                       // acc complete
                        // ---- prefix sums, coverage: four pixel rows per DPP sequence; the accumulators are cleared as they are read
                        {
                            int4* a0 = reinterpret_cast<int4*>(&acc[g][4 * cg]);
                            int4* a1 = reinterpret_cast<int4*>(&acc[g + 4][4 * cg]);
                            const int4 v0 = *a0, v1 = *a1;
                            *a0 = make_int4(0, 0, 0, 0); *a1 = make_int4(0, 0, 0, 0);
                            const int v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                            int ua[8], ch[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) { ua[j] = (int)((uint32_t)v[j] << 12) >> 12; ch[j] = (v[j] - ua[j]) >> 20; }
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                ch[4 * h + 1] += ch[4 * h]; ch[4 * h + 2] += ch[4 * h + 1]; ch[4 * h + 3] += ch[4 * h + 2];
                                int t = ch[4 * h + 3];                       // this lane's four columns; inclusive scan over the DPP row = the pixel row
#ifdef ABL_T_NOSCAN
                                const int ex = 0;
#else
                                t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, false);   // row_shr:1
                                t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, false);   // row_shr:2
                                t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, false);   // row_shr:4
                                t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, false);   // row_shr:8
                                const int ex = t - ch[4 * h + 3];
#endif
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int j = 4 * h + i;
                                    al[j] = (uint32_t)((__mul24(ch[j] + ex, 512 * 17) - __mul24(ua[j], 17) + 256) >> 9) & 255u;
                                }
                            }
                            // the converter's rectangle bounds what is painted (a cell at or beyond x_max is never emitted, so the coverage
                            // may not return to zero there); wave-uniform tests: most pairs lie inside
                            if (tx0 < e_xmin || tx0 + TILE_W > e_xmax) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) { const int cx = cx0 + (j & 3); if (cx < e_xmin || cx >= e_xmax) al[j] = 0u; }
                            }
                            if (row_lo > 0 || row_hi < STRIP_H) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) { const int rr = g + 4 * (j >> 2); if (rr < row_lo || rr >= row_hi) al[j] = 0u; }
                            }
                        }
                    } else {
                        // full cover: every in-frame pixel of the path's rows in this tile has coverage 255
                        const bool in0 = g >= row_lo && g < row_hi, in1 = g + 4 >= row_lo && g + 4 < row_hi;
                        if ((SHADERS == 0 || (eflags & BE_SOLID)) && ((eflags & BE_LERP) || (solid >> 24) == 0xffu)) {     // (wave-uniform) the colour itself
#pragma unroll
                            for (int j = 0; j < 4; ++j) { px[j] = in0 ? solid : px[j]; px[4 + j] = in1 ? solid : px[4 + j]; }
                            continue;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) { al[j] = in0 ? 255u : 0u; al[4 + j] = in1 ? 255u : 0u; }
                    }
#ifdef ABL_T_NOBLEND
                    for (int j = 0; j < 8; ++j) px[j] = al[j] == 255u ? solid : px[j];
#else
                    blend8<SHADERS>(px, al, eflags, solid, styles, style, bitmaps, cx0, cy0, bq, lane);
#endif
                }
                lds_barrier();                                           // ent / hdr are rewritten by the next round
            }
        }
        TRACE_NOWAIT(6);
        // ---- one store per pixel: premultiplied R,G,B,A bytes; a lane's four pixels of a row are one 16-byte store, the wavefront
        //      writes 256 contiguous bytes in each of four rows per instruction
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t p = px[j];
#ifdef SWFR_EMU
            px[j] = (p & 0xff00ff00u) | ((p >> 16) & 0xffu) | ((p & 0xffu) << 16);
#else
            px[j] = __builtin_amdgcn_perm(p, p, 0x03000102u);          // bytes 0 and 2 swapped
#endif
        }
        uint32_t* const fbp = fb_to ? fb_to : FR->fb;                     // (wave-uniform) this frame's own target, or the descriptor's
        const bool vec_ok = (width & 3) == 0 && ((uintptr_t)fbp & 15u) == 0u;      // (wave-uniform) every row of the frame starts 16-byte aligned
        if (vec_ok && tx0 + TILE_W <= width && ty0 + STRIP_H <= height) {
            // the strip lies inside the frame (wave-uniform: all but the last tile column / tile row): two stores, no per-lane tests
            uint32_t* rowp = fbp + (size_t)cy0 * (size_t)width + cx0;
            *reinterpret_cast<uint4*>(rowp) = make_uint4(px[0], px[1], px[2], px[3]);
            *reinterpret_cast<uint4*>(rowp + 4 * (size_t)width) = make_uint4(px[4], px[5], px[6], px[7]);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int y = cy0 + 4 * h;
                if (y >= height) continue;
                uint32_t* rowp = fbp + (size_t)y * (size_t)width + cx0;
                if (vec_ok && cx0 + 4 <= width) *reinterpret_cast<uint4*>(rowp) = make_uint4(px[4 * h], px[4 * h + 1], px[4 * h + 2], px[4 * h + 3]);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (cx0 + i < width) rowp[i] = px[4 * h + i];
                }
            }
        }
        TRACE(7);                                                        // stores acknowledged
        TRACE_OUT(2, w);
    }
}


#ifndef T2_WAVES
#define T2_WAVES 5                    // (96 VGPRs; round 4: five or six wavefronts per SIMD time alike on S1, five is a tenth faster on S0)
#endif
// (two entry points per kernel: one frame, its descriptor passed by value -- the fields arrive with the kernel arguments, no memory
//  round trip -- and a batch of frames, blockIdx.y indexing an array of descriptors in device memory)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(T2_WAVES))) void k2_tiles_solid_b(const Frame2* __restrict__ frames, uint32_t* fb_to) { tiles3_body<0>(FRAME_PTR(frames, blockIdx.y), fb_to); }
#ifndef T2_WAVES_SHADED
#define T2_WAVES_SHADED 4              // the samplers wait for texels: four wavefronts per SIMD (128 VGPRs) rather than the three 137 would allow
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(T2_WAVES_SHADED))) void k2_tiles_bitmap_b(const Frame2* __restrict__ frames, uint32_t* fb_to) { tiles3_body<1>(FRAME_PTR(frames, blockIdx.y), fb_to); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(T2_WAVES_SHADED))) void k2_tiles_shaded_b(const Frame2* __restrict__ frames, uint32_t* fb_to) { tiles3_body<2>(FRAME_PTR(frames, blockIdx.y), fb_to); }

// ---------------------------------------------------------------------------------------------
// launchers: `frames` is a device array of n_frames descriptors, blockIdx.y picks one
// ---------------------------------------------------------------------------------------------
// slow_kernels: 0 when the queued-row kernels will not be launched behind this k2_bin (their DevEdge records are then not written)
void launch2_bin(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t max_edges, uint32_t max_paths, uint32_t max_bands, uint32_t slow_kernels) {
    // tile-rows' workgroups, edge workgroups, path workgroups (a frame's own follow its own edge workgroups), + the eight that order the strips
    const uint32_t g = max_bands + (max_edges + BIN_THREADS - 1) / BIN_THREADS + (max_paths + BIN_THREADS - 1) / BIN_THREADS + XCDS;
    hipLaunchKernelGGL(k2_bin_b, dim3(g, n_frames), dim3(BIN_THREADS), 0, st, frames, slow_kernels);
}
void launch2_rows(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t max_chunks, uint32_t max_path_edges) {
    if (!max_chunks) return;
    if (max_path_edges > ROWS_STAGE) hipLaunchKernelGGL(k2_rows_wide_b, dim3(max_chunks, n_frames), dim3(64), 0, st, frames);
    else hipLaunchKernelGGL(k2_rows_b, dim3(max_chunks, n_frames), dim3(64), 0, st, frames);
}
void launch2_rows_slow(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t grid_slow, uint32_t grid_huge, uint32_t max_passes) {
    if (grid_slow) hipLaunchKernelGGL(k2_start_ranks_b, dim3(grid_slow / 4 + 1, n_frames), dim3(256), 0, st, frames);
    // a queued row whose edge-order history runs through another queued row is queued again for the next pass
    for (uint32_t pass = 0; pass < SLOW_PASSES; ++pass) {
        if (grid_slow) hipLaunchKernelGGL(k2_rows_slow_b, dim3(grid_slow, n_frames), dim3(64), 0, st, frames, pass);
        if (grid_huge) hipLaunchKernelGGL(k2_rows_huge_b, dim3(grid_huge, n_frames), dim3(HUGE_THREADS), 0, st, frames, pass);
        if (pass + 1 >= max_passes) break;
    }
}
// fb_to: where THIS launch's pixels go instead of the descriptors' framebuffer (one frame per launch only), or nullptr
void launch2_tiles(hipStream_t st, const Frame2* frames, uint32_t n_frames, uint32_t max_strips, uint32_t grid_cap, int shader_level, uint32_t* fb_to) {
    if (!max_strips) return;
    // Default shape (grid_cap == ~0u): a frame of more than T3_PAIR_FROM strips is launched as HALF as many
    // wavefronts as strips -- each paints slot k (the heavier half of the cost-ordered list) and slot k + grid (the lighter half), the
    // second strip's records fetched while the first is painted (round 4: S1 31.0-31.7 us per frame against 32.3; S2 2 %, the textured 4K frames 3 % faster);
    // smaller frames do not fill the GPU and keep one wavefront per strip.
    uint32_t g = max_strips < grid_cap ? max_strips : grid_cap;
    if (grid_cap == ~0u && max_strips > T3_PAIR_FROM) g = (max_strips + 1u) / 2u;
    if (shader_level >= 2) hipLaunchKernelGGL(k2_tiles_shaded_b, dim3(g, n_frames), dim3(64), 0, st, frames, fb_to);
    else if (shader_level == 1) hipLaunchKernelGGL(k2_tiles_bitmap_b, dim3(g, n_frames), dim3(64), 0, st, frames, fb_to);
    else hipLaunchKernelGGL(k2_tiles_solid_b, dim3(g, n_frames), dim3(64), 0, st, frames, fb_to);
}

}  // namespace swfr

#if defined(SWFR_TRACE) || defined(SWFR_TSTATS)
extern "C" __attribute__((visibility("default"))) int swfr_debug_trace(void* dst, size_t bytes) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const size_t n = bytes < sizeof(swfr::swfr_trace_buf) ? bytes : sizeof(swfr::swfr_trace_buf);
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(swfr::swfr_trace_buf), n, 0, hipMemcpyDeviceToHost) == hipSuccess ? (int)(n / 32) : -1;
}
#endif
