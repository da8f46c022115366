// device_types.hpp -- records shared by the HIP kernels and the host-side renderer.
#pragma once

#include <stdint.h>

#include "../../include/swfr.h"

namespace swfr {

constexpr int TILE_W = 64;   // one wavefront lane per pixel column
constexpr int TILE_H = 16;   // tile-rows ("bands") are the unit of band lists and of multi-GPU sharding
constexpr int STRIP_H = 8;   // pixel rows per k2_tiles wavefront: a 64x16 tile is rasterized as two independent 64x8 strips
constexpr int STRIPS_PER_TILE = TILE_H / STRIP_H;

// Per-edge constants of the tor scan converter (SURVEY.md A.5 make_edge), 96 bytes.
struct DevEdge {
    int32_t ytop, ybot;      // active sub-rows [ytop, ybot), 15 per pixel row, clamped to the path's rows
    int32_t x1, y1;          // upper end point of the line (24.8)
    int32_t dir;
    int32_t pad;             // paths with queued rows: position among the edges that start at the same sample row, in the order
                             // Cairo's sort of that bucket gives them (k2_start_ranks)
    int64_t ex;              // (x2 - x1) * 256
    int64_t dy;              // (y2 - y1) * 15 * 512, 0 for vertical edges
    int64_t dq, dr;          // per-sub-row slope: truncated quotient / remainder of ex*512 / dy
    double inv_dy;           // 1.0 / dy (correctly rounded): quotient estimates need one multiply, the integer fix-up makes them exact
    // constants of the rows the edge crosses completely (A.5 render_edge; all zero for vertical edges): over one pixel row x advances
    // by 7680 * ex / dy exactly, so the row's extent in units of 1/dy is the same for every such row
    double inv_dx;           // 1.0 / (7680 * |ex|)
    int64_t fr;              // floor_div(3840 * dy, 7680 * |ex|): sample rows per pixel column, quotient fq / remainder fr
    int64_t r15;             // floor_div(7680 * ex, dy): x advance per pixel row, quotient q15 / remainder r15 (edges at least 200/256 px tall)
    int32_t fq, q15;
};
static_assert(sizeof(DevEdge) == 96, "DevEdge layout");

// The same edge as the fast row routine of k2_rows uses it (rows3.hip): every remainder of the scan converter in units of
// 1 / D, D = 30 * (y2 - y1) -- Cairo's dy = 7680 * (y2 - y1) is 256 D and all of its remainders are multiples of 256 -- so that they
// fit 32 bits (D < 2^29 for end points within +-32768 px), with the steps in FLOOR form (quotient, remainder in [0, D): one add, one
// compare, one carry per step; the floor representation of a sum is unique, so this is Cairo's "normalise once").  Products that
// need more than 32 bits (A * DX < 2^53) are exact in double precision.  80 bytes, five 16-byte blocks in the order the kernel reads them.
struct FastEdge {
    int32_t x1, a0, DX, D;           // x1 (24.8); a0 = 256 - 30 y1, so that A(s) = (2 s + 1) 256 - 30 y1 = 512 s + a0; DX = x2 - x1 (0: vertical); D = 30 (y2 - y1)
    double invD;                     // 1 / D
    int32_t q15, r15;                // x advance per pixel row, 7680 DX / D (edges at least 200/256 px tall; else 0)
    int32_t hq, hr;                  // Cairo's half sample row (truncated halves of the truncated per-sample step), floor form
    int32_t dqf, drf;                // x advance per sample row, 512 DX / D, floor form
    int32_t ytop, ybot;              // active sample rows [ytop, ybot) (as DevEdge)
    int32_t dir, fq;                 // fq: sample rows per pixel column, floor(256 D / W), W = 512 |DX|
    double invW, fr;                 // 1 / W; fr = 256 D - fq W
};
static_assert(sizeof(FastEdge) == 80, "FastEdge layout");

using DevPath = swfr_path;   // 40 bytes: first_edge, n_edges, kind, fill_rule, style, lerp, pixel rect

enum : uint32_t { ROW_EMPTY = 0, ROW_FULL = 1, ROW_SUB = 2 };
// roles word of an (edge, row) pair inside the row kernels: FULL rows carry REC_FULL | 1 (left end of a span, +1) or | 2 (right
// end, -1); SUB rows carry 15 two-bit fields, field s = 1 (a span opens at this edge in sample row s) or 2 (closes)
constexpr uint32_t REC_FULL = 0x80000000u;
constexpr uint32_t REC_CELLS = 0x40000000u;    // (net height in bits 8..15)
constexpr int ROWS_CHUNK = 64;       // most pixel rows per k2_rows wavefront (one lane per row)

// flags of a band list entry (BandEntry2)
enum : uint32_t { BE_BOXES = 1u, BE_LERP = 2u, BE_SOLID = 4u, BE_OPAQUE_COVER = 8u /* solid, alpha 255, lerp blend */ };

// One k2_rows wavefront: `rows` (<= 64) consecutive pixel rows of one path.  rec_base is the path's first (edge, row) incidence
// (host-computed prefix): the chunk's cells go to a fixed region behind it, so cell allocation needs no global atomics.
struct ChunkInfo {
    uint32_t path, first_row, rec_base, rows;
    uint32_t slot0;          // index into band_slots of the (path, tile-row) pair of the chunk's first tile-row; ~0u if none
    uint32_t first_edge, n_edges;   // the path's edge range again: the wavefront's edge loads start with the chunk record, beside the path's
    uint32_t pad;
};

// One (path, tile-row) pair: where its BandEntry goes.  The host assigns the slots (painter's order inside every
// tile-row) from the paths' pixel rectangles while it sizes the band lists.
struct BandSlot {
    uint32_t path;
    uint32_t slot;           // index into band_list; the tile-row is recovered from band_off
    uint32_t band;
    uint32_t pad;
};

struct DevBitmap {
    const uint32_t* pixels;  // premultiplied ARGB, tight rows
    uint32_t width, height;
};

// CAIRO_FILTER_GOOD of a bitmap style when Cairo does not downgrade it to bilinear (minification below 0.75): pixman's
// separable convolution.  x_off / y_off index the handle's table of 16.16 weights: (1 << bits) phases x width taps.
// Every bitmap style also carries pixman's sample position: the pattern matrix rounded to 16.16 and anchored at the centre of the
// drawing operation's rectangle (host: pixman_transform_of).  Position of destination pixel (px, py), 16.16:
// base + px * (m00, m10) + py * (m01, m11) -- exactly what pixman_transform_point_3d gives for the pixel centre.
struct DevFilter {
    int32_t on, cw, ch, xbits, ybits;
    uint32_t x_off, y_off;
    int32_t m00, m01, m10, m11;
    int32_t pad;
    int64_t base_x, base_y;
    // what the bitmap shader needs of the style and of the bitmap table, so that one load of this record is all it waits for
    const uint32_t* pixels;  // premultiplied ARGB, tight rows (bitmap styles)
    uint32_t width, height;
    uint32_t extend;         // 0 none, 1 repeat
    uint32_t kind;           // SWFR_STYLE_*
    uint32_t pad2[2];
};
static_assert(sizeof(DevFilter) == 96, "DevFilter layout");

// A radial gradient as pixman holds it for one drawing operation (host: radial_of; cairo-image-source.c _pixman_image_for_gradient,
// pixman-radial-gradient.c, pixman-gradient-walker.c): the 16.16 sample position as for bitmaps, the circles after Cairo's
// fit-to-range scaling, the constant terms of the quadratic, and one single-precision colour ramp per interval between stops
// (PAD sentinels at both ends).  DevFilter::pad of the style holds its index + 1 into the handle's gradient table.
struct DevGradient {
    int64_t base_x, base_y;
    int32_t m00, m01, m10, m11;
    int32_t c1x, c1y, c1r, dx, dy, dr;       // 16.16
    int32_t n_intervals, pad;
    double a, inva, mindr;
    int32_t x[SWFR_MAX_STOPS + 2];           // interval boundaries: INT32_MIN, stop offsets, INT32_MAX
    float ramp[SWFR_MAX_STOPS + 1][8];       // a_s, a_b, r_s, r_b, g_s, g_b, b_s, b_b
};

// ---------------------------------------------------------------------------------------------------------------------------
// The row pass (raster2.hip) hands the tile pass CELLS, not edges.  A cell is one entry of Cairo's per-row cell list
// (covered_height / uncovered_area of one pixel column), already clipped to the converter's column range; the cells
// of one (path, pixel row) are contiguous, RowInfo2 says where.  The tile pass does no edge arithmetic at all.
// ---------------------------------------------------------------------------------------------------------------------------
// Packed into one word: column relative to the path's x_min (13 bits: a path is at most 8192 px wide -- the host refuses wider ones)
// and, in the low 19 bits, V = covered height * 16384 + uncovered area as one signed number (height: -15 .. 15 sample rows per edge,
// |area| <= 2 * 255 * 15 < 8192, so the area is V's low 14 bits sign-extended and the height what remains).  A producer whose area
// is a multiple of its height gets V with one multiplication.
struct Cell {
    uint32_t w;
};
static_assert(sizeof(Cell) == 4, "Cell layout");
constexpr int MAX_PATH_WIDTH = 8192;
constexpr int CELL_H = 16384;                                                 // V = height * CELL_H + area
constexpr Cell make_cell_v(int col_rel, int v) { return Cell{((uint32_t)col_rel << 19) | ((uint32_t)v & 0x7ffffu)}; }
constexpr Cell make_cell(int col_rel, int ch, int ua) { return make_cell_v(col_rel, ch * CELL_H + ua); }
constexpr int cell_col(Cell c) { return (int)(c.w >> 19); }                  // relative to the path's x_min
constexpr int cell_ua(Cell c) { return (int)(c.w << 18) >> 18; }
constexpr int cell_ch(Cell c) { return (((int)(c.w << 13) >> 13) - cell_ua(c)) >> 14; }
constexpr int MAX_CELLS_PER_EDGE_ROW = 17;   // a FULL-row edge has at most 15 + 2 cells with a non-zero height, a sampled one 15

// per (band entry, pixel row of its tile-row): indexed entry * TILE_H + (y % TILE_H), so a tile finds it from the band list position
struct RowInfo2 {
    uint32_t off;            // first cell
    uint16_t n;              // cells
    uint16_t mode;           // ROW_* (diagnostics)
};
enum : uint32_t { ROW_DEFER = 3,     // left to the slow-row kernel (coincident edges, or more active edges than the fast kernel keeps)
                  ROW_FOREIGN = 4 }; // another handle's tile-row (multi-GPU): never decided here

struct BandEntry2 {
    int16_t x_min, x_max, y_min, y_max;   // the converter's pixel rectangle
    uint32_t flags;                       // BE_* | tile-row << 8
    uint32_t solid;                       // premultiplied pixel of a solid style
    uint32_t style;
    uint32_t first_edge, n_edges;         // boxes paths
    uint32_t path;
};
static_assert(sizeof(BandEntry2) == 32, "BandEntry2 layout");

// a pixel row the fast row kernel leaves to the slow one
struct SlowRow {
    uint32_t path;
    int32_t row;             // absolute pixel row
    uint32_t ri;             // index of its RowInfo2
    uint32_t pad;
};

// the kernels' counters (per frame in flight)
enum : uint32_t { C2_ERROR = 0, C2_SLOW = 1, C2_HUGE = 2, C2_TIE_ROWS = 3, C2_TIE_PAIRTEST_SKIPPED = 4, C2_TIE_SORT_OVERFLOW = 5, C2_TIE_DEPTH = 6,
                  C2_CELLS = 7, C2_HEAD = 8 /* .. 15: cell allocation heads */, C2_PATHQ = 16,
                  C2_SLOWQ = 16 /* + pass (1..3): rows queued again for a later pass */, C2_HUGEQ = 20 /* + pass (1..3) */, C2_WORDS = 32 };
constexpr uint32_t SLOW_PASSES = 4;      // a queued row whose history runs through another queued row waits for the next pass
constexpr uint32_t C2_HEADS = 8;
// error bits
enum : uint32_t { E2_ACTIVE_EDGES = 1u, E2_ROW_TABLE = 2u, E2_CELL_RANGE = 4u, E2_CELL_ARENA = 8u, E2_SLOW_QUEUE = 16u, E2_START_GROUP = 32u };

// what the shader needs besides the style itself
struct Sources {
    const DevBitmap* bitmaps;
    const DevFilter* filters;    // per style index
    const int32_t* fparams;
    const DevGradient* gradients;
};

// one k2_tiles wavefront of the launch list: which strip, and its tile-row's slice of the band list
struct StripDesc {
    uint32_t wg;             // tile * STRIPS_PER_TILE + strip, tiles counted over the handle's own tile-rows
    uint32_t band_begin, n_b;
    uint32_t pad;            // the strip's tile column | its local tile-row << 16 (so that k2_tiles divides nothing)
};

// What the row pass (and k2_bin, for box paths) already knows about a strip as a whole, so that the tile pass need not walk anything
// for a strip without a partial (path, strip) pair: the position (1-based, in its tile-row's band list) of the topmost entry that
// paints anything in the strip, and of the topmost OPAQUE FULL COVER together with its colour -- maxima kept with atomics (the colour
// rides in the low half of a 64-bit maximum).  any == 0: nothing is painted; cover's position == any: the strip is that colour.
struct StripTop {
    uint32_t any, pad;
    unsigned long long cover;        // position << 32 | premultiplied pixel
};
static_assert(sizeof(StripTop) == 16, "StripTop layout");

// Everything the kernels need to know about one frame (device memory; blockIdx.y indexes an array of these).
constexpr uint32_t XCDS = 8;            // a launch's workgroups go round-robin over the chip's eight XCDs, each with its own L2
// slots of the tile pass's launch list: slot % XCDS = the XCD the hardware hands the workgroup to = (local tile-row) % XCDS, so that
// everything the strips of one tile-row read (its band list, class bytes, row headers, cells) is fetched into ONE L2; the classes
// are padded to the size of the largest (slots without a strip: wg = ~0u)
inline uint32_t strip_slots(uint32_t local_tile_rows, uint32_t strips_per_row) { return XCDS * ((local_tile_rows + XCDS - 1) / XCDS) * strips_per_row; }

struct Frame2 {
    // the scene (read-only)
    const swfr_edge* raw; const DevPath* paths;
    const swfr_style* styles;            // style i at byte offset i * style_stride: whole swfr_style records, or -- a scene of solid colours only -- just their first eight bytes {kind, pixel}
    Sources src;
    // layout of the tables below, from the paths' rectangles alone (host: prefix sums over the paths / the tile-rows)
    const uint32_t* path_chunks; const uint32_t* path_slots; const uint32_t* path_inc;   // per path: first chunk, first band slot, first (edge, row) pair
    const uint32_t* band_off;                                          // per tile-row: first entry of its band list
    const uint32_t* path_bands;                                        // per path: first | last << 16 tile-row of its rectangle (0xffff: none); 16 paths of padding behind the last
    // binning, per frame in flight (kernel-written: k2_bin, k2_rows)
    ChunkInfo* chunks; BandSlot* band_slots; StripDesc* strips;
    uint32_t* strip_cost;                                              // zero between frames (the ordering workgroups of k2_bin clear what they have read)
    // per frame in flight (kernel-written)
    DevEdge* edges; BandEntry2* band_list; uint8_t* cls; RowInfo2* rows; Cell* cells; SlowRow* slow; SlowRow* huge; uint32_t* counters;
    uint32_t* path_flag; uint32_t* path_queue;     // paths with queued rows: their edges get start ranks (k2_start_ranks)
    uint32_t* fb;
    StripTop* strip_top;                 // per strip of the handle (cleared at upload, and by the tile pass once read)
    uint32_t n_edges, n_paths, n_chunks, n_bands, n_strips, cell_slice, slow_cap;
    int32_t width, height, tiles_x;
    uint32_t fast_limit;     // active edges per row the fast routine of k2_rows keeps (<= 8)
    uint32_t style_stride;   // bytes between two styles (sizeof(swfr_style) or 8)
    uint32_t cell_main;      // cells [0, cell_main) belong to the chunk wavefronts of k2_rows, the rest to the slow rows' bump allocator
    uint32_t chunk_rows;     // pixel rows per k2_rows wavefront (16, 32 or 64)
    uint32_t chunk_cap;      // capacity of chunks[] (the host sizes it from the paths' rectangles)
    uint32_t strip_order;    // 0: strips in row-major order, 1: heaviest first
    uint32_t n_strip_slots;             // launch list slots of the tile pass: XCDS * ceil(local tile-rows / XCDS) * strips per tile-row (strip_slots())
    uint32_t band_first, band_stride;   // the handle's tile-rows: band_first + l * band_stride, l < n_strips / (STRIPS_PER_TILE * tiles_x)
                                        // (interleaved over the ranks: stride = ranks; one contiguous block per rank: stride = 1)
};

}  // namespace swfr
