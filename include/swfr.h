/*
 * swfr.h -- C-ABI of libswfr.so, the MI355X-native SWF vector rasterizer.
 *
 * This is the drop-in boundary for the reference's render(Stage) -> RGBA path.  The reference
 * (open-flash/swf-renderer) has no FFI for a rasterizer; its seam is the trait/interface layer,
 * and each entry point below names the reference interface it replaces:
 *
 *   swfr_create / swfr_destroy      HeadlessGfxRenderer::new(&instance, w, h)   rs/src/headless_renderer.rs:60-64
 *                                   new NodeCanvasRenderer(width, height)       ts/src/lib/renderers/node-canvas-renderer.ts:11-15
 *                                   createRenderer / destroyRenderer (handles)  rs/src/wasm.rs:60-76
 *   swfr_register_shape             ClientAssetStore::register_shape            rs/src/asset.rs:9-12
 *                                   (decode: ts/src/lib/shape/decode-swf-shape.ts:22-39)
 *   swfr_register_morph_shape       ClientAssetStore::register_morph_shape      rs/src/asset.rs:9-12
 *                                   (decode: ts/src/lib/shape/decode-swf-morph-shape.ts:21-41)
 *   swfr_register_bitmap(_tag)      Renderer.addBitmap(tag)                     ts/src/lib/renderer.ts:4-8,
 *                                   ts/src/lib/renderers/node-canvas-bitmap-service.ts:14-37
 *   swfr_render                     SwfRenderer::render(&mut self, Stage)       rs/src/swf_renderer.rs:3-5
 *                                   Renderer.render(stage)                      ts/src/lib/renderer.ts:4-8
 *                                   (scene walk: ts/src/lib/renderers/canvas-renderer.ts:69-350)
 *   swfr_read_image                 HeadlessGfxRenderer::get_image() -> Image   rs/src/headless_renderer.rs:233-244,
 *                                   Image{meta{width,height,stride},data}       rs/src/renderer.rs:88-103
 *   swfr_last_error                 Result<_, &'static str> / thrown Error      rs/src/headless_renderer.rs:64,233
 *
 * Conventions mirrored from the reference: definitions are borrowed for the call and copied
 * (rs/src/renderer.rs:24-64); render is synchronous and returns after the GPU is idle
 * (rs/src/headless_renderer.rs:703-712); one handle = one thread at a time; errors are an int
 * status plus a per-handle message, nothing is thrown across the boundary.
 *
 * Types mirror swf-tree 0.8 as evidenced by the reference fixtures (tests/<set>/<name>/ast.json).
 * Plain C: pointers + sizes only, no C++ or torch types.
 */
#ifndef SWFR_H
#define SWFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWFR_ABI_VERSION 1

typedef struct swfr_renderer swfr_renderer;               /* opaque handle */

enum {
    SWFR_OK = 0,
    SWFR_ERR_INVALID = 1,          /* bad argument / malformed definition (reference: "Invalid fill ID" etc.) */
    SWFR_ERR_NOT_IMPLEMENTED = 2,  /* reference throws NotImplementedFillStyle / NotImplementedLineStyle */
    SWFR_ERR_NOT_FOUND = 3,        /* unknown shape / bitmap id (reference: BitmapNotFound) */
    SWFR_ERR_NO_DEVICE = 4,        /* no HIP device, or a host-only handle was asked to rasterize */
    SWFR_ERR_DEVICE = 5,           /* HIP runtime error */
    SWFR_ERR_CAPACITY = 6          /* a capacity limit of the scan converter: more than 8192 active edges of one path in a pixel row (or
                                      starting at one sample row), or a limit of the replay of Cairo's
                                      edge-list order for coincident edges (swfr_get_stats) -- the frame is refused, never approximated */
};

/* ---- swf-tree value types ---------------------------------------------------------------- */
typedef struct { uint8_t r, g, b, a; } swfr_rgba8;                          /* StraightSRgba8 */
typedef struct { int32_t x_min, x_max, y_min, y_max; } swfr_rect;           /* twips */
typedef struct {                                                            /* swf_tree::Matrix */
    int32_t scale_x, scale_y, rotate_skew0, rotate_skew1;                   /* Sfixed16P16 epsilons */
    int32_t translate_x, translate_y;                                       /* twips */
} swfr_matrix;

enum { SWFR_FILL_SOLID = 0, SWFR_FILL_LINEAR_GRADIENT = 1, SWFR_FILL_RADIAL_GRADIENT = 2,
       SWFR_FILL_FOCAL_GRADIENT = 3, SWFR_FILL_BITMAP = 4 };

typedef struct { uint8_t ratio; swfr_rgba8 color; swfr_rgba8 morph_color; } swfr_color_stop;

typedef struct {
    uint32_t type;                       /* SWFR_FILL_* */
    swfr_rgba8 color, morph_color;       /* solid (morph_color: DefineMorphShape only) */
    swfr_matrix matrix;                  /* gradients, bitmap */
    uint32_t n_stops;                    /* gradients */
    const swfr_color_stop *stops;
    int32_t focal_point;                 /* focal gradient, Sfixed8P8 epsilons */
    uint32_t bitmap_id;                  /* bitmap */
    uint8_t repeating, smoothed;
} swfr_fill_style;

typedef struct {
    uint32_t width, morph_width;         /* twips; caps/joins of the SWF style are dropped by the
                                            reference (decode-swf-shape.ts:144-149) */
    swfr_fill_style fill;
} swfr_line_style;

typedef struct {
    uint32_t n_fill; const swfr_fill_style *fill;
    uint32_t n_line; const swfr_line_style *line;
} swfr_styles;

enum { SWFR_RECORD_EDGE = 0, SWFR_RECORD_STYLE_CHANGE = 1 };

typedef struct {
    uint32_t type;                       /* SWFR_RECORD_* */
    /* edge */
    int32_t delta_x, delta_y;
    uint8_t has_control_delta; int32_t control_delta_x, control_delta_y;
    int32_t morph_delta_x, morph_delta_y;                              /* morph shapes */
    uint8_t has_morph_control_delta; int32_t morph_control_delta_x, morph_control_delta_y;
    /* style change */
    uint8_t has_move_to; int32_t move_to_x, move_to_y;
    uint8_t has_morph_move_to; int32_t morph_move_to_x, morph_move_to_y;
    uint8_t has_left_fill; uint32_t left_fill;
    uint8_t has_right_fill; uint32_t right_fill;
    uint8_t has_line_style; uint32_t line_style;
    const swfr_styles *new_styles;       /* NULL when absent */
} swfr_shape_record;

typedef struct {                         /* tags::DefineShape / tags::DefineMorphShape */
    uint32_t id;
    swfr_rect bounds, morph_bounds;
    swfr_styles initial_styles;
    uint32_t n_records; const swfr_shape_record *records;
} swfr_define_shape;

/* ---- stage ------------------------------------------------------------------------------- */
enum { SWFR_OBJECT_SHAPE = 0, SWFR_OBJECT_MORPH_SHAPE = 1, SWFR_OBJECT_CONTAINER = 2 };

typedef struct swfr_display_object {
    uint32_t type;                       /* SWFR_OBJECT_* (ts/src/lib/display/display-object-type.ts) */
    uint32_t id;                         /* value returned by swfr_register_shape / swfr_register_morph_shape */
    uint8_t has_matrix; swfr_matrix matrix;
    double ratio;                        /* morph shapes: [0,1] (ts/src/lib/display/morph-shape.ts:9);
                                            Rust MorphRatio(u16) maps as ratio = v / 65535.0 */
    uint32_t n_children; const struct swfr_display_object *children;   /* containers */
} swfr_display_object;

typedef struct {
    uint32_t width, height;              /* carried for API parity; the frame size is the handle's */
    swfr_rgba8 background_color;         /* not painted by the reference (canvas-renderer.ts:72) */
    uint32_t n_children; const swfr_display_object *children;
} swfr_stage;

/* ---- configuration ------------------------------------------------------------------------ */
#define SWFR_DEVICE_HOST_ONLY (-1)       /* decode + geometry only; swfr_render fails with NO_DEVICE */
#define SWFR_FLAG_EVEN_ODD 1u            /* fill rule override (the reference always uses nonzero) */
#define SWFR_FLAG_BANDS_CONTIGUOUS 2u    /* multi-GPU: this handle's tile-rows are one contiguous block, see band_index below */

typedef struct {
    int32_t device;                      /* HIP device ordinal, or SWFR_DEVICE_HOST_ONLY */
    uint32_t flags;
    uint32_t band_index, band_count;     /* multi-GPU: this handle rasterizes the tile-rows (16 pixel rows each) t with
                                            t % band_count == band_index (0,0|1 = all), or, with SWFR_FLAG_BANDS_CONTIGUOUS,
                                            the block [band_index * n, (band_index + 1) * n), n = ceil(tile-rows / band_count):
                                            a rank's part of the frame is then one contiguous range of rows, which a gather can
                                            deposit in the assembled image without a copy */
} swfr_config;

/* ---- lifecycle / assets / render ----------------------------------------------------------- */
int  swfr_create(uint32_t width, uint32_t height, const swfr_config *cfg, swfr_renderer **out);
void swfr_destroy(swfr_renderer *r);
const char *swfr_last_error(const swfr_renderer *r);
uint32_t swfr_abi_version(void);

int  swfr_register_shape(swfr_renderer *r, const swfr_define_shape *tag, uint32_t *out_id);
int  swfr_register_morph_shape(swfr_renderer *r, const swfr_define_shape *tag, uint32_t *out_id);
int  swfr_register_bitmap(swfr_renderer *r, uint32_t id, uint32_t width, uint32_t height,
                          const uint8_t *rgba_straight, size_t stride);
/* Renderer.addBitmap(tag) with the tag's own bytes (ts/src/lib/renderers/node-canvas-bitmap-service.ts:14-37): media type
   "image/x-swf-bmp" is decoded by the library (decodeXSwfBmpSync, ts/src/lib/decode-x-swf-bmp.ts:9-41: format 3, zlib colour-mapped,
   rows padded to 4 bytes, opaque palette, an index past the palette is opaque black) and registered as by swfr_register_bitmap;
   any other media type, or another format id, is SWFR_ERR_NOT_IMPLEMENTED like the reference's "NotImplemented: Support for ...
   images" / "UnsupportedXSwfBmpFormatId"; a damaged stream is SWFR_ERR_INVALID. */
int  swfr_register_bitmap_tag(swfr_renderer *r, uint32_t id, const char *media_type, const uint8_t *data, size_t len);
/* The decoder alone (no handle, no device): dimensions, and -- when rgba is not NULL -- straight RGBA8 with tight rows into the
   caller's buffer of rgba_cap bytes (SWFR_ERR_CAPACITY when it is too small; call with NULL first to size it). */
int  swfr_decode_x_swf_bmp(const uint8_t *data, size_t len, uint32_t *width, uint32_t *height, uint8_t *rgba, size_t rgba_cap);
int  swfr_render(swfr_renderer *r, const swfr_stage *stage);               /* blocking */
int  swfr_read_image(swfr_renderer *r, uint8_t *dst, size_t dst_stride, int premultiplied);
/* Mapped read-back in two halves (HeadlessGfxRenderer::get_image maps its staging buffer the same way:
   rs/src/headless_renderer.rs:725-868): _async queues the device-to-host copy of the last frame into the handle's PINNED staging
   buffer behind the frame's kernels and returns; _wait blocks until it has arrived and hands out the staging buffer itself
   (width*4-byte rows, valid until the next read-back or swfr_destroy) -- no second copy on the host.  One read-back in flight per
   handle; ANY render call (swfr_render, swfr_render_resident(_async / _to), swfr_render_batch) may be issued before _wait: the
   host build overlaps the copy, and the streams of the handle's other frame sets are made to wait for the copy on the device before
   a frame is rasterized, so the image read back is never torn. */
int  swfr_read_image_async(swfr_renderer *r, int premultiplied);
int  swfr_read_image_wait(swfr_renderer *r, const uint8_t **data, size_t *stride);
/* render + mapped read-back of every frame, timed below the ABI (the reference's test loop: render, get_image). */
int  swfr_render_sequence_readback(swfr_renderer *r, const swfr_stage *stages, uint32_t n_stages, uint32_t repeat, int premultiplied,
                                   int overlap, double *seconds, uint64_t *checksum);

/* ---- low-level entry: the hot path proper (edge list -> RGBA8 in HBM) ---------------------- */
/* One edge of a flattened, limit-clipped polygon in 24.8 device coordinates: the line
   (x1,y1)-(x2,y2) with y1 < y2 is active for y in [top,bottom), y1 <= top < bottom <= y2 (an edge
   with top >= bottom is never active; any other edge active outside its own line is refused with
   SWFR_ERR_INVALID); dir is the winding direction.
   `reserved` is the index of the owning path (the device bins edges by it): swfr_build_frame fills
   it in, swfr_upload_edges / swfr_render_edges overwrite it from the paths' edge ranges. */
typedef struct { int32_t x1, y1, x2, y2, top, bottom, dir, reserved; } swfr_edge;

enum { SWFR_PATH_TOR = 0,   /* general polygon: Cairo "tor" 15x256 scan conversion */
       SWFR_PATH_BOXES = 1  /* rectilinear: `edges` hold disjoint boxes (x1,y1)-(x2,y2) */ };

typedef struct {
    uint32_t first_edge, n_edges;
    uint32_t kind;                       /* SWFR_PATH_* */
    uint32_t fill_rule;                  /* 0 nonzero, 1 even-odd */
    uint32_t style;                      /* index into styles */
    uint32_t lerp;                       /* 1: SOURCE-lerp blend (opaque source or clear surface), 0: OVER */
    int32_t x_min, y_min, x_max, y_max;  /* pixel rectangle of the converter (polygon extents ∩ frame) */
} swfr_path;

enum { SWFR_STYLE_SOLID = 0, SWFR_STYLE_RADIAL = 1, SWFR_STYLE_LINEAR = 2, SWFR_STYLE_BITMAP = 3 };
#define SWFR_MAX_STOPS 16

typedef struct {
    uint32_t kind;                       /* SWFR_STYLE_* */
    uint32_t pixel;                      /* solid: premultiplied 0xAARRGGBB */
    double inv[6];                       /* device -> pattern space (xx, yx, xy, yy, x0, y0): Cairo's pattern matrix.  A bitmap
                                            style belongs to one drawing operation: pixman's 16.16 transform is anchored at the
                                            centre of the operation's pixel rectangle, so paths sharing it must share x_min..y_max */
    double c0x, c0y, r0, c1x, c1y, r1;   /* radial (linear: c0 -> c1) */
    uint32_t n_stops;
    float stop_offset[SWFR_MAX_STOPS];
    float stop_rgba[SWFR_MAX_STOPS][4];  /* straight, 0..1 */
    uint32_t bitmap;                     /* registered bitmap id */
    uint32_t extend;                     /* 0 none, 1 repeat */
} swfr_style;

/* Upload a scene (kept resident in HBM), then rasterize it.  swfr_render_edges = upload + render. */
int  swfr_upload_edges(swfr_renderer *r, const swfr_edge *edges, size_t n_edges,
                       const swfr_path *paths, size_t n_paths,
                       const swfr_style *styles, size_t n_styles);
int  swfr_render_resident(swfr_renderer *r, uint32_t frames);              /* blocking; frames >= 1 */
/* Measurement entry: the resident scene as `frames_per_launch` (1..64) frames per kernel launch, every frame with its own
   kernel-written buffers and framebuffer, `launches` launches back to back (after one warm-up launch); *total_ms = HIP-event time of
   the `launches` launches.  The last frame is readable with swfr_read_image.  The saturated-GPU figure of bench.py.  The handle keeps
   the frames_per_launch framebuffers and work buffers (grow-only, like the frame sets': 64 x 33 MB at 4K) until swfr_destroy. */
int  swfr_render_resident_batched(swfr_renderer *r, uint32_t frames_per_launch, uint32_t launches, float *total_ms);
int  swfr_render_edges(swfr_renderer *r, const swfr_edge *edges, size_t n_edges,
                       const swfr_path *paths, size_t n_paths,
                       const swfr_style *styles, size_t n_styles);

/* Host half of swfr_render without the device: builds the frame's edge list exactly as
   swfr_render would upload it.  Arrays are owned by the handle until the next call. */
int  swfr_build_frame(swfr_renderer *r, const swfr_stage *stage,
                      const swfr_edge **edges, size_t *n_edges,
                      const swfr_path **paths, size_t *n_paths,
                      const swfr_style **styles, size_t *n_styles);

/* Decoded paths of a registered (morph) shape as JSON text in the format of the reference's
   decode goldens (tests/<set>/<name>/shape.ts.json).  String owned by the handle. */
int  swfr_shape_json(swfr_renderer *r, uint32_t id, int morph, const char **json);

/* A batch of different frames in one call (the reference renders its 256 morph ratios by calling render 256 times,
   ts/src/test/node-canvas-renderer.spec.ts:86-113; rs/src/lib.rs:116-130).  With a device destination the frames are rendered
   in groups of up to SWFR_BATCH_FRAMES (default 64) by ONE launch per kernel and group, two groups alternating so that the host
   builds one while the GPU renders the other; frame i lands, premultiplied RGBA8 with tight rows, at device_dst + i * frame_stride
   (DEVICE memory, e.g. a torch tensor).  With device_dst == NULL the frames are rendered one after the other and only the last is
   kept for swfr_read_image.  Blocking: returns when every frame is finished. */
int  swfr_render_batch(swfr_renderer *r, const swfr_stage *stages, uint32_t n_stages, void *device_dst, size_t frame_stride);

/* Timing of the last swfr_render / swfr_render_resident, from HIP events on the handle's stream. */
typedef struct {
    float total_ms;                      /* first kernel start -> last kernel end, all frames */
    float setup_ms, rows_ms, tiles_ms;   /* per-kernel sums over the timed_frames frames that carried events */
    uint32_t frames;
    uint64_t n_edges, n_paths, n_row_tasks, n_records;
    uint32_t timed_frames;               /* frames / SWFR_EVENT_STRIDE (default 16) */
} swfr_timing;
int  swfr_last_timing(swfr_renderer *r, swfr_timing *out);

/* Where the time of the last swfr_render went (the reference's calling pattern: one blocking call per frame). */
typedef struct {
    double build_ms;                     /* host: Stage -> edge list (scene walk, flattening, stroking, clipping) */
    double upload_host_ms;               /* host: staging copy + table layout (prefix sums over the paths) */
    double h2d_ms;                       /* the scene's one host-to-device copy (HIP events) */
    double device_ms;                    /* first kernel start -> last kernel end (HIP events) */
    double total_ms;                     /* wall clock of the whole call, synchronisation included */
    uint64_t h2d_bytes;
} swfr_path_timing;
int  swfr_last_path_timing(swfr_renderer *r, swfr_path_timing *out);

/* What the row kernels met since the handle was created (summed over every frame whose counters came back): rows the fast kernel
   left to the general ones, rows that needed the replay of Cairo's edge-list order for coincident edges, and how often a capacity
   limit of that replay was reached -- each such frame was refused with SWFR_ERR_CAPACITY, never rendered approximately. */
typedef struct swfr_stats {
    uint64_t frames;                     /* frames whose counters were read back */
    uint64_t queued_rows;                /* (path, pixel row) pairs left to k2_rows_slow (coincident edges or > 8 active edges) */
    uint64_t crowded_rows;               /* ... of which handed on to k2_rows_huge (> 64 active edges) */
    uint64_t tie_rows;                   /* rows whose edge order came from the list-order replay */
    uint64_t pairtest_limit;             /* frames refused: crossing test over more than 2^21 edge pairs */
    uint64_t start_group_limit;          /* frames refused: more than 8192 edges of a path start at one sample row / are active in a row */
    uint64_t history_limit;              /* frames refused: order of two older coincident edges needs history deeper than two levels */
    uint64_t reserved;
} swfr_stats;
int  swfr_get_stats(swfr_renderer *r, swfr_stats *out);

/* The reference's animation loop in one call: for (rep < repeat) for (i < n_stages) swfr_render(stages[i]) -- every frame built,
   uploaded, binned, rasterized and waited for exactly as by swfr_render -- timed on the host side of the C-ABI.  `seconds`
   receives the wall clock of the loop; `sum` (optional) the per-stage times added up. */
int  swfr_render_sequence(swfr_renderer *r, const swfr_stage *stages, uint32_t n_stages, uint32_t repeat, double *seconds,
                          swfr_path_timing *sum);

/* Multi-GPU: copy this handle's band slab (its tile-rows, packed in order) into a caller-owned
   DEVICE buffer (e.g. a torch tensor's data_ptr) so that the caller can gather it with RCCL. */
size_t swfr_band_slab_bytes(const swfr_renderer *r);
int  swfr_copy_band_slab(swfr_renderer *r, void *device_dst);

/* Multi-GPU / pipelined use.  swfr_set_targets: the frames of the resident scene are rendered straight into caller-owned DEVICE
   buffers (width*height*4 bytes each, premultiplied RGBA8, tight rows; e.g. torch tensors) -- frame set k of the handle (there are
   SWFR_FRAMES_IN_FLIGHT of them, rotating) writes targets[k]; takes effect with the next swfr_upload_edges.  A handle that owns only
   some tile-rows writes only those rows.  swfr_render_resident_async queues one frame on the next frame set without waiting and
   reports which set (and so which target and which stream) it used; swfr_stream_handle gives that set's hipStream_t so that the
   caller's own streams can be ordered behind / before it with events; swfr_wait blocks until everything queued has finished and
   reports the kernels' error state like swfr_render_resident does. */
int  swfr_set_targets(swfr_renderer *r, void *const *device_targets, uint32_t n_targets);
int  swfr_render_resident_async(swfr_renderer *r, uint32_t *out_set);
/* The same, with THIS frame's pixels going to block_target: a device buffer that holds only the handle's own block of tile-rows
   (a handle with SWFR_FLAG_BANDS_CONTIGUOUS: rows_of_the_block * width * 4 bytes, the block's first pixel row first).  Any number of
   such buffers may be in flight -- they are not tied to the handle's frame sets -- which is what an exchange of SEVERAL frames at once
   needs (frame f assembled on rank f mod N, the N frames' blocks sent in one grouped all-to-all: DESIGN.md, multi-GPU). */
int  swfr_render_resident_async_to(swfr_renderer *r, void *block_target, uint32_t *out_set);
/* n_frames such frames (<= 64), one per block target, queued in one call; *sets_used: bit k set when frame set k's stream carries
   one of them (the caller orders its exchange behind exactly those streams). */
int  swfr_render_resident_group_to(swfr_renderer *r, void *const *block_targets, uint32_t n_frames, uint32_t *sets_used);
void *swfr_stream_handle(swfr_renderer *r, uint32_t set);
int  swfr_wait(swfr_renderer *r);

/* Diagnostics (tools/soak_case.py): copies an intermediate buffer of the last rendered frame to the host.
   what = 0: row headers (8 bytes per (band list entry, row of its tile-row): first cell u32, cells u16, mode u16);
   1: cells (4 bytes each: column relative to the path's x_min << 19 | covered height (5 bits signed) << 14 | uncovered area (14 bits signed)).
   Returns the number of bytes copied (at most `bytes`), or a negative SWFR_ERR_*. */
long swfr_debug_copy(swfr_renderer *r, int what, void *dst, size_t bytes);
/* Device pointer of the premultiplied RGBA8 framebuffer (width*height*4 bytes, tight rows). */
void *swfr_device_framebuffer(swfr_renderer *r);

#ifdef __cplusplus
}
#endif
#endif /* SWFR_H */
