// renderer.cpp -- the C-ABI of include/swfr.h: handle, asset store, host frame build, device pipeline.
//
// There is NO CPU rasterization fallback in this library: without a HIP device swfr_create fails
// (unless the caller explicitly asks for a host-only handle, which can decode and build edge lists
// but refuses to render).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/swfr.h"
#include "device_types.hpp"
#include "frame_builder.hpp"
#include "shape_decoder.hpp"
#include "bitmap_decode.hpp"

namespace swfr {
void launch2_bin(hipStream_t, const Frame2*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
void launch2_rows(hipStream_t, const Frame2*, uint32_t, uint32_t, uint32_t);
void launch2_rows_slow(hipStream_t, const Frame2*, uint32_t, uint32_t, uint32_t, uint32_t);
void launch2_tiles(hipStream_t, const Frame2*, uint32_t, uint32_t, uint32_t, int, uint32_t*);
void launch_unpremultiply(hipStream_t, const uint32_t*, uint32_t*, size_t);
void launch_pack_band(hipStream_t, const uint32_t*, uint32_t*, int, int, uint32_t, uint32_t, uint32_t);
}  // namespace swfr

using namespace swfr;

namespace {

constexpr size_t COUNTER_WORDS = C2_WORDS;   // per frame set

struct HipError {
    hipError_t code;
    const char* what;
};
#define HIP_CHECK(expr)                                 \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) throw HipError{_e, #expr}; \
    } while (0)

// grow-only device buffer
template <class T>
struct DevBuf {
    T* ptr = nullptr;
    size_t cap = 0;
    bool view = false;                       // points into the handle's scene arena: not owned
    void set_view(void* p) {
        if (ptr && !view) (void)hipFree(ptr);
        ptr = static_cast<T*>(p);
        cap = 0;
        view = true;
    }
    void reserve(size_t n) {
        if (view) { ptr = nullptr; view = false; cap = 0; }
        if (n <= cap) return;
        if (ptr) HIP_CHECK(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(n, 64);
        want += want / 2;
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ptr), want * sizeof(T)));
        cap = want;
    }
    void release() {
        if (ptr && !view) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        view = false;
    }
};

// The read-only arrays of an uploaded scene live in one device allocation and are filled by one H2D copy from one pinned
// staging buffer (nine small pageable copies cost ~70 us of host time per frame; one pinned copy a fraction of that).
struct SceneArena {
    uint8_t* dev = nullptr;
    uint8_t* host = nullptr;
    size_t cap = 0, used = 0;
    hipEvent_t copied = nullptr;             // recorded behind the H2D: the staging buffer may be rewritten after it
    hipEvent_t copy_begin = nullptr, copy_end = nullptr;   // timing of the copy (swfr_last_path_timing)
    void begin(size_t bytes) {
        if (copied) HIP_CHECK(hipEventSynchronize(copied));
        if (bytes > cap) {
            if (dev) (void)hipFree(dev);
            if (host) (void)hipHostFree(host);
            dev = host = nullptr;
            const size_t want = bytes + bytes / 2 + 4096;
            HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dev), want));
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host), want, hipHostMallocDefault));
            cap = want;
        }
        used = 0;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    // copies `bytes` into the staging buffer and returns the device address they will land at
    void* push(const void* src, size_t bytes) {
        void* d = dev + used;
        if (bytes && src) std::memcpy(host + used, src, bytes);     // (src == nullptr: the caller has written the bytes in place)
        used += padded(bytes);
        return d;
    }
    void flush(hipStream_t st, bool timed = false) {
        if (timed) {
            if (!copy_begin) { HIP_CHECK(hipEventCreate(&copy_begin)); HIP_CHECK(hipEventCreate(&copy_end)); }
            HIP_CHECK(hipEventRecord(copy_begin, st));
        }
        if (used) HIP_CHECK(hipMemcpyAsync(dev, host, used, hipMemcpyHostToDevice, st));
        if (timed) HIP_CHECK(hipEventRecord(copy_end, st));
        if (!copied) HIP_CHECK(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(copied, st));
    }
    void release() {
        if (copied) (void)hipEventDestroy(copied);
        if (copy_begin) { (void)hipEventDestroy(copy_begin); (void)hipEventDestroy(copy_end); }
        if (dev) (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
        dev = host = nullptr; copied = nullptr; copy_begin = copy_end = nullptr; cap = used = 0;
    }
};

struct DeviceBitmap {
    uint32_t* pixels = nullptr;
    uint32_t width = 0, height = 0;
};

}  // namespace

struct swfr_renderer {
    uint32_t width = 0, height = 0;
    swfr_config cfg{};
    bool has_device = false;
    std::string error;
    std::unique_ptr<FrameBuilder> builder;
    std::string json_scratch;

    // device state
    hipStream_t stream = nullptr;           // == fs[0].stream
    uint32_t* h_counters = nullptr;         // pinned: the kernels' counters of up to four frame sets
    std::vector<hipEvent_t> ev;             // 4 per frame of the last swfr_render_resident call
    // The read-only arrays of one uploaded scene (views into its arena) and what the launches need to know about it.
    // Resident rendering uses scene 0; swfr_render_batch keeps one scene per frame in flight.
    struct Scene {
        SceneArena arena;
        swfr_edge* raw = nullptr; DevPath* paths = nullptr; swfr_style* styles = nullptr;
        DevFilter* filters = nullptr; int32_t* filter_params = nullptr; DevGradient* gradients = nullptr;
        size_t n_edges = 0, n_paths = 0, n_styles = 0, n_rows = 0, n_chunks = 0, n_bands = 0, chunk_rows = 64;
        bool any_shader = false;
        int shader_level = 0;
        size_t n_incidences = 0, n_strips = 0, n_strip_slots = 0;   // (edge, pixel row) pairs: bounds the cells of a frame; k2_tiles wavefronts / launch list slots
        size_t n_slots = 0, cell_total = 0;
        uint32_t max_path_edges = 0;              // picks the row kernel's instance (edges staged in LDS per chunk)
        Frame2 proto{};                           // the scene fields and sizes of a frame descriptor (per-frame buffers not filled in)
        Frame2* frames_dev = nullptr;            // one descriptor per frame set (contiguous, in the arena)
        uint32_t slow_passes = SLOW_PASSES;      // passes of the slow-row kernels the scene needs (known after a frame of a resident scene)
        bool slow_verified = false;              // slow_state / slow_passes come from this scene's own counters, not from the previous scene
        int slow_state = 0;                      // 0: unknown (both slow-row kernels are launched), 1: the scene has no queued rows, 2: none with > 64 edges
    };
    Scene scn[4];
    // One set of kernel-written per-frame buffers and the stream they are used on.  With SWFR_FRAMES_IN_FLIGHT = n consecutive
    // frames rotate over n sets, so the front of frame f+1 overlaps the tail of frame f (every frame recomputes everything).
    struct FrameSet {
        hipStream_t stream = nullptr;
        DevBuf<DevEdge> d_edges;
        DevBuf<uint8_t> d_cls;
        uint32_t* counters = nullptr;            // this set's COUNTER_WORDS of swfr_renderer::d_counters (one buffer: one copy brings all sets' back)
        DevBuf<uint32_t> d_fb;
        DevBuf<BandEntry2> d_band2;
        DevBuf<RowInfo2> d_rows2;
        DevBuf<Cell> d_cells;
        DevBuf<SlowRow> d_slow, d_huge;
        DevBuf<uint32_t> d_path_flag, d_path_queue;
        DevBuf<ChunkInfo> d_chunks;
        DevBuf<BandSlot> d_band_slots;
        DevBuf<StripDesc> d_strips;
        DevBuf<uint32_t> d_strip_cost;
        // the kernels of one frame of the resident scene on this set, captured once per uploaded scene: a frame is then ONE
        // hipGraphLaunch for the host instead of three to twelve kernel launches
        hipGraph_t graph = nullptr;
        hipGraphExec_t graph_exec = nullptr;
        uint64_t graph_key = 0;                  // scene generation | slow_state | slow_passes the graph was captured for (0: none)
    };
    FrameSet fs[4];
    uint64_t scene_gen = 0;                 // counts uploads into scene slot 0
    int resident_batch = 2;                 // SWFR_RESIDENT_BATCH: frames per kernel launch of swfr_render_resident (blockIdx.y = frame; 0/1: one launch chain per frame).
                                            // Two since round 4 (calls without per-kernel events only): half the launches fill the pipeline sooner -- 20 frames 640 us instead of 690,
                                            // 300 frames alike (tools/short_run_probe.py); four per launch lose the overlap of consecutive groups (33.9 us per frame)
#ifdef SWFR_EMU
    int use_graphs = 0;                     // (the emulator's runtime has no graphs)
#else
    int use_graphs = 0;                     // SWFR_GRAPHS=1: frames of a resident scene as graph launches.  Off by default: measured on MI355X /
                                            // ROCm 7.2 a graph launch per frame is SLOWER than its three kernel launches (S1: 162 000 instead of
                                            // 218 000 Mpx/s at 300 frames, 143 000 instead of 178 000 at 20: profiles/r03p_graphs.txt)
#endif
    DevBuf<DevBitmap> d_bitmap_table;
    DevBuf<uint32_t> d_tmp;
    DevBuf<uint32_t> d_counters;            // 4 x COUNTER_WORDS: the frame sets' counters, contiguous
    int in_flight = 4;
    bool scene_from_builder = false;        // the scene in slot 0 is the frame builder's last frame (its arrays are still there)
    uint32_t sets_ready = 0;                // frame sets whose descriptors belong to the scene in slot 0 (swfr_render prepares one, swfr_upload_edges all)
    int hint_slow_state = 0; uint32_t hint_slow_passes = SLOW_PASSES;   // what the last rendered scene needed of the queued-row kernels
    uint32_t* fb_cur = nullptr;             // framebuffer of the last completed frame
    std::map<uint32_t, DeviceBitmap> bitmaps;
    std::vector<DevBitmap> bitmap_table;   // indexed by bitmap id
    bool bitmap_table_dirty = false, bitmap_table_dirty_copied = false;
    bool scene_ready = false, fb_valid = false;
    swfr_timing timing{};
    swfr_path_timing path_timing{};
    int force_chunk_rows = 0;               // SWFR_CHUNK_ROWS: test knob
    int strip_order = 1;                    // SWFR_STRIP_ORDER=0: launch the k2_tiles wavefronts in row-major order (per XCD class)
    int event_stride = 16;                  // SWFR_EVENT_STRIDE: per-kernel HIP events on every n-th resident frame
    bool event_stride_given = false;        // (set explicitly: short runs do not lower it)
    int fast_limit = 16;                    // rows with more active edges go to k2_rows_slow; the row kernel's instance caps it at its 8 or 16 slots (SWFR_FAST_LIMIT: test knob)
    int tiles_grid = 0;                     // SWFR_TILES_GRID: persistent k2_tiles wavefronts per frame (0 = default)
    // swfr_render_batch: groups of frames rendered by ONE launch per kernel (blockIdx.y = frame); two groups alternate,
    // the host builds one while the GPU works on the other
    struct BatchGroup {
        SceneArena arena;                   // the frames' edge lists, tables and descriptors: one H2D copy per group
        DevBuf<uint8_t> work, cls;          // the kernel-written buffers of every frame of the group, carved from two allocations
        hipStream_t stream = nullptr;
        uint32_t* h_counters = nullptr;     // pinned
        size_t h_counters_cap = 0;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // around the group's kernels
    };
    BatchGroup groups[2];
    int batch_frames = 64;                  // SWFR_BATCH_FRAMES: frames per launch in swfr_render_batch
    uint32_t* targets[4] = {nullptr, nullptr, nullptr, nullptr};   // swfr_set_targets: frame set k renders into targets[k]
    swfr_stats stats = {};
    // swfr_render_resident_batched: B copies of the kernel-written buffers + B framebuffers, the B descriptors, events
    DevBuf<uint8_t> rb_work, rb_cls;
    DevBuf<Frame2> rb_frames;
    DevBuf<uint32_t> rb_fb;
    uint32_t rb_count = 0;
    hipEvent_t rb_ev[2] = {nullptr, nullptr};
    // swfr_read_image: pinned staging (a pageable destination costs 7x the copy time), swfr_read_image_async: the copy in flight
    uint8_t* h_image = nullptr; size_t h_image_cap = 0;
    hipEvent_t read_done = nullptr;
    bool read_pending = false;
    uint32_t n_targets = 0, async_next = 0, async_used = 0;   // async_used: bit k = frame set k has run since the last wait

    ~swfr_renderer() {
        if (has_device) {
            (void)hipSetDevice(cfg.device);
            d_bitmap_table.release(); d_tmp.release(); d_counters.release();
            rb_work.release(); rb_cls.release(); rb_frames.release(); rb_fb.release();
            for (auto& e : rb_ev) if (e) (void)hipEventDestroy(e);
            if (h_image) (void)hipHostFree(h_image);
            if (read_done) (void)hipEventDestroy(read_done);
            for (int k = 0; k < 4; ++k) {
                FrameSet& x = fs[k];
                x.d_edges.release(); x.d_cls.release(); x.d_fb.release();
                x.d_band2.release(); x.d_rows2.release(); x.d_cells.release(); x.d_slow.release(); x.d_huge.release(); x.d_path_flag.release(); x.d_path_queue.release(); x.d_chunks.release(); x.d_band_slots.release(); x.d_strips.release();
                x.d_strip_cost.release();
                if (x.graph_exec) (void)hipGraphExecDestroy(x.graph_exec);
                if (x.graph) (void)hipGraphDestroy(x.graph);
                if (k > 0 && x.stream) (void)hipStreamDestroy(x.stream);
                scn[k].arena.release();
            }
            if (h_counters) (void)hipHostFree(h_counters);
            for (auto& g : groups) {
                g.arena.release(); g.work.release(); g.cls.release();
                if (g.stream) (void)hipStreamDestroy(g.stream);
                if (g.h_counters) (void)hipHostFree(g.h_counters);
                if (g.ev_begin) { (void)hipEventDestroy(g.ev_begin); (void)hipEventDestroy(g.ev_end); }
            }
            for (auto& kv : bitmaps) if (kv.second.pixels) (void)hipFree(kv.second.pixels);
            for (auto& e : ev) if (e) (void)hipEventDestroy(e);
            if (stream) (void)hipStreamDestroy(stream);
        }
    }
};

namespace {

int fail(swfr_renderer* r, int code, const std::string& msg) {
    if (r) r->error = msg;
    return code;
}

template <class F>
int guarded(swfr_renderer* r, F&& f) {
    try {
        if (r && r->has_device) HIP_CHECK(hipSetDevice(r->cfg.device));
        return f();
    } catch (const StatusError& e) {
        return fail(r, e.code, e.message);
    } catch (const HipError& e) {
        return fail(r, SWFR_ERR_DEVICE, std::string(e.what) + ": " + hipGetErrorString(e.code));
    } catch (const std::bad_alloc&) {
        return fail(r, SWFR_ERR_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(r, SWFR_ERR_INVALID, e.what());
    }
}

// the handle's share of the frame's tile-rows: local tile-row l is frame tile-row first + l * stride, l < count; `padded` = the
// share of the best-served rank (slabs are gathered at that common size)
struct BandShare { uint32_t first, stride, count, padded; };
BandShare band_share(const swfr_renderer* r) {
    const uint32_t tile_rows = (r->height + TILE_H - 1) / TILE_H;
    const uint32_t bc = r->cfg.band_count > 1 ? r->cfg.band_count : 1, bi = r->cfg.band_count > 1 ? r->cfg.band_index : 0;
    if (bc > 1 && (r->cfg.flags & SWFR_FLAG_BANDS_CONTIGUOUS)) {
        const uint32_t n = (tile_rows + bc - 1) / bc, first = bi * n;
        return BandShare{first, 1u, first >= tile_rows ? 0u : std::min(n, tile_rows - first), n};
    }
    return BandShare{bi, bc, tile_rows <= bi ? 0u : (tile_rows - bi + bc - 1) / bc, (tile_rows + bc - 1) / bc};
}
uint32_t local_tile_rows(const swfr_renderer* r) { return band_share(r).count; }

// Validate a caller-supplied scene so that no kernel can index out of bounds.
// The class bytes of a frame and, behind them (16-byte aligned), one StripTop record per strip of the handle: both are cleared when a
// scene is uploaded (the tile pass clears a strip's record again when it has read it).
inline size_t cls_bytes_of(size_t n_slots, size_t tiles_x) { return (size_t(STRIPS_PER_TILE) * n_slots * tiles_x + 64 + 15) & ~size_t(15); }
inline size_t cls_region_bytes(size_t n_slots, size_t tiles_x, size_t n_strips) { return cls_bytes_of(n_slots, tiles_x) + (n_strips + 1) * sizeof(StripTop); }

void validate_scene(const swfr_renderer* r, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
                    const swfr_style* styles, size_t n_styles) {
    // end points within +-32768 px (2^23 in 24.8): inside that range the int64 products of the closed-form edge evaluation cannot overflow
    for (size_t i = 0; i < n_edges; ++i) {
        const swfr_edge& e = edges[i];
        const int32_t lim = 1 << 23;
        if (e.x1 < -lim || e.x1 > lim || e.x2 < -lim || e.x2 > lim || e.y1 < -lim || e.y1 > lim || e.y2 < -lim || e.y2 > lim)
            throw StatusError{SWFR_ERR_INVALID, "edge end point outside +-32768 px"};
    }
    // every edge belongs to exactly one path: the paths' edge ranges are disjoint and cover the edge list (the kernels and the host
    // layout index the path table with the edge's owner)
    {
        static thread_local std::vector<uint8_t> owned;
        owned.assign(n_edges, 0);
        for (size_t i = 0; i < n_paths; ++i) {
            const swfr_path& p = paths[i];
            if (size_t(p.first_edge) + p.n_edges > n_edges) throw StatusError{SWFR_ERR_INVALID, "path edge range out of bounds"};
            for (size_t k = 0; k < p.n_edges; ++k) {
                if (owned[p.first_edge + k]) throw StatusError{SWFR_ERR_INVALID, "paths share an edge (their edge ranges overlap)"};
                owned[p.first_edge + k] = 1;
            }
        }
        for (size_t i = 0; i < n_edges; ++i)
            if (!owned[i]) throw StatusError{SWFR_ERR_INVALID, "an edge belongs to no path"};
    }
    for (size_t i = 0; i < n_paths; ++i) {
        const swfr_path& p = paths[i];
        if (size_t(p.first_edge) + p.n_edges > n_edges) throw StatusError{SWFR_ERR_INVALID, "path edge range out of bounds"};
        if (p.style >= n_styles) throw StatusError{SWFR_ERR_INVALID, "path style index out of bounds"};
        if (p.kind > SWFR_PATH_BOXES) throw StatusError{SWFR_ERR_INVALID, "unknown path kind"};
        if (p.kind == SWFR_PATH_TOR)
            // an edge of the scan converter: a line (x1, y1)-(x2, y2) running downwards, active over [top, bottom) INSIDE its own extent
            // (what the frame builder and Cairo's clipper produce); the per-row stepping of k2_rows is exact only there
            for (size_t k = 0; k < p.n_edges; ++k) {
                const swfr_edge& e = edges[p.first_edge + k];
                if (e.top >= e.bottom) continue;                              // never active
                if (!(e.y1 < e.y2 && e.y1 <= e.top && e.bottom <= e.y2))
                    throw StatusError{SWFR_ERR_INVALID, "edge active outside its line: need y1 < y2 and y1 <= top < bottom <= y2"};
            }
        if (p.x_min < 0 || p.y_min < 0 || p.x_max > int(r->width) || p.y_max > int(r->height) || p.x_min > p.x_max || p.y_min > p.y_max)
            throw StatusError{SWFR_ERR_INVALID, "path pixel rectangle outside the frame"};
    }
    for (size_t i = 0; i < n_styles; ++i) {
        const swfr_style& s = styles[i];
        if (s.kind > SWFR_STYLE_BITMAP) throw StatusError{SWFR_ERR_INVALID, "unknown style kind"};
        if (s.n_stops > SWFR_MAX_STOPS) throw StatusError{SWFR_ERR_INVALID, "too many gradient stops"};
        if (s.kind == SWFR_STYLE_BITMAP && !r->bitmaps.count(s.bitmap)) throw StatusError{SWFR_ERR_NOT_FOUND, "BitmapNotFound"};
    }
}

// CAIRO_FILTER_GOOD for a surface pattern (cairo-pattern.c _cairo_pattern_analyze_filter, cairo-image-source.c
// create_separable_convolution): bilinear unless an axis is minified below 0.75; then a box (x) tent kernel per axis,
// sampled at 2^bits phases, weights in 16.16, normalised with the rounding error put on the centre tap.
double good_box_kernel(double x, double r) { return std::max(0.0, std::min(std::min(r, 1.0), std::min((r + 1) / 2 - x, (r + 1) / 2 + x))); }
void good_axis(double r, int& width, int& bits, std::vector<int32_t>& out) {
    width = r < 1.0 ? 2 : int(std::ceil(r + 1));
    bits = 0;
    if (width > 1) while (r * double(1 << bits) <= 128.0) ++bits;
    const int n_phases = 1 << bits;
    const double step = 1.0 / n_phases;
    for (int i = 0; i < n_phases; ++i) {
        const size_t base = out.size();
        if (width <= 1) { out.push_back(65536); continue; }
        const double frac = (i + .5) * step;
        const double x1 = std::ceil(frac - width / 2.0 - 0.5) - frac + 0.5;   // centre of the left-most tap
        double total = 0;
        for (int j = 0; j < width; ++j) { const double v = good_box_kernel(x1 + j, r); total += v; out.push_back(int32_t(v * 65536.0)); }
        total = 1 / total;
        int32_t new_total = 0;
        for (int j = 0; j < width; ++j) new_total += (out[base + size_t(j)] = int32_t(out[base + size_t(j)] * total));
        out[base + size_t(width / 2)] += 65536 - new_total;
    }
}
bool good_use_bilinear(double x, double y, double t) {
    const double h = x * x + y * y;                                   // device -> pattern matrix row
    if (h < 1.0 / (0.75 * 0.75)) return true;
    if (h > 3.99 && h < 4.01 && to_fixed(x * y) == 0 && (to_fixed(t) & 255) == 0) return true;   // exactly 1/2, axis-parallel, integer offset
    return false;
}
int64_t fixed_16_16(double d) { return int64_t(std::nearbyint(d * 65536.0)); }   // _cairo_fixed_16_16_from_double: ties to even

// The pattern matrix as pixman gets it (cairo-matrix.c _cairo_matrix_to_pixman_matrix_offset, cairo-image-source.c
// _pixman_image_set_properties): an integer translation is split off so that what remains is small, the matrix is rounded to
// 16.16 and its translation is corrected until the centre of the operation's rectangle maps where the double matrix puts it.
struct PixmanPosition { int64_t base_x, base_y; int32_t m00, m01, m10, m11; };
PixmanPosition pixman_transform_of(Affine m, const int rect[4]) {
    PixmanPosition f;
    const double xc = rect[0] + (rect[2] - rect[0]) / 2., yc = rect[1] + (rect[3] - rect[1]) / 2.;
    int64_t ox = 0, oy = 0;
    if (m.x0 != 0.0 || m.y0 != 0.0) {
        double tx = m.x0, ty = m.y0, norm = std::max(std::fabs(tx), std::fabs(ty));
        for (int i = -1; i < 2; i += 2)
            for (int j = -1; j < 2; j += 2) {
                double den = (m.xx + i) * (m.yy + j) - m.xy * m.yx;
                if (std::fabs(den) < DBL_EPSILON) continue;
                double x = m.y0 * m.xy - m.x0 * (m.yy + j), y = m.x0 * m.yx - m.y0 * (m.xx + i);
                den = 1 / den;
                x *= den;
                y *= den;
                const double new_norm = std::max(std::fabs(x), std::fabs(y));
                if (norm > new_norm) { norm = new_norm; tx = x; ty = y; }
            }
        tx = std::floor(tx);
        ty = std::floor(ty);
        ox = int64_t(-tx);
        oy = int64_t(-ty);
        Affine t;
        t.x0 = tx; t.y0 = ty;
        m = t.then(m);                                              // cairo_matrix_translate
    }
    int64_t p[2][3] = {{fixed_16_16(m.xx), fixed_16_16(m.xy), fixed_16_16(m.x0)}, {fixed_16_16(m.yx), fixed_16_16(m.yy), fixed_16_16(m.y0)}};
    const double eps = 1.0 / 256.0, det = m.det();
    const bool unity = std::fabs(det * det - 1.0) < eps &&
                       ((std::fabs(m.xy) < eps && std::fabs(m.yx) < eps) || (std::fabs(m.xx) < eps && std::fabs(m.yy) < eps));
    Affine inv = m;
    if (!unity && inv.invert_cairo()) {
        for (int it = 0; it < 5; ++it) {
            const int64_t vx = fixed_16_16(xc), vy = fixed_16_16(yc);
            const int64_t tx = (p[0][0] * vx + p[0][1] * vy + p[0][2] * 65536 + 0x8000) >> 16;
            const int64_t ty = (p[1][0] * vx + p[1][1] * vy + p[1][2] * 65536 + 0x8000) >> 16;
            // pixman_transform_point_3d fails when the centre does not map into 16.16 (|coordinate| >= 32768): Cairo then leaves
            // the translation as rounded ("If we can't transform the reference point, skip the adjustment")
            if (tx != (int32_t)tx || ty != (int32_t)ty) break;
            double x = double(tx) / 65536.0, y = double(ty) / 65536.0;
            inv.apply(x, y);
            x -= xc;
            y -= yc;
            m.apply_distance(x, y);
            const int64_t dx = fixed_16_16(x), dy = fixed_16_16(y);
            p[0][2] -= dx;
            p[1][2] -= dy;
            if (dx == 0 && dy == 0) break;
        }
    }
    // pixman_transform_point_3d of pixel (px, py)'s centre: (p·(X, Y, 1) + 0x8000) >> 16 with X = (px + ox + .5) in 16.16; the
    // per-pixel and per-row steps are whole multiples of 65536 inside the sum, so they come out of the shift exactly
    const int64_t X0 = ox * 65536 + 0x8000, Y0 = oy * 65536 + 0x8000;
    f.base_x = (p[0][0] * X0 + p[0][1] * Y0 + p[0][2] * 65536 + 0x8000) >> 16;
    f.base_y = (p[1][0] * X0 + p[1][1] * Y0 + p[1][2] * 65536 + 0x8000) >> 16;
    f.m00 = int32_t(p[0][0]); f.m01 = int32_t(p[0][1]); f.m10 = int32_t(p[1][0]); f.m11 = int32_t(p[1][1]);
    return f;
}
Affine pattern_matrix(const swfr_style& st) {
    Affine m;
    m.xx = st.inv[0]; m.yx = st.inv[1]; m.xy = st.inv[2]; m.yy = st.inv[3]; m.x0 = st.inv[4]; m.y0 = st.inv[5];
    return m;
}

// A radial gradient as cairo 1.16 hands it to pixman 0.40 and as pixman evaluates it: circles scaled into +-16383 (the matrix
// takes the inverse factor), 16.16 circles and stops, 16-bit colours, single-precision ramps per interval (gradient_walker_reset),
// PAD sentinels.  The device evaluates B and C of the quadratic as exact 64-bit integers, the root in doubles, the ramp in floats.
DevGradient radial_of(const swfr_style& st, const int rect[4]) {
    DevGradient g;
    std::memset(&g, 0, sizeof g);
    double c0x = st.c0x, c0y = st.c0y, c0r = st.r0, c1x = st.c1x, c1y = st.c1y, c1r = st.r1;
    double dim = std::fabs(c0x);
    for (double v : {c0y, c0r, c1x, c1y, c1r, c0x - c1x, c0y - c1y, c0r - c1r}) dim = std::max(dim, std::fabs(v));
    Affine m = pattern_matrix(st);
    if (dim > 16383.0) {                                            // PIXMAN_MAX_INT >> 1
        dim = 16383.0 / dim;
        c0x *= dim; c0y *= dim; c0r *= dim; c1x *= dim; c1y *= dim; c1r *= dim;
        m = m.then(Affine::scale(dim, dim));
    }
    const PixmanPosition pos = pixman_transform_of(m, rect);
    g.base_x = pos.base_x; g.base_y = pos.base_y; g.m00 = pos.m00; g.m01 = pos.m01; g.m10 = pos.m10; g.m11 = pos.m11;
    const int64_t f1x = fixed_16_16(c0x), f1y = fixed_16_16(c0y), f1r = fixed_16_16(c0r);
    const int64_t dx = fixed_16_16(c1x) - f1x, dy = fixed_16_16(c1y) - f1y, dr = fixed_16_16(c1r) - f1r;
    g.c1x = int32_t(f1x); g.c1y = int32_t(f1y); g.c1r = int32_t(f1r); g.dx = int32_t(dx); g.dy = int32_t(dy); g.dr = int32_t(dr);
    g.a = double(dx * dx + dy * dy - dr * dr);
    g.inva = g.a != 0 ? 1. * 65536 / g.a : 0;
    g.mindr = -1. * 65536 * double(f1r);
    const int n = int(st.n_stops);
    g.n_intervals = n ? n + 1 : 0;
    uint16_t col[SWFR_MAX_STOPS + 2][4] = {};
    for (int i = 0; i < n; ++i) {
        g.x[i + 1] = int32_t(fixed_16_16(double(st.stop_offset[i])));
        // cairo_pattern_add_color_stop_rgba keeps doubles; _cairo_color_double_to_short = (uint16)(v * 65535 + 0.5)
        col[i + 1][0] = uint16_t(double(st.stop_rgba[i][3]) * 65535.0 + 0.5);
        col[i + 1][1] = uint16_t(double(st.stop_rgba[i][0]) * 65535.0 + 0.5);
        col[i + 1][2] = uint16_t(double(st.stop_rgba[i][1]) * 65535.0 + 0.5);
        col[i + 1][3] = uint16_t(double(st.stop_rgba[i][2]) * 65535.0 + 0.5);
    }
    g.x[0] = INT32_MIN; g.x[n + 1] = INT32_MAX;
    if (n) { std::memcpy(col[0], col[1], sizeof col[0]); std::memcpy(col[n + 1], col[n], sizeof col[0]); }
    for (int k = 0; k <= n && n; ++k) {
        const int64_t left_x = g.x[k], right_x = g.x[k + 1];
        const float lx = left_x * (1.0f / 65536.0f), rx = right_x * (1.0f / 65536.0f);
        for (int ch = 0; ch < 4; ++ch) {
            const float l = col[k][ch] * (1.0f / 257.0f), rr = col[k + 1][ch] * (1.0f / 257.0f);
            if ((-FLT_MIN < (rx - lx) && (rx - lx) < FLT_MIN) || left_x == INT32_MIN || right_x == INT32_MAX) {
                g.ramp[k][2 * ch] = 0.0f;
                g.ramp[k][2 * ch + 1] = (l + rr) / 510.0f;
            } else {
                const float w_rec = 1.0f / (rx - lx);
                g.ramp[k][2 * ch + 1] = (l * rx - rr * lx) * w_rec * (1.0f / 255.0f);
                g.ramp[k][2 * ch] = (rr - l) * w_rec * (1.0f / 255.0f);
            }
        }
    }
    return g;
}

DevFilter good_filter(const swfr_style& st, const int rect[4], std::vector<int32_t>& params) {
    DevFilter f{};
    if (st.kind != SWFR_STYLE_BITMAP) return f;
    const PixmanPosition pos = pixman_transform_of(pattern_matrix(st), rect);
    f.base_x = pos.base_x; f.base_y = pos.base_y; f.m00 = pos.m00; f.m01 = pos.m01; f.m10 = pos.m10; f.m11 = pos.m11;
    const double xx = st.inv[0], yx = st.inv[1], xy = st.inv[2], yy = st.inv[3], x0 = st.inv[4], y0 = st.inv[5];
    if (good_use_bilinear(xx, xy, x0) && good_use_bilinear(yx, yy, y0)) return f;
    double dx = std::hypot(xx, xy), dy = std::hypot(yx, yy);
    dx = std::min(dx, 16.0); dy = std::min(dy, 16.0);
    if (dx < 1.0 / 0.75) dx = 1.0;
    if (dy < 1.0 / 0.75) dy = 1.0;
    f.on = 1;
    f.x_off = uint32_t(params.size());
    good_axis(dx, f.cw, f.xbits, params);
    f.y_off = uint32_t(params.size());
    good_axis(dy, f.ch, f.ybits, params);
    return f;
}

// pixman's view of every bitmap / radial-gradient style of a scene (sample positions, filter tables, colour ramps)
void prepare_sources(const swfr_path* paths, size_t n_paths, const swfr_style* styles, size_t n_styles, std::vector<DevFilter>& filters,
                     std::vector<DevGradient>& gradients, std::vector<int32_t>& fparams, const std::vector<DevBitmap>& bitmap_table) {
    filters.assign(n_styles, DevFilter{});
    // a bitmap style belongs to one drawing operation: pixman's transform is anchored at the centre of that operation's rectangle
    std::vector<int> rect(4 * n_styles, 0);
    std::vector<uint8_t> seen(n_styles, 0);
    for (size_t i = 0; i < n_paths; ++i) {
        const swfr_path& p = paths[i];
        if (styles[p.style].kind != SWFR_STYLE_BITMAP && styles[p.style].kind != SWFR_STYLE_RADIAL) continue;
        int* q = &rect[4 * size_t(p.style)];
        if (seen[p.style] && (q[0] != p.x_min || q[1] != p.y_min || q[2] != p.x_max || q[3] != p.y_max))
            throw StatusError{SWFR_ERR_INVALID, "paths that share a bitmap or radial-gradient style must share the pixel rectangle (one drawing operation)"};
        seen[p.style] = 1;
        q[0] = p.x_min; q[1] = p.y_min; q[2] = p.x_max; q[3] = p.y_max;
    }
    for (size_t i = 0; i < n_styles; ++i) {
        filters[i] = good_filter(styles[i], &rect[4 * i], fparams);
        filters[i].kind = styles[i].kind; filters[i].extend = styles[i].extend;
        if (styles[i].kind == SWFR_STYLE_BITMAP && styles[i].bitmap < bitmap_table.size()) {
            const DevBitmap& bm = bitmap_table[styles[i].bitmap];
            filters[i].pixels = bm.pixels; filters[i].width = bm.width; filters[i].height = bm.height;
        }
        if (styles[i].kind == SWFR_STYLE_RADIAL) {
            gradients.push_back(radial_of(styles[i], &rect[4 * i]));
            filters[i].pad = int32_t(gradients.size());            // index + 1 into the gradient table
        }
    }
}

// What the host works out about a scene: the LAYOUT of the tables the device fills -- how many row chunks, band
// entries and cells there can be and where each path's share starts (prefix sums over the paths' rectangles and edge row spans,
// and over the tile-rows) -- plus pixman's view of the bitmap / gradient styles.  No binning: that is the device's work, per frame.
struct SceneLayout {
    uint32_t chunk_rows = ROWS_CHUNK;
    size_t n_chunks = 0, n_slots = 0, n_rows = 0, n_bands = 0, n_strips = 0, n_strip_slots = 0, incidences = 0, cell_main = 0, cell_total = 0;
    bool any_shader = false;
    uint32_t max_path_edges = 0;
    int shader_level = 0;       // 0 solid colours only, 1 + bitmap fills, 2 + gradients: picks the tile kernel's instance
    std::vector<uint32_t> chunk_base, slot_base, inc_base, band_off;
    std::vector<uint32_t> band_span;     // per path: first tile-row | last tile-row << 16 of its rectangle (0xffff | 0 << 16: none); padded to a multiple of 16 paths
    std::vector<DevFilter> filters;
    std::vector<DevGradient> gradients;
    std::vector<int32_t> fparams;
};
// `edges` carry their path index in `reserved`
// src_paths: the paths as the caller drew them, when `paths` holds the column blocks of wide ones (split_wide_paths): a bitmap or
// gradient style is anchored at the rectangle of its drawing operation, not of a block
void layout_scene(const swfr_renderer* r, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
                  const swfr_style* styles, size_t n_styles, SceneLayout& L, const swfr_path* src_paths = nullptr, size_t n_src_paths = 0) {
    L.n_bands = (r->height + TILE_H - 1) / TILE_H;
    const uint32_t tiles_x = (r->width + TILE_W - 1) / TILE_W;
    // rows per k2_rows wavefront: 64 when that already gives the GPU a thousand wavefronts, fewer (whole tile-rows) for scenes made
    // of a few tall paths
    auto count_chunks = [&](uint32_t cr) {
        size_t n = 0;
        for (size_t i = 0; i < n_paths; ++i)
            if (paths[i].kind == SWFR_PATH_TOR && paths[i].y_max > paths[i].y_min)
                n += (size_t(paths[i].y_max) - size_t(paths[i].y_min) / TILE_H * TILE_H + cr - 1) / cr;
        return n;
    };
    // (counted over the whole frame also for a handle that owns a share of the tile-rows: with chunks sized for the share alone
    //  -- 16 rows at an eighth of S1 -- k2_bin writes four times the chunk descriptors and loses what k2_rows gains:
    //  profiles/r03o_blocks_timing.json)
    L.chunk_rows = ROWS_CHUNK;
    // (down to the rows of ONE STRIP for a frame of a few tall paths: its row kernel is as slow as its slowest wavefront, and a wavefront
    //  whose rows all need sample passes runs one pass per four rows)
    if (r->force_chunk_rows == 8 || r->force_chunk_rows == 16 || r->force_chunk_rows == 32 || r->force_chunk_rows == 64) L.chunk_rows = uint32_t(r->force_chunk_rows);
    else while (L.chunk_rows > uint32_t(STRIP_H) && count_chunks(L.chunk_rows) < 1024) L.chunk_rows >>= 1;
    L.any_shader = false; L.shader_level = 0;
    for (size_t i = 0; i < n_styles; ++i) {
        L.any_shader = L.any_shader || styles[i].kind != SWFR_STYLE_SOLID;
        L.shader_level = std::max(L.shader_level, styles[i].kind == SWFR_STYLE_SOLID ? 0 : (styles[i].kind == SWFR_STYLE_BITMAP ? 1 : 2));
    }
    // exclusive prefixes over the paths (first chunk, first band slot) and over the tile-rows (band list offsets, by a difference
    // array over the paths' tile-row ranges)
    L.n_chunks = L.n_slots = L.n_rows = 0; L.max_path_edges = 0;
    L.chunk_base.assign(n_paths + 1, 0); L.slot_base.assign(n_paths + 1, 0); L.inc_base.assign(n_paths + 1, 0); L.band_off.assign(L.n_bands + 2, 0);
    L.band_span.assign((n_paths + 15) / 16 * 16 + 16, 0xffffu);
    for (size_t i = 0; i < n_paths; ++i) {
        const swfr_path& p = paths[i];
        L.chunk_base[i] = uint32_t(L.n_chunks); L.slot_base[i] = uint32_t(L.n_slots);
        if (p.kind == SWFR_PATH_TOR && p.x_max - p.x_min > MAX_PATH_WIDTH)
            throw StatusError{SWFR_ERR_CAPACITY, "a path wider than 8192 px reached the layout unsplit (cell columns are kept in 13 bits relative to the path)"};
        if (p.kind == SWFR_PATH_TOR && p.y_max > p.y_min) L.n_chunks += (size_t(p.y_max) - size_t(p.y_min) / TILE_H * TILE_H + L.chunk_rows - 1) / L.chunk_rows;
        if (p.y_max > p.y_min && p.x_max > p.x_min) {
            const size_t b0 = size_t(p.y_min / TILE_H), b1 = size_t((p.y_max - 1) / TILE_H);
            L.n_slots += b1 - b0 + 1;
            ++L.band_off[b0 + 1]; --L.band_off[b1 + 2];
            L.band_span[i] = uint32_t(b0) | (uint32_t(b1) << 16);
        }
        if (p.kind == SWFR_PATH_TOR) { L.n_rows += size_t(p.y_max - p.y_min); L.max_path_edges = std::max(L.max_path_edges, p.n_edges); }
    }
    L.chunk_base[n_paths] = uint32_t(L.n_chunks); L.slot_base[n_paths] = uint32_t(L.n_slots);
    for (size_t b = 1; b <= L.n_bands + 1; ++b) L.band_off[b] += L.band_off[b - 1];       // difference array -> counts, shifted by one
    for (size_t b = 1; b <= L.n_bands + 1; ++b) L.band_off[b] += L.band_off[b - 1];       // counts -> exclusive prefix: entries of tile-rows < b
    // pixel rows an edge can have a sample row in (a bound: +1 for the rounding of the sample grid): every (edge, row) pair yields
    // at most MAX_CELLS_PER_EDGE_ROW cells
    for (size_t i = 0; i < n_edges; ++i) {
        const swfr_edge& e = edges[i];
        const swfr_path& p = paths[e.reserved];
        if (p.kind != SWFR_PATH_TOR) continue;
        const int64_t top = std::max<int64_t>(e.top, int64_t(p.y_min) * 256), bot = std::min<int64_t>(e.bottom, int64_t(p.y_max) * 256);
        if (bot > top) L.inc_base[size_t(e.reserved) + 1] += uint32_t(((bot + 255) >> 8) - (top >> 8)) + 1;
    }
    for (size_t i = 0; i < n_paths; ++i) L.inc_base[i + 1] += L.inc_base[i];
    L.incidences = L.inc_base[n_paths];
    L.cell_main = L.incidences * MAX_CELLS_PER_EDGE_ROW;
    L.cell_total = L.cell_main * 2 + 4096;
    if (L.cell_total > 0xfffffff0ull) throw StatusError{SWFR_ERR_CAPACITY, "scene too large for 32-bit cell offsets"};
    L.n_strips = size_t(local_tile_rows(r)) * tiles_x * STRIPS_PER_TILE;
    L.n_strip_slots = strip_slots(local_tile_rows(r), uint32_t(tiles_x * STRIPS_PER_TILE));
    L.filters.clear(); L.gradients.clear(); L.fparams.clear();
    if (src_paths) prepare_sources(src_paths, n_src_paths, styles, n_styles, L.filters, L.gradients, L.fparams, r->bitmap_table);
    else prepare_sources(paths, n_paths, styles, n_styles, L.filters, L.gradients, L.fparams, r->bitmap_table);
}
// A tor path wider than the 13-bit column field of a cell (only possible in frames wider than 8192 px) is rasterized as several
// paths over the SAME edges, one per block of 8192 pixel columns: the scan converter clips a path's cells to its column range
// exactly the way a tile does -- a cell left of the range only adds its height to the range's first column, a cell right of it is
// dropped -- so the blocks' pixels are the pixels of the whole path, and the blocks do not overlap.  (A bitmap or gradient style stays
// anchored at the rectangle of the unsplit path: layout_scene gets the original paths for that.)  Returns false when the scene
// has no such path; otherwise the new edge list (tagged with the owning path) and path table.
bool split_wide_paths(const swfr_edge* edges, const swfr_path* paths, size_t n_paths, std::vector<swfr_edge>& out_e, std::vector<swfr_path>& out_p) {
    bool any = false;
    for (size_t i = 0; i < n_paths && !any; ++i) any = paths[i].kind == SWFR_PATH_TOR && paths[i].x_max - paths[i].x_min > MAX_PATH_WIDTH;
    if (!any) return false;
    out_e.clear(); out_p.clear();
    for (size_t i = 0; i < n_paths; ++i) {
        const swfr_path& p = paths[i];
        const bool wide = p.kind == SWFR_PATH_TOR && p.x_max - p.x_min > MAX_PATH_WIDTH;
        const int blocks = wide ? (p.x_max - p.x_min + MAX_PATH_WIDTH - 1) / MAX_PATH_WIDTH : 1;
        for (int k = 0; k < blocks; ++k) {
            swfr_path q = p;
            if (wide) { q.x_min = p.x_min + k * MAX_PATH_WIDTH; q.x_max = std::min(p.x_max, q.x_min + MAX_PATH_WIDTH); }
            q.first_edge = uint32_t(out_e.size());
            for (uint32_t j = 0; j < p.n_edges; ++j) {
                swfr_edge e = edges[p.first_edge + j];
                e.reserved = int32_t(out_p.size());
                out_e.push_back(e);
            }
            out_p.push_back(q);
        }
    }
    return true;
}

// bytes of the scene's read-only part in an arena (raw arrays + layout prefixes + sources), each piece padded
size_t scene_arena_bytes(const SceneLayout& L, size_t n_edges, size_t n_paths, size_t n_styles) {
    auto P = SceneArena::padded;
    // (a scene of solid colours only: the styles' {kind, pixel} heads, no filter records)
    return P(n_edges * sizeof(swfr_edge)) + P(n_paths * sizeof(swfr_path)) + (L.shader_level == 0 ? P(n_styles * 8) : P(n_styles * sizeof(swfr_style)) + P(n_styles * sizeof(DevFilter))) +
           P(L.fparams.size() * sizeof(int32_t)) + P(L.gradients.size() * sizeof(DevGradient)) + 3 * P((n_paths + 1) * sizeof(uint32_t)) +
           P((L.n_bands + 2) * sizeof(uint32_t)) + P(L.band_span.size() * sizeof(uint32_t));
}
// pushes that part and fills the descriptor's scene fields; the edge array's host staging copy is returned (for tagging)
swfr_edge* push_scene(SceneArena& A, const SceneLayout& L, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
                      const swfr_style* styles, size_t n_styles, Frame2& f) {
    swfr_edge* staged = reinterpret_cast<swfr_edge*>(A.host + A.used);
    f.raw = static_cast<swfr_edge*>(A.push(edges, n_edges * sizeof(swfr_edge)));
    f.paths = static_cast<DevPath*>(A.push(paths, n_paths * sizeof(swfr_path)));
    static_assert(offsetof(swfr_style, kind) == 0 && offsetof(swfr_style, pixel) == 4, "the head of a style is {kind, pixel}");
    if (L.shader_level == 0) {
        // solid colours only: the kernels read a style's kind and pixel and nothing else, so only those eight bytes travel
        uint32_t* heads = reinterpret_cast<uint32_t*>(A.host + A.used);
        for (size_t i = 0; i < n_styles; ++i) { heads[2 * i] = styles[i].kind; heads[2 * i + 1] = styles[i].pixel; }
        f.styles = reinterpret_cast<const swfr_style*>(A.push(nullptr, n_styles * 8));
        f.style_stride = 8;
        f.src.filters = nullptr;
    } else {
        f.styles = static_cast<swfr_style*>(A.push(styles, n_styles * sizeof(swfr_style)));
        f.style_stride = uint32_t(sizeof(swfr_style));
        f.src.filters = static_cast<DevFilter*>(A.push(L.filters.data(), n_styles * sizeof(DevFilter)));
    }
    f.src.fparams = static_cast<int32_t*>(A.push(L.fparams.data(), L.fparams.size() * sizeof(int32_t)));
    f.src.gradients = static_cast<DevGradient*>(A.push(L.gradients.data(), L.gradients.size() * sizeof(DevGradient)));
    f.path_chunks = static_cast<uint32_t*>(A.push(L.chunk_base.data(), (n_paths + 1) * sizeof(uint32_t)));
    f.path_slots = static_cast<uint32_t*>(A.push(L.slot_base.data(), (n_paths + 1) * sizeof(uint32_t)));
    f.path_inc = static_cast<uint32_t*>(A.push(L.inc_base.data(), (n_paths + 1) * sizeof(uint32_t)));
    f.band_off = static_cast<uint32_t*>(A.push(L.band_off.data(), (L.n_bands + 2) * sizeof(uint32_t)));
    f.path_bands = static_cast<uint32_t*>(A.push(L.band_span.data(), L.band_span.size() * sizeof(uint32_t)));
    return staged;
}
void fill_frame_sizes(const swfr_renderer* r, const SceneLayout& L, size_t n_edges, size_t n_paths, Frame2& f) {
    f.n_edges = uint32_t(n_edges); f.n_paths = uint32_t(n_paths); f.n_chunks = uint32_t(L.n_chunks);
    f.n_bands = uint32_t(L.n_bands); f.n_strips = uint32_t(L.n_strips); f.n_strip_slots = uint32_t(L.n_strip_slots); f.cell_slice = uint32_t(L.cell_total); f.slow_cap = uint32_t(L.n_rows + 64);
    f.width = int32_t(r->width); f.height = int32_t(r->height); f.tiles_x = int32_t((r->width + TILE_W - 1) / TILE_W);
    f.fast_limit = uint32_t(std::min(std::max(r->fast_limit, 0), 16));       // (the row kernel's instance caps it at its own slots: 8 or 16)
    const BandShare bs = band_share(r);
    f.band_first = bs.first; f.band_stride = bs.stride;
    f.cell_main = uint32_t(L.cell_main);
    f.chunk_rows = L.chunk_rows; f.chunk_cap = uint32_t(L.n_chunks + 1); f.strip_order = r->strip_order ? 1u : 0u;
}

// Uploads a scene -- the raw edge list, the paths and the styles, plus the layout above -- and sizes the buffers the
// kernels write.
int upload2(swfr_renderer* r, int si, bool all_sets, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
            const swfr_style* styles, size_t n_styles, uint32_t* fb_override, bool edges_tagged) {
    swfr_renderer::Scene& sc = r->scn[si];
    if (si == 0) r->scene_ready = false;
    if (r->async_used) {
        // frames queued by swfr_render_resident_async may still read the scene and the buffers this call rewrites
        for (int k = 0; k < 4; ++k) if (r->fs[k].stream) HIP_CHECK(hipStreamSynchronize(r->fs[k].stream));
    }
    // which of the queued-row kernels the frame needs: assumed to be what the previous frame needed (an animation's frames are
    // alike), checked against the frame's own counters afterwards (render_resident renders again with everything if not)
    sc.slow_verified = false;
    sc.slow_state = si == 0 ? r->hint_slow_state : 0; sc.slow_passes = si == 0 ? r->hint_slow_passes : SLOW_PASSES;
    if (si > 0 && !r->fs[si].stream) HIP_CHECK(hipStreamCreateWithFlags(&r->fs[si].stream, hipStreamNonBlocking));
    const hipStream_t up_stream = r->fs[si].stream;
    const uint32_t tiles_x = (r->width + TILE_W - 1) / TILE_W;
    std::vector<swfr_edge> tagged;
    if (!edges_tagged) {
        tagged.assign(edges, edges + n_edges);
        for (size_t i = 0; i < n_paths; ++i)
            for (uint32_t k = 0; k < paths[i].n_edges; ++k) tagged[paths[i].first_edge + k].reserved = int32_t(i);
        edges = tagged.data();
    }
    static thread_local std::vector<swfr_edge> split_e;
    static thread_local std::vector<swfr_path> split_p;
    const swfr_path* src_paths = nullptr; size_t n_src_paths = 0;
    if (split_wide_paths(edges, paths, n_paths, split_e, split_p)) {             // (frames wider than 8192 px only)
        src_paths = paths; n_src_paths = n_paths;
        edges = split_e.data(); n_edges = split_e.size(); paths = split_p.data(); n_paths = split_p.size();
    }
    static thread_local SceneLayout layout_scratch;        // (vectors keep their capacity from frame to frame)
    SceneLayout& L = layout_scratch;
    layout_scene(r, edges, n_edges, paths, n_paths, styles, n_styles, L, src_paths, n_src_paths);
    sc.n_edges = n_edges; sc.n_paths = n_paths; sc.n_styles = n_styles; sc.any_shader = L.any_shader; sc.shader_level = L.shader_level;
    sc.n_chunks = L.n_chunks; sc.chunk_rows = L.chunk_rows; sc.n_bands = L.n_bands; sc.n_rows = L.n_rows;
    sc.n_strips = L.n_strips; sc.n_strip_slots = L.n_strip_slots; sc.n_incidences = L.incidences;
    sc.n_slots = L.n_slots; sc.cell_total = L.cell_total; sc.max_path_edges = L.max_path_edges;
    SceneArena& A = sc.arena;
    A.begin(scene_arena_bytes(L, n_edges, n_paths, n_styles) + SceneArena::padded(4 * sizeof(Frame2)) + 4096);
    Frame2 proto;
    std::memset(&proto, 0, sizeof proto);
    push_scene(A, L, edges, n_edges, paths, n_paths, styles, n_styles, proto);
    sc.raw = const_cast<swfr_edge*>(proto.raw); sc.paths = const_cast<DevPath*>(proto.paths); sc.styles = const_cast<swfr_style*>(proto.styles);
    fill_frame_sizes(r, L, n_edges, n_paths, proto);
    const size_t n_slots = L.n_slots, n_rows = L.n_rows, n_chunks = L.n_chunks;
    // ---- per-frame (kernel-written) buffers: grow-only allocations
    const int n_sets = std::max(1, std::min(r->in_flight, 4));
    auto reserve_zeroed = [&](DevBuf<uint32_t>& b, size_t n) {
        const uint32_t* before = b.ptr;
        b.reserve(n);
        if (b.ptr != before) HIP_CHECK(hipMemsetAsync(b.ptr, 0, b.cap * sizeof(uint32_t), up_stream));
    };
    for (int k = 0; k < 4; ++k) {
        if (all_sets ? k >= n_sets : k != si) continue;
        auto& x = r->fs[k];
        if (!x.stream) HIP_CHECK(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
        x.d_edges.reserve(2 * n_edges + 1); x.d_band2.reserve(n_slots); x.d_rows2.reserve(n_slots * TILE_H + 64); x.d_cells.reserve(L.cell_total);
        x.d_slow.reserve(2 * (n_rows + 64)); x.d_huge.reserve(2 * (n_rows + 64));     // (two queues each: a pass reads one and refills the other)
        x.d_path_flag.reserve(n_paths + 64); x.d_path_queue.reserve(n_paths + 64);
        x.d_chunks.reserve(n_chunks + 1); x.d_band_slots.reserve(n_slots + 8); x.d_strips.reserve(L.n_strip_slots + 1);
        reserve_zeroed(x.d_strip_cost, L.n_strips + 1);                                 // (zero between frames: the ordering workgroup clears what it has read)
        r->d_counters.reserve(4 * COUNTER_WORDS);
        x.counters = r->d_counters.ptr + size_t(k) * COUNTER_WORDS;
        x.d_cls.reserve(cls_region_bytes(n_slots, tiles_x, L.n_strips));
        // class bytes outside the paths' rectangles are never written by a kernel: cleared once per uploaded scene
        HIP_CHECK(hipMemsetAsync(x.d_cls.ptr, 0, cls_region_bytes(n_slots, tiles_x, L.n_strips), up_stream));
        if (!x.d_fb.ptr && !r->n_targets) {                                             // (a handle with caller targets never renders into a buffer of its own)
            x.d_fb.reserve(size_t(r->width) * r->height);
            HIP_CHECK(hipMemsetAsync(x.d_fb.ptr, 0, size_t(r->width) * r->height * 4, up_stream));
        }
    }
    if (r->bitmap_table_dirty) r->d_bitmap_table.reserve(r->bitmap_table.size());     // (filled below; the address is what the descriptor needs)
    proto.src.bitmaps = r->d_bitmap_table.ptr;
    sc.proto = proto;
    Frame2 fr[4];
    std::memset(fr, 0, sizeof fr);
    for (int k = 0; k < 4; ++k) {
        if (all_sets ? k >= n_sets : k != si) continue;
        auto& x = r->fs[k];
        Frame2& f = fr[k];
        f = proto;
        f.chunks = x.d_chunks.ptr; f.band_slots = x.d_band_slots.ptr; f.strips = x.d_strips.ptr;
        f.strip_cost = x.d_strip_cost.ptr;
        f.edges = x.d_edges.ptr; f.band_list = x.d_band2.ptr; f.cls = x.d_cls.ptr; f.strip_top = reinterpret_cast<StripTop*>(x.d_cls.ptr + cls_bytes_of(n_slots, tiles_x)); f.rows = x.d_rows2.ptr; f.cells = x.d_cells.ptr;
        f.slow = x.d_slow.ptr; f.huge = x.d_huge.ptr; f.counters = x.counters;
        f.path_flag = x.d_path_flag.ptr; f.path_queue = x.d_path_queue.ptr;
        f.fb = (fb_override && k == si) ? fb_override : (r->n_targets ? r->targets[uint32_t(k) % r->n_targets] : x.d_fb.ptr);
    }
    sc.frames_dev = static_cast<Frame2*>(A.push(fr, sizeof fr));
    A.flush(up_stream, si == 0);
    if (all_sets) {
        // the copy and the memsets above ran on set `si`'s stream: the other sets' streams start their frames behind them
        for (int k = 0; k < n_sets; ++k)
            if (k != si && r->fs[k].stream) HIP_CHECK(hipStreamWaitEvent(r->fs[k].stream, A.copied, 0));
    }
    if (r->bitmap_table_dirty) {
        if (!r->bitmap_table.empty())
            HIP_CHECK(hipMemcpyAsync(r->d_bitmap_table.ptr, r->bitmap_table.data(), r->bitmap_table.size() * sizeof(DevBitmap),
                                     hipMemcpyHostToDevice, r->stream));
        r->bitmap_table_dirty = false;
        r->bitmap_table_dirty_copied = true;
    }
    if (r->bitmap_table_dirty_copied) { HIP_CHECK(hipStreamSynchronize(r->stream)); r->bitmap_table_dirty_copied = false; }   // bitmap_table may be edited next
    if (si == 0) { r->scene_ready = true; r->sets_ready = all_sets ? uint32_t(n_sets) : 1u; r->scene_from_builder = edges_tagged; ++r->scene_gen; }
    return SWFR_OK;
}

// Uploads a caller-supplied scene (validated first) into scene slot `si`; see upload2.
int upload(swfr_renderer* r, int si, bool all_sets, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
           const swfr_style* styles, size_t n_styles, uint32_t* fb_override = nullptr) {
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    validate_scene(r, edges, n_edges, paths, n_paths, styles, n_styles);
    return upload2(r, si, all_sets, edges, n_edges, paths, n_paths, styles, n_styles, fb_override, false);
}

// the kernels of one frame of scene `sc` on frame set `F`, writing the framebuffer `fb`; e (optional): four events around them
// overlapped: other frames run beside this one (several frame sets in flight) -- the tile pass is then launched in its paired shape (two
// strips per wavefront, better for the frame rate); a frame that has the GPU to itself (a blocking swfr_render, a handle with one frame
// in flight) gets one wavefront per strip, which finishes 2-3 us sooner ("Two strips per wavefront")
void launch_frame(swfr_renderer* r, const swfr_renderer::Scene& sc, swfr_renderer::FrameSet& F, uint32_t* fb, hipEvent_t* e, bool overlapped) {
    const hipStream_t st = F.stream;
    {
        // the frame's descriptor (scene arrays, this set's buffers, the framebuffer) was written with the scene
        const Frame2* fh = sc.frames_dev + (&F - r->fs);        // the set's descriptor, uploaded with the scene
        if (e) HIP_CHECK(hipEventRecord(e[0], st));
        launch2_bin(st, fh, 1, uint32_t(sc.n_edges), uint32_t(sc.n_paths), uint32_t(sc.n_bands), (sc.n_chunks && sc.slow_state != 1) ? 1u : 0u);
        if (e) HIP_CHECK(hipEventRecord(e[1], st));
        launch2_rows(st, fh, 1, uint32_t(sc.n_chunks), sc.max_path_edges);
        // the queued rows (coincident edges, crowded rows): skipped once a frame of this resident scene has shown there are none
        if (sc.n_chunks && sc.slow_state != 1) launch2_rows_slow(st, fh, 1, 1024u, sc.slow_state == 2 ? 0u : 256u, sc.slow_passes);
        if (e) HIP_CHECK(hipEventRecord(e[2], st));
        const uint32_t grid = r->tiles_grid > 0 ? uint32_t(r->tiles_grid) : (overlapped ? ~0u : 0x7fffffffu);   // (~0u: the default shape; any other value caps the wavefronts)
        launch2_tiles(st, fh, 1, uint32_t(sc.n_strip_slots), grid, sc.shader_level, fb);      // (fb: this frame's own target, else the descriptor's)
        if (e) HIP_CHECK(hipEventRecord(e[3], st));
    }
}

int check_counters(swfr_renderer* r, const uint32_t* counters) {
    {
        const uint32_t err = counters[C2_ERROR];
        r->stats.frames += 1; r->stats.queued_rows += counters[C2_SLOW]; r->stats.crowded_rows += counters[C2_HUGE];
        r->stats.tie_rows += counters[C2_TIE_ROWS];
        r->stats.pairtest_limit += counters[C2_TIE_PAIRTEST_SKIPPED] ? 1 : 0;
        r->stats.start_group_limit += ((err & (E2_ACTIVE_EDGES | E2_START_GROUP)) || counters[C2_TIE_SORT_OVERFLOW]) ? 1 : 0;
        r->stats.history_limit += counters[C2_TIE_DEPTH] ? 1 : 0;
        if (err & ~(E2_ACTIVE_EDGES | E2_START_GROUP | E2_CELL_ARENA)) {
            r->fb_valid = false;
            return fail(r, SWFR_ERR_DEVICE, "internal consistency check failed in the raster kernels (code " + std::to_string(err) + ")");
        }
        if (err) {
            r->fb_valid = false;
            return fail(r, SWFR_ERR_CAPACITY, err & E2_CELL_ARENA ? "cell arena exhausted (uneven allocator shares)" :
                        "a pixel row has more than 8192 active edges of one path, or more than 8192 of its edges start at one sample row (scan converter capacity)");
        }
        // the replay of Cairo's edge-list order for coincident edges has capacity limits; a scene that reaches one is refused,
        // never rendered approximately
        if (counters[C2_TIE_PAIRTEST_SKIPPED] | counters[C2_TIE_SORT_OVERFLOW] | counters[C2_TIE_DEPTH]) {
            r->fb_valid = false;
            return fail(r, SWFR_ERR_CAPACITY, std::string("coincident edges beyond the capacity of the list-order replay (") +
                        (counters[C2_TIE_PAIRTEST_SKIPPED] ? "crossing test over more than 2^21 edge pairs; " : "") +
                        (counters[C2_TIE_SORT_OVERFLOW] ? "more than 16 edges starting at one sample row; " : "") +
                        (counters[C2_TIE_DEPTH] ? "tie history deeper than two levels; " : "") + ")");
        }
        return SWFR_OK;
    }
}

// One frame of the resident scene on frame set F as a graph launch: the set's kernels (with the queued-row kernels the scene is
// known to need) are captured from the set's own stream the first time and replayed afterwards; a new upload, or a change in what
// the scene needs of the queued-row kernels, captures again.
void ensure_frame_graph(swfr_renderer* r, const swfr_renderer::Scene& sc, swfr_renderer::FrameSet& F) {
    const uint64_t key = (r->scene_gen << 16) | (uint64_t(sc.slow_state & 3) << 8) | uint64_t(sc.slow_passes & 0xffu) | (uint64_t(1) << 63);
    if (F.graph_key != key) {
        if (F.graph_exec) { (void)hipGraphExecDestroy(F.graph_exec); F.graph_exec = nullptr; }
        if (F.graph) { (void)hipGraphDestroy(F.graph); F.graph = nullptr; }
        F.graph_key = 0;
        HIP_CHECK(hipStreamBeginCapture(F.stream, hipStreamCaptureModeRelaxed));
        try {
            launch_frame(r, sc, F, nullptr, nullptr, true);
        } catch (...) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(F.stream, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        HIP_CHECK(hipStreamEndCapture(F.stream, &F.graph));
        HIP_CHECK(hipGraphInstantiate(&F.graph_exec, F.graph, nullptr, nullptr, 0));
        F.graph_key = key;
    }
}
void launch_frame_graph(swfr_renderer* r, const swfr_renderer::Scene& sc, swfr_renderer::FrameSet& F) {
    ensure_frame_graph(r, sc, F);
    HIP_CHECK(hipGraphLaunch(F.graph_exec, F.stream));
}

// A mapped read-back that nobody has waited for yet (swfr_read_image_async) reads the last frame's framebuffer -- or its
// un-premultiplied copy -- on the handle's stream; frame sets 1..3 render on streams of their own and are not ordered behind it.
// Before anything is launched on them they wait for the copy ON THE DEVICE (no host wait; nothing at all when no read is pending),
// so that a frame can never be rasterized into the buffer the copy is still reading.
void order_frames_behind_pending_read(swfr_renderer* r) {
    if (!r->read_pending || !r->read_done) return;
    for (int k = 1; k < 4; ++k)
        if (r->fs[k].stream) HIP_CHECK(hipStreamWaitEvent(r->fs[k].stream, r->read_done, 0));
}

int render_resident(swfr_renderer* r, uint32_t frames) {
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    if (!r->scene_ready) return fail(r, SWFR_ERR_INVALID, "no scene uploaded");
    if (frames == 0) frames = 1;
    if (frames > 1 && r->scene_from_builder && r->sets_ready < uint32_t(std::max(1, std::min(r->in_flight, 4)))) {
        // swfr_render prepared one frame set for its one frame; several frames of the same scene rotate over all of them
        const auto& e = r->builder->edges(); const auto& p = r->builder->paths(); const auto& st = r->builder->styles();
        const int rc = upload2(r, 0, true, e.data(), e.size(), p.data(), p.size(), st.data(), st.size(), nullptr, true);
        if (rc != SWFR_OK) return rc;
    }
    if (frames > 1 && r->use_graphs && !r->scn[0].slow_verified) {
        // which queued-row kernels the scene needs is learnt from one blocking frame, so that the graphs of the frames behind it
        // are captured once (with exactly those kernels) and a later call finds them ready
        const int rc = render_resident(r, 1);
        if (rc != SWFR_OK) return rc;
        --frames;
    }
    const swfr_renderer::Scene& sc = r->scn[0];
    float setup_ms = 0, rows_ms = 0, tiles_ms = 0, total_ms = 0;
    uint32_t counters[COUNTER_WORDS] = {};
    if (frames > 4096) frames = 4096;
    // all frames are queued back to back, rotating over the frame sets; events bracket every kernel on the stream the frame runs on
    uint32_t n_sets = 1;
    if (frames > 1)
        while (n_sets < uint32_t(std::min(r->in_flight, 4)) && n_sets < r->sets_ready && r->fs[n_sets].stream && (r->n_targets || r->fs[n_sets].d_fb.ptr)) ++n_sets;
    uint32_t stride = r->event_stride < 1 ? 1u : uint32_t(r->event_stride);
    if (frames < 64 && !r->event_stride_given) stride = std::min(stride, 8u);   // (a short run still gets two or three frames with per-kernel times, unless SWFR_EVENT_STRIDE says otherwise)
    const uint32_t first_timed = std::min(stride / 2, frames - 1);      // (not frame 0: the first frames run before the pipeline is full)
    const uint32_t n_timed = (frames - first_timed + stride - 1) / stride;
    // events: four per timed frame (indexed by the timed frame's ordinal), then begin, end and the joins; created once, never inside
    // a later call's timed region unless it times more frames than any call before
    while (r->ev.size() < size_t(n_timed) * 4 + 5) {
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreate(&e));
        r->ev.push_back(e);
    }
    hipEvent_t ev_begin = r->ev[size_t(n_timed) * 4], ev_end = r->ev[size_t(n_timed) * 4 + 1];
    hipEvent_t* ev_join = &r->ev[size_t(n_timed) * 4 + 2];
    // (no clearing of the counters here: k2_bin zeroes its frame's counters itself)
    // every frame set's graph exists before the first launch: a call never captures one between two frames (a short first call --
    // a warm-up of a few frames -- would otherwise leave the capture of the sets it launched with events to the next call)
    if (frames > 1 && r->use_graphs && sc.slow_verified && !(r->resident_batch >= 2 && stride > frames))
        for (uint32_t k = 0; k < n_sets; ++k) ensure_frame_graph(r, sc, r->fs[k]);
    HIP_CHECK(hipEventRecord(ev_begin, r->stream));
    order_frames_behind_pending_read(r);
#ifdef SWFR_BEGIN_WAIT
    for (uint32_t k = 1; k < n_sets; ++k) HIP_CHECK(hipStreamWaitEvent(r->fs[k].stream, ev_begin, 0));
#endif
    // (the other sets' streams are not ordered behind ev_begin: every call ends with all streams joined and waited for, an upload orders
    //  them behind its copy, and three more API calls before the first launch cost a short call a microsecond per frame)
    // Frames per launch: the frame sets are cut into groups of `rb` (their descriptors are contiguous), a group's frames are one launch
    // per kernel (blockIdx.y = frame) on the group's first stream, and the groups alternate -- a third of the host's launches per
    // frame (three launches take the host about as long as a frame takes the GPU) and the GPU still has two streams to overlap.
    // Only without per-kernel events (a timed frame is launched alone).
    uint32_t rb = (r->resident_batch >= 2 && stride > frames) ? uint32_t(r->resident_batch) : 1u;
    while (rb > 1 && (rb > n_sets || n_sets % rb != 0)) --rb;
    const bool batched = rb > 1 && frames >= 2;
    uint32_t last_set = (frames - 1) % n_sets;
    int64_t last_on[4] = {-1, -1, -1, -1};                 // per frame set's stream: the last launch of this call it carries (none: -1)
    if (batched) {
        const uint32_t groups = n_sets / rb;
        uint32_t gi = 0;
        for (uint32_t f = 0; f < frames; f += rb, ++gi) {
            const uint32_t g = gi % groups, cnt = std::min(rb, frames - f);
            const hipStream_t st = r->fs[g * rb].stream;
            const Frame2* fh = sc.frames_dev + g * rb;
            launch2_bin(st, fh, cnt, uint32_t(sc.n_edges), uint32_t(sc.n_paths), uint32_t(sc.n_bands), (sc.n_chunks && sc.slow_state != 1) ? 1u : 0u);
            launch2_rows(st, fh, cnt, uint32_t(sc.n_chunks), sc.max_path_edges);
            if (sc.n_chunks && sc.slow_state != 1) launch2_rows_slow(st, fh, cnt, 1024u, sc.slow_state == 2 ? 0u : 256u, sc.slow_passes);
            launch2_tiles(st, fh, cnt, uint32_t(sc.n_strip_slots), r->tiles_grid > 0 ? uint32_t(r->tiles_grid) : ~0u, sc.shader_level, nullptr);
            last_set = g * rb + cnt - 1;
            last_on[g * rb] = int64_t(gi);
        }
    }
    for (uint32_t f = 0; f < frames && !batched; ++f) {
        swfr_renderer::FrameSet& F = r->fs[f % n_sets];
        const bool timed = f >= first_timed && (f - first_timed) % stride == 0;   // per-kernel events on every stride-th frame (each costs a queue packet)
        if (!timed && frames > 1 && r->use_graphs && sc.slow_verified) launch_frame_graph(r, sc, F);
        else launch_frame(r, sc, F, nullptr, timed ? &r->ev[size_t((f - first_timed) / stride) * 4] : nullptr, n_sets > 1 && frames > 1);
        last_on[f % n_sets] = int64_t(f);
    }
    for (uint32_t k = 1; k < n_sets; ++k) {                // (only the streams that carried something join)
        if (last_on[k] < 0) continue;
        HIP_CHECK(hipEventRecord(ev_join[k - 1], r->fs[k].stream));
        HIP_CHECK(hipStreamWaitEvent(r->stream, ev_join[k - 1], 0));
    }
    HIP_CHECK(hipEventRecord(ev_end, r->stream));
    HIP_CHECK(hipGetLastError());
    // the counters of every set come back through pinned memory behind the last kernel: one synchronisation for everything
    if (!r->h_counters) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&r->h_counters), 4 * COUNTER_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    const uint32_t used_sets = std::min(frames, n_sets);               // (a set that rendered no frame of this call holds an older frame's counters)
    HIP_CHECK(hipMemcpyAsync(r->h_counters, r->d_counters.ptr, size_t(used_sets) * COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, r->stream));
    // The other sets' streams are waited for too, in the order their last frames finish: the runtime then knows them idle, and a
    // hipDeviceSynchronize behind this call (torch.cuda.synchronize()) costs 3 us instead of 47 (it drains every stream it has not seen
    // idle, 12 us apiece).  All but the last of these waits return while the GPU is still at the call's last frames.
    for (;;) {
        int pick = -1;
        for (uint32_t k = 1; k < n_sets; ++k)
            if (last_on[k] >= 0 && (pick < 0 || last_on[k] < last_on[pick])) pick = int(k);
        if (pick < 0) break;
        if (r->fs[pick].stream) HIP_CHECK(hipStreamSynchronize(r->fs[pick].stream));
        last_on[pick] = -1;
    }
    HIP_CHECK(hipStreamSynchronize(r->stream));
    r->fb_cur = r->n_targets ? r->targets[last_set % r->n_targets] : r->fs[last_set].d_fb.ptr;
    uint32_t timed_frames = 0;
    for (uint32_t f = batched ? frames : first_timed; f < frames; f += stride, ++timed_frames) {      // (frames per launch: no per-kernel events)
        hipEvent_t* e = &r->ev[size_t(timed_frames) * 4];
        float a = 0, b = 0, c = 0;
        HIP_CHECK(hipEventElapsedTime(&a, e[0], e[1]));
        HIP_CHECK(hipEventElapsedTime(&b, e[1], e[2]));
        HIP_CHECK(hipEventElapsedTime(&c, e[2], e[3]));
        setup_ms += a; rows_ms += b; tiles_ms += c;
    }
    HIP_CHECK(hipEventElapsedTime(&total_ms, ev_begin, ev_end));
    std::memcpy(counters, r->h_counters, sizeof counters);
    for (uint32_t k = 1; k < used_sets; ++k)               // (set 0's counts, every set's error and limit flags)
        for (uint32_t w : {uint32_t(C2_ERROR), uint32_t(C2_TIE_PAIRTEST_SKIPPED), uint32_t(C2_TIE_SORT_OVERFLOW), uint32_t(C2_TIE_DEPTH)})
            counters[w] |= r->h_counters[k * COUNTER_WORDS + w];
    r->timing = swfr_timing{total_ms, setup_ms, rows_ms, tiles_ms, frames, sc.n_edges, sc.n_paths, sc.n_rows, 0, timed_frames};
    r->fb_valid = true;
    {
        uint32_t slow = 0, huge = 0;
        for (uint32_t k = 0; k < used_sets; ++k) { slow |= r->h_counters[k * COUNTER_WORDS + C2_SLOW]; huge |= r->h_counters[k * COUNTER_WORDS + C2_HUGE]; }
        uint32_t passes = 1;
        for (uint32_t k = 0; k < used_sets; ++k)
            for (uint32_t q = 1; q < SLOW_PASSES; ++q) {
                const uint32_t* c = r->h_counters + k * COUNTER_WORDS;
                if (c[C2_SLOWQ + q] | c[C2_HUGEQ + q]) { passes = std::max(passes, q + 1); huge |= c[C2_HUGEQ + q]; }
            }
        // were the launches enough?  (a resident scene starts with everything, or with the previous scene's needs as a guess)
        const bool short_launch = (sc.slow_state == 1 && slow) || (sc.slow_state == 2 && huge) || passes > sc.slow_passes;
        r->scn[0].slow_state = slow == 0 ? 1 : (huge == 0 ? 2 : 0);     // what the next frames of this resident scene can skip
        r->scn[0].slow_passes = passes;
        r->hint_slow_state = r->scn[0].slow_state; r->hint_slow_passes = passes;
        r->scn[0].slow_verified = !short_launch;
        if (short_launch) {                                    // rows were left in a queue nobody read: the frames are not valid
            r->scn[0].slow_state = 0; r->scn[0].slow_passes = SLOW_PASSES;
            r->fb_valid = false;
            return render_resident(r, frames);
        }
        r->timing.n_records = 0;
        for (uint32_t h = 0; h < C2_HEADS; ++h) r->timing.n_records += counters[C2_HEAD + h];     // cells of the last frame of set 0
    }
    return check_counters(r, counters);
}

// A batch of different frames (BASELINE config 3: 256 morph ratios): frame i is built on the host, uploaded into the scene slot
// of frame set i mod n and rasterized on that set's stream straight into device_dst + i * frame_stride, while the host already
// builds frame i+1.  Nothing waits for the GPU until the end (a scene slot's pinned staging buffer is reused only after its
// previous H2D copy has completed).
// swfr_render_batch with a device destination: the frames are rendered in groups of SWFR_BATCH_FRAMES (default 64) by
// one launch per kernel and group (blockIdx.y = frame of the group, every frame with its own descriptor, tables and buffers), so a
// batch of small frames -- the 256 ratios of a morph shape -- fills the GPU instead of paying a launch chain per frame.  Two groups
// alternate: the host builds group g + 1 (scene walk, flattening, layout) while the GPU rasterizes group g.
int render_batch2(swfr_renderer* r, const swfr_stage* stages, uint32_t n, void* device_dst, size_t frame_stride) {
    struct FrameData { std::vector<swfr_edge> e; std::vector<swfr_path> p; std::vector<swfr_style> s; SceneLayout L; };
    static thread_local std::vector<FrameData> fd;
    // frames per group: SWFR_BATCH_FRAMES at most; a call with fewer frames is still cut into four groups, so that building group
    // g + 1 overlaps rasterizing group g, as long as a group keeps enough pixels (32 Mpx: four 4K frames, thirty-two 1024x1024 ones)
    // for one launch per kernel to fill the GPU
    const uint32_t fill = uint32_t((size_t(32) << 20) / std::max<size_t>(size_t(r->width) * r->height, 1)) + 1;
    const uint32_t B = std::min(uint32_t(std::max(1, r->batch_frames)), std::max((n + 3) / 4, fill));
    const uint32_t tiles_x = (r->width + TILE_W - 1) / TILE_W;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    double t_build = 0, t_wait = 0, t_stage = 0, t_launch = 0;
    const bool trace = std::getenv("SWFR_BATCH_TRACE") != nullptr;          // host time per stage of the call, to stderr
    if (r->bitmap_table_dirty) {
        r->d_bitmap_table.reserve(r->bitmap_table.size());
        if (!r->bitmap_table.empty()) HIP_CHECK(hipMemcpy(r->d_bitmap_table.ptr, r->bitmap_table.data(), r->bitmap_table.size() * sizeof(DevBitmap), hipMemcpyHostToDevice));
        r->bitmap_table_dirty = false;
    }
    int rc = SWFR_OK;
    double device_ms = 0;                                   // kernels only, summed over the groups (they do not overlap each other much)
    uint32_t gi = 0;
    struct Pending { uint32_t first, count; };
    Pending pend[2] = {{0, 0}, {0, 0}};
    auto finish_group = [&](int g) {                        // waits for a group's launches and checks its frames' counters
        auto& G = r->groups[g];
        if (!pend[g].count) return;
        HIP_CHECK(hipStreamSynchronize(G.stream));
        float ms = 0;
        if (hipEventElapsedTime(&ms, G.ev_begin, G.ev_end) == hipSuccess) device_ms += ms;
        for (uint32_t k = 0; k < pend[g].count && rc == SWFR_OK; ++k) rc = check_counters(r, G.h_counters + size_t(k) * COUNTER_WORDS);
        pend[g].count = 0;
    };
    for (uint32_t first = 0; first < n && rc == SWFR_OK; first += B, ++gi) {
        const uint32_t cnt = std::min(B, n - first);
        const int g = int(gi & 1);
        auto& G = r->groups[g];
        if (!G.stream) HIP_CHECK(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
        // ---- host: build the group's frames (the other group is being rasterized meanwhile)
        if (fd.size() < cnt) fd.resize(cnt);
        size_t arena_bytes = pad(cnt * sizeof(Frame2)) + 4096, work_bytes = 0, cls_bytes = 0;
        size_t max_e = 0, max_p = 0, max_bands = 0, max_chunks = 0, max_strips = 0;
        int shader_level = 0;
        uint32_t max_pe = 0;
        auto t0 = clk::now();
        for (uint32_t k = 0; k < cnt; ++k) {
            FrameData& F = fd[k];
            r->builder->build(stages[first + k]);
            F.e = r->builder->edges(); F.p = r->builder->paths(); F.s = r->builder->styles();
            validate_scene(r, F.e.data(), F.e.size(), F.p.data(), F.p.size(), F.s.data(), F.s.size());
            {
                std::vector<swfr_edge> se; std::vector<swfr_path> sp;
                if (split_wide_paths(F.e.data(), F.p.data(), F.p.size(), se, sp)) {
                    const std::vector<swfr_path> orig = F.p;
                    F.e.swap(se); F.p.swap(sp);
                    layout_scene(r, F.e.data(), F.e.size(), F.p.data(), F.p.size(), F.s.data(), F.s.size(), F.L, orig.data(), orig.size());
                } else layout_scene(r, F.e.data(), F.e.size(), F.p.data(), F.p.size(), F.s.data(), F.s.size(), F.L);
            }
            arena_bytes += scene_arena_bytes(F.L, F.e.size(), F.p.size(), F.s.size());
            const SceneLayout& L = F.L;
            work_bytes += pad((2 * F.e.size() + 1) * sizeof(DevEdge)) + pad(L.n_slots * sizeof(BandEntry2)) + pad((L.n_slots * TILE_H + 64) * sizeof(RowInfo2)) +
                          pad(L.cell_total * sizeof(Cell)) + 2 * pad(2 * (L.n_rows + 64) * sizeof(SlowRow)) + 2 * pad((F.p.size() + 64) * sizeof(uint32_t)) +
                          pad((L.n_chunks + 1) * sizeof(ChunkInfo)) + pad((L.n_slots + 8) * sizeof(BandSlot)) + pad((L.n_strip_slots + 1) * sizeof(StripDesc)) +
                          pad((L.n_strips + 1) * sizeof(uint32_t)) + pad(COUNTER_WORDS * sizeof(uint32_t));
            cls_bytes += pad(cls_region_bytes(L.n_slots, tiles_x, L.n_strips));
            max_e = std::max(max_e, F.e.size()); max_p = std::max(max_p, F.p.size()); max_bands = std::max(max_bands, L.n_bands);
            max_chunks = std::max(max_chunks, L.n_chunks); max_strips = std::max(max_strips, L.n_strip_slots);
            shader_level = std::max(shader_level, L.shader_level);
            max_pe = std::max(max_pe, L.max_path_edges);
        }
        t_build += ms_since(t0); t0 = clk::now();
        // ---- this group's previous use must be over before its staging and device buffers are rewritten
        finish_group(g);
        t_wait += ms_since(t0); t0 = clk::now();
        if (rc != SWFR_OK) break;
        G.arena.begin(arena_bytes);
        const uint8_t* work_before = G.work.ptr;
        G.work.reserve(work_bytes + 4096); G.cls.reserve(cls_bytes + 4096);
        if (G.work.ptr != work_before) HIP_CHECK(hipMemsetAsync(G.work.ptr, 0, G.work.cap, G.stream));      // (fresh memory: the strip costs start at zero)
        HIP_CHECK(hipMemsetAsync(G.cls.ptr, 0, cls_bytes, G.stream));                                        // class bytes outside the paths' rectangles
        if (G.h_counters_cap < cnt) {
            if (G.h_counters) (void)hipHostFree(G.h_counters);
            HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&G.h_counters), size_t(B) * COUNTER_WORDS * sizeof(uint32_t), hipHostMallocDefault));
            G.h_counters_cap = B;
        }
        std::vector<Frame2> fr(cnt);
        uint8_t* w = G.work.ptr;
        uint8_t* c = G.cls.ptr;
        auto carve = [&](size_t bytes) { uint8_t* q = w; w += pad(bytes); return q; };
        for (uint32_t k = 0; k < cnt; ++k) {
            FrameData& F = fd[k];
            const SceneLayout& L = F.L;
            Frame2& f = fr[k];
            std::memset(&f, 0, sizeof f);
            push_scene(G.arena, L, F.e.data(), F.e.size(), F.p.data(), F.p.size(), F.s.data(), F.s.size(), f);
            fill_frame_sizes(r, L, F.e.size(), F.p.size(), f);
            f.src.bitmaps = r->d_bitmap_table.ptr;
            f.edges = reinterpret_cast<DevEdge*>(carve((2 * F.e.size() + 1) * sizeof(DevEdge)));
            f.band_list = reinterpret_cast<BandEntry2*>(carve(L.n_slots * sizeof(BandEntry2)));
            f.rows = reinterpret_cast<RowInfo2*>(carve((L.n_slots * TILE_H + 64) * sizeof(RowInfo2)));
            f.cells = reinterpret_cast<Cell*>(carve(L.cell_total * sizeof(Cell)));
            f.slow = reinterpret_cast<SlowRow*>(carve(2 * (L.n_rows + 64) * sizeof(SlowRow)));
            f.huge = reinterpret_cast<SlowRow*>(carve(2 * (L.n_rows + 64) * sizeof(SlowRow)));
            f.path_flag = reinterpret_cast<uint32_t*>(carve((F.p.size() + 64) * sizeof(uint32_t)));
            f.path_queue = reinterpret_cast<uint32_t*>(carve((F.p.size() + 64) * sizeof(uint32_t)));
            f.chunks = reinterpret_cast<ChunkInfo*>(carve((L.n_chunks + 1) * sizeof(ChunkInfo)));
            f.band_slots = reinterpret_cast<BandSlot*>(carve((L.n_slots + 8) * sizeof(BandSlot)));
            f.strips = reinterpret_cast<StripDesc*>(carve((L.n_strip_slots + 1) * sizeof(StripDesc)));
            f.strip_cost = reinterpret_cast<uint32_t*>(carve((L.n_strips + 1) * sizeof(uint32_t)));
            f.counters = reinterpret_cast<uint32_t*>(carve(COUNTER_WORDS * sizeof(uint32_t)));
            f.cls = c; f.strip_top = reinterpret_cast<StripTop*>(c + cls_bytes_of(L.n_slots, tiles_x)); c += pad(cls_region_bytes(L.n_slots, tiles_x, L.n_strips));
            f.strip_order = 0;                              // (a frame's buffers held another frame before: no cost history to order by)
            f.fb = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(device_dst) + size_t(first + k) * frame_stride);
        }
        const Frame2* frames_dev = static_cast<Frame2*>(G.arena.push(fr.data(), cnt * sizeof(Frame2)));
        G.arena.flush(G.stream);
        t_stage += ms_since(t0); t0 = clk::now();
        if (!G.ev_begin) { HIP_CHECK(hipEventCreate(&G.ev_begin)); HIP_CHECK(hipEventCreate(&G.ev_end)); }
        HIP_CHECK(hipEventRecord(G.ev_begin, G.stream));
        launch2_bin(G.stream, frames_dev, cnt, uint32_t(max_e), uint32_t(max_p), uint32_t(max_bands), 1u);
        launch2_rows(G.stream, frames_dev, cnt, uint32_t(max_chunks), max_pe);
        if (max_chunks) launch2_rows_slow(G.stream, frames_dev, cnt, 256u, 64u, SLOW_PASSES);
        launch2_tiles(G.stream, frames_dev, cnt, uint32_t(max_strips), ~0u, shader_level, nullptr);
        HIP_CHECK(hipEventRecord(G.ev_end, G.stream));
        for (uint32_t k = 0; k < cnt; ++k)
            HIP_CHECK(hipMemcpyAsync(G.h_counters + size_t(k) * COUNTER_WORDS, fr[k].counters, COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, G.stream));
        HIP_CHECK(hipGetLastError());
        pend[g] = Pending{first, cnt};
        r->fb_cur = fr[cnt - 1].fb;
        t_launch += ms_since(t0);
    }
    auto t_end = clk::now();
    finish_group(0);
    finish_group(1);
    if (trace) std::fprintf(stderr, "[swfr] batch of %u frames in groups of %u: build+layout %.3f ms, waiting for a group's buffers %.3f, staging %.3f, launches %.3f, final wait %.3f, kernels %.3f\n",
                            n, B, t_build, t_wait, t_stage, t_launch, ms_since(t_end), device_ms);
    r->scene_ready = false;
    r->fb_valid = rc == SWFR_OK;
    r->timing = swfr_timing{float(device_ms), 0, 0, 0, n, 0, 0, 0, 0, 0};     // (swfr_last_timing: total_ms = the groups' kernel time)
    return rc;
}

// The resident scene as `per_launch` frames per kernel launch (blockIdx.y = frame, every frame with its own kernel-written buffers
// and its own framebuffer), `launches` times back to back on one stream: the GPU saturated by one scene -- what frames in flight
// approximate with streams.  Every frame recomputes everything from the raw edge list, as in swfr_render_resident.
int render_resident_batched(swfr_renderer* r, uint32_t per_launch, uint32_t launches, float* ms_out) {
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    if (!r->scene_ready) return fail(r, SWFR_ERR_INVALID, "no scene uploaded");
    if (per_launch == 0 || per_launch > 64 || launches == 0) return fail(r, SWFR_ERR_INVALID, "1..64 frames per launch, at least one launch");
    // which queued-row kernels the scene needs: from a blocking frame of the scene itself
    if (!r->scn[0].slow_verified) { const int rc = render_resident(r, 1); if (rc != SWFR_OK) return rc; }
    const swfr_renderer::Scene& sc = r->scn[0];
    const uint32_t B = per_launch, tiles_x = (r->width + TILE_W - 1) / TILE_W;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t n_px = size_t(r->width) * r->height;
    const size_t work_one = pad((2 * sc.n_edges + 1) * sizeof(DevEdge)) + pad(sc.n_slots * sizeof(BandEntry2)) + pad((sc.n_slots * TILE_H + 64) * sizeof(RowInfo2)) +
                            pad(sc.cell_total * sizeof(Cell)) + 2 * pad(2 * (sc.n_rows + 64) * sizeof(SlowRow)) + 2 * pad((sc.n_paths + 64) * sizeof(uint32_t)) +
                            pad((sc.n_chunks + 1) * sizeof(ChunkInfo)) + pad((sc.n_slots + 8) * sizeof(BandSlot)) + pad((sc.n_strip_slots + 1) * sizeof(StripDesc)) +
                            pad((sc.n_strips + 1) * sizeof(uint32_t)) + pad(COUNTER_WORDS * sizeof(uint32_t));
    const size_t cls_one = pad(cls_region_bytes(sc.n_slots, tiles_x, sc.n_strips));
    const hipStream_t st = r->stream;
    r->fb_valid = false;                                       // (fb_cur may point into a buffer the next lines reallocate; set again below)
    r->fb_cur = nullptr;
    const uint32_t* rb_fb_before = r->rb_fb.ptr;
    r->rb_work.reserve(work_one * B + 4096); r->rb_cls.reserve(cls_one * B + 4096); r->rb_frames.reserve(B); r->rb_fb.reserve(n_px * B);
    // (a handle that owns only some tile-rows writes only those: the rest of a fresh framebuffer is cleared once, as upload2 does for d_fb)
    if (r->rb_fb.ptr != rb_fb_before) HIP_CHECK(hipMemsetAsync(r->rb_fb.ptr, 0, r->rb_fb.cap * sizeof(uint32_t), st));
    HIP_CHECK(hipMemsetAsync(r->rb_work.ptr, 0, work_one * B, st));           // (strip costs start at zero; class bytes outside the paths' rectangles)
    HIP_CHECK(hipMemsetAsync(r->rb_cls.ptr, 0, cls_one * B, st));
    std::vector<Frame2> fr(B);
    uint8_t* w = r->rb_work.ptr;
    auto carve = [&](size_t bytes) { uint8_t* q = w; w += pad(bytes); return q; };
    for (uint32_t k = 0; k < B; ++k) {
        Frame2& f = fr[k];
        f = sc.proto;
        f.edges = reinterpret_cast<DevEdge*>(carve((2 * sc.n_edges + 1) * sizeof(DevEdge)));
        f.band_list = reinterpret_cast<BandEntry2*>(carve(sc.n_slots * sizeof(BandEntry2)));
        f.rows = reinterpret_cast<RowInfo2*>(carve((sc.n_slots * TILE_H + 64) * sizeof(RowInfo2)));
        f.cells = reinterpret_cast<Cell*>(carve(sc.cell_total * sizeof(Cell)));
        f.slow = reinterpret_cast<SlowRow*>(carve(2 * (sc.n_rows + 64) * sizeof(SlowRow)));
        f.huge = reinterpret_cast<SlowRow*>(carve(2 * (sc.n_rows + 64) * sizeof(SlowRow)));
        f.path_flag = reinterpret_cast<uint32_t*>(carve((sc.n_paths + 64) * sizeof(uint32_t)));
        f.path_queue = reinterpret_cast<uint32_t*>(carve((sc.n_paths + 64) * sizeof(uint32_t)));
        f.chunks = reinterpret_cast<ChunkInfo*>(carve((sc.n_chunks + 1) * sizeof(ChunkInfo)));
        f.band_slots = reinterpret_cast<BandSlot*>(carve((sc.n_slots + 8) * sizeof(BandSlot)));
        f.strips = reinterpret_cast<StripDesc*>(carve((sc.n_strip_slots + 1) * sizeof(StripDesc)));
        f.strip_cost = reinterpret_cast<uint32_t*>(carve((sc.n_strips + 1) * sizeof(uint32_t)));
        f.counters = reinterpret_cast<uint32_t*>(carve(COUNTER_WORDS * sizeof(uint32_t)));
        f.cls = r->rb_cls.ptr + cls_one * k; f.strip_top = reinterpret_cast<StripTop*>(f.cls + cls_bytes_of(sc.n_slots, tiles_x));
        f.fb = r->rb_fb.ptr + n_px * k;
    }
    HIP_CHECK(hipMemcpyAsync(r->rb_frames.ptr, fr.data(), B * sizeof(Frame2), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));                       // (fr is a local)
    if (!r->rb_ev[0]) { HIP_CHECK(hipEventCreate(&r->rb_ev[0])); HIP_CHECK(hipEventCreate(&r->rb_ev[1])); }
    auto one_launch = [&]() {
        launch2_bin(st, r->rb_frames.ptr, B, uint32_t(sc.n_edges), uint32_t(sc.n_paths), uint32_t(sc.n_bands), (sc.n_chunks && sc.slow_state != 1) ? 1u : 0u);
        launch2_rows(st, r->rb_frames.ptr, B, uint32_t(sc.n_chunks), sc.max_path_edges);
        if (sc.n_chunks && sc.slow_state != 1) launch2_rows_slow(st, r->rb_frames.ptr, B, 256u, sc.slow_state == 2 ? 0u : 64u, sc.slow_passes);
        launch2_tiles(st, r->rb_frames.ptr, B, uint32_t(sc.n_strip_slots), ~0u, sc.shader_level, nullptr);
    };
    one_launch();                                              // warm-up: leaves a cost history for the strip order
    HIP_CHECK(hipEventRecord(r->rb_ev[0], st));
    for (uint32_t l = 0; l < launches; ++l) one_launch();
    HIP_CHECK(hipEventRecord(r->rb_ev[1], st));
    HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> hc(size_t(B) * COUNTER_WORDS);
    for (uint32_t k = 0; k < B; ++k)
        HIP_CHECK(hipMemcpyAsync(hc.data() + size_t(k) * COUNTER_WORDS, fr[k].counters, COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, r->rb_ev[0], r->rb_ev[1]));
    if (ms_out) *ms_out = ms;
    r->rb_count = B;
    r->fb_cur = fr[B - 1].fb;
    r->fb_valid = true;
    int rc = SWFR_OK;
    for (uint32_t k = 0; k < B && rc == SWFR_OK; ++k) {
        const uint32_t* c = hc.data() + size_t(k) * COUNTER_WORDS;
        // a frame that queued rows for kernels this call did not launch is not valid
        if ((sc.slow_state == 1 && c[C2_SLOW]) || (sc.slow_state == 2 && c[C2_HUGE])) rc = fail(r, SWFR_ERR_DEVICE, "queued rows without their kernels in a batched launch");
        if (rc == SWFR_OK) rc = check_counters(r, c);
    }
    return rc;
}

int render_batch(swfr_renderer* r, const swfr_stage* stages, uint32_t n, void* device_dst, size_t frame_stride) {
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    if (n == 0) return SWFR_OK;
    if (device_dst && frame_stride < size_t(r->width) * r->height * 4) return fail(r, SWFR_ERR_INVALID, "frame stride smaller than a frame");
    if (device_dst && r->batch_frames > 1) return render_batch2(r, stages, n, device_dst, frame_stride);
    order_frames_behind_pending_read(r);                       // (frames without a device_dst land in the frame sets' own framebuffers)
    const uint32_t n_sets = uint32_t(std::max(1, std::min(r->in_flight, 4)));
    std::vector<uint32_t*> pinned_counters;
    uint32_t* hc = nullptr;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hc), size_t(n) * COUNTER_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    int rc = SWFR_OK;
    try {
        for (uint32_t i = 0; i < n; ++i) {
            const int k = int(i % n_sets);
            r->builder->build(stages[i]);
            const auto& e = r->builder->edges();
            const auto& p = r->builder->paths();
            const auto& s = r->builder->styles();
            uint32_t* fb_dst = device_dst ? reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(device_dst) + size_t(i) * frame_stride) : nullptr;
            validate_scene(r, e.data(), e.size(), p.data(), p.size(), s.data(), s.size());
            rc = upload2(r, k, false, e.data(), e.size(), p.data(), p.size(), s.data(), s.size(), fb_dst, true);   // (the builder's edges carry their path index)
            if (rc != SWFR_OK) break;
            // nothing renders a frame of a batch again: every frame launches all the queued-row kernels (upload2 seeds set 0 with what
            // the previous scene needed, a guess only render_resident checks against the frame's counters)
            r->scn[k].slow_state = 0; r->scn[k].slow_passes = SLOW_PASSES;
            swfr_renderer::FrameSet& F = r->fs[k];
            if (k > 0 && i < n_sets) HIP_CHECK(hipStreamSynchronize(r->stream));   // first use of the set: bitmap table etc. are in place
            uint32_t* fb = device_dst ? reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(device_dst) + size_t(i) * frame_stride) : F.d_fb.ptr;
            launch_frame(r, r->scn[k], F, nullptr, nullptr, n_sets > 1 && n > 1);     // (the descriptor carries fb)
            HIP_CHECK(hipMemcpyAsync(hc + size_t(i) * COUNTER_WORDS, F.counters, COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, F.stream));
            r->fb_cur = fb;
        }
        HIP_CHECK(hipGetLastError());
        for (uint32_t k = 0; k < n_sets; ++k)
            if (r->fs[k].stream) HIP_CHECK(hipStreamSynchronize(r->fs[k].stream));
    } catch (...) {
        (void)hipHostFree(hc);
        throw;
    }
    r->scene_ready = false;                 // scene 0 now holds some frame of the batch, not a scene the caller uploaded
    r->fb_valid = true;
    if (rc == SWFR_OK)
        for (uint32_t i = 0; i < n && rc == SWFR_OK; ++i) rc = check_counters(r, hc + size_t(i) * COUNTER_WORDS);
    (void)hipHostFree(hc);
    return rc;
}

// Mapped read-back of the last frame: the framebuffer (or its un-premultiplied copy) goes to the handle's PINNED staging buffer by
// one asynchronous copy on the handle's stream behind the frame's kernels, and the caller reads it there (a pageable destination makes
// the runtime stage the copy through its own pinned pieces and a host memcpy: 2.3 ms for a 4K frame instead of 0.6).
int start_read(swfr_renderer* r, int premultiplied) {
    if (r->read_pending) { HIP_CHECK(hipEventSynchronize(r->read_done)); r->read_pending = false; }      // (an earlier read-back nobody waited for)
    const size_t n = size_t(r->width) * r->height, bytes = n * 4;
    const uint32_t* src = r->fb_cur ? r->fb_cur : r->fs[0].d_fb.ptr;
    if (!src) return fail(r, SWFR_ERR_INVALID, "no framebuffer");
    if (!premultiplied) {
        r->d_tmp.reserve(n);
        launch_unpremultiply(r->stream, src, r->d_tmp.ptr, n);
        src = r->d_tmp.ptr;
    }
    if (r->h_image_cap < bytes) {
        if (r->h_image) (void)hipHostFree(r->h_image);
        r->h_image = nullptr; r->h_image_cap = 0;
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&r->h_image), bytes, hipHostMallocDefault));
        r->h_image_cap = bytes;
    }
    if (!r->read_done) HIP_CHECK(hipEventCreateWithFlags(&r->read_done, hipEventDisableTiming));
    HIP_CHECK(hipMemcpyAsync(r->h_image, src, bytes, hipMemcpyDeviceToHost, r->stream));
    HIP_CHECK(hipEventRecord(r->read_done, r->stream));
    r->read_pending = true;
    return SWFR_OK;
}
int finish_read(swfr_renderer* r) {
    if (!r->read_pending) return fail(r, SWFR_ERR_INVALID, "no read-back in flight");
    HIP_CHECK(hipEventSynchronize(r->read_done));
    r->read_pending = false;
    return SWFR_OK;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

uint32_t swfr_abi_version(void) { return SWFR_ABI_VERSION; }

const char* swfr_last_error(const swfr_renderer* r) { return r ? r->error.c_str() : "null handle"; }

int swfr_create(uint32_t width, uint32_t height, const swfr_config* cfg, swfr_renderer** out) {
    if (!out) return SWFR_ERR_INVALID;
    *out = nullptr;
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return SWFR_ERR_INVALID;
    std::unique_ptr<swfr_renderer> r(new (std::nothrow) swfr_renderer);
    if (!r) return SWFR_ERR_CAPACITY;
    r->width = width;
    r->height = height;
    if (cfg) r->cfg = *cfg;
    if (r->cfg.band_count > 1 && r->cfg.band_index >= r->cfg.band_count) return SWFR_ERR_INVALID;
    r->builder.reset(new FrameBuilder(width, height, (r->cfg.flags & SWFR_FLAG_EVEN_ODD) != 0));
    if (const char* fl = std::getenv("SWFR_FAST_LIMIT")) r->fast_limit = std::atoi(fl);
    if (const char* tg = std::getenv("SWFR_TILES_GRID")) r->tiles_grid = std::atoi(tg);
    if (const char* bf = std::getenv("SWFR_BATCH_FRAMES")) r->batch_frames = std::max(1, std::atoi(bf));
    if (const char* cr = std::getenv("SWFR_CHUNK_ROWS")) r->force_chunk_rows = std::atoi(cr);
    if (const char* so = std::getenv("SWFR_STRIP_ORDER")) r->strip_order = std::atoi(so);
    if (const char* fi = std::getenv("SWFR_FRAMES_IN_FLIGHT")) r->in_flight = std::atoi(fi);
    if (const char* es = std::getenv("SWFR_EVENT_STRIDE")) { r->event_stride = std::atoi(es); r->event_stride_given = true; }
    if (const char* ug = std::getenv("SWFR_GRAPHS")) r->use_graphs = std::atoi(ug);
#ifdef SWFR_EMU
    r->use_graphs = 0;                                              // (the emulator's runtime has no graphs)
#endif
    if (const char* rb = std::getenv("SWFR_RESIDENT_BATCH")) r->resident_batch = std::atoi(rb);
    if (r->cfg.device == SWFR_DEVICE_HOST_ONLY) {
        *out = r.release();
        return SWFR_OK;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SWFR_ERR_NO_DEVICE;
    if (r->cfg.device < 0 || r->cfg.device >= count) return SWFR_ERR_NO_DEVICE;
    r->has_device = true;
    swfr_renderer* raw = r.get();
    const int rc = guarded(raw, [&]() {
        HIP_CHECK(hipStreamCreateWithFlags(&raw->stream, hipStreamNonBlocking));
        raw->fs[0].stream = raw->stream;
        // the frame sets' streams first: HIP deals streams onto a few hardware queues in creation order, and the sets must not share one
        for (int k = 1; k < std::max(1, std::min(raw->in_flight, 4)); ++k) HIP_CHECK(hipStreamCreateWithFlags(&raw->fs[k].stream, hipStreamNonBlocking));
        for (int i = 0; i < 4 * 32 + 5; ++i) { hipEvent_t e = nullptr; HIP_CHECK(hipEventCreate(&e)); raw->ev.push_back(e); }   // (so that no render call has to create its timing events inside somebody's timed region)
        raw->fs[0].d_fb.reserve(size_t(width) * height);
        HIP_CHECK(hipMemsetAsync(raw->fs[0].d_fb.ptr, 0, size_t(width) * height * 4, raw->stream));
        raw->d_counters.reserve(4 * COUNTER_WORDS);
        raw->fs[0].counters = raw->d_counters.ptr;
        HIP_CHECK(hipStreamSynchronize(raw->stream));
        return int(SWFR_OK);
    });
    if (rc != SWFR_OK) return rc;
    *out = r.release();
    return SWFR_OK;
}

void swfr_destroy(swfr_renderer* r) { delete r; }

int swfr_register_shape(swfr_renderer* r, const swfr_define_shape* tag, uint32_t* out_id) {
    if (!r || !tag || !out_id) return fail(r, SWFR_ERR_INVALID, "null argument");
    return guarded(r, [&]() {
        *out_id = r->builder->add_shape(decode_shape(*tag, false));
        return int(SWFR_OK);
    });
}

int swfr_register_morph_shape(swfr_renderer* r, const swfr_define_shape* tag, uint32_t* out_id) {
    if (!r || !tag || !out_id) return fail(r, SWFR_ERR_INVALID, "null argument");
    return guarded(r, [&]() {
        *out_id = r->builder->add_morph_shape(decode_shape(*tag, true));
        return int(SWFR_OK);
    });
}

int swfr_decode_x_swf_bmp(const uint8_t* data, size_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, size_t rgba_cap) {
    if (!data || !width || !height) return SWFR_ERR_INVALID;
    try {
        std::vector<uint8_t> px;
        const XSwfBmpStatus st = decode_x_swf_bmp(data, len, *width, *height, px);
        if (st == XSwfBmpStatus::UnsupportedFormat) return SWFR_ERR_NOT_IMPLEMENTED;
        if (st != XSwfBmpStatus::Ok) return SWFR_ERR_INVALID;
        if (rgba) {
            if (rgba_cap < px.size()) return SWFR_ERR_CAPACITY;
            std::memcpy(rgba, px.data(), px.size());
        }
        return SWFR_OK;
    } catch (...) { return SWFR_ERR_DEVICE; }
}

int swfr_register_bitmap_tag(swfr_renderer* r, uint32_t id, const char* media_type, const uint8_t* data, size_t len) {
    if (!r || !media_type || !data) return fail(r, SWFR_ERR_INVALID, "null argument");
    if (std::strcmp(media_type, "image/x-swf-bmp") != 0)
        return fail(r, SWFR_ERR_NOT_IMPLEMENTED, std::string("NotImplemented: Support for ") + media_type + " images");   // node-canvas-bitmap-service.ts:34-35
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> px;
    XSwfBmpStatus st = XSwfBmpStatus::Corrupt;
    try { st = decode_x_swf_bmp(data, len, w, h, px); } catch (...) { return fail(r, SWFR_ERR_DEVICE, "out of memory decoding a bitmap"); }
    if (st == XSwfBmpStatus::UnsupportedFormat) return fail(r, SWFR_ERR_NOT_IMPLEMENTED, std::string("UnsupportedXSwfBmpFormatId: ") + std::to_string(len ? int(data[0]) : -1));
    if (st != XSwfBmpStatus::Ok) return fail(r, SWFR_ERR_INVALID, "corrupt image/x-swf-bmp data");
    return swfr_register_bitmap(r, id, w, h, px.data(), size_t(w) * 4);
}

int swfr_register_bitmap(swfr_renderer* r, uint32_t id, uint32_t width, uint32_t height, const uint8_t* rgba, size_t stride) {
    if (!r || !rgba || width == 0 || height == 0 || stride < size_t(width) * 4) return fail(r, SWFR_ERR_INVALID, "bad bitmap");
    if (id > 65535) return fail(r, SWFR_ERR_INVALID, "bitmap id out of range");
    return guarded(r, [&]() {
        r->builder->add_bitmap(id, BitmapInfo{width, height});
        if (!r->has_device) return int(SWFR_OK);
        // putImageData: straight RGBA -> premultiplied ARGB (c * a / 255)
        std::vector<uint32_t> argb(size_t(width) * height);
        for (uint32_t y = 0; y < height; ++y) {
            const uint8_t* row = rgba + size_t(y) * stride;
            for (uint32_t x = 0; x < width; ++x) {
                const uint32_t a = row[4 * x + 3];
                const uint32_t pr = row[4 * x] * a / 255, pg = row[4 * x + 1] * a / 255, pb = row[4 * x + 2] * a / 255;
                argb[size_t(y) * width + x] = (a << 24) | (pr << 16) | (pg << 8) | pb;
            }
        }
        DeviceBitmap& slot = r->bitmaps[id];
        if (slot.pixels) HIP_CHECK(hipFree(slot.pixels));
        slot = DeviceBitmap{};
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&slot.pixels), argb.size() * 4));
        HIP_CHECK(hipMemcpy(slot.pixels, argb.data(), argb.size() * 4, hipMemcpyHostToDevice));
        slot.width = width;
        slot.height = height;
        if (r->bitmap_table.size() <= id) r->bitmap_table.resize(id + 1, DevBitmap{nullptr, 0, 0});
        r->bitmap_table[id] = DevBitmap{slot.pixels, width, height};
        r->bitmap_table_dirty = true;
        return int(SWFR_OK);
    });
}

int swfr_build_frame(swfr_renderer* r, const swfr_stage* stage, const swfr_edge** edges, size_t* n_edges, const swfr_path** paths,
                     size_t* n_paths, const swfr_style** styles, size_t* n_styles) {
    if (!r || !stage) return fail(r, SWFR_ERR_INVALID, "null argument");
    return guarded(r, [&]() {
        r->scene_from_builder = false;                          // (the builder's arrays are about to hold another frame than scene 0)
        r->builder->build(*stage);
        if (edges) *edges = r->builder->edges().data();
        if (n_edges) *n_edges = r->builder->edges().size();
        if (paths) *paths = r->builder->paths().data();
        if (n_paths) *n_paths = r->builder->paths().size();
        if (styles) *styles = r->builder->styles().data();
        if (n_styles) *n_styles = r->builder->styles().size();
        return int(SWFR_OK);
    });
}

int swfr_upload_edges(swfr_renderer* r, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
                      const swfr_style* styles, size_t n_styles) {
    if (!r || (n_edges && !edges) || (n_paths && !paths) || (n_styles && !styles)) return fail(r, SWFR_ERR_INVALID, "null argument");
    return guarded(r, [&]() { return upload(r, 0, true, edges, n_edges, paths, n_paths, styles, n_styles); });
}

int swfr_render_resident(swfr_renderer* r, uint32_t frames) {
    if (!r) return SWFR_ERR_INVALID;
    return guarded(r, [&]() { return render_resident(r, frames); });
}

int swfr_render_edges(swfr_renderer* r, const swfr_edge* edges, size_t n_edges, const swfr_path* paths, size_t n_paths,
                      const swfr_style* styles, size_t n_styles) {
    const int rc = swfr_upload_edges(r, edges, n_edges, paths, n_paths, styles, n_styles);
    if (rc != SWFR_OK) return rc;
    return swfr_render_resident(r, 1);
}

int swfr_render(swfr_renderer* r, const swfr_stage* stage) {
    if (!r || !stage) return fail(r, SWFR_ERR_INVALID, "null argument");
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    return guarded(r, [&]() {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t0 = clk::now();
        r->builder->build(*stage);
        const auto t1 = clk::now();
        const auto& e = r->builder->edges();
        const auto& p = r->builder->paths();
        const auto& s = r->builder->styles();
        validate_scene(r, e.data(), e.size(), p.data(), p.size(), s.data(), s.size());
        // (one frame set: this frame is rendered alone; the builder's edges carry their path index)
        int rc = upload2(r, 0, false, e.data(), e.size(), p.data(), p.size(), s.data(), s.size(), nullptr, true);
        const auto t2 = clk::now();
        if (rc != SWFR_OK) return rc;
        rc = render_resident(r, 1);
        const auto t3 = clk::now();
        swfr_path_timing& pt = r->path_timing;
        pt.build_ms = ms(t0, t1); pt.upload_host_ms = ms(t1, t2); pt.device_ms = r->timing.total_ms; pt.total_ms = ms(t0, t3);
        pt.h2d_bytes = r->scn[0].arena.used; pt.h2d_ms = 0;
        if (r->scn[0].arena.copy_begin) {
            float h = 0;
            if (hipEventElapsedTime(&h, r->scn[0].arena.copy_begin, r->scn[0].arena.copy_end) == hipSuccess) pt.h2d_ms = h;
        }
        return rc;
    });
}

int swfr_last_path_timing(swfr_renderer* r, swfr_path_timing* out) {
    if (!r || !out) return SWFR_ERR_INVALID;
    *out = r->path_timing;
    return SWFR_OK;
}

int swfr_render_sequence(swfr_renderer* r, const swfr_stage* stages, uint32_t n_stages, uint32_t repeat, double* seconds, swfr_path_timing* sum) {
    if (!r || (!stages && n_stages) || !seconds) return fail(r, SWFR_ERR_INVALID, "null argument");
    swfr_path_timing acc{};
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t rep = 0; rep < repeat; ++rep)
        for (uint32_t i = 0; i < n_stages; ++i) {
            const int rc = swfr_render(r, &stages[i]);
            if (rc != SWFR_OK) return rc;
            const swfr_path_timing& pt = r->path_timing;
            acc.build_ms += pt.build_ms; acc.upload_host_ms += pt.upload_host_ms; acc.h2d_ms += pt.h2d_ms; acc.device_ms += pt.device_ms;
            acc.total_ms += pt.total_ms; acc.h2d_bytes += pt.h2d_bytes;
        }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (sum) *sum = acc;
    return SWFR_OK;
}

// render + get_image per frame, the loop of the reference's own tests (HeadlessGfxRenderer::get_image after every render,
// rs/src/headless_renderer.rs:233-244): the mapped read-back of every frame, waited for at once (overlap == 0) or after the NEXT
// frame's swfr_render has returned -- its copy then overlaps that frame's host build (overlap != 0).  *checksum: the sum of one
// pixel per frame, so that every frame's image has been looked at.
int swfr_render_sequence_readback(swfr_renderer* r, const swfr_stage* stages, uint32_t n_stages, uint32_t repeat, int premultiplied, int overlap,
                                  double* seconds, uint64_t* checksum) {
    if (!r || (!stages && n_stages) || !seconds) return fail(r, SWFR_ERR_INVALID, "null argument");
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t sum = 0;
    bool pending = false;
    const uint8_t* data = nullptr;
    const size_t mid = (size_t(r->height / 2) * r->width + r->width / 2) * 4;
    for (uint32_t rep = 0; rep < repeat; ++rep)
        for (uint32_t i = 0; i < n_stages; ++i) {
            int rc = swfr_render(r, &stages[i]);
            if (rc != SWFR_OK) return rc;
            if (pending) { rc = swfr_read_image_wait(r, &data, nullptr); if (rc != SWFR_OK) return rc; sum += data[mid] + data[mid + 3]; pending = false; }
            rc = swfr_read_image_async(r, premultiplied);
            if (rc != SWFR_OK) return rc;
            pending = true;
            if (!overlap) { rc = swfr_read_image_wait(r, &data, nullptr); if (rc != SWFR_OK) return rc; sum += data[mid] + data[mid + 3]; pending = false; }
        }
    if (pending) { const int rc = swfr_read_image_wait(r, &data, nullptr); if (rc != SWFR_OK) return rc; sum += data[mid] + data[mid + 3]; }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (checksum) *checksum = sum;
    return SWFR_OK;
}

int swfr_render_batch(swfr_renderer* r, const swfr_stage* stages, uint32_t n_stages, void* device_dst, size_t frame_stride) {
    if (!r || (!stages && n_stages)) return fail(r, SWFR_ERR_INVALID, "null argument");
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    return guarded(r, [&]() { return render_batch(r, stages, n_stages, device_dst, frame_stride); });
}

int swfr_read_image(swfr_renderer* r, uint8_t* dst, size_t dst_stride, int premultiplied) {
    if (!r || !dst || dst_stride < size_t(r->width) * 4) return fail(r, SWFR_ERR_INVALID, "bad destination");
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle has no image");
    if (!r->fb_valid) return fail(r, SWFR_ERR_INVALID, "nothing rendered yet");
    return guarded(r, [&]() {
        const size_t n = size_t(r->width) * r->height;
        const uint32_t* src = r->fb_cur ? r->fb_cur : r->fs[0].d_fb.ptr;
        if (!premultiplied) {
            r->d_tmp.reserve(n);
            launch_unpremultiply(r->stream, src, r->d_tmp.ptr, n);
            src = r->d_tmp.ptr;
        }
        HIP_CHECK(hipMemcpy2DAsync(dst, dst_stride, src, size_t(r->width) * 4, size_t(r->width) * 4, r->height, hipMemcpyDeviceToHost,
                                   r->stream));
        HIP_CHECK(hipStreamSynchronize(r->stream));
        return int(SWFR_OK);
    });
}

int swfr_read_image_async(swfr_renderer* r, int premultiplied) {
    if (!r) return SWFR_ERR_INVALID;
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle has no image");
    if (!r->fb_valid) return fail(r, SWFR_ERR_INVALID, "nothing rendered yet");
    return guarded(r, [&]() { return start_read(r, premultiplied); });
}

int swfr_read_image_wait(swfr_renderer* r, const uint8_t** data, size_t* stride) {
    if (!r || !data) return fail(r, SWFR_ERR_INVALID, "null argument");
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle has no image");
    return guarded(r, [&]() {
        const int rc = finish_read(r);
        if (rc == SWFR_OK) { *data = r->h_image; if (stride) *stride = size_t(r->width) * 4; }
        return rc;
    });
}

int swfr_render_resident_batched(swfr_renderer* r, uint32_t frames_per_launch, uint32_t launches, float* total_ms) {
    if (!r) return SWFR_ERR_INVALID;
    return guarded(r, [&]() { return render_resident_batched(r, frames_per_launch, launches, total_ms); });
}

int swfr_shape_json(swfr_renderer* r, uint32_t id, int morph, const char** json) {
    if (!r || !json) return fail(r, SWFR_ERR_INVALID, "null argument");
    return guarded(r, [&]() {
        const DecodedShape* s = r->builder->shape(id, morph != 0);
        if (!s) throw StatusError{SWFR_ERR_NOT_FOUND, "unknown shape id"};
        r->json_scratch = shape_to_json(*s);
        *json = r->json_scratch.c_str();
        return int(SWFR_OK);
    });
}

int swfr_last_timing(swfr_renderer* r, swfr_timing* out) {
    if (!r || !out) return SWFR_ERR_INVALID;
    *out = r->timing;
    return SWFR_OK;
}

size_t swfr_band_slab_bytes(const swfr_renderer* r) {
    if (!r) return 0;
    const BandShare bs = band_share(r);
    return size_t(bs.stride == 1 && r->cfg.band_count > 1 ? bs.padded : bs.count) * TILE_H * r->width * 4;    // (contiguous blocks: every rank's slab has the common size)
}

int swfr_copy_band_slab(swfr_renderer* r, void* device_dst) {
    if (!r || !device_dst) return fail(r, SWFR_ERR_INVALID, "null argument");
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle");
    return guarded(r, [&]() {
        const BandShare bs = band_share(r);
        if (bs.stride == 1 && r->cfg.band_count > 1) {         // one contiguous block of rows: a plain copy, zero padding behind it
            const uint32_t* fb = r->fb_cur ? r->fb_cur : r->fs[0].d_fb.ptr;
            const size_t row_bytes = size_t(r->width) * 4, y0 = size_t(bs.first) * TILE_H;
            const size_t rows = y0 >= r->height ? 0 : std::min<size_t>(size_t(bs.count) * TILE_H, r->height - y0);
            if (rows) HIP_CHECK(hipMemcpyAsync(device_dst, reinterpret_cast<const uint8_t*>(fb) + y0 * row_bytes, rows * row_bytes, hipMemcpyDeviceToDevice, r->stream));
            const size_t total = size_t(bs.padded) * TILE_H;
            if (total > rows) HIP_CHECK(hipMemsetAsync(static_cast<uint8_t*>(device_dst) + rows * row_bytes, 0, (total - rows) * row_bytes, r->stream));
            HIP_CHECK(hipStreamSynchronize(r->stream));
            return int(SWFR_OK);
        }
        const uint32_t bc = r->cfg.band_count > 1 ? r->cfg.band_count : 1, bi = r->cfg.band_count > 1 ? r->cfg.band_index : 0;
        launch_pack_band(r->stream, r->fb_cur ? r->fb_cur : r->fs[0].d_fb.ptr, static_cast<uint32_t*>(device_dst), int(r->width), int(r->height), bi, bc, local_tile_rows(r));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(r->stream));
        return int(SWFR_OK);
    });
}

int swfr_set_targets(swfr_renderer* r, void* const* device_targets, uint32_t n_targets) {
    if (!r || n_targets > 4 || (n_targets && !device_targets)) return fail(r, SWFR_ERR_INVALID, "bad targets");
    r->n_targets = n_targets;
    for (uint32_t k = 0; k < n_targets; ++k) r->targets[k] = static_cast<uint32_t*>(device_targets[k]);
    r->scene_ready = false;                               // the frame descriptors carry the framebuffer address: upload again
    return SWFR_OK;
}

int swfr_render_resident_async(swfr_renderer* r, uint32_t* out_set) {
    if (!r) return SWFR_ERR_INVALID;
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    if (!r->scene_ready) return fail(r, SWFR_ERR_INVALID, "no scene uploaded");
    return guarded(r, [&]() {
        uint32_t n_sets = 1;
        while (n_sets < uint32_t(std::min(r->in_flight, 4)) && n_sets < r->sets_ready && r->fs[n_sets].stream && r->fs[n_sets].d_cells.ptr) ++n_sets;
        const uint32_t k = r->async_next++ % n_sets;
        r->async_used |= 1u << k;
        swfr_renderer::FrameSet& F = r->fs[k];
        if (!r->scn[0].slow_verified) { r->scn[0].slow_state = 0; r->scn[0].slow_passes = SLOW_PASSES; }   // (nothing checks an async frame's queues: launch everything unless a blocking frame of this scene has shown what it needs)
        order_frames_behind_pending_read(r);
        launch_frame(r, r->scn[0], F, nullptr, nullptr, n_sets > 1);
        HIP_CHECK(hipGetLastError());
        r->fb_cur = r->n_targets ? r->targets[k % r->n_targets] : F.d_fb.ptr;
        if (out_set) *out_set = k;
        return int(SWFR_OK);
    });
}

int swfr_render_resident_async_to(swfr_renderer* r, void* block_target, uint32_t* out_set) {
    if (!r || !block_target) return SWFR_ERR_INVALID;
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle cannot rasterize");
    if (!r->scene_ready) return fail(r, SWFR_ERR_INVALID, "no scene uploaded");
    const BandShare bs = band_share(r);
    if (bs.stride != 1) return fail(r, SWFR_ERR_INVALID, "a block target needs a handle that owns one contiguous block of tile-rows");
    return guarded(r, [&]() {
        uint32_t n_sets = 1;
        while (n_sets < uint32_t(std::min(r->in_flight, 4)) && n_sets < r->sets_ready && r->fs[n_sets].stream && r->fs[n_sets].d_cells.ptr) ++n_sets;
        const uint32_t k = r->async_next++ % n_sets;
        r->async_used |= 1u << k;
        swfr_renderer::FrameSet& F = r->fs[k];
        if (!r->scn[0].slow_verified) { r->scn[0].slow_state = 0; r->scn[0].slow_passes = SLOW_PASSES; }
        // row y of the frame is row y - (first tile-row) * 16 of the block: the kernels address the frame, so they get the block's
        // address moved up by the rows above it (only the handle's own rows are ever written)
        uint32_t* fb = static_cast<uint32_t*>(block_target) - size_t(bs.first) * TILE_H * r->width;
        order_frames_behind_pending_read(r);
        launch_frame(r, r->scn[0], F, fb, nullptr, true);
        HIP_CHECK(hipGetLastError());
        r->fb_cur = nullptr; r->fb_valid = false;              // (the frame is the caller's: nothing to read back from the handle)
        if (out_set) *out_set = k;
        return int(SWFR_OK);
    });
}

int swfr_render_resident_group_to(swfr_renderer* r, void* const* block_targets, uint32_t n_frames, uint32_t* sets_used) {
    if (!r || !block_targets || n_frames == 0 || n_frames > 64) return SWFR_ERR_INVALID;
    uint32_t used = 0;
    for (uint32_t i = 0; i < n_frames; ++i) {
        uint32_t k = 0;
        const int rc = swfr_render_resident_async_to(r, block_targets[i], &k);
        if (rc != SWFR_OK) return rc;
        used |= 1u << k;
    }
    if (sets_used) *sets_used = used;
    return SWFR_OK;
}

int swfr_get_stats(swfr_renderer* r, swfr_stats* out) {
    if (!r || !out) return SWFR_ERR_INVALID;
    *out = r->stats;
    return SWFR_OK;
}

void* swfr_stream_handle(swfr_renderer* r, uint32_t set) { return (r && r->has_device && set < 4) ? static_cast<void*>(r->fs[set].stream) : nullptr; }

int swfr_wait(swfr_renderer* r) {
    if (!r) return SWFR_ERR_INVALID;
    if (!r->has_device) return fail(r, SWFR_ERR_NO_DEVICE, "host-only handle");
    return guarded(r, [&]() {
        if (!r->h_counters) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&r->h_counters), 4 * COUNTER_WORDS * sizeof(uint32_t), hipHostMallocDefault));
        int rc = SWFR_OK;
        for (uint32_t k = 0; k < 4; ++k) {
            if (!r->fs[k].stream || !r->fs[k].counters) continue;
            if (!(r->async_used >> k & 1u)) { HIP_CHECK(hipStreamSynchronize(r->fs[k].stream)); continue; }
            HIP_CHECK(hipMemcpyAsync(r->h_counters + k * COUNTER_WORDS, r->fs[k].counters, COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, r->fs[k].stream));
            HIP_CHECK(hipStreamSynchronize(r->fs[k].stream));
            if (rc == SWFR_OK) rc = check_counters(r, r->h_counters + k * COUNTER_WORDS);
        }
        r->async_used = 0;
        r->fb_valid = rc == SWFR_OK;
        return rc;
    });
}

long swfr_debug_copy(swfr_renderer* r, int what, void* dst, size_t bytes) {
    if (!r || !dst) return -long(SWFR_ERR_INVALID);
    if (!r->has_device) return -long(SWFR_ERR_NO_DEVICE);
    const swfr_renderer::FrameSet& F = r->fs[0];
    const void* src = what == 0 ? static_cast<const void*>(F.d_rows2.ptr) : static_cast<const void*>(F.d_cells.ptr);
    const size_t have = what == 0 ? F.d_rows2.cap * sizeof(RowInfo2) : F.d_cells.cap * sizeof(Cell);
    const size_t n = std::min(bytes, have);
    if (!src || !n) return 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) return -long(SWFR_ERR_DEVICE);
    return long(n);
}

void* swfr_device_framebuffer(swfr_renderer* r) { return (r && r->has_device) ? (r->fb_cur ? r->fb_cur : r->fs[0].d_fb.ptr) : nullptr; }

#pragma GCC visibility pop
}  // extern "C"
