// frame_builder.hpp -- host scene walk: Stage -> painter-ordered edge list + path/style tables.
//
// Mirrors CanvasRenderer (ts/src/lib/renderers/canvas-renderer.ts:61-350): reset CTM, clear,
// scale(1/20), matrix stack over containers/shapes/morph shapes, per path beginPath + commands +
// fill() / stroke().  Instead of calling a Canvas it emits what the GPU scan converter consumes.
#pragma once

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/swfr.h"
#include "geometry.hpp"
#include "shape_decoder.hpp"

namespace swfr {

struct BitmapInfo {
    uint32_t width = 0, height = 0;
};

struct StatusError {
    int code;
    std::string message;
};

// Frames with many top-level display objects are built by several threads: the children are cut into contiguous ranges, every range
// is walked by a worker builder of its own (same code, its own output arrays), and the pieces are joined in painter's order.  The one
// thing a display object's output depends on besides itself -- whether the surface is still clear, which turns the first translucent
// paint into a SOURCE-rule lerp -- is marked in the paths (`lerp` = 2) and settled when the pieces are joined.
class FrameBuilder {
public:
    FrameBuilder(uint32_t width, uint32_t height, bool even_odd);
    ~FrameBuilder();
    FrameBuilder(const FrameBuilder&) = delete;
    FrameBuilder& operator=(const FrameBuilder&) = delete;
    void set_threads(int n) { threads_ = n; }          // 0 / 1: single thread; default: SWFR_BUILD_THREADS or min(8, cores)

    uint32_t add_shape(DecodedShape s) { shapes_.push_back(std::move(s)); return uint32_t(shapes_.size() - 1); }
    uint32_t add_morph_shape(DecodedShape s) { morphs_.push_back(std::move(s)); return uint32_t(morphs_.size() - 1); }
    void add_bitmap(uint32_t id, BitmapInfo info) { bitmaps_[id] = info; }
    const DecodedShape* shape(uint32_t id, bool morph) const;

    // Throws StatusError.  Results stay valid until the next build().
    void build(const swfr_stage& stage);
    const std::vector<swfr_edge>& edges() const { return edges_; }
    const std::vector<swfr_path>& paths() const { return paths_; }
    const std::vector<swfr_style>& styles() const { return styles_; }

private:
    struct State {
        Affine ctm;
        Affine inv;               // Cairo keeps the inverse beside the CTM and updates it factor by factor (_cairo_gstate_transform)
        double line_width = 1.0;  // node-canvas creates its context with line width 1
        int cap = 0, join = 0;
    };
    void draw(const swfr_display_object& obj, int depth);
    void draw_path(const StyledPath& p, bool morph, double ratio);
    void trace(const StyledPath& p, bool morph, double ratio);
    void emit_fill(const OwnedFill& f, bool morph, double ratio);
    void emit_stroke(const StyledPath& p, bool morph, double ratio);
    void emit_polygon(Polygon& poly, bool rectilinear, uint32_t style, bool opaque_solid, int bx0 = 0, int by0 = 0, int bx1 = INT32_MAX, int by1 = INT32_MAX);
    uint32_t push_solid(uint32_t pixel);
    bool frame_bounds(Pt lo, Pt hi, bool& needs_clip) const;
    bool transform(const Affine& m);  // context.transform(m); false: singular
    static Affine matrix_of(const swfr_matrix& m);

    // ---- multi-threaded build
    struct Pool;                                         // the worker threads (created on first use)
    void build_range(const swfr_stage& stage, uint32_t lo, uint32_t hi);    // worker: children [lo, hi) into this builder's arrays
    void copy_piece(FrameBuilder& dst, size_t edge_off, size_t path_off, size_t style_off, bool clear_at_start) const;
    const FrameBuilder* store() const { return parent_ ? parent_ : this; }  // where shapes and bitmaps are registered
    const FrameBuilder* parent_ = nullptr;
    std::unique_ptr<Pool> pool_;
    int threads_ = -1;
    bool failed_ = false;
    StatusError failure_{0, ""};

    uint32_t w_, h_;
    bool even_odd_;
    std::vector<DecodedShape> shapes_, morphs_;
    std::map<uint32_t, BitmapInfo> bitmaps_;
    std::vector<State> stack_;
    DevicePath path_;
    Polygon poly_;
    bool surface_clear_ = true;
    std::vector<swfr_edge> edges_;
    std::vector<swfr_path> paths_;
    std::vector<swfr_style> styles_;
};

}  // namespace swfr
