// frame_builder.cpp -- see frame_builder.hpp.
#include "frame_builder.hpp"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <functional>

#include <algorithm>
#include <cstring>

namespace swfr {
namespace {

inline double lerp(double a, double b, double r) { return b * r + a * (1 - r); }  // canvas-renderer.ts:24-26

struct Css {
    int r, g, b, a;
};
// fromNormalizedColor (ts/src/lib/css-color.ts:11-13) then node-canvas' CSS colour parse: R is
// `& 0xff`-truncated, G/B truncate to ints, alpha is quantised to 8 bits in float.
Css css_color(double r, double g, double b, double a) {
    Css c;
    c.r = int(r * 255.0) & 0xff;
    c.g = std::clamp(int(g * 255.0), 0, 255);
    c.b = std::clamp(int(b * 255.0), 0, 255);
    const float af = float(std::clamp(a, 0.0, 1.0));
    c.a = int(af * 255.0f);
    return c;
}
// cairo_set_source_rgba(r/255, g/255, b/255, a/255): premultiply in doubles, 16-bit shorts, >> 8.
uint32_t premultiplied_pixel(const Css& c) {
    const double r = c.r / 255.0, g = c.g / 255.0, b = c.b / 255.0, a = c.a / 255.0;
    auto sh = [](double v) { return uint32_t(uint16_t(v * 65535.0 + 0.5)) >> 8; };
    return (sh(a) << 24) | (sh(r * a) << 16) | (sh(g * a) << 8) | sh(b * a);
}
Css morph_color(const swfr_rgba8& s, const swfr_rgba8& e, bool morph, double ratio) {
    if (!morph) return css_color(s.r / 255.0, s.g / 255.0, s.b / 255.0, s.a / 255.0);
    return css_color(lerp(s.r / 255.0, e.r / 255.0, ratio), lerp(s.g / 255.0, e.g / 255.0, ratio),
                     lerp(s.b / 255.0, e.b / 255.0, ratio), lerp(s.a / 255.0, e.a / 255.0, ratio));
}

}  // namespace

Affine FrameBuilder::matrix_of(const swfr_matrix& m) {
    // applyMatrix (canvas-renderer.ts:179-188): transform(scaleX, rotateSkew0, rotateSkew1, scaleY, tx, ty)
    Affine a;
    a.xx = m.scale_x / 65536.0;
    a.yx = m.rotate_skew0 / 65536.0;
    a.xy = m.rotate_skew1 / 65536.0;
    a.yy = m.scale_y / 65536.0;
    a.x0 = m.translate_x;
    a.y0 = m.translate_y;
    return a;
}

const DecodedShape* FrameBuilder::shape(uint32_t id, bool morph) const {
    const auto& v = morph ? store()->morphs_ : store()->shapes_;
    return id < v.size() ? &v[id] : nullptr;
}

// ---- worker threads: a fixed set, woken per frame (generation counter), each running one job index of the current task
struct FrameBuilder::Pool {
    std::vector<std::thread> threads;                          // thread i (1-based) runs job i of a task when the task has that many jobs
    std::vector<std::unique_ptr<FrameBuilder>> builders;       // one builder per piece
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::atomic<uint64_t> generation{0};
    std::atomic<int> pending{0};
    std::atomic<bool> quit{false};
    int n_jobs = 0;
    std::function<void(int)> task;
    // Frames arrive a few hundred microseconds apart while an animation is rendered: a worker that has just finished keeps polling for
    // that long before it blocks (a sleeping core takes longer to wake than a piece takes to build), and the caller polls for the
    // pieces' completion likewise.
    static constexpr long SPIN_NS = 400000;
    static long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    void run(int jobs, std::function<void(int)> fn) {           // jobs 1 .. jobs-1 on the pool, job 0 here; returns when all are done
        task = std::move(fn);
        n_jobs = jobs;
        // every worker takes part in every generation (one without a job only counts itself off): run() cannot return -- and the next
        // run() cannot rewrite task / n_jobs -- while a worker is still between noticing the generation and reading them
        pending.store(int(threads.size()), std::memory_order_relaxed);
        generation.fetch_add(1, std::memory_order_release);    // (publishes task / n_jobs / pending)
        { std::lock_guard<std::mutex> lk(m); }                  // a worker between its predicate check and its wait sees the new generation or gets the notification
        cv_go.notify_all();
        task(0);
        const long t0 = now_ns();
        while (pending.load(std::memory_order_acquire) != 0) {
            if (now_ns() - t0 > SPIN_NS) {
                std::unique_lock<std::mutex> lk(m);
                cv_done.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0; });
                break;
            }
            relax();
        }
    }
    void worker(int index) {
        uint64_t seen = 0;
        for (;;) {
            const long t0 = now_ns();
            uint64_t g;
            while ((g = generation.load(std::memory_order_acquire)) == seen && !quit.load(std::memory_order_relaxed)) {
                if (now_ns() - t0 > SPIN_NS) {
                    std::unique_lock<std::mutex> lk(m);
                    cv_go.wait(lk, [&] { return quit.load() || generation.load(std::memory_order_acquire) != seen; });
                } else relax();
            }
            if (quit.load()) return;
            seen = g;
            if (index < n_jobs) task(index);                    // (a task with fewer jobs: nothing to do but to count off)
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                { std::lock_guard<std::mutex> lk(m); }
                cv_done.notify_one();
            }
        }
    }
};

FrameBuilder::FrameBuilder(uint32_t width, uint32_t height, bool even_odd) : w_(width), h_(height), even_odd_(even_odd) {}

FrameBuilder::~FrameBuilder() {
    if (pool_) {
        pool_->quit.store(true);
        { std::lock_guard<std::mutex> lk(pool_->m); }
        pool_->cv_go.notify_all();
        for (auto& t : pool_->threads) t.join();
    }
}

void FrameBuilder::build_range(const swfr_stage& stage, uint32_t lo, uint32_t hi) {
    edges_.clear();
    paths_.clear();
    styles_.clear();
    stack_.clear();
    failed_ = false;
    surface_clear_ = true;  // clearRect over the whole canvas (canvas-renderer.ts:70-71); a later piece learns the truth when joined
    State s;
    s.ctm = Affine::scale(1.0 / 20.0, 1.0 / 20.0);  // twips -> px (:74)
    s.inv = Affine::scale(1.0 / (1.0 / 20.0), 1.0 / (1.0 / 20.0));  // cairo_scale: ctm_inverse *= scale(1/sx, 1/sy)
    stack_.push_back(s);
    try {
        for (uint32_t i = lo; i < hi; ++i) draw(stage.children[i], 0);
    } catch (const StatusError& e) {
        failed_ = true;
        failure_ = e;
    }
}

// this piece's arrays into their place in the joined frame: indices shifted, the "still clear?" lerps settled
void FrameBuilder::copy_piece(FrameBuilder& dst, size_t edge_off, size_t path_off, size_t style_off, bool clear_at_start) const {
    for (size_t i = 0; i < edges_.size(); ++i) {
        swfr_edge e = edges_[i];
        e.reserved += int32_t(path_off);
        dst.edges_[edge_off + i] = e;
    }
    for (size_t i = 0; i < paths_.size(); ++i) {
        swfr_path p = paths_[i];
        p.first_edge += uint32_t(edge_off);
        p.style += uint32_t(style_off);
        if (p.lerp == 2) p.lerp = clear_at_start ? 1 : 0;       // (inside the piece the flag is only ever 2 while nothing was painted)
        dst.paths_[path_off + i] = p;
    }
    if (!styles_.empty()) std::memcpy(&dst.styles_[style_off], styles_.data(), styles_.size() * sizeof(swfr_style));
}

void FrameBuilder::build(const swfr_stage& stage) {
    int want = threads_;
    if (want < 0) {
        const char* env = std::getenv("SWFR_BUILD_THREADS");
        want = env ? std::atoi(env) : int(std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
        threads_ = want;
    }
    // pieces of at least 64 top-level display objects each
    const int pieces = int(std::min<uint32_t>(uint32_t(std::max(want, 1)), stage.n_children / 64));
    if (pieces < 2) {
        build_range(stage, 0, stage.n_children);
        if (failed_) throw failure_;
        for (swfr_path& p : paths_) if (p.lerp == 2) p.lerp = 1;
        return;
    }
    if (!pool_) pool_.reset(new Pool);
    Pool& P = *pool_;
    while (int(P.builders.size()) < pieces) {                 // (piece k is built by builders[k-1] for k >= 1; piece 0 needs a scratch builder, too)
        P.builders.emplace_back(new FrameBuilder(w_, h_, even_odd_));
        P.builders.back()->parent_ = this;
        P.builders.back()->threads_ = 0;
    }
    while (int(P.threads.size()) < pieces - 1) {
        const int index = int(P.threads.size()) + 1;
        P.threads.emplace_back([&P, index] { P.worker(index); });
    }
    auto lo_of = [&](int k) { return uint32_t(uint64_t(stage.n_children) * uint64_t(k) / uint64_t(pieces)); };
    // ---- phase A: every piece into its own builder
    P.run(pieces, [&](int k) { P.builders[k]->build_range(stage, lo_of(k), lo_of(k + 1)); });
    size_t ne = 0, np = 0, ns = 0;
    std::vector<size_t> eo(pieces), po(pieces), so(pieces);
    std::vector<char> clear_at(pieces);
    bool clear = true;
    for (int k = 0; k < pieces; ++k) {
        const FrameBuilder& B = *P.builders[k];
        if (B.failed_) throw B.failure_;                      // the first failing display object in painter's order, as a single walk would report
        eo[k] = ne; po[k] = np; so[k] = ns; clear_at[k] = clear;
        ne += B.edges_.size(); np += B.paths_.size(); ns += B.styles_.size();
        clear = clear && B.surface_clear_;
    }
    edges_.resize(ne); paths_.resize(np); styles_.resize(ns);
    surface_clear_ = clear;
    // ---- phase B: the pieces copied to their places, in parallel
    P.run(pieces, [&](int k) { P.builders[k]->copy_piece(*this, eo[k], po[k], so[k], clear_at[k] != 0); });
}

void FrameBuilder::draw(const swfr_display_object& obj, int depth) {
    if (depth > 256) throw StatusError{SWFR_ERR_INVALID, "display tree too deep"};
    stack_.push_back(stack_.back());  // context.save()
    struct Pop {
        std::vector<State>& s;
        ~Pop() { s.pop_back(); }
    } pop{stack_};
    if (obj.has_matrix) transform(matrix_of(obj.matrix));  // a singular matrix puts Cairo's context in an error state; here it is ignored
    switch (obj.type) {
        case SWFR_OBJECT_CONTAINER:
            for (uint32_t i = 0; i < obj.n_children; ++i) draw(obj.children[i], depth + 1);
            break;
        case SWFR_OBJECT_SHAPE: {
            const DecodedShape* sh = shape(obj.id, false);
            if (!sh) throw StatusError{SWFR_ERR_NOT_FOUND, "unknown shape id"};
            for (const StyledPath& p : sh->paths) draw_path(p, false, 0.0);
            break;
        }
        case SWFR_OBJECT_MORPH_SHAPE: {
            const DecodedShape* sh = shape(obj.id, true);
            if (!sh) throw StatusError{SWFR_ERR_NOT_FOUND, "unknown morph shape id"};
            for (const StyledPath& p : sh->paths) draw_path(p, true, obj.ratio);
            break;
        }
        default:
            throw StatusError{SWFR_ERR_INVALID, "UnexpectedDisplayObjectType"};
    }
}

bool FrameBuilder::transform(const Affine& m) {
    Affine t = m;
    if (!t.invert_cairo()) return false;
    State& st = stack_.back();
    st.ctm = m.then(st.ctm);
    st.inv = st.inv.then(t);
    return true;
}

void FrameBuilder::trace(const StyledPath& p, bool morph, double ratio) {
    const Affine& ctm = stack_.back().ctm;
    auto val = [&](const Coord& c) { return morph ? lerp(c.s, c.e, ratio) : c.s; };
    auto dev = [&](double x, double y) {
        ctm.apply(x, y);
        // 24.8 device coordinates within +-32768 px: beyond that neither Cairo's fixed point nor the kernels' int64 products hold
        if (!(std::fabs(x) <= 32768.0 && std::fabs(y) <= 32768.0)) throw StatusError{SWFR_ERR_INVALID, "geometry outside +-32768 device pixels"};
        return Pt{to_fixed(x), to_fixed(y)};
    };
    path_.clear();
    for (const PathCommand& c : p.commands) {
        switch (c.kind) {
            case PathCommand::MoveTo:
                path_.move_to(dev(val(c.x), val(c.y)));
                break;
            case PathCommand::LineTo:
                path_.line_to(dev(val(c.x), val(c.y)));
                break;
            case PathCommand::CurveTo: {
                // node-canvas quadraticCurveTo: the current point is the quantised device point mapped
                // back to user space; (0,0) stands for "no current point" and selects the control point
                double x = 0, y = 0;
                if (path_.has_current_point()) {
                    Affine inv = ctm;
                    x = from_fixed(path_.current_point().x);
                    y = from_fixed(path_.current_point().y);
                    if (inv.invert()) inv.apply(x, y);
                }
                const double x1 = val(c.cx), y1 = val(c.cy), x2 = val(c.x), y2 = val(c.y);
                if (x == 0 && y == 0) {
                    x = x1;
                    y = y1;
                }
                const double k = 2.0 / 3.0;
                path_.cubic_to(dev(x + k * (x1 - x), y + k * (y1 - y)), dev(x2 + k * (x1 - x2), y2 + k * (y1 - y2)), dev(x2, y2));
                break;
            }
        }
    }
}

void FrameBuilder::draw_path(const StyledPath& p, bool morph, double ratio) {
    if ((!p.has_fill && !p.has_line) || p.commands.empty()) return;
    trace(p, morph, ratio);
    if (p.has_fill) emit_fill(p.fill, morph, ratio);
    if (p.has_line) emit_stroke(p, morph, ratio);
}

bool FrameBuilder::frame_bounds(Pt lo, Pt hi, bool& needs_clip) const {
    if (!(lo.x < hi.x && lo.y < hi.y)) return false;
    int x0 = floor_px(lo.x), y0 = floor_px(lo.y), x1 = ceil_px(hi.x), y1 = ceil_px(hi.y);
    needs_clip = !(x0 >= 0 && y0 >= 0 && x1 <= int(w_) && y1 <= int(h_));
    x0 = std::max(x0, 0);
    y0 = std::max(y0, 0);
    x1 = std::min(x1, int(w_));
    y1 = std::min(y1, int(h_));
    return x0 < x1 && y0 < y1;
}

uint32_t FrameBuilder::push_solid(uint32_t pixel) {
    swfr_style st;
    std::memset(&st, 0, sizeof st);
    st.kind = SWFR_STYLE_SOLID;
    st.pixel = pixel;
    styles_.push_back(st);
    return uint32_t(styles_.size() - 1);
}

void FrameBuilder::emit_polygon(Polygon& poly, bool rectilinear, uint32_t style, bool opaque_solid, int bx0, int by0, int bx1, int by1) {
    const bool lerp_blend = opaque_solid || surface_clear_;
    // Cairo: geometry whose extents miss the operation's rectangle is NOTHING_TO_DO and leaves the surface's "clear" state alone
    // (trim_extents_to_polygon / _to_boxes) -- e.g. a stroke whose approximate extents touch the frame while its outline does
    // not; a rectilinear path that yields no boxes at all counts as drawn (clip_and_composite_boxes)
    if (poly.empty()) {
        if (rectilinear) surface_clear_ = false;
        return;
    }
    swfr_path p;
    std::memset(&p, 0, sizeof p);
    p.first_edge = uint32_t(edges_.size());
    p.fill_rule = even_odd_ ? 1 : 0;
    p.style = style;
    p.lerp = opaque_solid ? 1 : (surface_clear_ ? 2 : 0);      // 2: a lerp only because the surface is still clear (settled by build())
    (void)lerp_blend;
    // converter rectangle: the polygon's extents inside the operation's bounded rectangle (the frame for fills)
    p.x_min = std::max(floor_px(poly.ext_min().x), std::max(bx0, 0));
    p.y_min = std::max(floor_px(poly.ext_min().y), std::max(by0, 0));
    p.x_max = std::min(ceil_px(poly.ext_max().x), std::min(bx1, int(w_)));
    p.y_max = std::min(ceil_px(poly.ext_max().y), std::min(by1, int(h_)));
    if (p.x_min >= p.x_max || p.y_min >= p.y_max) return;
    surface_clear_ = false;
    if (rectilinear) {
        p.kind = SWFR_PATH_BOXES;
        rectilinear_to_boxes(poly, even_odd_, edges_);
    } else {
        p.kind = SWFR_PATH_TOR;
        edges_.insert(edges_.end(), poly.edges().begin(), poly.edges().end());
    }
    p.n_edges = uint32_t(edges_.size()) - p.first_edge;
    if (p.n_edges) {
        for (uint32_t k = p.first_edge; k < p.first_edge + p.n_edges; ++k) edges_[k].reserved = int32_t(paths_.size());   // owning path: the device bins by it
        paths_.push_back(p);
    } else edges_.resize(p.first_edge);
}

void FrameBuilder::emit_fill(const OwnedFill& f, bool morph, double ratio) {
    const swfr_fill_style& s = f.style;
    uint32_t style_index = 0;
    bool opaque_solid = false;
    // context.save(); <source>; fill(); context.restore()  (canvas-renderer.ts:292-336)
    if (s.type == SWFR_FILL_SOLID) {
        const uint32_t px = premultiplied_pixel(morph_color(s.color, s.morph_color, morph, ratio));
        if ((px >> 24) == 0) return;  // Cairo: OVER with a clear source is a no-op
        opaque_solid = (px >> 24) == 0xff;
        style_index = push_solid(px);
    } else if (s.type == SWFR_FILL_BITMAP) {
        auto it = store()->bitmaps_.find(s.bitmap_id);
        if (it == store()->bitmaps_.end()) throw StatusError{SWFR_ERR_NOT_FOUND, "BitmapNotFound: " + std::to_string(s.bitmap_id)};
        swfr_style st;
        std::memset(&st, 0, sizeof st);
        st.kind = SWFR_STYLE_BITMAP;
        // context.save(); transform(fill.matrix): the pattern matrix is the inverse CTM at fill() time
        Affine fm = matrix_of(s.matrix);
        if (!fm.invert_cairo()) return;
        const Affine inv = stack_.back().inv.then(fm);
        const double m[6] = {inv.xx, inv.yx, inv.xy, inv.yy, inv.x0, inv.y0};
        std::memcpy(st.inv, m, sizeof m);
        st.bitmap = s.bitmap_id;
        st.extend = s.repeating ? 1 : 0;
        styles_.push_back(st);
        style_index = uint32_t(styles_.size() - 1);
    } else {
        // Radial == focal with focalPoint 0 (decode-swf-shape.ts:127-133); createRadialGradient(
        // f*16384, 0, 0, 0, 0, 16384) under fill.matrix (canvas-renderer.ts:320-331).  Linear gradients
        // throw NotImplementedFillStyle in the reference (:332-333); here they are the documented
        // extension createLinearGradient(-16384, 0, 16384, 0).
        if (f.stops.size() > SWFR_MAX_STOPS) throw StatusError{SWFR_ERR_CAPACITY, "too many gradient stops"};
        swfr_style st;
        std::memset(&st, 0, sizeof st);
        Affine fm = matrix_of(s.matrix);
        if (!fm.invert_cairo()) return;
        const Affine inv = stack_.back().inv.then(fm);
        const double m[6] = {inv.xx, inv.yx, inv.xy, inv.yy, inv.x0, inv.y0};
        std::memcpy(st.inv, m, sizeof m);
        const double R = 16384.0;
        if (s.type == SWFR_FILL_LINEAR_GRADIENT) {
            st.kind = SWFR_STYLE_LINEAR;
            st.c0x = -R; st.c1x = R;
        } else {
            st.kind = SWFR_STYLE_RADIAL;
            const double focal = s.type == SWFR_FILL_FOCAL_GRADIENT ? s.focal_point / 256.0 : 0.0;
            st.c0x = lerp(0, R, focal);
            st.r0 = 0; st.r1 = R;
        }
        // stops sorted by offset, stable (cairo_pattern_add_color_stop keeps them ordered)
        std::vector<swfr_color_stop> stops = f.stops;
        std::stable_sort(stops.begin(), stops.end(), [](const swfr_color_stop& a, const swfr_color_stop& b) { return a.ratio < b.ratio; });
        st.n_stops = uint32_t(stops.size());
        for (size_t i = 0; i < stops.size(); ++i) {
            const Css c = css_color(stops[i].color.r / 255.0, stops[i].color.g / 255.0, stops[i].color.b / 255.0, stops[i].color.a / 255.0);
            st.stop_offset[i] = float(stops[i].ratio / 255.0);
            st.stop_rgba[i][0] = float(c.r / 255.0);
            st.stop_rgba[i][1] = float(c.g / 255.0);
            st.stop_rgba[i][2] = float(c.b / 255.0);
            st.stop_rgba[i][3] = float(c.a / 255.0);
        }
        // a gradient whose stops are all transparent is a clear source (_cairo_pattern_is_clear -> _gradient_is_clear): OVER with
        // it is a no-op that leaves the surface's "clear" state alone, like the transparent solid above
        if (std::all_of(stops.begin(), stops.end(), [](const swfr_color_stop& c) { return c.color.a == 0; })) return;
        styles_.push_back(st);
        style_index = uint32_t(styles_.size() - 1);
    }
    if (path_.empty_extents()) return;
    // the operation's bounded rectangle: path extents rounded out (the mask), inside the frame, inside the source's extents
    const Pt plo = path_.box_min(), phi = path_.box_max();
    if (!(plo.x < phi.x && plo.y < phi.y)) return;  // nothing to do: surface stays clear
    int x0 = floor_px(plo.x), y0 = floor_px(plo.y), x1 = ceil_px(phi.x), y1 = ceil_px(phi.y);
    const int mask_w = x1 - x0, mask_h = y1 - y0;
    x0 = std::max(x0, 0);
    y0 = std::max(y0, 0);
    x1 = std::min(x1, int(w_));
    y1 = std::min(y1, int(h_));
    if (s.type == SWFR_FILL_BITMAP && !s.repeating) {
        // OVER is bounded by its source (_cairo_pattern_get_extents of an EXTEND_NONE surface pattern): the bitmap's rectangle in
        // device space; an axis the filter magnifies is padded by half a source pixel and rounded to the nearest pixel edge,
        // the others are rounded out
        const swfr_style& st = styles_[style_index];
        Affine pm;
        pm.xx = st.inv[0]; pm.yx = st.inv[1]; pm.xy = st.inv[2]; pm.yy = st.inv[3]; pm.x0 = st.inv[4]; pm.y0 = st.inv[5];
        const BitmapInfo& bi = store()->bitmaps_.find(s.bitmap_id)->second;
        double sx0 = 0, sy0 = 0, sx1 = bi.width, sy1 = bi.height;
        bool round_x = false, round_y = false;
        if (std::hypot(pm.xx, pm.yx) < 1.0) { sx0 -= 0.5; sx1 += 0.5; round_x = true; }
        if (std::hypot(pm.xy, pm.yy) < 1.0) { sy0 -= 0.5; sy1 += 0.5; round_y = true; }
        Affine im = pm;
        if (im.invert_cairo()) {
            double bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
            for (int k = 0; k < 4; ++k) {
                double x = (k & 1) ? sx1 : sx0, y = (k & 2) ? sy1 : sy0;
                im.apply(x, y);
                if (!k || x < bx0) bx0 = x;
                if (!k || x > bx1) bx1 = x;
                if (!k || y < by0) by0 = y;
                if (!k || y > by1) by1 = y;
            }
            if (!round_x) { bx0 -= 0.5; bx1 += 0.5; }
            if (!round_y) { by0 -= 0.5; by1 += 0.5; }
            bx0 = std::floor(bx0 + 0.5); by0 = std::floor(by0 + 0.5); bx1 = std::floor(bx1 + 0.5); by1 = std::floor(by1 + 0.5);
            if (x0 < bx0) x0 = int(bx0);
            if (y0 < by0) y0 = int(by0);
            if (x1 > bx1) x1 = int(bx1);
            if (y1 > by1) y1 = int(by1);
        }
    }
    if (x0 >= x1 || y0 >= y1) return;                   // nothing to do: surface stays clear
    const bool needs_clip = mask_w > x1 - x0 || mask_h > y1 - y0;
    Polygon& poly = poly_;                            // (reused: its edge vector keeps its capacity from shape to shape)
    const Pt lo{fixed_t(x0) * 256, fixed_t(y0) * 256}, hi{fixed_t(x1) * 256, fixed_t(y1) * 256};  // the limits are the bounded rectangle
    poly.reset(needs_clip, lo, hi);
    if (needs_clip) fill_to_polygon_clipped(path_, poly, lo, hi); else fill_to_polygon(path_, poly);
    emit_polygon(poly, path_.fill_is_rectilinear(), style_index, opaque_solid, x0, y0, x1, y1);
}

void FrameBuilder::emit_stroke(const StyledPath& p, bool morph, double ratio) {
    if (p.fill.style.type != SWFR_FILL_SOLID) throw StatusError{SWFR_ERR_NOT_IMPLEMENTED, "NotImplementedLineStyle"};
    State& st = stack_.back();
    const double width = morph ? lerp(p.width, p.morph_width, ratio) : double(p.width);
    if (width > 0) st.line_width = width;  // node-canvas ignores non-positive widths
    const uint32_t px = premultiplied_pixel(morph_color(p.fill.style.color, p.fill.style.morph_color, morph, ratio));
    if (morph) st.cap = st.join = 1;  // lineCap = lineJoin = "round" (canvas-renderer.ts:263-264)
    if ((px >> 24) == 0) return;
    if (path_.empty_extents()) return;
    // _cairo_compositor_stroke: a pen that degenerates to one vertex (line width <= tolerance/2 = 0.05 device pixels under the CTM)
    // paints nothing, whatever the stroker; the surface stays untouched
    if (stroke_pen_vertices(st.line_width, st.ctm) <= 1) return;
    // approximate stroke extents: path box grown by the style's maximum distance from the path
    double expansion = 0.5;
    if (st.cap == 2) expansion = M_SQRT1_2;
    if (st.join == 0 && !path_.stroke_is_rectilinear() && expansion < M_SQRT2 * 10.0) expansion = M_SQRT2 * 10.0;
    expansion *= st.line_width;
    const Affine& c = st.ctm;
    const double eps = 1.0 / 256.0, det = c.det();
    const bool unity = std::fabs(det * det - 1.0) < eps &&
                       ((std::fabs(c.xy) < eps && std::fabs(c.yx) < eps) || (std::fabs(c.xx) < eps && std::fabs(c.yy) < eps));
    const double gx = unity ? expansion : expansion * std::hypot(c.xx, c.xy);
    const double gy = unity ? expansion : expansion * std::hypot(c.yy, c.yx);
    Pt lo = path_.box_min(), hi = path_.box_max();
    lo.x -= to_fixed(gx); lo.y -= to_fixed(gy);
    hi.x += to_fixed(gx); hi.y += to_fixed(gy);
    bool needs_clip = false;
    if (!frame_bounds(lo, hi, needs_clip)) return;
    // the operation is bounded by these extents (rounded out, inside the frame): what the stroker produces beyond them is not painted
    const int bx0 = std::max(floor_px(lo.x), 0), by0 = std::max(floor_px(lo.y), 0);
    const int bx1 = std::min(ceil_px(hi.x), int(w_)), by1 = std::min(ceil_px(hi.y), int(h_));
    const Pt frame_lo{0, 0}, frame_hi{fixed_t(w_) * 256, fixed_t(h_) * 256};
    StrokeParams sp;
    sp.line_width = st.line_width;
    sp.cap = st.cap;
    sp.join = st.join;
    const bool opaque = (px >> 24) == 0xff;
    // strokes are filled non-zero regardless of the configured fill rule
    const bool saved = even_odd_;
    even_odd_ = false;
    Polygon& poly = poly_;
    bool done = false;
    if (path_.stroke_is_rectilinear()) {
        // Cairo's box stroker when it accepts the style: the union of one box per segment, painted like a rectilinear fill.  Its
        // boxes are NOT clipped against the frame first: a stroke whose boxes all lie outside is then "boxes that miss the operation's
        // rectangle" (nothing drawn, the surface keeps its clear state) and not "no boxes at all" (which counts as drawn) -- found by
        // the soak (mixed 7100/2196: the next translucent fill is then still composited with the SOURCE rule)
        poly.reset(false, frame_lo, frame_hi);
        if (stroke_rectilinear_to_boxes(path_, sp, st.ctm, poly)) {
            emit_polygon(poly, true, push_solid(px), opaque, bx0, by0, bx1, by1);
            done = true;
        }
    }
    if (!done) {
        poly.reset(needs_clip, frame_lo, frame_hi);
        if (needs_clip) {
            sp.has_bounds = true;
            sp.bounds_lo = Pt{frame_lo.x - to_fixed(gx), frame_lo.y - to_fixed(gy)};
            sp.bounds_hi = Pt{frame_hi.x + to_fixed(gx), frame_hi.y + to_fixed(gy)};
        }
        if (!stroke_to_polygon(path_, sp, st.ctm, poly)) {
            even_odd_ = saved;
            return;                                            // singular CTM: nothing can be stroked
        }
        emit_polygon(poly, false, push_solid(px), opaque, bx0, by0, bx1, by1);
    }
    even_odd_ = saved;
}

}  // namespace swfr
