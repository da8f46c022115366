// geometry.hpp -- host-side geometry of the hot path: user space -> 24.8 device space -> polygons.
//
// Everything the reference's Canvas2D backend does to a path before scan conversion happens here
// on the CPU (north star: "decodes shape records to an edge list on CPU"): CTM application and
// 24.8 quantisation, collinear merge, cubic flattening, stroke expansion (pen, all joins and caps,
// closed sub-paths, curves, rectilinear strokes), clipping of edges against the frame, and
// tessellation of rectilinear paths into boxes.
// The arithmetic follows SURVEY.md Appendix A.1-A.4, A.5b, A.6, A.8 (Cairo 1.16 as used through
// ts/src/lib/renderers/canvas-renderer.ts:207-350); results feed swfr_edge/swfr_path arrays.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/swfr.h"

namespace swfr {

using fixed_t = int32_t;  // 24.8

struct Pt {
    fixed_t x = 0, y = 0;
    bool operator==(const Pt& o) const { return x == o.x && y == o.y; }
    bool operator!=(const Pt& o) const { return !(*this == o); }
};

inline fixed_t to_fixed(double v) { return static_cast<fixed_t>(std::nearbyint(v * 256.0)); }
inline double from_fixed(fixed_t v) { return static_cast<double>(v) / 256.0; }
inline int floor_px(fixed_t v) { return v >> 8; }
inline int ceil_px(fixed_t v) { return (v + 255) >> 8; }

// Affine transform in Cairo's convention: x' = xx*x + xy*y + x0, y' = yx*x + yy*y + y0.
struct Affine {
    double xx = 1, yx = 0, xy = 0, yy = 1, x0 = 0, y0 = 0;
    static Affine scale(double sx, double sy) { return {sx, 0, 0, sy, 0, 0}; }
    // "this" is applied first, then `after` (cairo_matrix_multiply(result, this, after)).
    Affine then(const Affine& after) const;
    void apply(double& x, double& y) const;
    void apply_distance(double& dx, double& dy) const;
    double det() const { return xx * yy - yx * xy; }
    bool invert();
    // cairo_matrix_invert operation by operation (scale-only shortcut; otherwise adjoint times 1/det): pattern matrices are
    // rounded to 16.16 for pixman afterwards, so they are derived exactly as Cairo derives them
    bool invert_cairo();
    bool is_identity() const { return xx == 1 && yx == 0 && xy == 0 && yy == 1 && x0 == 0 && y0 == 0; }
};

// A path in device space, built the way cairo_path_fixed_t is: consecutive collinear line
// segments merge, degenerate segments vanish, and the rectilinear flags are tracked.
class DevicePath {
public:
    enum Verb : uint8_t { Move, Line, Cubic, Close };

    void clear();
    void move_to(Pt p);
    void line_to(Pt p);
    void cubic_to(Pt c1, Pt c2, Pt end);
    bool has_current_point() const { return has_cur_; }
    Pt current_point() const { return cur_; }

    const std::vector<Verb>& verbs() const { return verbs_; }
    const std::vector<Pt>& points() const { return pts_; }
    bool empty_extents() const { return !has_box_; }
    Pt box_min() const { return lo_; }
    Pt box_max() const { return hi_; }
    bool fill_is_rectilinear() const;
    bool stroke_is_rectilinear() const { return stroke_rect_; }

private:
    void begin_subpath();
    void flush_move();
    void grow(Pt p);
    void pop_line();
    std::vector<Verb> verbs_;
    std::vector<Pt> pts_;
    Pt cur_{}, start_{}, lo_{}, hi_{};
    bool has_cur_ = false, pending_move_ = true, has_box_ = false;
    bool fill_rect_ = true, stroke_rect_ = true;
};

// Polygon = bag of oriented edges, optionally clipped against a limit box (the frame).
class Polygon {
public:
    void reset(bool clip, Pt lim_lo, Pt lim_hi);
    void add_segment(Pt a, Pt b, int dir);  // drops horizontals, orients, clips
    const std::vector<swfr_edge>& edges() const { return edges_; }
    bool empty() const { return edges_.empty(); }
    Pt ext_min() const { return emin_; }
    Pt ext_max() const { return emax_; }

private:
    void push(Pt p1, Pt p2, fixed_t top, fixed_t bottom, int dir);
    void push_clipped(Pt p1, Pt p2, fixed_t top, fixed_t bottom, int dir);
    std::vector<swfr_edge> edges_;
    bool clip_ = false;
    Pt llo_{}, lhi_{};
    Pt emin_{INT32_MAX, INT32_MAX}, emax_{INT32_MIN, INT32_MIN};
};

struct StrokeParams {
    double line_width = 1.0;
    double miter_limit = 10.0;
    int cap = 0;   // 0 butt, 1 round, 2 square (cairo_line_cap_t)
    int join = 0;  // 0 miter, 1 round, 2 bevel (cairo_line_join_t)
    // Stroker bounds (Cairo sets them when the stroke may leave the surface: the limits grown by the style's reach):
    // fans whose centre lies outside are skipped, curves that cannot touch the box become chords.
    bool has_bounds = false;
    Pt bounds_lo{}, bounds_hi{};
};

constexpr double kTolerance = 0.1;  // Cairo's default flattening tolerance (device pixels)

// Fill: implicit close of every sub-path, curves flattened.
void fill_to_polygon(const DevicePath& path, Polygon& out);
// Same, for a polygon that is being clipped to [lo,hi]: curves that cannot touch the box become chords.
void fill_to_polygon_clipped(const DevicePath& path, Polygon& out, Pt lo, Pt hi);
// Stroke -> outline polygon (filled non-zero): Cairo 1.16's polygon stroker with its pen (round joins and caps, fans inside
// curves), miter / bevel joins, butt / square caps, closed sub-paths and tangent-following curve decomposition.
// Returns false only for a singular CTM.
bool stroke_to_polygon(const DevicePath& path, const StrokeParams& sp, const Affine& ctm, Polygon& out);
// Vertices of the stroking pen for this width under the CTM (cairo-pen.c); <= 1 means the stroke paints nothing at all.
int stroke_pen_vertices(double line_width, const Affine& ctm);
// Cairo's rectilinear stroker (axis-aligned paths, miter joins, butt/square caps, scale-only CTM): one box per segment as
// left/right edges in `out`; the union (non-zero) is the stroke.  Returns false when Cairo would decline (the polygon
// stroker is used then).
bool stroke_rectilinear_to_boxes(const DevicePath& path, const StrokeParams& sp, const Affine& ctm, Polygon& out);
// Rectilinear fill region -> disjoint boxes (x1,y1)-(x2,y2), stored in swfr_edge records.
void rectilinear_to_boxes(const Polygon& poly, bool even_odd, std::vector<swfr_edge>& boxes);

}  // namespace swfr
