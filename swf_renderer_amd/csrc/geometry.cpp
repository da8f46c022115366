// geometry.cpp -- see geometry.hpp.  Arithmetic per SURVEY.md Appendix A (Cairo 1.16 image backend).
#include "geometry.hpp"

#include <algorithm>
#include <functional>

namespace swfr {

// ---------------------------------------------------------------------------------------------
// Affine
// ---------------------------------------------------------------------------------------------
Affine Affine::then(const Affine& b) const {
    Affine r;
    r.xx = xx * b.xx + yx * b.xy;
    r.yx = xx * b.yx + yx * b.yy;
    r.xy = xy * b.xx + yy * b.xy;
    r.yy = xy * b.yx + yy * b.yy;
    r.x0 = x0 * b.xx + y0 * b.xy + b.x0;
    r.y0 = x0 * b.yx + y0 * b.yy + b.y0;
    return r;
}
void Affine::apply(double& x, double& y) const {
    double nx = xx * x + xy * y + x0, ny = yx * x + yy * y + y0;
    x = nx;
    y = ny;
}
void Affine::apply_distance(double& dx, double& dy) const {
    double nx = xx * dx + xy * dy, ny = yx * dx + yy * dy;
    dx = nx;
    dy = ny;
}
bool Affine::invert() {
    double d = det();
    if (d == 0 || !std::isfinite(d)) return false;
    Affine r;
    r.xx = yy / d;
    r.yx = -yx / d;
    r.xy = -xy / d;
    r.yy = xx / d;
    r.x0 = (xy * y0 - yy * x0) / d;
    r.y0 = (yx * x0 - xx * y0) / d;
    *this = r;
    return true;
}

bool Affine::invert_cairo() {
    if (xy == 0. && yx == 0.) {
        x0 = -x0;
        y0 = -y0;
        if (xx != 1.) {
            if (xx == 0.) return false;
            xx = 1. / xx;
            x0 *= xx;
        }
        if (yy != 1.) {
            if (yy == 0.) return false;
            yy = 1. / yy;
            y0 *= yy;
        }
        return true;
    }
    const double d = det();
    if (d == 0 || !std::isfinite(d)) return false;
    const double a = xx, b = yx, c = xy, e = yy, tx = x0, ty = y0, k = 1 / d;
    xx = e * k;
    yx = -b * k;
    xy = -c * k;
    yy = a * k;
    x0 = (c * ty - e * tx) * k;
    y0 = (b * tx - a * ty) * k;
    return true;
}

// ---------------------------------------------------------------------------------------------
// DevicePath (A.2)
// ---------------------------------------------------------------------------------------------
void DevicePath::clear() {
    verbs_.clear();
    pts_.clear();
    cur_ = start_ = Pt{};
    has_cur_ = false;
    pending_move_ = true;
    has_box_ = false;
    fill_rect_ = stroke_rect_ = true;
}
void DevicePath::grow(Pt p) {
    if (!has_box_) {
        lo_ = hi_ = p;
        has_box_ = true;
        return;
    }
    lo_.x = std::min(lo_.x, p.x);
    lo_.y = std::min(lo_.y, p.y);
    hi_.x = std::max(hi_.x, p.x);
    hi_.y = std::max(hi_.y, p.y);
}
void DevicePath::begin_subpath() {
    if (!pending_move_) {
        // the sub-path just ended is implicitly closed when filled
        if (fill_rect_) fill_rect_ = cur_.x == start_.x || cur_.y == start_.y;
        pending_move_ = true;
    }
    has_cur_ = false;
}
void DevicePath::move_to(Pt p) {
    begin_subpath();
    has_cur_ = true;
    cur_ = start_ = p;
}
void DevicePath::flush_move() {
    if (!pending_move_) return;
    pending_move_ = false;
    grow(cur_);
    start_ = cur_;
    verbs_.push_back(Move);
    pts_.push_back(cur_);
}
void DevicePath::pop_line() {
    verbs_.pop_back();
    pts_.pop_back();
}
void DevicePath::line_to(Pt p) {
    if (!has_cur_) {
        move_to(p);
        return;
    }
    flush_move();
    const bool after_move = verbs_.back() == Move;
    if (!after_move && p == cur_) return;  // degenerate segments survive only right after a move
    if (verbs_.back() == Line) {
        const Pt prev = pts_[pts_.size() - 2];
        if (prev == cur_) {
            pop_line();  // previous segment was degenerate: replace it
        } else {
            const int64_t ax = int64_t(cur_.x) - prev.x, ay = int64_t(cur_.y) - prev.y;
            const int64_t bx = int64_t(p.x) - cur_.x, by = int64_t(p.y) - cur_.y;
            const bool same_slope = ay * bx == by * ax;
            const bool backwards = ((ax * bx) >> 8) + ((ay * by) >> 8) < 0;
            if (same_slope && !backwards) pop_line();  // collinear continuation: extend instead
        }
    }
    if (stroke_rect_) {
        stroke_rect_ = cur_.x == p.x || cur_.y == p.y;
        fill_rect_ = fill_rect_ && stroke_rect_;
    }
    cur_ = p;
    grow(p);
    verbs_.push_back(Line);
    pts_.push_back(p);
}
void DevicePath::cubic_to(Pt c1, Pt c2, Pt end) {
    if (has_cur_ && cur_ == end && c1 == end && c2 == end) {
        line_to(end);
        return;
    }
    if (!has_cur_) move_to(c1);
    flush_move();
    if (verbs_.back() == Line && pts_[pts_.size() - 2] == cur_) pop_line();
    grow(c1);
    grow(c2);
    grow(end);  // control box: superset of Cairo's tight curve box (only gates frame clipping)
    cur_ = end;
    fill_rect_ = stroke_rect_ = false;
    verbs_.push_back(Cubic);
    pts_.push_back(c1);
    pts_.push_back(c2);
    pts_.push_back(end);
}
bool DevicePath::fill_is_rectilinear() const {
    if (!fill_rect_) return false;
    if (!has_cur_ || pending_move_) return true;
    return cur_.x == start_.x || cur_.y == start_.y;
}

// ---------------------------------------------------------------------------------------------
// Polygon + frame clipping (A.5b)
// ---------------------------------------------------------------------------------------------
namespace {
// Cairo's *_mul_div_floor helpers are plain truncating divisions.
fixed_t line_x_at_y(Pt p1, Pt p2, fixed_t y) {
    if (y == p1.y) return p1.x;
    if (y == p2.y) return p2.x;
    const int64_t dy = int64_t(p2.y) - p1.y;
    if (dy == 0) return p1.x;
    return p1.x + fixed_t((int64_t(y) - p1.y) * (int64_t(p2.x) - p1.x) / dy);
}
fixed_t line_y_at_x(Pt p1, Pt p2, fixed_t x) {
    if (x == p1.x) return p1.y;
    if (x == p2.x) return p2.y;
    const int64_t dx = int64_t(p2.x) - p1.x;
    if (dx == 0) return p1.y;
    return p1.y + fixed_t((int64_t(x) - p1.x) * (int64_t(p2.y) - p1.y) / dx);
}
}  // namespace

void Polygon::reset(bool clip, Pt lim_lo, Pt lim_hi) {
    edges_.clear();
    clip_ = clip;
    llo_ = lim_lo;
    lhi_ = lim_hi;
    emin_ = Pt{INT32_MAX, INT32_MAX};
    emax_ = Pt{INT32_MIN, INT32_MIN};
}
void Polygon::push(Pt p1, Pt p2, fixed_t top, fixed_t bottom, int dir) {
    swfr_edge e{p1.x, p1.y, p2.x, p2.y, top, bottom, dir, 0};
    edges_.push_back(e);
    emin_.y = std::min(emin_.y, top);
    emax_.y = std::max(emax_.y, bottom);
    auto widen = [&](Pt p, fixed_t at, fixed_t own) {
        if (p.x < emin_.x || p.x > emax_.x) {
            fixed_t x = p.x;
            if (at != own) x = line_x_at_y(p1, p2, at);
            emin_.x = std::min(emin_.x, x);
            emax_.x = std::max(emax_.x, x);
        }
    };
    widen(p1, top, p1.y);
    widen(p2, bottom, p2.y);
}
void Polygon::push_clipped(Pt p1, Pt p2, fixed_t top, fixed_t bottom, int dir) {
    if (top >= lhi_.y || bottom <= llo_.y) return;
    const Pt left_top = llo_, left_bot{llo_.x, lhi_.y}, right_top{lhi_.x, llo_.y}, right_bot = lhi_;
    fixed_t ty = std::max(top, llo_.y), by = std::min(bottom, lhi_.y);
    const fixed_t xl = std::min(p1.x, p2.x), xr = std::max(p1.x, p2.x);
    if (llo_.x <= xl && xr <= lhi_.x) return push(p1, p2, ty, by, dir);
    if (xr <= llo_.x) return push(left_top, left_bot, ty, by, dir);       // all left: rides the left side
    if (lhi_.x <= xl) return push(right_top, right_bot, ty, by, dir);     // all right
    const bool down_right = (p1.x <= p2.x) == (p1.y <= p2.y);
    auto cross_left = [&](int adjust) {
        fixed_t y = line_y_at_x(p1, p2, llo_.x);
        if (line_x_at_y(p1, p2, y) < llo_.x) y += adjust;
        return y;
    };
    auto cross_right = [&](int adjust) {
        fixed_t y = line_y_at_x(p1, p2, lhi_.x);
        if (line_x_at_y(p1, p2, y) > lhi_.x) y += adjust;
        return y;
    };
    if (down_right) {
        fixed_t ly = xl >= llo_.x ? ty : cross_left(+1);
        ly = std::min(ly, by);
        if (ty < ly) {
            push(left_top, left_bot, ty, ly, dir);
            ty = ly;
        }
        fixed_t ry = xr <= lhi_.x ? by : cross_right(-1);
        ry = std::max(ry, ty);
        if (by > ry) {
            push(right_top, right_bot, ry, by, dir);
            by = ry;
        }
    } else {
        fixed_t ry = xr <= lhi_.x ? ty : cross_right(+1);
        ry = std::min(ry, by);
        if (ty < ry) {
            push(right_top, right_bot, ty, ry, dir);
            ty = ry;
        }
        fixed_t ly = xl >= llo_.x ? by : cross_left(-1);
        ly = std::max(ly, ty);
        if (by > ly) {
            push(left_top, left_bot, ly, by, dir);
            by = ly;
        }
    }
    if (ty != by) push(p1, p2, ty, by, dir);
}
void Polygon::add_segment(Pt a, Pt b, int dir) {
    if (a.y == b.y) return;
    if (a.y > b.y) {
        std::swap(a, b);
        dir = -dir;
    }
    if (clip_) {
        if (b.y <= llo_.y || a.y >= lhi_.y) return;
        push_clipped(a, b, a.y, b.y, dir);
    } else {
        push(a, b, a.y, b.y, dir);
    }
}

// ---------------------------------------------------------------------------------------------
// Cubic flattening (A.3)
// ---------------------------------------------------------------------------------------------
namespace {
struct Knots {
    Pt a, b, c, d;
};
inline Pt midpoint(Pt p, Pt q) { return Pt{p.x + ((q.x - p.x) >> 1), p.y + ((q.y - p.y) >> 1)}; }

double flatness2(const Knots& k) {
    double bx = from_fixed(k.b.x - k.a.x), by = from_fixed(k.b.y - k.a.y);
    double cx = from_fixed(k.c.x - k.a.x), cy = from_fixed(k.c.y - k.a.y);
    if (k.a != k.d) {
        const double dx = from_fixed(k.d.x - k.a.x), dy = from_fixed(k.d.y - k.a.y);
        const double v = dx * dx + dy * dy;
        auto project = [&](double& px, double& py) {
            const double u = px * dx + py * dy;
            if (u <= 0) return;
            if (u >= v) {
                px -= dx;
                py -= dy;
            } else {
                px -= u / v * dx;
                py -= u / v * dy;
            }
        };
        project(bx, by);
        project(cx, cy);
    }
    return std::max(bx * bx + by * by, cx * cx + cy * cy);
}

class Flattener {
public:
    Flattener(Pt start, std::function<void(Pt)> sink) : last_(start), sink_(std::move(sink)) {}
    // false: the cubic is the straight line a->d
    bool run(Pt a, Pt b, Pt c, Pt d) {
        if (a == b && c == d) return false;
        Knots k{a, b, c, d};
        split(k);
        sink_(d);
        return true;
    }

private:
    void emit(Pt p) {
        if (p == last_) return;
        last_ = p;
        sink_(p);
    }
    void split(Knots k) {
        if (flatness2(k) < kTolerance * kTolerance) {
            emit(k.a);
            return;
        }
        const Pt ab = midpoint(k.a, k.b), bc = midpoint(k.b, k.c), cd = midpoint(k.c, k.d);
        const Pt abbc = midpoint(ab, bc), bccd = midpoint(bc, cd), mid = midpoint(abbc, bccd);
        split(Knots{k.a, ab, abbc, mid});
        split(Knots{mid, bccd, cd, k.d});
    }
    Pt last_;
    std::function<void(Pt)> sink_;
};

bool cubic_touches_box(Pt a, Pt b, Pt c, Pt d, Pt lo, Pt hi) {
    auto inside = [&](Pt p) { return p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y; };
    if (inside(a) || inside(b) || inside(c) || inside(d)) return true;
    const fixed_t x0 = std::min({a.x, b.x, c.x, d.x}), x1 = std::max({a.x, b.x, c.x, d.x});
    const fixed_t y0 = std::min({a.y, b.y, c.y, d.y}), y1 = std::max({a.y, b.y, c.y, d.y});
    return !(x1 <= lo.x || x0 >= hi.x || y1 <= lo.y || y0 >= hi.y);
}
}  // namespace

// A polygon being clipped exposes its limits through this small view so the filler can skip
// flattening of curves that cannot touch the frame (Cairo does the same).
struct ClipView {
    bool on;
    Pt lo, hi;
};

static void fill_walk(const DevicePath& path, Polygon& out, const ClipView& cv) {
    Pt cur{}, start{};
    const auto& pts = path.points();
    size_t ip = 0;
    auto seg = [&](Pt to) {
        out.add_segment(cur, to, +1);
        cur = to;
    };
    for (DevicePath::Verb v : path.verbs()) {
        switch (v) {
            case DevicePath::Move:
                seg(start);
                cur = start = pts[ip++];
                break;
            case DevicePath::Line:
                seg(pts[ip++]);
                break;
            case DevicePath::Cubic: {
                const Pt c1 = pts[ip], c2 = pts[ip + 1], end = pts[ip + 2];
                ip += 3;
                if (cv.on && !cubic_touches_box(cur, c1, c2, end, cv.lo, cv.hi)) {
                    seg(end);
                } else {
                    Flattener f(cur, [&](Pt p) { seg(p); });
                    if (!f.run(cur, c1, c2, end)) seg(end);
                }
                break;
            }
            case DevicePath::Close:
                seg(start);
                break;
        }
    }
    seg(start);
}

// ---------------------------------------------------------------------------------------------
// Stroker (A.8): Cairo 1.16 cairo-path-stroke-polygon.c + cairo-pen.c + the tangent decomposition of
// cairo-spline.c, and the rectilinear stroker of cairo-path-stroke-boxes.c.  The oracle holds the same
// arithmetic in C (oracle/swfr_oracle.c) and is pinned against libcairo; tests compare the two edge for edge.
// ---------------------------------------------------------------------------------------------
namespace {
struct Face {
    Pt ccw, at, cw;
    int64_t vx = 0, vy = 0;   // device vector of the segment (fixed deltas)
    double ux = 0, uy = 0;    // unit device slope
    double sx = 0, sy = 0;    // unit user-space slope (square caps)
};

double unit(double& dx, double& dy) {
    const double x = dx, y = dy;
    double mag;
    if (x == 0.0) {
        dx = 0.0;
        mag = y > 0.0 ? y : -y;
        dy = y > 0.0 ? 1.0 : -1.0;
    } else if (y == 0.0) {
        dy = 0.0;
        mag = x > 0.0 ? x : -x;
        dx = x > 0.0 ? 1.0 : -1.0;
    } else {
        mag = std::hypot(x, y);
        dx = x / mag;
        dy = y / mag;
    }
    return mag;
}

// _cairo_slope_compare on fixed deltas: < 0 when a turns clockwise into b
int slope_order(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
    const int64_t l = ay * bx, r = by * ax;
    if (l != r) return l < r ? -1 : 1;
    const bool az = ax == 0 && ay == 0, bz = bx == 0 && by == 0;
    if (az && bz) return 0;
    if (az) return 1;
    if (bz) return -1;
    if ((ax ^ bx) < 0 || (ay ^ by) < 0) return (ax > 0 || (ax == 0 && ay > 0)) ? -1 : 1;
    return 0;
}
int turn_direction(const Face& a, const Face& b) { return slope_order(a.vx, a.vy, b.vx, b.vy); }
int cross_sign(double ax, double ay, double bx, double by) {
    const double c = ax * by - bx * ay;
    return c > 0 ? 1 : c < 0 ? -1 : 0;
}

bool has_unity_scale(const Affine& m) {
    const double eps = 1.0 / 256.0, det = m.det();
    if (std::fabs(det * det - 1.0) < eps) {
        if (std::fabs(m.xy) < eps && std::fabs(m.yx) < eps) return true;
        if (std::fabs(m.xx) < eps && std::fabs(m.yy) < eps) return true;
    }
    return false;
}
double circle_major_axis(const Affine& m, double radius) {
    if (has_unity_scale(m)) return radius;
    const double i = m.xx * m.xx + m.yx * m.yx, j = m.xy * m.xy + m.yy * m.yy;
    const double f = 0.5 * (i + j), g = 0.5 * (i - j), h = m.xx * m.xy + m.yx * m.yy;
    return radius * std::sqrt(f + std::hypot(g, h));
}
int pen_vertices_needed(double tolerance, double radius, const Affine& m) {
    const double major = circle_major_axis(m, radius);
    if (tolerance >= 4 * major) return 1;
    if (tolerance >= major) return 4;
    int n = int(std::ceil(2 * M_PI / std::acos(1 - tolerance / major)));
    if (n % 2) ++n;
    return std::max(n, 4);
}

// The pen: a polygonal circle of the stroke's radius under the CTM; joins, caps and curve cusps copy runs of its vertices.
class Pen {
public:
    void init(double radius, double tolerance, const Affine& ctm) {
        const bool reflect = ctm.det() < 0.0;
        const int n = pen_vertices_needed(tolerance, radius, ctm);
        v_.resize(size_t(n));
        for (int i = 0; i < n; ++i) {
            const double theta = 2 * M_PI * i / double(n);
            double dx = radius * std::cos(reflect ? -theta : theta), dy = radius * std::sin(reflect ? -theta : theta);
            ctm.apply_distance(dx, dy);
            v_[size_t(i)].pt = Pt{to_fixed(dx), to_fixed(dy)};
        }
        for (int i = 0; i < n; ++i) {
            const Vertex& prev = v_[size_t((i + n - 1) % n)];
            const Vertex& next = v_[size_t((i + 1) % n)];
            Vertex& v = v_[size_t(i)];
            v.cw_x = int64_t(v.pt.x) - prev.pt.x; v.cw_y = int64_t(v.pt.y) - prev.pt.y;
            v.ccw_x = int64_t(next.pt.x) - v.pt.x; v.ccw_y = int64_t(next.pt.y) - v.pt.y;
        }
    }
    int size() const { return int(v_.size()); }
    Pt offset(int i) const { return v_[size_t(i)].pt; }
    // vertices strictly between the incoming and outgoing directions, walking clockwise / counter-clockwise
    void active_cw(int64_t ix, int64_t iy, int64_t ox, int64_t oy, int& start, int& stop) const {
        const int n = size();
        int lo = 0, hi = n, i = (lo + hi) >> 1;
        do {
            if (slope_order(v_[size_t(i)].cw_x, v_[size_t(i)].cw_y, ix, iy) < 0) lo = i; else hi = i;
            i = (lo + hi) >> 1;
        } while (hi - lo > 1);
        if (slope_order(v_[size_t(i)].cw_x, v_[size_t(i)].cw_y, ix, iy) < 0)
            if (++i == n) i = 0;
        start = i;
        if (slope_order(ox, oy, v_[size_t(i)].ccw_x, v_[size_t(i)].ccw_y) >= 0) {
            lo = i; hi = i + n; i = (lo + hi) >> 1;
            do {
                const int j = i >= n ? i - n : i;
                if (slope_order(v_[size_t(j)].cw_x, v_[size_t(j)].cw_y, ox, oy) > 0) hi = i; else lo = i;
                i = (lo + hi) >> 1;
            } while (hi - lo > 1);
            if (i >= n) i -= n;
        }
        stop = i;
    }
    void active_ccw(int64_t ix, int64_t iy, int64_t ox, int64_t oy, int& start, int& stop) const {
        const int n = size();
        int lo = 0, hi = n, i = (lo + hi) >> 1;
        do {
            if (slope_order(ix, iy, v_[size_t(i)].ccw_x, v_[size_t(i)].ccw_y) < 0) lo = i; else hi = i;
            i = (lo + hi) >> 1;
        } while (hi - lo > 1);
        if (slope_order(ix, iy, v_[size_t(i)].ccw_x, v_[size_t(i)].ccw_y) < 0)
            if (++i == n) i = 0;
        start = i;
        if (slope_order(v_[size_t(i)].cw_x, v_[size_t(i)].cw_y, ox, oy) <= 0) {
            lo = i; hi = i + n; i = (lo + hi) >> 1;
            do {
                const int j = i >= n ? i - n : i;
                if (slope_order(ox, oy, v_[size_t(j)].ccw_x, v_[size_t(j)].ccw_y) > 0) hi = i; else lo = i;
                i = (lo + hi) >> 1;
            } while (hi - lo > 1);
            if (i >= n) i -= n;
        }
        stop = i;
    }

private:
    struct Vertex {
        Pt pt;
        int64_t cw_x = 0, cw_y = 0, ccw_x = 0, ccw_y = 0;   // slopes of the pen edges ending / starting here
    };
    std::vector<Vertex> v_;
};

class OutlineBuilder {
public:
    OutlineBuilder(const StrokeParams& sp, const Affine& ctm, Polygon& out) : sp_(sp), ctm_(ctm), out_(out) {
        inv_ = ctm;
        invertible_ = inv_.invert();
        identity_ = inv_.is_identity();
        det_positive_ = ctm.det() >= 0.0;
        half_ = sp.line_width / 2.0;
        // a round join between two pieces of a flattened curve is only needed when the chord misses the arc by more than
        // the tolerance: cos(turn) < 2 (1 - tol/half)^2 - 1
        cusp_ = 1 - kTolerance / half_;
        cusp_ *= cusp_;
        cusp_ *= 2;
        cusp_ -= 1;
        pen_.init(half_, kTolerance, ctm);
    }
    bool ok() const { return invertible_; }
    void move_to(Pt p) {
        finish_subpath();
        have_first_ = have_current_ = started_ = false;
        first_point_ = p;
        current_.at = p;
    }
    void line_to(Pt p) {
        const Pt from = current_.at;
        started_ = true;
        if (from == p) return;
        const int64_t vx = int64_t(p.x) - from.x, vy = int64_t(p.y) - from.y;
        Face start = make_face(from, vx, vy);
        if (have_current_) {
            const int turn = turn_direction(current_, start);
            if (turn != 0) {
                const bool clockwise = turn < 0;
                // (Cairo 1.16's proximity test that would skip tiny joins is compiled out.)
                join_outer(current_, start, clockwise);
                join_inner(current_, start, clockwise);
            }
        } else {
            begin_with(start);
        }
        current_ = start;
        current_.at = p;
        current_.ccw.x += fixed_t(vx);
        current_.ccw.y += fixed_t(vy);
        current_.cw.x += fixed_t(vx);
        current_.cw.y += fixed_t(vy);
        right_.push_back(current_.cw);
        left_.push_back(current_.ccw);
    }
    void curve_to(Pt b, Pt c, Pt d) {
        const Pt a = current_.at;
        if (sp_.has_bounds && !cubic_touches_box(a, b, c, d, sp_.bounds_lo, sp_.bounds_hi)) return line_to(d);
        if (a == b && c == d) return line_to(d);
        // initial and final tangents (the first / last non-degenerate control leg)
        int64_t ix, iy, fx, fy;
        if (a != b) { ix = int64_t(b.x) - a.x; iy = int64_t(b.y) - a.y; }
        else if (a != c) { ix = int64_t(c.x) - a.x; iy = int64_t(c.y) - a.y; }
        else if (a != d) { ix = int64_t(d.x) - a.x; iy = int64_t(d.y) - a.y; }
        else return line_to(d);
        if (c != d) { fx = int64_t(d.x) - c.x; fy = int64_t(d.y) - c.y; }
        else if (b != d) { fx = int64_t(d.x) - b.x; fy = int64_t(d.y) - b.y; }
        else return line_to(d);
        Face face = make_face(a, ix, iy);
        if (have_current_) {
            const bool clockwise = turn_direction(current_, face) < 0;
            join_outer(current_, face, clockwise);
            join_inner(current_, face, clockwise);
        } else {
            begin_with(face);
        }
        current_ = face;
        spline_last_ = a;
        decompose(Knots{a, b, c, d});
        spline_to(d, fx, fy);
    }
    void close() {
        line_to(first_point_);
        if (have_first_ && have_current_) {
            close_outer(current_, first_);
            close_inner(current_, first_);
            emit(right_, +1);
            emit(left_, -1);
            right_.clear();
            left_.clear();
        } else {
            finish_subpath();
        }
        started_ = have_first_ = have_current_ = false;
    }
    // caps of the sub-path that ends here (add_caps)
    void finish_subpath() {
        if (started_ && !have_first_ && !have_current_ && sp_.cap == 1) {
            // degenerate sub-path with round caps: a dot
            const Face f = make_face(first_point_, 256, 0);
            cap(reversed(f), left_);
            cap(f, left_);
            if (!left_.empty()) left_.push_back(left_.front());
            emit(left_, -1);
            left_.clear();
            return;
        }
        if (have_current_) cap(current_, left_);           // trailing cap
        emit(left_, -1);
        left_.clear();
        if (have_first_) {
            left_.push_back(first_.cw);                    // leading cap: first.cw -> ... -> first.ccw
            cap(reversed(first_), left_);
            emit(left_, -1);
            left_.clear();
        }
        emit(right_, +1);
        right_.clear();
    }

private:
    static Face reversed(const Face& f) {
        Face r = f;
        r.sx = -r.sx; r.sy = -r.sy; r.vx = -r.vx; r.vy = -r.vy;
        std::swap(r.cw, r.ccw);
        return r;
    }
    void begin_with(const Face& start) {
        if (!have_first_) {
            first_ = start;
            have_first_ = true;
        }
        have_current_ = true;
        right_.push_back(start.cw);
        left_.push_back(start.ccw);
    }
    Face make_face(Pt at, int64_t vx, int64_t vy) const {
        Face f;
        double sx = double(vx) / 256.0, sy = double(vy) / 256.0;
        unit(sx, sy);
        f.ux = sx;
        f.uy = sy;
        double fx, fy;
        if (!identity_) {
            inv_.apply_distance(sx, sy);
            unit(sx, sy);
            if (det_positive_) {
                fx = -sy * half_;
                fy = sx * half_;
            } else {
                fx = sy * half_;
                fy = -sx * half_;
            }
            ctm_.apply_distance(fx, fy);
        } else {
            fx = -sy * half_;
            fy = sx * half_;
        }
        const fixed_t ox = to_fixed(fx), oy = to_fixed(fy);
        f.ccw = Pt{at.x + ox, at.y + oy};
        f.at = at;
        f.cw = Pt{at.x - ox, at.y - oy};
        f.sx = sx;
        f.sy = sy;
        f.vx = vx;
        f.vy = vy;
        return f;
    }
    // pen vertices between two directions around `mid`, appended to `side`
    void fan(int64_t ix, int64_t iy, int64_t ox, int64_t oy, Pt mid, bool clockwise, std::vector<Pt>& side) const {
        if (sp_.has_bounds && !(sp_.bounds_lo.x <= mid.x && mid.x <= sp_.bounds_hi.x && sp_.bounds_lo.y <= mid.y && mid.y <= sp_.bounds_hi.y))
            return;
        int start, stop;
        const int n = pen_.size();
        if (clockwise) {
            pen_.active_cw(ix, iy, ox, oy, start, stop);
            while (start != stop) {
                const Pt o = pen_.offset(start);
                side.push_back(Pt{mid.x + o.x, mid.y + o.y});
                if (++start == n) start = 0;
            }
        } else {
            pen_.active_ccw(ix, iy, ox, oy, start, stop);
            while (start != stop) {
                const Pt o = pen_.offset(start);
                side.push_back(Pt{mid.x + o.x, mid.y + o.y});
                if (start-- == 0) start += n;
            }
        }
    }
    void join_inner(const Face& in, const Face& out, bool clockwise) {
        std::vector<Pt>& side = clockwise ? left_ : right_;
        side.push_back(in.at);
        side.push_back(clockwise ? out.ccw : out.cw);
    }
    void close_inner(const Face& in, const Face& out) {
        const bool clockwise = turn_direction(in, out) < 0;
        std::vector<Pt>& side = clockwise ? left_ : right_;
        side.push_back(in.at);
        side.push_back(clockwise ? out.ccw : out.cw);
        side.front() = side.back();
    }
    // the miter tip when the limit allows it and it lies between the two faces
    bool miter_tip(const Face& in, const Face& out, Pt a, Pt b, Pt& tip) const {
        const double dot = in.ux * out.ux + in.uy * out.uy, ml = sp_.miter_limit;
        if (!(2 <= ml * ml * (1 + dot))) return false;
        const double x1 = from_fixed(a.x), y1 = from_fixed(a.y), dx1 = in.ux, dy1 = in.uy;
        const double x2 = from_fixed(b.x), y2 = from_fixed(b.y), dx2 = out.ux, dy2 = out.uy;
        const double my = ((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1);
        const double mx = std::fabs(dy1) >= std::fabs(dy2) ? (my - y1) * dx1 / dy1 + x1 : (my - y2) * dx2 / dy2 + x2;
        const double ix = from_fixed(in.at.x), iy = from_fixed(in.at.y);
        if (cross_sign(x1 - ix, y1 - iy, mx - ix, my - iy) == cross_sign(x2 - ix, y2 - iy, mx - ix, my - iy)) return false;
        tip = Pt{to_fixed(mx), to_fixed(my)};
        return true;
    }
    void join_outer(const Face& in, const Face& out, bool clockwise) {
        if (in.cw == out.cw && in.ccw == out.ccw) return;
        const Pt a = clockwise ? in.cw : in.ccw, b = clockwise ? out.cw : out.ccw;
        std::vector<Pt>& side = clockwise ? right_ : left_;
        if (sp_.join == 1) {
            fan(in.vx, in.vy, out.vx, out.vy, in.at, clockwise, side);
        } else if (sp_.join == 0) {
            Pt tip;
            if (miter_tip(in, out, a, b, tip)) {
                side.back() = tip;
                return;
            }
        }
        side.push_back(b);  // bevel, a rejected miter, or the end of the fan
    }
    // the join that closes a sub-path: a round join too flat to need a fan falls through to the miter code (as Cairo does;
    // its tip can leave the stroke's approximate extents, which bound what is painted)
    void close_outer(const Face& in, const Face& out) {
        if (in.cw == out.cw && in.ccw == out.ccw) return;
        const bool clockwise = turn_direction(in, out) < 0;
        const Pt a = clockwise ? in.cw : in.ccw, b = clockwise ? out.cw : out.ccw;
        std::vector<Pt>& side = clockwise ? right_ : left_;
        if (sp_.join == 1 && (in.ux * out.ux + in.uy * out.uy) < cusp_) {
            fan(in.vx, in.vy, out.vx, out.vy, in.at, clockwise, side);
        } else if (sp_.join != 2) {
            Pt tip;
            if (miter_tip(in, out, a, b, tip)) {
                side.back() = tip;
                side.front() = tip;
                return;
            }
        }
        side.push_back(b);
    }
    void cap(const Face& f, std::vector<Pt>& side) const {
        if (sp_.cap == 1) {
            fan(f.vx, f.vy, -f.vx, -f.vy, f.at, false, side);
        } else if (sp_.cap == 2) {
            double dx = f.sx * half_, dy = f.sy * half_;
            ctm_.apply_distance(dx, dy);
            const fixed_t vx = to_fixed(dx), vy = to_fixed(dy);
            side.push_back(Pt{f.ccw.x + vx, f.ccw.y + vy});
            side.push_back(Pt{f.cw.x + vx, f.cw.y + vy});
        }
        side.push_back(f.cw);
    }
    // one point of a flattened curve with the curve's tangent there
    void spline_to(Pt p, int64_t tx, int64_t ty) {
        Face face;
        if ((tx | ty) == 0) {                                 // cusp: turn around with a fan
            face = reversed(current_);
            const bool clockwise = turn_direction(current_, face) < 0;
            fan(current_.vx, current_.vy, face.vx, face.vy, current_.at, clockwise, clockwise ? right_ : left_);
        } else {
            face = make_face(p, tx, ty);
            if ((face.ux * current_.ux + face.uy * current_.uy) < cusp_) {
                const bool clockwise = turn_direction(current_, face) < 0;
                current_.cw.x += face.at.x - current_.at.x;
                current_.cw.y += face.at.y - current_.at.y;
                right_.push_back(current_.cw);
                current_.ccw.x += face.at.x - current_.at.x;
                current_.ccw.y += face.at.y - current_.at.y;
                left_.push_back(current_.ccw);
                fan(current_.vx, current_.vy, face.vx, face.vy, current_.at, clockwise, clockwise ? right_ : left_);
            }
            right_.push_back(face.cw);
            left_.push_back(face.ccw);
        }
        current_ = face;
    }
    void spline_point(Pt p, Pt knot) {
        if (p == spline_last_) return;
        spline_last_ = p;
        spline_to(p, int64_t(knot.x) - p.x, int64_t(knot.y) - p.y);
    }
    void decompose(Knots k) {
        if (flatness2(k) < kTolerance * kTolerance) return spline_point(k.a, k.b);
        const Pt ab = midpoint(k.a, k.b), bc = midpoint(k.b, k.c), cd = midpoint(k.c, k.d);
        const Pt abbc = midpoint(ab, bc), bccd = midpoint(bc, cd), mid = midpoint(abbc, bccd);
        decompose(Knots{k.a, ab, abbc, mid});
        decompose(Knots{mid, bccd, cd, k.d});
    }
    void emit(const std::vector<Pt>& contour, int dir) {
        for (size_t i = 1; i < contour.size(); ++i) out_.add_segment(contour[i - 1], contour[i], dir);
    }

    StrokeParams sp_;
    Affine ctm_, inv_;
    Polygon& out_;
    Pen pen_;
    bool invertible_ = true, identity_ = false, det_positive_ = true;
    double half_ = 0.5, cusp_ = 0.0;
    std::vector<Pt> right_, left_;  // cw contour (direction +1), ccw contour (direction -1)
    Face current_, first_;
    Pt first_point_{}, spline_last_{};
    bool have_first_ = false, have_current_ = false, started_ = false;
};
}  // namespace

void fill_to_polygon(const DevicePath& path, Polygon& out) {
    // the polygon was reset by the caller (with or without clipping)
    ClipView cv{false, {}, {}};
    fill_walk(path, out, cv);
}

// Variant used by the frame builder when clipping is on (needs the limits for the curve shortcut).
void fill_to_polygon_clipped(const DevicePath& path, Polygon& out, Pt lo, Pt hi) {
    ClipView cv{true, lo, hi};
    fill_walk(path, out, cv);
}

bool stroke_to_polygon(const DevicePath& path, const StrokeParams& sp, const Affine& ctm, Polygon& out) {
    OutlineBuilder ob(sp, ctm, out);
    if (!ob.ok()) return false;
    const auto& pts = path.points();
    size_t ip = 0;
    for (DevicePath::Verb v : path.verbs()) {
        switch (v) {
            case DevicePath::Move:
                ob.move_to(pts[ip++]);
                break;
            case DevicePath::Line:
                ob.line_to(pts[ip++]);
                break;
            case DevicePath::Cubic:
                ob.curve_to(pts[ip], pts[ip + 1], pts[ip + 2]);
                ip += 3;
                break;
            case DevicePath::Close:
                ob.close();
                break;
        }
    }
    ob.finish_subpath();
    return true;
}

int stroke_pen_vertices(double line_width, const Affine& ctm) { return pen_vertices_needed(kTolerance, line_width / 2.0, ctm); }

bool stroke_rectilinear_to_boxes(const DevicePath& path, const StrokeParams& sp, const Affine& ctm, Polygon& out) {
    if (sp.join != 0 || sp.miter_limit < M_SQRT2 || !(sp.cap == 0 || sp.cap == 2)) return false;
    if (!(ctm.xy == 0.0 && ctm.yx == 0.0)) return false;      // scale-only matrices
    const fixed_t hx = to_fixed(std::fabs(ctm.xx) * sp.line_width / 2.0), hy = to_fixed(std::fabs(ctm.yy) * sp.line_width / 2.0);
    struct Seg {
        Pt a, b;
        bool horizontal;
    };
    std::vector<Seg> segs;
    bool open_sub_path = false;
    Pt cur{}, first{};
    auto box = [&](fixed_t x1, fixed_t y1, fixed_t x2, fixed_t y2) {
        if (x1 == x2 || y1 == y2) return;
        out.add_segment(Pt{x1, y1}, Pt{x1, y2}, +1);
        out.add_segment(Pt{x2, y1}, Pt{x2, y2}, -1);
    };
    auto flush = [&]() {
        const size_t n = segs.size();
        for (size_t i = 0; i < n; ++i) {
            Pt a = segs[i].a, b = segs[i].b;
            // lengthen towards a perpendicular neighbour (the miter) or for a square cap at an open end
            bool grow_a = segs[i].horizontal != segs[i == 0 ? n - 1 : i - 1].horizontal;
            bool grow_b = segs[i].horizontal != segs[i == n - 1 ? 0 : i + 1].horizontal;
            if (open_sub_path) {
                if (i == 0) grow_a = sp.cap != 0;
                if (i == n - 1) grow_b = sp.cap != 0;
            }
            if (a.y == b.y) {
                if (a.x < b.x) { if (grow_a) a.x -= hx; if (grow_b) b.x += hx; }
                else { if (grow_a) a.x += hx; if (grow_b) b.x -= hx; }
                a.y -= hy; b.y += hy;
            } else {
                if (a.y < b.y) { if (grow_a) a.y -= hy; if (grow_b) b.y += hy; }
                else { if (grow_a) a.y += hy; if (grow_b) b.y -= hy; }
                a.x -= hx; b.x += hx;
            }
            box(std::min(a.x, b.x), std::min(a.y, b.y), std::max(a.x, b.x), std::max(a.y, b.y));
        }
        segs.clear();
    };
    auto line = [&](Pt b) {
        if (cur == b) return;
        segs.push_back(Seg{cur, b, cur.y == b.y});
        cur = b;
        open_sub_path = true;
    };
    const auto& pts = path.points();
    size_t ip = 0;
    for (DevicePath::Verb v : path.verbs()) {
        switch (v) {
            case DevicePath::Move:
                flush();
                cur = first = pts[ip++];
                open_sub_path = false;
                break;
            case DevicePath::Line:
                line(pts[ip++]);
                break;
            case DevicePath::Cubic:
                ip += 3;                                       // cannot happen: the path is rectilinear
                break;
            case DevicePath::Close:
                if (open_sub_path) {
                    line(first);
                    open_sub_path = false;
                    flush();
                }
                break;
        }
    }
    flush();
    return true;
}

// ---------------------------------------------------------------------------------------------
// Rectilinear region -> disjoint boxes (A.6)
// ---------------------------------------------------------------------------------------------
void rectilinear_to_boxes(const Polygon& poly, bool even_odd, std::vector<swfr_edge>& boxes) {
    struct V {
        fixed_t x, top, bottom;
        int dir;
    };
    std::vector<V> vs;
    std::vector<fixed_t> ys;
    for (const swfr_edge& e : poly.edges()) {
        vs.push_back(V{e.x1, e.top, e.bottom, e.dir});
        ys.push_back(e.top);
        ys.push_back(e.bottom);
    }
    std::sort(ys.begin(), ys.end());
    ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
    std::stable_sort(vs.begin(), vs.end(), [](const V& a, const V& b) { return a.x < b.x; });
    const unsigned mask = even_odd ? 1u : ~0u;
    for (size_t s = 0; s + 1 < ys.size(); ++s) {
        const fixed_t ya = ys[s], yb = ys[s + 1];
        int winding = 0;
        bool inside = false;
        fixed_t xs = 0;
        for (const V& v : vs) {
            if (!(v.top <= ya && v.bottom >= yb)) continue;
            winding += v.dir;
            const bool now = (unsigned(winding) & mask) != 0;
            if (now && !inside) {
                xs = v.x;
                inside = true;
            } else if (!now && inside) {
                inside = false;
                if (xs != v.x) {
                    // merge with the box directly above when it has the same x-range
                    bool merged = false;
                    for (auto it = boxes.rbegin(); it != boxes.rend() && it->y2 >= ya; ++it) {
                        if (it->y2 == ya && it->x1 == xs && it->x2 == v.x) {
                            it->y2 = yb;
                            it->bottom = yb;
                            merged = true;
                            break;
                        }
                    }
                    if (!merged) boxes.push_back(swfr_edge{xs, ya, v.x, yb, ya, yb, 0, 0});
                }
            }
        }
    }
}

}  // namespace swfr
