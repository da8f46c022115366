/*
 * swfr_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into the product.
 *
 * CPU restatement (plain C, single thread) of the arithmetic the reference's pixel-correct
 * path delegates to: node-canvas 2.6.1 -> Cairo 1.16 image backend -> pixman 0.40
 * (pinned in /root/reference/ts/yarn.lock:835-837; call sites
 * ts/src/lib/renderers/canvas-renderer.ts:69-78,179-188,207-350).  That code is a
 * third-party dependency which is NOT under /root/reference, so this file restates its
 * published algorithm behind a mini "cairo context" API that
 * oracle/oracle_backend.py drives with the same call sequence the reference issues.
 *
 * Pinning: tests/test_oracle_goldens.py checks this file against every golden the
 * reference's own tests hold for the path (tests/flat-shapes/<asterisk>/shape.png,
 * tests/flat-morph-shapes/homestuck-beta-29/{0,32768,65536}.png) and, when the container's
 * libcairo.so.2 (1.16.0) is present, tests/test_oracle_vs_cairo.py fuzzes it against that.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Sections:  [A.1] fixed point  [A.2] path  [A.3] spline  [A.5b] polygon + limits
 *            [A.8] stroker      [A.5] tor scan converter  [A.6] boxes  [A.7] compositing
 */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <float.h>

#define EXPORT __attribute__((visibility("default")))

typedef int32_t fx_t;                       /* 24.8 fixed */
typedef struct { fx_t x, y; } pt_t;
typedef struct { double xx, yx, xy, yy, x0, y0; } mat_t;

#define FX_ONE 256
#define GRID_X 256
#define GRID_Y 15

/* ------------------------------------------------------------------ [A.1] fixed point */
static fx_t fx_from_double(double d) { return (fx_t)nearbyint(d * 256.0); }   /* rint, ties-to-even */
static double fx_to_double(fx_t f) { return (double)f / 256.0; }
static int fx_floor_i(fx_t f) { return f >> 8; }
static int fx_ceil_i(fx_t f) { return (f + 255) >> 8; }

static void mat_identity(mat_t *m) { m->xx = 1; m->yx = 0; m->xy = 0; m->yy = 1; m->x0 = 0; m->y0 = 0; }
/* r = a * b : apply a first, then b (cairo_matrix_multiply) */
static void mat_multiply(mat_t *r, const mat_t *a, const mat_t *b)
{
    mat_t t;
    t.xx = a->xx * b->xx + a->yx * b->xy;
    t.yx = a->xx * b->yx + a->yx * b->yy;
    t.xy = a->xy * b->xx + a->yy * b->xy;
    t.yy = a->xy * b->yx + a->yy * b->yy;
    t.x0 = a->x0 * b->xx + a->y0 * b->xy + b->x0;
    t.y0 = a->x0 * b->yx + a->y0 * b->yy + b->y0;
    *r = t;
}
static void mat_point(const mat_t *m, double *x, double *y)
{
    double nx = m->xx * *x + m->xy * *y + m->x0;
    double ny = m->yx * *x + m->yy * *y + m->y0;
    *x = nx; *y = ny;
}
static void mat_distance(const mat_t *m, double *dx, double *dy)
{
    double nx = m->xx * *dx + m->xy * *dy;
    double ny = m->yx * *dx + m->yy * *dy;
    *dx = nx; *dy = ny;
}
static double mat_det(const mat_t *m) { return m->xx * m->yy - m->yx * m->xy; }
static int mat_invert(mat_t *m)
{
    double det = mat_det(m);
    mat_t r;
    if (det == 0 || !isfinite(det)) return 0;
    r.xx = m->yy / det;  r.yx = -m->yx / det;
    r.xy = -m->xy / det; r.yy = m->xx / det;
    r.x0 = (m->xy * m->y0 - m->yy * m->x0) / det;
    r.y0 = (m->yx * m->x0 - m->xx * m->y0) / det;
    *m = r;
    return 1;
}
/* cairo_matrix_invert, operation by operation (the pattern matrix is rounded to 16.16 afterwards, so the order matters in principle) */
static int mat_invert_cairo(mat_t *m)
{
    if (m->xy == 0. && m->yx == 0.) {
        m->x0 = -m->x0; m->y0 = -m->y0;
        if (m->xx != 1.) { if (m->xx == 0.) return 0; m->xx = 1. / m->xx; m->x0 *= m->xx; }
        if (m->yy != 1.) { if (m->yy == 0.) return 0; m->yy = 1. / m->yy; m->y0 *= m->yy; }
        return 1;
    }
    double det = mat_det(m);
    if (det == 0 || !isfinite(det)) return 0;
    const double a = m->xx, b = m->yx, cc = m->xy, d = m->yy, tx = m->x0, ty = m->y0, k = 1 / det;
    m->xx = d * k; m->yx = -b * k; m->xy = -cc * k; m->yy = a * k; m->x0 = (cc * ty - d * tx) * k; m->y0 = (b * tx - a * ty) * k;
    return 1;
}
static int mat_is_identity(const mat_t *m)
{
    return m->xx == 1 && m->yx == 0 && m->xy == 0 && m->yy == 1 && m->x0 == 0 && m->y0 == 0;
}

/* ------------------------------------------------------------------ [A.2] path */
enum { OP_MOVE = 0, OP_LINE = 1, OP_CURVE = 2, OP_CLOSE = 3 };

typedef struct {
    uint8_t *ops; int nops, cops;
    pt_t *pts; int npts, cpts;
    pt_t cur, last_move;
    int has_cur, needs_move_to, has_extents;
    int fill_is_rect, stroke_is_rect;
    pt_t e1, e2;                            /* extents of all op points */
} path_t;

static void path_reset(path_t *p)
{
    p->nops = p->npts = 0;
    p->has_cur = 0; p->needs_move_to = 1; p->has_extents = 0;
    p->fill_is_rect = p->stroke_is_rect = 1;
    p->cur.x = p->cur.y = 0; p->last_move = p->cur;
}
static void path_ext(path_t *p, pt_t q)
{
    if (!p->has_extents) { p->e1 = p->e2 = q; p->has_extents = 1; return; }
    if (q.x < p->e1.x) p->e1.x = q.x;
    if (q.x > p->e2.x) p->e2.x = q.x;
    if (q.y < p->e1.y) p->e1.y = q.y;
    if (q.y > p->e2.y) p->e2.y = q.y;
}
static void path_add(path_t *p, int op, const pt_t *pts, int n)
{
    if (p->nops == p->cops) { p->cops = p->cops ? 2 * p->cops : 64; p->ops = realloc(p->ops, p->cops); }
    if (p->npts + n > p->cpts) { p->cpts = p->cpts ? 2 * p->cpts + n : 128; p->pts = realloc(p->pts, sizeof(pt_t) * p->cpts); }
    p->ops[p->nops++] = (uint8_t)op;
    for (int i = 0; i < n; i++) p->pts[p->npts++] = pts[i];
}
static int path_last_op(const path_t *p) { return p->nops ? p->ops[p->nops - 1] : -1; }
static void path_drop_line_to(path_t *p) { p->nops--; p->npts--; }

static void path_new_sub_path(path_t *p)
{
    if (!p->needs_move_to) {
        if (p->fill_is_rect)   /* implicit close for fill */
            p->fill_is_rect = p->cur.x == p->last_move.x || p->cur.y == p->last_move.y;
        p->needs_move_to = 1;
    }
    p->has_cur = 0;
}
static void path_move_to(path_t *p, fx_t x, fx_t y)
{
    path_new_sub_path(p);
    p->has_cur = 1;
    p->cur.x = x; p->cur.y = y;
    p->last_move = p->cur;
}
static void path_move_to_apply(path_t *p)
{
    if (!p->needs_move_to) return;
    p->needs_move_to = 0;
    path_ext(p, p->cur);
    p->last_move = p->cur;
    path_add(p, OP_MOVE, &p->cur, 1);
}
static void path_line_to(path_t *p, fx_t x, fx_t y)
{
    pt_t q = { x, y };
    if (!p->has_cur) { path_move_to(p, x, y); return; }
    path_move_to_apply(p);
    /* a degenerate line is kept only directly after a MOVE_TO */
    if (path_last_op(p) != OP_MOVE && x == p->cur.x && y == p->cur.y) return;
    /* collinear merge / replacement of a degenerate previous line */
    if (path_last_op(p) == OP_LINE) {
        pt_t pp = p->pts[p->npts - 2];
        if (pp.x == p->cur.x && pp.y == p->cur.y) {
            path_drop_line_to(p);
        } else {
            int64_t pdx = (int64_t)p->cur.x - pp.x, pdy = (int64_t)p->cur.y - pp.y;
            int64_t sdx = (int64_t)x - p->cur.x, sdy = (int64_t)y - p->cur.y;
            if (pdy * sdx == sdy * pdx &&
                !(((pdx * sdx) >> 8) + ((pdy * sdy) >> 8) < 0))       /* not backwards */
                path_drop_line_to(p);
        }
    }
    if (p->stroke_is_rect) {
        p->stroke_is_rect = p->cur.x == x || p->cur.y == y;
        p->fill_is_rect &= p->stroke_is_rect;
    }
    p->cur = q;
    path_ext(p, q);
    path_add(p, OP_LINE, &q, 1);
}
static void path_curve_to(path_t *p, fx_t x0, fx_t y0, fx_t x1, fx_t y1, fx_t x2, fx_t y2)
{
    pt_t q[3] = { { x0, y0 }, { x1, y1 }, { x2, y2 } };
    if (p->has_cur && p->cur.x == x2 && p->cur.y == y2 &&
        x1 == x2 && x0 == x2 && y1 == y2 && y0 == y2) { path_line_to(p, x2, y2); return; }
    if (!p->has_cur) path_move_to(p, x0, y0);
    path_move_to_apply(p);
    if (path_last_op(p) == OP_LINE) {
        pt_t pp = p->pts[p->npts - 2];
        if (pp.x == p->cur.x && pp.y == p->cur.y) path_drop_line_to(p);
    }
    /* extents: control-point box (a superset of cairo's tight curve box; only used to decide
       whether polygon limits apply, where a superset is equivalent -- see DESIGN.md) */
    path_ext(p, q[0]); path_ext(p, q[1]); path_ext(p, q[2]);
    p->cur = q[2];
    p->fill_is_rect = p->stroke_is_rect = 0;
    path_add(p, OP_CURVE, q, 3);
}
static void path_close(path_t *p)
{
    if (!p->has_cur) return;
    path_line_to(p, p->last_move.x, p->last_move.y);
    /* cairo drops a trailing line_to that the close implies; irrelevant for fills, and the
       reference never calls closePath() (canvas-renderer.ts has no closePath call) */
    p->needs_move_to = 1;
    path_add(p, OP_CLOSE, NULL, 0);
}
static int path_fill_is_rectilinear(const path_t *p)
{
    if (!p->fill_is_rect) return 0;
    if (!p->has_cur || p->needs_move_to) return 1;
    return p->cur.x == p->last_move.x || p->cur.y == p->last_move.y;
}

/* ------------------------------------------------------------------ polygon */
typedef struct { pt_t p1, p2; fx_t top, bottom; int dir; } pedge_t;
typedef struct {
    pedge_t *e; int n, cap;
    int has_limits; pt_t l1, l2;
    pt_t x1, x2;                            /* extents (min,max) */
} polygon_t;

static void polygon_init(polygon_t *g, int has_limits, pt_t l1, pt_t l2)
{
    g->n = 0; g->has_limits = has_limits; g->l1 = l1; g->l2 = l2;
    g->x1.x = g->x1.y = INT32_MAX; g->x2.x = g->x2.y = INT32_MIN;
}
static fx_t edge_x_for_y(const pt_t *p1, const pt_t *p2, fx_t y)
{
    if (y == p1->y) return p1->x;
    if (y == p2->y) return p2->x;
    int64_t dy = (int64_t)p2->y - p1->y;
    fx_t x = p1->x;
    if (dy != 0) {
        int64_t num = ((int64_t)y - p1->y) * ((int64_t)p2->x - p1->x);
        /* _cairo_fixed_mul_div_floor is a plain C (truncating) division despite its name:
           pinned by fuzzing off-frame polygons against libcairo (floor: 10/6000 differ) */
        x += (fx_t)(num / dy);
    }
    return x;
}
static fx_t edge_y_for_x(const pt_t *p1, const pt_t *p2, fx_t x)
{
    if (x == p1->x) return p1->y;
    if (x == p2->x) return p2->y;
    int64_t dx = (int64_t)p2->x - p1->x;
    fx_t y = p1->y;
    if (dx != 0) {
        int64_t num = ((int64_t)x - p1->x) * ((int64_t)p2->y - p1->y);
        y += (fx_t)(num / dx);
    }
    return y;
}
static void polygon_raw_add(polygon_t *g, const pt_t *p1, const pt_t *p2, fx_t top, fx_t bottom, int dir)
{
    if (g->n == g->cap) { g->cap = g->cap ? 2 * g->cap : 64; g->e = realloc(g->e, sizeof(pedge_t) * g->cap); }
    pedge_t *e = &g->e[g->n++];
    e->p1 = *p1; e->p2 = *p2; e->top = top; e->bottom = bottom; e->dir = dir;
    if (top < g->x1.y) g->x1.y = top;
    if (bottom > g->x2.y) g->x2.y = bottom;
    if (p1->x < g->x1.x || p1->x > g->x2.x) {
        fx_t x = p1->x;
        if (top != p1->y) x = edge_x_for_y(p1, p2, top);
        if (x < g->x1.x) g->x1.x = x;
        if (x > g->x2.x) g->x2.x = x;
    }
    if (p2->x < g->x1.x || p2->x > g->x2.x) {
        fx_t x = p2->x;
        if (bottom != p2->y) x = edge_x_for_y(p1, p2, bottom);
        if (x < g->x1.x) g->x1.x = x;
        if (x > g->x2.x) g->x2.x = x;
    }
}
/* [A.5b] _add_clipped_edge */
static void polygon_add_clipped(polygon_t *g, const pt_t *p1, const pt_t *p2, fx_t top, fx_t bottom, int dir)
{
    pt_t l1 = g->l1, l2 = g->l2, bot_left = { l1.x, l2.y }, top_right = { l2.x, l1.y };
    if (top >= l2.y || bottom <= l1.y) return;
    fx_t top_y = top > l1.y ? top : l1.y, bot_y = bottom < l2.y ? bottom : l2.y;
    fx_t pleft = p1->x < p2->x ? p1->x : p2->x, pright = p1->x > p2->x ? p1->x : p2->x;
    if (l1.x <= pleft && pright <= l2.x) { polygon_raw_add(g, p1, p2, top_y, bot_y, dir); return; }
    if (pright <= l1.x) { polygon_raw_add(g, &l1, &bot_left, top_y, bot_y, dir); return; }
    if (l2.x <= pleft) { polygon_raw_add(g, &top_right, &l2, top_y, bot_y, dir); return; }
    fx_t left_y, right_y;
    int tlbr = (p1->x <= p2->x) == (p1->y <= p2->y);
    if (tlbr) {
        if (pleft >= l1.x) left_y = top_y;
        else { left_y = edge_y_for_x(p1, p2, l1.x); if (edge_x_for_y(p1, p2, left_y) < l1.x) left_y++; }
        if (left_y > bot_y) left_y = bot_y;
        if (top_y < left_y) { polygon_raw_add(g, &l1, &bot_left, top_y, left_y, dir); top_y = left_y; }
        if (pright <= l2.x) right_y = bot_y;
        else { right_y = edge_y_for_x(p1, p2, l2.x); if (edge_x_for_y(p1, p2, right_y) > l2.x) right_y--; }
        if (right_y < top_y) right_y = top_y;
        if (bot_y > right_y) { polygon_raw_add(g, &top_right, &l2, right_y, bot_y, dir); bot_y = right_y; }
    } else {
        if (pright <= l2.x) right_y = top_y;
        else { right_y = edge_y_for_x(p1, p2, l2.x); if (edge_x_for_y(p1, p2, right_y) > l2.x) right_y++; }
        if (right_y > bot_y) right_y = bot_y;
        if (top_y < right_y) { polygon_raw_add(g, &top_right, &l2, top_y, right_y, dir); top_y = right_y; }
        if (pleft >= l1.x) left_y = bot_y;
        else { left_y = edge_y_for_x(p1, p2, l1.x); if (edge_x_for_y(p1, p2, left_y) < l1.x) left_y--; }
        if (left_y < top_y) left_y = top_y;
        if (bot_y > left_y) { polygon_raw_add(g, &l1, &bot_left, left_y, bot_y, dir); bot_y = left_y; }
    }
    if (top_y != bot_y) polygon_raw_add(g, p1, p2, top_y, bot_y, dir);
}
static void polygon_add_edge(polygon_t *g, const pt_t *a, const pt_t *b, int dir)
{
    const pt_t *p1 = a, *p2 = b;
    if (p1->y == p2->y) return;                         /* horizontal edges are dropped */
    if (p1->y > p2->y) { p1 = b; p2 = a; dir = -dir; }
    if (g->has_limits) {
        if (p2->y <= g->l1.y || p1->y >= g->l2.y) return;
        polygon_add_clipped(g, p1, p2, p1->y, p2->y, dir);
    } else
        polygon_raw_add(g, p1, p2, p1->y, p2->y, dir);
}

/* ------------------------------------------------------------------ [A.3] spline */
typedef void (*add_point_fn)(void *closure, const pt_t *p);
typedef struct { pt_t a, b, c, d; } knots_t;
typedef struct { add_point_fn fn; void *closure; pt_t last; } spline_t;

static void spline_add_point(spline_t *s, const pt_t *p)
{
    if (p->x == s->last.x && p->y == s->last.y) return;
    s->last = *p;
    s->fn(s->closure, p);
}
static void lerp_half(const pt_t *a, const pt_t *b, pt_t *r)
{
    r->x = a->x + ((b->x - a->x) >> 1);
    r->y = a->y + ((b->y - a->y) >> 1);
}
static double spline_error_squared(const knots_t *k)
{
    double bdx = fx_to_double(k->b.x - k->a.x), bdy = fx_to_double(k->b.y - k->a.y);
    double cdx = fx_to_double(k->c.x - k->a.x), cdy = fx_to_double(k->c.y - k->a.y);
    if (k->a.x != k->d.x || k->a.y != k->d.y) {
        double dx = fx_to_double(k->d.x - k->a.x), dy = fx_to_double(k->d.y - k->a.y);
        double v = dx * dx + dy * dy, u;
        u = bdx * dx + bdy * dy;
        if (u <= 0) { /* keep */ } else if (u >= v) { bdx -= dx; bdy -= dy; }
        else { bdx -= u / v * dx; bdy -= u / v * dy; }
        u = cdx * dx + cdy * dy;
        if (u <= 0) { /* keep */ } else if (u >= v) { cdx -= dx; cdy -= dy; }
        else { cdx -= u / v * dx; cdy -= u / v * dy; }
    }
    double berr = bdx * bdx + bdy * bdy, cerr = cdx * cdx + cdy * cdy;
    return berr > cerr ? berr : cerr;
}
static void spline_decompose_into(knots_t *s1, double tol2, spline_t *out)
{
    if (spline_error_squared(s1) < tol2) { spline_add_point(out, &s1->a); return; }
    knots_t s2;
    pt_t ab, bc, cd, abbc, bccd, fin;
    lerp_half(&s1->a, &s1->b, &ab); lerp_half(&s1->b, &s1->c, &bc); lerp_half(&s1->c, &s1->d, &cd);
    lerp_half(&ab, &bc, &abbc); lerp_half(&bc, &cd, &bccd); lerp_half(&abbc, &bccd, &fin);
    s2.a = fin; s2.b = bccd; s2.c = cd; s2.d = s1->d;
    s1->b = ab; s1->c = abbc; s1->d = fin;
    spline_decompose_into(s1, tol2, out);
    spline_decompose_into(&s2, tol2, out);
}
/* returns 0 when the spline degenerates to the line a->d (caller emits line_to(d)) */
static int spline_flatten(const pt_t *a, const pt_t *b, const pt_t *c, const pt_t *d, double tol,
                          add_point_fn fn, void *closure)
{
    if (a->x == b->x && a->y == b->y && c->x == d->x && c->y == d->y) return 0;
    spline_t s = { fn, closure, *a };
    knots_t k = { *a, *b, *c, *d };
    spline_decompose_into(&k, tol * tol, &s);
    fn(closure, d);                                       /* final point is always emitted */
    return 1;
}

/* ------------------------------------------------------------------ filler */
typedef struct { polygon_t *g; pt_t cur, last_move; } filler_t;
static void filler_line_to(void *c, const pt_t *p)
{
    filler_t *f = c;
    polygon_add_edge(f->g, &f->cur, p, 1);
    f->cur = *p;
}
static int box_contains(const pt_t *l1, const pt_t *l2, const pt_t *p)
{
    return p->x >= l1->x && p->x <= l2->x && p->y >= l1->y && p->y <= l2->y;
}
static int spline_intersects(const pt_t *a, const pt_t *b, const pt_t *c, const pt_t *d, const pt_t *l1, const pt_t *l2)
{
    if (box_contains(l1, l2, a) || box_contains(l1, l2, b) || box_contains(l1, l2, c) || box_contains(l1, l2, d)) return 1;
    pt_t m1 = *a, m2 = *a;
    const pt_t *q[3] = { b, c, d };
    for (int i = 0; i < 3; i++) {
        if (q[i]->x < m1.x) m1.x = q[i]->x;
        if (q[i]->x > m2.x) m2.x = q[i]->x;
        if (q[i]->y < m1.y) m1.y = q[i]->y;
        if (q[i]->y > m2.y) m2.y = q[i]->y;
    }
    if (m2.x <= l1->x || m1.x >= l2->x || m2.y <= l1->y || m1.y >= l2->y) return 0;
    return 1;
}
static void path_fill_to_polygon(const path_t *p, double tol, polygon_t *g)
{
    filler_t f = { g, { 0, 0 }, { 0, 0 } };
    const pt_t *pts = p->pts;
    for (int i = 0; i < p->nops; i++) {
        switch (p->ops[i]) {
        case OP_MOVE:
            filler_line_to(&f, &f.last_move);             /* close current sub-path */
            f.cur = f.last_move = pts[0]; pts += 1; break;
        case OP_LINE:
            filler_line_to(&f, &pts[0]); pts += 1; break;
        case OP_CURVE:
            if (g->has_limits && !spline_intersects(&f.cur, &pts[0], &pts[1], &pts[2], &g->l1, &g->l2))
                filler_line_to(&f, &pts[2]);
            else if (!spline_flatten(&f.cur, &pts[0], &pts[1], &pts[2], tol, filler_line_to, &f))
                filler_line_to(&f, &pts[2]);
            pts += 3; break;
        case OP_CLOSE:
            filler_line_to(&f, &f.last_move); break;
        }
    }
    filler_line_to(&f, &f.last_move);
}

/* ------------------------------------------------------------------ [A.8] stroker */
/* Restates cairo 1.16 cairo-path-stroke-polygon.c (+ cairo-pen.c, cairo-spline.c's tangent decomposition and the
   rectilinear stroker of cairo-path-stroke-boxes.c).  Pinned by fuzzing against libcairo (tests/test_oracle_vs_cairo.py). */
typedef struct { pt_t *p; int n, cap; int dir; } contour_t;
typedef struct {
    pt_t ccw, point, cw;
    int64_t dvx, dvy;                       /* dev_vector (fixed deltas) */
    double dsx, dsy;                        /* unit device slope */
    double usx, usy;                        /* unit user-space slope (square caps) */
    double length;
} face_t;
typedef struct { pt_t pt; int64_t cw_dx, cw_dy, ccw_dx, ccw_dy; } pen_vertex_t;
typedef struct { int n; pen_vertex_t *v; } pen_t;
typedef struct {
    contour_t cw, ccw;
    polygon_t *g;
    const mat_t *ctm, *inv;
    int ctm_det_positive, ctm_identity;
    double half_width, miter_limit, tol, spline_cusp_tol;
    int join, cap;                          /* join: 0 miter 1 round 2 bevel; cap: 0 butt 1 round 2 square */
    int64_t contour_tol;
    pen_t pen;
    int has_bounds; pt_t b1, b2;
    pt_t first_point;
    int has_initial_sub_path, has_current_face, has_first_face;
    face_t current_face, first_face;
    int unsupported;
} stroker_t;

static void contour_reset(contour_t *c) { c->n = 0; }
static void contour_push(contour_t *c, const pt_t *p)
{
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 64; c->p = realloc(c->p, sizeof(pt_t) * c->cap); }
    c->p[c->n++] = *p;
}
static int within_tolerance(const pt_t *a, const pt_t *b, int64_t tol)
{
    /* Cairo 1.16.0's within_tolerance() starts with an unconditional `return FALSE`: fuzzing
       against libcairo shows 65/3000 differing strokes with the distance test and 0/3000 without. */
    (void)a; (void)b; (void)tol;
    return 0;
}
static void contour_add_point(stroker_t *s, contour_t *c, const pt_t *p)
{
    if (c->n && within_tolerance(p, &c->p[c->n - 1], s->contour_tol)) return;
    contour_push(c, p);
}
static void polygon_add_contour(polygon_t *g, const contour_t *c)
{
    if (c->n <= 1) return;
    for (int i = 1; i < c->n; i++) polygon_add_edge(g, &c->p[i - 1], &c->p[i], c->dir);
}
static double normalize_slope(double *dx, double *dy)
{
    double dx0 = *dx, dy0 = *dy, mag;
    if (dx0 == 0.0) { *dx = 0.0; if (dy0 > 0.0) { mag = dy0; *dy = 1.0; } else { mag = -dy0; *dy = -1.0; } }
    else if (dy0 == 0.0) { *dy = 0.0; if (dx0 > 0.0) { mag = dx0; *dx = 1.0; } else { mag = -dx0; *dx = -1.0; } }
    else { mag = hypot(dx0, dy0); *dx = dx0 / mag; *dy = dy0 / mag; }
    return mag;
}
static void compute_face(const pt_t *point, int64_t ddx, int64_t ddy, stroker_t *s, face_t *f)
{
    double sdx = (double)ddx / 256.0, sdy = (double)ddy / 256.0, fdx, fdy;
    f->length = normalize_slope(&sdx, &sdy);
    f->dsx = sdx; f->dsy = sdy;
    if (!s->ctm_identity) {
        mat_distance(s->inv, &sdx, &sdy);
        normalize_slope(&sdx, &sdy);
        if (s->ctm_det_positive) { fdx = -sdy * s->half_width; fdy = sdx * s->half_width; }
        else { fdx = sdy * s->half_width; fdy = -sdx * s->half_width; }
        mat_distance(s->ctm, &fdx, &fdy);
    } else { fdx = -sdy * s->half_width; fdy = sdx * s->half_width; }
    fx_t ox = fx_from_double(fdx), oy = fx_from_double(fdy);
    f->ccw.x = point->x + ox; f->ccw.y = point->y + oy;
    f->point = *point;
    f->cw.x = point->x - ox; f->cw.y = point->y - oy;
    f->usx = sdx; f->usy = sdy;
    f->dvx = ddx; f->dvy = ddy;
}
static int slope_compare(int64_t adx, int64_t ady, int64_t bdx, int64_t bdy)
{
    int64_t l = ady * bdx, r = bdy * adx;
    if (l != r) return l < r ? -1 : 1;
    if (adx == 0 && ady == 0 && bdx == 0 && bdy == 0) return 0;
    if (adx == 0 && ady == 0) return 1;
    if (bdx == 0 && bdy == 0) return -1;
    if (((adx ^ bdx) < 0) || ((ady ^ bdy) < 0)) return (adx > 0 || (adx == 0 && ady > 0)) ? -1 : 1;
    return 0;
}
static int slope_compare_sgn(double dx1, double dy1, double dx2, double dy2)
{
    double c = dx1 * dy2 - dx2 * dy1;
    return c > 0 ? 1 : c < 0 ? -1 : 0;
}

/* ---- pen (cairo-pen.c): a polygonal circle of radius half_width under the CTM, fine enough for `tolerance` */
static int mat_has_unity_scale(const mat_t *m)
{
    const double eps = 1.0 / 256.0;
    double det = mat_det(m);
    if (fabs(det * det - 1.0) < eps) {
        if (fabs(m->xy) < eps && fabs(m->yx) < eps) return 1;
        if (fabs(m->xx) < eps && fabs(m->yy) < eps) return 1;
    }
    return 0;
}
static double mat_circle_major_axis(const mat_t *m, double radius)
{
    if (mat_has_unity_scale(m)) return radius;
    double a = m->xx, b = m->yx, c = m->xy, d = m->yy;
    double i = a * a + b * b, j = c * c + d * d;
    double f = 0.5 * (i + j), g = 0.5 * (i - j), h = a * c + b * d;
    return radius * sqrt(f + hypot(g, h));
}
static int pen_vertices_needed(double tolerance, double radius, const mat_t *m)
{
    double major_axis = mat_circle_major_axis(m, radius);
    int n;
    if (tolerance >= 4 * major_axis) n = 1;
    else if (tolerance >= major_axis) n = 4;
    else {
        n = (int)ceil(2 * M_PI / acos(1 - tolerance / major_axis));
        if (n % 2) n++;
        if (n < 4) n = 4;
    }
    return n;
}
static void pen_init(pen_t *pen, double radius, double tolerance, const mat_t *ctm)
{
    int reflect = mat_det(ctm) < 0.0;
    pen->n = pen_vertices_needed(tolerance, radius, ctm);
    pen->v = malloc(sizeof(pen_vertex_t) * pen->n);
    for (int i = 0; i < pen->n; i++) {
        double theta = 2 * M_PI * i / (double)pen->n, dx, dy;
        dx = radius * cos(reflect ? -theta : theta);
        dy = radius * sin(reflect ? -theta : theta);
        mat_distance(ctm, &dx, &dy);
        pen->v[i].pt.x = fx_from_double(dx);
        pen->v[i].pt.y = fx_from_double(dy);
    }
    for (int i = 0; i < pen->n; i++) {
        const pen_vertex_t *prev = &pen->v[(i + pen->n - 1) % pen->n], *next = &pen->v[(i + 1) % pen->n];
        pen_vertex_t *v = &pen->v[i];
        v->cw_dx = (int64_t)v->pt.x - prev->pt.x; v->cw_dy = (int64_t)v->pt.y - prev->pt.y;
        v->ccw_dx = (int64_t)next->pt.x - v->pt.x; v->ccw_dy = (int64_t)next->pt.y - v->pt.y;
    }
}
static void pen_find_active_cw(const pen_t *pen, int64_t idx, int64_t idy, int64_t odx, int64_t ody, int *start, int *stop)
{
    int lo = 0, hi = pen->n, i = (lo + hi) >> 1;
    do {
        if (slope_compare(pen->v[i].cw_dx, pen->v[i].cw_dy, idx, idy) < 0) lo = i; else hi = i;
        i = (lo + hi) >> 1;
    } while (hi - lo > 1);
    if (slope_compare(pen->v[i].cw_dx, pen->v[i].cw_dy, idx, idy) < 0)
        if (++i == pen->n) i = 0;
    *start = i;
    if (slope_compare(odx, ody, pen->v[i].ccw_dx, pen->v[i].ccw_dy) >= 0) {
        lo = i; hi = i + pen->n; i = (lo + hi) >> 1;
        do {
            int j = i; if (j >= pen->n) j -= pen->n;
            if (slope_compare(pen->v[j].cw_dx, pen->v[j].cw_dy, odx, ody) > 0) hi = i; else lo = i;
            i = (lo + hi) >> 1;
        } while (hi - lo > 1);
        if (i >= pen->n) i -= pen->n;
    }
    *stop = i;
}
static void pen_find_active_ccw(const pen_t *pen, int64_t idx, int64_t idy, int64_t odx, int64_t ody, int *start, int *stop)
{
    int lo = 0, hi = pen->n, i = (lo + hi) >> 1;
    do {
        if (slope_compare(idx, idy, pen->v[i].ccw_dx, pen->v[i].ccw_dy) < 0) lo = i; else hi = i;
        i = (lo + hi) >> 1;
    } while (hi - lo > 1);
    if (slope_compare(idx, idy, pen->v[i].ccw_dx, pen->v[i].ccw_dy) < 0)
        if (++i == pen->n) i = 0;
    *start = i;
    if (slope_compare(pen->v[i].cw_dx, pen->v[i].cw_dy, odx, ody) <= 0) {
        lo = i; hi = i + pen->n; i = (lo + hi) >> 1;
        do {
            int j = i; if (j >= pen->n) j -= pen->n;
            if (slope_compare(odx, ody, pen->v[j].ccw_dx, pen->v[j].ccw_dy) > 0) hi = i; else lo = i;
            i = (lo + hi) >> 1;
        } while (hi - lo > 1);
        if (i >= pen->n) i -= pen->n;
    }
    *stop = i;
}
static void add_fan(stroker_t *s, int64_t idx, int64_t idy, int64_t odx, int64_t ody, const pt_t *mid, int clockwise, contour_t *c)
{
    const pen_t *pen = &s->pen;
    int start, stop;
    if (s->has_bounds && !(s->b1.x <= mid->x && mid->x <= s->b2.x && s->b1.y <= mid->y && mid->y <= s->b2.y)) return;
    if (clockwise) {
        pen_find_active_cw(pen, idx, idy, odx, ody, &start, &stop);
        while (start != stop) {
            pt_t p = { mid->x + pen->v[start].pt.x, mid->y + pen->v[start].pt.y };
            contour_add_point(s, c, &p);
            if (++start == pen->n) start = 0;
        }
    } else {
        pen_find_active_ccw(pen, idx, idy, odx, ody, &start, &stop);
        while (start != stop) {
            pt_t p = { mid->x + pen->v[start].pt.x, mid->y + pen->v[start].pt.y };
            contour_add_point(s, c, &p);
            if (start-- == 0) start += pen->n;
        }
    }
}
static int join_is_clockwise(const face_t *in, const face_t *out) { return slope_compare(in->dvx, in->dvy, out->dvx, out->dvy) < 0; }

static void inner_join(stroker_t *s, const face_t *in, const face_t *out, int clockwise)
{
    contour_t *inner = clockwise ? &s->ccw : &s->cw;
    const pt_t *outpt = clockwise ? &out->ccw : &out->cw;
    contour_add_point(s, inner, &in->point);
    contour_add_point(s, inner, outpt);
}
static void inner_close(stroker_t *s, const face_t *in, const face_t *out)
{
    int clockwise = join_is_clockwise(in, out);
    contour_t *inner = clockwise ? &s->ccw : &s->cw;
    const pt_t *inpt = clockwise ? &out->ccw : &out->cw;
    contour_add_point(s, inner, &in->point);
    contour_add_point(s, inner, inpt);
    inner->p[0] = inner->p[inner->n - 1];
}
/* the miter point when the limit allows it and it lies between the two faces; 0 otherwise */
static int miter_point(stroker_t *s, const face_t *in, const face_t *out, const pt_t *inpt, const pt_t *outpt, pt_t *res)
{
    double in_dot_out = in->dsx * out->dsx + in->dsy * out->dsy, ml = s->miter_limit;
    if (2 <= ml * ml * (1 + in_dot_out)) {
        double x1 = fx_to_double(inpt->x), y1 = fx_to_double(inpt->y), dx1 = in->dsx, dy1 = in->dsy;
        double x2 = fx_to_double(outpt->x), y2 = fx_to_double(outpt->y), dx2 = out->dsx, dy2 = out->dsy;
        double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
        double mx = fabs(dy1) >= fabs(dy2) ? (my - y1) * dx1 / dy1 + x1 : (my - y2) * dx2 / dy2 + x2;
        double ix = fx_to_double(in->point.x), iy = fx_to_double(in->point.y);
        double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
        if (slope_compare_sgn(fdx1, fdy1, mdx, mdy) != slope_compare_sgn(fdx2, fdy2, mdx, mdy)) {
            res->x = fx_from_double(mx); res->y = fx_from_double(my);
            return 1;
        }
    }
    return 0;
}
static void outer_close(stroker_t *s, const face_t *in, const face_t *out)
{
    if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x && in->ccw.y == out->ccw.y) return;
    int clockwise = join_is_clockwise(in, out);
    const pt_t *inpt = clockwise ? &in->cw : &in->ccw, *outpt = clockwise ? &out->cw : &out->ccw;
    contour_t *outer = clockwise ? &s->cw : &s->ccw;
    if (within_tolerance(inpt, outpt, s->contour_tol)) { outer->p[0] = outer->p[outer->n - 1]; return; }
    if (s->join == 1 && (in->dsx * out->dsx + in->dsy * out->dsy) < s->spline_cusp_tol) {
        add_fan(s, in->dvx, in->dvy, out->dvx, out->dvy, &in->point, clockwise, outer);
    } else if (s->join != 2) {                           /* MITER, or a ROUND join too flat for a fan: cairo falls through to the
                                                            miter code; its tip may leave the approximate extents (see op_bounds) */
        pt_t p;
        if (miter_point(s, in, out, inpt, outpt, &p)) {
            outer->p[outer->n - 1] = p;
            outer->p[0] = p;
            return;
        }
    }
    contour_add_point(s, outer, outpt);
}
static void outer_join(stroker_t *s, const face_t *in, const face_t *out, int clockwise)
{
    if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x && in->ccw.y == out->ccw.y) return;
    const pt_t *inpt = clockwise ? &in->cw : &in->ccw, *outpt = clockwise ? &out->cw : &out->ccw;
    contour_t *outer = clockwise ? &s->cw : &s->ccw;
    if (s->join == 1) {                                  /* ROUND: fan around the common midpoint */
        add_fan(s, in->dvx, in->dvy, out->dvx, out->dvy, &in->point, clockwise, outer);
    } else if (s->join == 0) {                           /* MITER */
        pt_t p;
        if (miter_point(s, in, out, inpt, outpt, &p)) { outer->p[outer->n - 1] = p; return; }
    }
    contour_add_point(s, outer, outpt);                  /* BEVEL, a rejected miter, or the end of the fan */
}
static void add_cap(stroker_t *s, const face_t *f, contour_t *c)
{
    if (s->cap == 1) {                                   /* ROUND */
        add_fan(s, f->dvx, f->dvy, -f->dvx, -f->dvy, &f->point, 0, c);
    } else if (s->cap == 2) {                            /* SQUARE */
        double dx = f->usx * s->half_width, dy = f->usy * s->half_width;
        mat_distance(s->ctm, &dx, &dy);
        fx_t vx = fx_from_double(dx), vy = fx_from_double(dy);
        pt_t p = { f->ccw.x + vx, f->ccw.y + vy };
        contour_add_point(s, c, &p);
        p.x = f->cw.x + vx; p.y = f->cw.y + vy;
        contour_add_point(s, c, &p);
    }
    contour_add_point(s, c, &f->cw);
}
static void add_leading_cap(stroker_t *s, const face_t *face, contour_t *c)
{
    face_t r = *face;
    r.usx = -r.usx; r.usy = -r.usy; r.dvx = -r.dvx; r.dvy = -r.dvy;
    pt_t t = r.cw; r.cw = r.ccw; r.ccw = t;
    add_cap(s, &r, c);
}
static void add_caps(stroker_t *s)
{
    if (s->has_initial_sub_path && !s->has_first_face && !s->has_current_face && s->cap == 1) {
        /* degenerate sub-path with round caps: a dot */
        face_t face;
        compute_face(&s->first_point, 256, 0, s, &face);
        add_leading_cap(s, &face, &s->ccw);
        add_cap(s, &face, &s->ccw);
        if (s->ccw.n) { pt_t first = s->ccw.p[0]; contour_push(&s->ccw, &first); }
        polygon_add_contour(s->g, &s->ccw);
        contour_reset(&s->ccw);
    } else {
        if (s->has_current_face) add_cap(s, &s->current_face, &s->ccw);
        polygon_add_contour(s->g, &s->ccw);
        contour_reset(&s->ccw);
        if (s->has_first_face) {
            contour_push(&s->ccw, &s->first_face.cw);
            add_leading_cap(s, &s->first_face, &s->ccw);
            polygon_add_contour(s->g, &s->ccw);
            contour_reset(&s->ccw);
        }
        polygon_add_contour(s->g, &s->cw);
        contour_reset(&s->cw);
    }
}
static void stroker_move_to(stroker_t *s, const pt_t *p)
{
    add_caps(s);
    s->has_first_face = s->has_current_face = s->has_initial_sub_path = 0;
    s->first_point = *p;
    s->current_face.point = *p;
}
static void stroker_line_to(void *closure, const pt_t *point)
{
    stroker_t *s = closure;
    face_t start;
    pt_t *p1 = &s->current_face.point;
    s->has_initial_sub_path = 1;
    if (p1->x == point->x && p1->y == point->y) return;
    int64_t ddx = (int64_t)point->x - p1->x, ddy = (int64_t)point->y - p1->y;
    compute_face(p1, ddx, ddy, s, &start);
    if (s->has_current_face) {
        int cw = slope_compare(s->current_face.dvx, s->current_face.dvy, start.dvx, start.dvy);
        if (cw) {
            cw = cw < 0;
            if (!within_tolerance(&s->current_face.ccw, &start.ccw, s->contour_tol) ||
                !within_tolerance(&s->current_face.cw, &start.cw, s->contour_tol)) {
                outer_join(s, &s->current_face, &start, cw);
                inner_join(s, &s->current_face, &start, cw);
            }
        }
    } else {
        if (!s->has_first_face) { s->first_face = start; s->has_first_face = 1; }
        s->has_current_face = 1;
        contour_add_point(s, &s->cw, &start.cw);
        contour_add_point(s, &s->ccw, &start.ccw);
    }
    s->current_face = start;
    s->current_face.point = *point;
    s->current_face.ccw.x += (fx_t)ddx; s->current_face.ccw.y += (fx_t)ddy;
    s->current_face.cw.x += (fx_t)ddx; s->current_face.cw.y += (fx_t)ddy;
    contour_add_point(s, &s->cw, &s->current_face.cw);
    contour_add_point(s, &s->ccw, &s->current_face.ccw);
}
/* one point of a flattened curve with the curve's tangent there (cairo spline_to) */
static void stroker_spline_to(stroker_t *s, const pt_t *point, int64_t tdx, int64_t tdy)
{
    face_t face;
    if ((tdx | tdy) == 0) {                              /* cusp: turn around with a fan */
        face = s->current_face;
        face.usx = -face.usx; face.usy = -face.usy; face.dvx = -face.dvx; face.dvy = -face.dvy;
        pt_t t = face.cw; face.cw = face.ccw; face.ccw = t;
        int clockwise = join_is_clockwise(&s->current_face, &face);
        contour_t *outer = clockwise ? &s->cw : &s->ccw;
        add_fan(s, s->current_face.dvx, s->current_face.dvy, face.dvx, face.dvy, &s->current_face.point, clockwise, outer);
    } else {
        compute_face(point, tdx, tdy, s, &face);
        if ((face.dsx * s->current_face.dsx + face.dsy * s->current_face.dsy) < s->spline_cusp_tol) {
            int clockwise = join_is_clockwise(&s->current_face, &face);
            s->current_face.cw.x += face.point.x - s->current_face.point.x;
            s->current_face.cw.y += face.point.y - s->current_face.point.y;
            contour_add_point(s, &s->cw, &s->current_face.cw);
            s->current_face.ccw.x += face.point.x - s->current_face.point.x;
            s->current_face.ccw.y += face.point.y - s->current_face.point.y;
            contour_add_point(s, &s->ccw, &s->current_face.ccw);
            contour_t *outer = clockwise ? &s->cw : &s->ccw;
            add_fan(s, s->current_face.dvx, s->current_face.dvy, face.dvx, face.dvy, &s->current_face.point, clockwise, outer);
        }
        contour_add_point(s, &s->cw, &face.cw);
        contour_add_point(s, &s->ccw, &face.ccw);
    }
    s->current_face = face;
}
typedef struct { stroker_t *s; pt_t last; } spline_tan_t;
static void spline_tan_add(spline_tan_t *sp, const pt_t *p, const pt_t *knot)
{
    if (p->x == sp->last.x && p->y == sp->last.y) return;
    sp->last = *p;
    stroker_spline_to(sp->s, p, (int64_t)knot->x - p->x, (int64_t)knot->y - p->y);
}
static void spline_tan_decompose_into(knots_t *s1, double tol2, spline_tan_t *out)
{
    if (spline_error_squared(s1) < tol2) { spline_tan_add(out, &s1->a, &s1->b); return; }
    knots_t s2;
    pt_t ab, bc, cd, abbc, bccd, fin;
    lerp_half(&s1->a, &s1->b, &ab); lerp_half(&s1->b, &s1->c, &bc); lerp_half(&s1->c, &s1->d, &cd);
    lerp_half(&ab, &bc, &abbc); lerp_half(&bc, &cd, &bccd); lerp_half(&abbc, &bccd, &fin);
    s2.a = fin; s2.b = bccd; s2.c = cd; s2.d = s1->d;
    s1->b = ab; s1->c = abbc; s1->d = fin;
    spline_tan_decompose_into(s1, tol2, out);
    spline_tan_decompose_into(&s2, tol2, out);
}
static void stroker_curve_to(stroker_t *s, const pt_t *b, const pt_t *c, const pt_t *d)
{
    const pt_t a = s->current_face.point;
    if (s->has_bounds && !spline_intersects(&a, b, c, d, &s->b1, &s->b2)) { stroker_line_to(s, d); return; }
    /* _cairo_spline_init: initial / final slopes; degenerate splines are lines */
    if (a.x == b->x && a.y == b->y && c->x == d->x && c->y == d->y) { stroker_line_to(s, d); return; }
    int64_t isx, isy, fsx, fsy;
    if (a.x != b->x || a.y != b->y) { isx = (int64_t)b->x - a.x; isy = (int64_t)b->y - a.y; }
    else if (a.x != c->x || a.y != c->y) { isx = (int64_t)c->x - a.x; isy = (int64_t)c->y - a.y; }
    else if (a.x != d->x || a.y != d->y) { isx = (int64_t)d->x - a.x; isy = (int64_t)d->y - a.y; }
    else { stroker_line_to(s, d); return; }
    if (c->x != d->x || c->y != d->y) { fsx = (int64_t)d->x - c->x; fsy = (int64_t)d->y - c->y; }
    else if (b->x != d->x || b->y != d->y) { fsx = (int64_t)d->x - b->x; fsy = (int64_t)d->y - b->y; }
    else { stroker_line_to(s, d); return; }
    face_t face;
    compute_face(&a, isx, isy, s, &face);
    if (s->has_current_face) {
        int clockwise = join_is_clockwise(&s->current_face, &face);
        outer_join(s, &s->current_face, &face, clockwise);
        inner_join(s, &s->current_face, &face, clockwise);
    } else {
        if (!s->has_first_face) { s->first_face = face; s->has_first_face = 1; }
        s->has_current_face = 1;
        contour_add_point(s, &s->cw, &face.cw);
        contour_add_point(s, &s->ccw, &face.ccw);
    }
    s->current_face = face;
    spline_tan_t sp = { s, a };
    knots_t k = { a, *b, *c, *d };
    spline_tan_decompose_into(&k, s->tol * s->tol, &sp);
    stroker_spline_to(s, d, fsx, fsy);
}
static void stroker_close_path(stroker_t *s)
{
    stroker_line_to(s, &s->first_point);
    if (s->has_first_face && s->has_current_face) {
        outer_close(s, &s->current_face, &s->first_face);
        inner_close(s, &s->current_face, &s->first_face);
        polygon_add_contour(s->g, &s->cw);
        polygon_add_contour(s->g, &s->ccw);
        contour_reset(&s->cw); contour_reset(&s->ccw);
    } else
        add_caps(s);
    s->has_initial_sub_path = 0;
    s->has_first_face = 0;
    s->has_current_face = 0;
}
static int path_stroke_to_polygon(const path_t *p, stroker_t *s)
{
    const pt_t *pts = p->pts;
    s->spline_cusp_tol = 1 - s->tol / s->half_width;
    s->spline_cusp_tol *= s->spline_cusp_tol;
    s->spline_cusp_tol *= 2;
    s->spline_cusp_tol -= 1;
    s->pen.n = 0; s->pen.v = NULL;
    pen_init(&s->pen, s->half_width, s->tol, s->ctm);    /* >= 4 vertices: the caller dropped strokes whose pen degenerates */
    for (int i = 0; i < p->nops; i++) {
        switch (p->ops[i]) {
        case OP_MOVE: stroker_move_to(s, &pts[0]); pts += 1; break;
        case OP_LINE: stroker_line_to(s, &pts[0]); pts += 1; break;
        case OP_CURVE: stroker_curve_to(s, &pts[0], &pts[1], &pts[2]); pts += 3; break;
        case OP_CLOSE: stroker_close_path(s); break;
        }
    }
    add_caps(s);
    free(s->pen.v); s->pen.v = NULL;
    return s->unsupported;
}

/* ---- rectilinear strokes (cairo-path-stroke-boxes.c, undashed): one box per segment, lengthened for joins / square caps;
        the union of the boxes is what gets painted (cairo tessellates them into disjoint boxes, A.6 sums exact areas) */
typedef struct { pt_t p1, p2; int horizontal; } rseg_t;
typedef struct {
    polygon_t *g; fx_t hx, hy; int cap;
    rseg_t *seg; int n, capn;
    pt_t cur, first; int open_sub_path;
} rstroker_t;
static void rs_add_box(polygon_t *g, fx_t x1, fx_t y1, fx_t x2, fx_t y2)
{
    if (x1 == x2 || y1 == y2) return;
    pt_t a = { x1, y1 }, b = { x1, y2 }, c2 = { x2, y1 }, d = { x2, y2 };
    polygon_add_edge(g, &a, &b, 1);
    polygon_add_edge(g, &c2, &d, -1);
}
static void rs_emit_segments(rstroker_t *r)
{
    for (int i = 0; i < r->n; i++) {
        pt_t a = r->seg[i].p1, b = r->seg[i].p2;
        int j = i == 0 ? r->n - 1 : i - 1;
        int lengthen_initial = r->seg[i].horizontal != r->seg[j].horizontal;
        j = i == r->n - 1 ? 0 : i + 1;
        int lengthen_final = r->seg[i].horizontal != r->seg[j].horizontal;
        if (r->open_sub_path) {
            if (i == 0) lengthen_initial = r->cap != 0;
            if (i == r->n - 1) lengthen_final = r->cap != 0;
        }
        if (lengthen_initial | lengthen_final) {
            if (a.y == b.y) {
                if (a.x < b.x) { if (lengthen_initial) a.x -= r->hx; if (lengthen_final) b.x += r->hx; }
                else { if (lengthen_initial) a.x += r->hx; if (lengthen_final) b.x -= r->hx; }
            } else {
                if (a.y < b.y) { if (lengthen_initial) a.y -= r->hy; if (lengthen_final) b.y += r->hy; }
                else { if (lengthen_initial) a.y += r->hy; if (lengthen_final) b.y -= r->hy; }
            }
        }
        if (a.y == b.y) { a.y -= r->hy; b.y += r->hy; } else { a.x -= r->hx; b.x += r->hx; }
        rs_add_box(r->g, a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y);
    }
    r->n = 0;
}
static void rs_line_to(rstroker_t *r, const pt_t *b)
{
    const pt_t a = r->cur;
    if (a.x == b->x && a.y == b->y) return;
    if (r->n == r->capn) { r->capn = r->capn ? 2 * r->capn : 16; r->seg = realloc(r->seg, sizeof(rseg_t) * r->capn); }
    r->seg[r->n].p1 = a; r->seg[r->n].p2 = *b; r->seg[r->n].horizontal = a.y == b->y; r->n++;
    r->cur = *b;
    r->open_sub_path = 1;
}
/* returns 0 when cairo's rectilinear stroker declines (the caller then uses the polygon stroker) */
static int path_stroke_rectilinear(const path_t *p, const mat_t *ctm, double line_width, int join, int cap, double miter_limit, polygon_t *g)
{
    if (join != 0) return 0;
    if (miter_limit < M_SQRT2) return 0;
    if (!(cap == 0 || cap == 2)) return 0;
    if (!(ctm->xy == 0.0 && ctm->yx == 0.0)) return 0;   /* _cairo_matrix_is_scale */
    rstroker_t r; memset(&r, 0, sizeof(r));
    r.g = g; r.cap = cap;
    r.hx = fx_from_double(fabs(ctm->xx) * line_width / 2.0);
    r.hy = fx_from_double(fabs(ctm->yy) * line_width / 2.0);
    const pt_t *pts = p->pts;
    for (int i = 0; i < p->nops; i++) {
        switch (p->ops[i]) {
        case OP_MOVE: rs_emit_segments(&r); r.cur = r.first = pts[0]; r.open_sub_path = 0; pts += 1; break;
        case OP_LINE: rs_line_to(&r, &pts[0]); pts += 1; break;
        case OP_CURVE: pts += 3; break;                 /* cannot happen: the path is rectilinear */
        case OP_CLOSE:
            if (r.open_sub_path) { rs_line_to(&r, &r.first); r.open_sub_path = 0; rs_emit_segments(&r); }
            break;
        }
    }
    rs_emit_segments(&r);
    free(r.seg);
    return 1;
}

/* ------------------------------------------------------------------ sources + surface */
enum { SRC_SOLID = 0, SRC_RADIAL = 1, SRC_LINEAR = 2, SRC_SURFACE = 3 };
typedef struct { double t; double r, g, b, a; } stop_t;
typedef struct {
    int kind;
    uint32_t pixel;                          /* SOLID: premultiplied ARGB */
    mat_t inv;                               /* device -> pattern space */
    double cx0, cy0, r0, cx1, cy1, r1;       /* RADIAL / LINEAR (x0,y0,x1,y1) */
    stop_t *stops; int nstops;
    const uint32_t *tex; int tw, th, extend; /* SURFACE: premultiplied ARGB, extend 0 none / 1 repeat */
    /* SURFACE, CAIRO_FILTER_GOOD when it is not downgraded to bilinear: pixman separable convolution tables */
    int good, cw, ch, xbits, ybits;
    int32_t *xpar, *ypar;                    /* (1 << bits) phases x width taps, 16.16 */
    /* SURFACE: what pixman is given for one drawing operation (source_prepare_pixman): 16.16 transform + integer offset */
    int64_t pm[2][3]; int pox, poy;
    /* RADIAL as pixman holds it (source_prepare_radial): circles in 16.16 after cairo's fit-to-range scaling, the quadratic's
       constant terms, and one colour ramp per interval between consecutive stops (sentinels at both ends: PAD) */
    int64_t g_c1x, g_c1y, g_c1r, g_dx, g_dy, g_dr;
    double g_a, g_inva, g_mindr;
    int g_n;                                 /* intervals = stops + 1 */
    int64_t *g_x;                            /* g_n + 1 boundaries: INT32_MIN, stop offsets (16.16), INT32_MAX */
    float *g_ramp;                           /* per interval: a_s, a_b, r_s, r_b, g_s, g_b, b_s, b_b */
} source_t;

/* ---- the pattern matrix as pixman gets it (cairo-matrix.c _cairo_matrix_to_pixman_matrix_offset, cairo-image-source.c
        _pixman_image_set_properties): an integer translation is split off so that what remains is small, the matrix is rounded
        to 16.16 (ties to even) and its translation is then corrected so that the centre of the operation's rectangle maps where
        the double matrix would put it.  x0..y1 is that rectangle (bounded extents of the fill / stroke, in pixels). */
static int32_t f16_from_double(double d) { return (int32_t)(int64_t)nearbyint(d * 65536.0); }
static void source_prepare_pixman_m(source_t *s, mat_t m, int x0, int y0, int x1, int y1)
{
    const double xc = x0 + (x1 - x0) / 2., yc = y0 + (y1 - y0) / 2.;
    s->pox = s->poy = 0;
    if (m.x0 != 0.0 || m.y0 != 0.0) {
        /* spread the offset between the integer part and the matrix: solutions of |x| = |x*xx + y*xy + x0|, |y| = |...| */
        double tx = m.x0, ty = m.y0, norm = fmax(fabs(tx), fabs(ty));
        for (int i = -1; i < 2; i += 2)
            for (int j = -1; j < 2; j += 2) {
                double den = (m.xx + i) * (m.yy + j) - m.xy * m.yx;
                if (fabs(den) < DBL_EPSILON) continue;
                double x = m.y0 * m.xy - m.x0 * (m.yy + j), y = m.x0 * m.yx - m.y0 * (m.xx + i);
                den = 1 / den; x *= den; y *= den;
                double new_norm = fmax(fabs(x), fabs(y));
                if (norm > new_norm) { norm = new_norm; tx = x; ty = y; }
            }
        tx = floor(tx); ty = floor(ty);
        s->pox = (int)-tx; s->poy = (int)-ty;
        mat_t t = { 1, 0, 0, 1, tx, ty };
        mat_multiply(&m, &t, &m);                                   /* cairo_matrix_translate */
    }
    s->pm[0][0] = f16_from_double(m.xx); s->pm[0][1] = f16_from_double(m.xy); s->pm[0][2] = f16_from_double(m.x0);
    s->pm[1][0] = f16_from_double(m.yx); s->pm[1][1] = f16_from_double(m.yy); s->pm[1][2] = f16_from_double(m.y0);
    if (mat_has_unity_scale(&m)) return;
    mat_t inv = m;
    if (!mat_invert_cairo(&inv)) return;
    for (int it = 0; it < 5; it++) {
        const int64_t vx = f16_from_double(xc), vy = f16_from_double(yc);
        const int64_t tx = (s->pm[0][0] * vx + s->pm[0][1] * vy + s->pm[0][2] * 65536 + 0x8000) >> 16;
        const int64_t ty = (s->pm[1][0] * vx + s->pm[1][1] * vy + s->pm[1][2] * 65536 + 0x8000) >> 16;
        /* cairo-matrix.c "If we can't transform the reference point, skip the adjustment": pixman_transform_point_3d (pixman-matrix.c)
           reports failure when a coordinate of the result does not fit 16.16 -- gradients scaled to +-16383 whose centre of
           operation maps beyond +-32768 */
        if (tx != (int32_t)tx || ty != (int32_t)ty) return;
        double x = (double)tx / 65536.0, y = (double)ty / 65536.0;
        mat_point(&inv, &x, &y);
        x -= xc; y -= yc;
        mat_distance(&m, &x, &y);
        const int32_t dx = f16_from_double(x), dy = f16_from_double(y);
        s->pm[0][2] -= dx; s->pm[1][2] -= dy;
        if (dx == 0 && dy == 0) break;
    }
}
static void source_prepare_pixman(source_t *s, int x0, int y0, int x1, int y1)
{
    source_prepare_pixman_m(s, s->inv, x0, y0, x1, y1);
    if (getenv("SWFO_TRACE_SOURCE"))
        fprintf(stderr, "surface transform %d %d %d | %d %d %d offset %d %d rect %d %d %d %d\n", (int)s->pm[0][0], (int)s->pm[0][1], (int)s->pm[0][2],
                (int)s->pm[1][0], (int)s->pm[1][1], (int)s->pm[1][2], s->pox, s->poy, x0, y0, x1, y1);
}

/* ---- radial gradients exactly as cairo 1.16 + pixman 0.40 compute them (cairo-image-source.c _pixman_image_for_gradient,
        cairo-pattern.c _cairo_gradient_pattern_fit_to_range, pixman-radial-gradient.c, pixman-gradient-walker.c): the circles are
        scaled into +-16383 (the matrix takes the inverse factor) and rounded to 16.16; B and C of the quadratic are exact 64-bit
        integers of the pixel's 16.16 sample position; the root is taken in doubles; the colour ramp between two stops is evaluated
        in single precision and premultiplied there.  Extend is PAD (cairo's default for gradients). */
static void source_prepare_radial(source_t *s, int x0, int y0, int x1, int y1)
{
    double c0x = s->cx0, c0y = s->cy0, c0r = s->r0, c1x = s->cx1, c1y = s->cy1, c1r = s->r1;
    double dim = fabs(c0x);
    dim = fmax(dim, fabs(c0y)); dim = fmax(dim, fabs(c0r)); dim = fmax(dim, fabs(c1x)); dim = fmax(dim, fabs(c1y)); dim = fmax(dim, fabs(c1r));
    dim = fmax(dim, fabs(c0x - c1x)); dim = fmax(dim, fabs(c0y - c1y)); dim = fmax(dim, fabs(c0r - c1r));
    mat_t m = s->inv;
    if (dim > 16383.0) {                                             /* PIXMAN_MAX_INT >> 1 */
        dim = 16383.0 / dim;
        c0x *= dim; c0y *= dim; c0r *= dim; c1x *= dim; c1y *= dim; c1r *= dim;
        mat_t sc = { dim, 0, 0, dim, 0, 0 };
        mat_multiply(&m, &s->inv, &sc);
    }
    source_prepare_pixman_m(s, m, x0, y0, x1, y1);
    if (getenv("SWFO_TRACE_SOURCE"))                                 /* diagnostic: compare with what libcairo hands pixman */
        fprintf(stderr, "radial transform %d %d %d | %d %d %d offset %d %d rect %d %d %d %d\n", (int)s->pm[0][0], (int)s->pm[0][1], (int)s->pm[0][2],
                (int)s->pm[1][0], (int)s->pm[1][1], (int)s->pm[1][2], s->pox, s->poy, x0, y0, x1, y1);
    s->g_c1x = f16_from_double(c0x); s->g_c1y = f16_from_double(c0y); s->g_c1r = f16_from_double(c0r);
    s->g_dx = f16_from_double(c1x) - s->g_c1x; s->g_dy = f16_from_double(c1y) - s->g_c1y; s->g_dr = f16_from_double(c1r) - s->g_c1r;
    s->g_a = (double)(s->g_dx * s->g_dx + s->g_dy * s->g_dy - s->g_dr * s->g_dr);
    s->g_inva = s->g_a != 0 ? 1. * 65536 / s->g_a : 0;
    s->g_mindr = -1. * 65536 * (double)s->g_c1r;
    /* stops with the two PAD sentinels; one ramp per interval (gradient_walker_reset) */
    const int n = s->nstops;
    free(s->g_x); free(s->g_ramp);
    s->g_n = n + 1;
    s->g_x = malloc(sizeof(int64_t) * (size_t)(n + 2));
    s->g_ramp = malloc(sizeof(float) * 8 * (size_t)(n + 1));
    uint16_t (*col)[4] = malloc(sizeof(uint16_t[4]) * (size_t)(n + 2));
    for (int i = 0; i < n; i++) {
        s->g_x[i + 1] = f16_from_double(s->stops[i].t);
        col[i + 1][0] = (uint16_t)(s->stops[i].a * 65535.0 + 0.5); col[i + 1][1] = (uint16_t)(s->stops[i].r * 65535.0 + 0.5);
        col[i + 1][2] = (uint16_t)(s->stops[i].g * 65535.0 + 0.5); col[i + 1][3] = (uint16_t)(s->stops[i].b * 65535.0 + 0.5);
    }
    s->g_x[0] = INT32_MIN; s->g_x[n + 1] = INT32_MAX;
    if (n) { memcpy(col[0], col[1], sizeof col[0]); memcpy(col[n + 1], col[n], sizeof col[0]); }
    for (int k = 0; k <= n && n; k++) {
        const int64_t left_x = s->g_x[k], right_x = s->g_x[k + 1];
        const uint16_t *lc = col[k], *rc = col[k + 1];
        float *w = s->g_ramp + 8 * k;
        const float lx = left_x * (1.0f / 65536.0f), rx = right_x * (1.0f / 65536.0f);
        for (int ch = 0; ch < 4; ch++) {
            const float l = lc[ch] * (1.0f / 257.0f), r = rc[ch] * (1.0f / 257.0f);
            if ((-FLT_MIN < (rx - lx) && (rx - lx) < FLT_MIN) || left_x == INT32_MIN || right_x == INT32_MAX) {
                w[2 * ch] = 0.0f; w[2 * ch + 1] = (l + r) / 510.0f;
            } else {
                const float w_rec = 1.0f / (rx - lx);
                w[2 * ch + 1] = (l * rx - r * lx) * w_rec * (1.0f / 255.0f);
                w[2 * ch] = (r - l) * w_rec * (1.0f / 255.0f);
            }
        }
    }
    free(col);
}
static uint32_t radial_walker_pixel(const source_t *s, int64_t x)
{
    /* the interval: first stop with x < stop.x ends it (a position equal to a stop belongs to the interval on its right) */
    int k = 0;
    while (k < s->g_n - 1 && !(x < s->g_x[k + 1])) k++;
    const float *w = s->g_ramp + 8 * k;
    const float y = x * (1.0f / 65536.0f);
    const float fa = 255.f * (w[0] * y + w[1]);
    const float fr = fa * (w[2] * y + w[3]), fg = fa * (w[4] * y + w[5]), fb = fa * (w[6] * y + w[7]);
    return (((uint32_t)(fa + .5f) << 24) & 0xff000000u) | (((uint32_t)(fr + .5f) << 16) & 0x00ff0000u) |
           (((uint32_t)(fg + .5f) << 8) & 0x0000ff00u) | ((uint32_t)(fb + .5f) & 0x000000ffu);
}
static void source_pixman_position(const source_t *s, int px, int py, int64_t *vx, int64_t *vy);
static uint32_t sample_radial_pixman(const source_t *s, int px, int py)
{
    if (!s->nstops) return 0;
    int64_t vx, vy;
    source_pixman_position(s, px, py, &vx, &vy);
    vx -= s->g_c1x; vy -= s->g_c1y;
    const int64_t bi = vx * s->g_dx + vy * s->g_dy + s->g_c1r * s->g_dr;
    const int64_t ci = vx * vx + vy * vy - s->g_c1r * s->g_c1r;
    const double a = s->g_a, b = (double)bi, c = (double)ci, dr = (double)s->g_dr;
    if (a == 0) {
        if (b == 0) return 0;
        const double t = 65536 / 2 * c / b;
        if (t * dr >= s->g_mindr) return radial_walker_pixel(s, (int64_t)t);
        return 0;
    }
    const double discr = b * b + a * -c;
    if (discr >= 0) {
        const double sq = sqrt(discr), t0 = (b + sq) * s->g_inva, t1 = (b - sq) * s->g_inva;
        if (t0 * dr >= s->g_mindr) return radial_walker_pixel(s, (int64_t)t0);
        else if (t1 * dr >= s->g_mindr) return radial_walker_pixel(s, (int64_t)t1);
    }
    return 0;
}

/* pixman's sample position of destination pixel (px, py), 16.16 (pixman_transform_point_3d of the pixel centre; stepping along
   the scanline by the matrix column is exact, so every pixel can be evaluated by itself) */
static void source_pixman_position(const source_t *s, int px, int py, int64_t *vx, int64_t *vy)
{
    const int64_t X = ((int64_t)(px + s->pox) << 16) + 0x8000, Y = ((int64_t)(py + s->poy) << 16) + 0x8000;
    *vx = (s->pm[0][0] * X + s->pm[0][1] * Y + s->pm[0][2] * 65536 + 0x8000) >> 16;
    *vy = (s->pm[1][0] * X + s->pm[1][1] * Y + s->pm[1][2] * 65536 + 0x8000) >> 16;
}

/* ---- CAIRO_FILTER_GOOD for surface patterns (cairo-pattern.c _cairo_pattern_analyze_filter, cairo-image-source.c
        create_separable_convolution, pixman bits_image_fetch_pixel_separable_convolution): pixman's integer tables and
        accumulation at pixman's own 16.16 sample positions. */
static double good_box_kernel(double x, double r) { return fmax(0.0, fmin(fmin(r, 1.0), fmin((r + 1) / 2 - x, (r + 1) / 2 + x))); }
static int good_box_width(double r) { return r < 1.0 ? 2 : (int)ceil(r + 1); }
static void good_get_filter(double r, int width, int subsample, int32_t *out)
{
    int n_phases = 1 << subsample;
    double step = 1.0 / n_phases;
    int32_t *p = out;
    if (width <= 1) { for (int i = 0; i < n_phases; i++) *p++ = 65536; return; }
    for (int i = 0; i < n_phases; i++) {
        double frac = (i + .5) * step;
        double x1 = ceil(frac - width / 2.0 - 0.5) - frac + 0.5;      /* centre of the left-most pixel */
        double total = 0;
        int32_t new_total = 0;
        for (int j = 0; j < width; j++) { double v = good_box_kernel(x1 + j, r); total += v; p[j] = (int32_t)(v * 65536.0); }
        total = 1 / total;
        for (int j = 0; j < width; j++) new_total += (p[j] = (int32_t)(p[j] * total));
        p[width / 2] += 65536 - new_total;                           /* any error goes on the centre pixel */
        p += width;
    }
}
static int good_use_bilinear(double x, double y, double t)
{
    double h = x * x + y * y;                                        /* this is the inverse (device -> pattern) matrix */
    if (h < 1.0 / (0.75 * 0.75)) return 1;                           /* scale > .75 */
    if ((h > 3.99 && h < 4.01) && !fx_from_double(x * y) && (fx_from_double(t) & 255) == 0) return 1;   /* exactly 1/2, axis-parallel, integer offset */
    return 0;
}
static void source_setup_filter(source_t *s)
{
    free(s->xpar); free(s->ypar); s->xpar = s->ypar = NULL; s->good = 0;
    if (s->kind != SRC_SURFACE) return;
    const mat_t *m = &s->inv;
    if (good_use_bilinear(m->xx, m->xy, m->x0) && good_use_bilinear(m->yx, m->yy, m->y0)) return;
    double dx = hypot(m->xx, m->xy), dy = hypot(m->yx, m->yy);
    if (dx > 16.0) dx = 16.0;
    if (dy > 16.0) dy = 16.0;
    if (dx < 1.0 / 0.75) dx = 1.0;                                   /* match the bilinear filter for scales > .75 */
    if (dy < 1.0 / 0.75) dy = 1.0;
    s->cw = good_box_width(dx); s->xbits = 0;
    if (s->cw > 1) while (dx * (1 << s->xbits) <= 128.0) s->xbits++;
    s->ch = good_box_width(dy); s->ybits = 0;
    if (s->ch > 1) while (dy * (1 << s->ybits) <= 128.0) s->ybits++;
    s->xpar = malloc(sizeof(int32_t) * (size_t)(s->cw << s->xbits));
    s->ypar = malloc(sizeof(int32_t) * (size_t)(s->ch << s->ybits));
    good_get_filter(dx, s->cw, s->xbits, s->xpar);
    good_get_filter(dy, s->ch, s->ybits, s->ypar);
    s->good = 1;
}
static uint32_t sample_good(const source_t *s, int64_t x, int64_t y)
{
    const int xsh = 16 - s->xbits, ysh = 16 - s->ybits;
    const int64_t x_off = (((int64_t)s->cw << 16) - 65536) >> 1, y_off = (((int64_t)s->ch << 16) - 65536) >> 1;
    /* round to the middle of the closest phase */
    x = ((x >> xsh) << xsh) + ((1 << xsh) >> 1);
    y = ((y >> ysh) << ysh) + ((1 << ysh) >> 1);
    const int px = (int)((x & 0xffff) >> xsh), py = (int)((y & 0xffff) >> ysh);
    const int32_t *yp = s->ypar + py * s->ch;
    const int x1 = (int)((x - 1 - x_off) >> 16), y1 = (int)((y - 1 - y_off) >> 16);
    int64_t sr = 0, sg = 0, sb = 0, sa = 0;
    for (int i = y1; i < y1 + s->ch; i++) {
        int64_t fy = *yp++;
        const int32_t *xp = s->xpar + px * s->cw;
        if (!fy) continue;
        for (int j = x1; j < x1 + s->cw; j++) {
            int32_t fx = *xp++;
            if (!fx) continue;
            int rx = j, ry = i;
            uint32_t pixel;
            if (s->extend == 1) { rx = ((rx % s->tw) + s->tw) % s->tw; ry = ((ry % s->th) + s->th) % s->th; pixel = s->tex[ry * s->tw + rx]; }
            else pixel = (rx < 0 || ry < 0 || rx >= s->tw || ry >= s->th) ? 0 : s->tex[ry * s->tw + rx];
            int32_t f = (int32_t)((fy * fx + 0x8000) >> 16);
            sr += (int)((pixel >> 16) & 255) * f; sg += (int)((pixel >> 8) & 255) * f; sb += (int)(pixel & 255) * f; sa += (int)(pixel >> 24) * f;
        }
    }
    sa = (sa + 0x8000) >> 16; sr = (sr + 0x8000) >> 16; sg = (sg + 0x8000) >> 16; sb = (sb + 0x8000) >> 16;
    if (sa < 0) sa = 0; if (sa > 255) sa = 255; if (sr < 0) sr = 0; if (sr > 255) sr = 255;
    if (sg < 0) sg = 0; if (sg > 255) sg = 255; if (sb < 0) sb = 0; if (sb > 255) sb = 255;
    return ((uint32_t)sa << 24) | ((uint32_t)sr << 16) | ((uint32_t)sg << 8) | (uint32_t)sb;
}

typedef struct {
    mat_t ctm, ctm_inverse; double line_width; int cap, join; double miter_limit; int fill_rule;
} gstate_t;

typedef struct swfo_ctx {
    int w, h; uint32_t *px; int is_clear;
    int bx0, by0, bx1, by1;                 /* bounded rectangle of the current operation (op_bounds) */
    gstate_t gs[64]; int ngs;
    path_t path;
    source_t src;
    int last_unsupported, trace_rows;
    /* scratch */
    int32_t *ch, *ua; int *touched; int ntouched; uint8_t *tmark;
    uint8_t *rowcov;
    int32_t *last_poly; int last_poly_n, last_poly_rect;   /* edges of the last fill/stroke polygon (tests) */
} swfo_ctx;

/* ------------------------------------------------------------------ [A.7] compositing */
static inline uint32_t mul8x2_8(uint32_t a, uint8_t b)
{
    uint32_t t = (a & 0xff00ff) * b + 0x7f007f;
    return ((t + ((t >> 8) & 0xff00ff)) >> 8) & 0xff00ff;
}
static inline uint32_t add8x2_8x2(uint32_t a, uint32_t b)
{
    uint32_t t = a + b;
    t |= 0x1000100 - ((t >> 8) & 0xff00ff);
    return t & 0xff00ff;
}
static inline uint32_t lerp8x4(uint32_t src, uint8_t a, uint32_t dst)
{
    return add8x2_8x2(mul8x2_8(src, a), mul8x2_8(dst, (uint8_t)~a)) |
           (add8x2_8x2(mul8x2_8(src >> 8, a), mul8x2_8(dst >> 8, (uint8_t)~a)) << 8);
}
/* pixman: UN8x4_MUL_UN8 / over */
static inline uint32_t pm_mul_un8(uint32_t x, uint8_t a)
{
    uint32_t rb = (x & 0xff00ff) * a + 0x800080;
    rb = ((rb + ((rb >> 8) & 0xff00ff)) >> 8) & 0xff00ff;
    uint32_t ag = ((x >> 8) & 0xff00ff) * a + 0x800080;
    ag = ((ag + ((ag >> 8) & 0xff00ff)) >> 8) & 0xff00ff;
    return rb | (ag << 8);
}
static inline uint32_t pm_add_sat(uint32_t x, uint32_t y)
{
    uint32_t rb = (x & 0xff00ff) + (y & 0xff00ff);
    rb |= 0x1000100 - ((rb >> 8) & 0xff00ff); rb &= 0xff00ff;
    uint32_t ag = ((x >> 8) & 0xff00ff) + ((y >> 8) & 0xff00ff);
    ag |= 0x1000100 - ((ag >> 8) & 0xff00ff); ag &= 0xff00ff;
    return rb | (ag << 8);
}
static inline uint32_t pm_over(uint32_t src, uint32_t dst)
{
    return pm_add_sat(pm_mul_un8(dst, (uint8_t)(~src >> 24)), src);
}

static uint32_t color_to_pixel(double r, double g, double b, double a)
{
    /* _cairo_color_init_rgba: premultiply in doubles, _cairo_color_double_to_short, >> 8 */
    uint32_t rs = (uint32_t)(uint16_t)(r * a * 65535.0 + 0.5), gs = (uint32_t)(uint16_t)(g * a * 65535.0 + 0.5);
    uint32_t bs = (uint32_t)(uint16_t)(b * a * 65535.0 + 0.5), as = (uint32_t)(uint16_t)(a * 65535.0 + 0.5);
    return ((as >> 8) << 24) | ((rs >> 8) << 16) | ((gs >> 8) << 8) | (bs >> 8);
}

/* gradient / texture sampling: float64 model of pixman's general path (+-1 LSB) */
static uint32_t gradient_color(const source_t *s, double t)
{
    const stop_t *st = s->stops; int n = s->nstops;
    double r, g, b, a;
    if (n == 0) return 0;
    if (t <= st[0].t) { r = st[0].r; g = st[0].g; b = st[0].b; a = st[0].a; }
    else if (t >= st[n - 1].t) { r = st[n - 1].r; g = st[n - 1].g; b = st[n - 1].b; a = st[n - 1].a; }
    else {
        int i = 0;
        while (i + 1 < n && st[i + 1].t <= t) i++;
        double span = st[i + 1].t - st[i].t, f = span > 0 ? (t - st[i].t) / span : 0;
        r = st[i].r + (st[i + 1].r - st[i].r) * f; g = st[i].g + (st[i + 1].g - st[i].g) * f;
        b = st[i].b + (st[i + 1].b - st[i].b) * f; a = st[i].a + (st[i + 1].a - st[i].a) * f;
    }
    uint32_t A = (uint32_t)(a * 255.0 + 0.5), R = (uint32_t)(r * a * 255.0 + 0.5);
    uint32_t G = (uint32_t)(g * a * 255.0 + 0.5), B = (uint32_t)(b * a * 255.0 + 0.5);
    return (A << 24) | (R << 16) | (G << 8) | B;
}
static uint32_t sample_source(const source_t *s, int px, int py)
{
    double x = px + 0.5, y = py + 0.5;
    if (s->kind == SRC_LINEAR) mat_point(&s->inv, &x, &y);
    if (s->kind == SRC_RADIAL) return sample_radial_pixman(s, px, py);
    if (s->kind == SRC_RADIAL) {                           /* (float64 model, kept for reference: within +-1 of the above) */
        /* |p - c(t)| = r(t), larger root, PAD extend */
        double cdx = s->cx1 - s->cx0, cdy = s->cy1 - s->cy0, dr = s->r1 - s->r0;
        double pdx = x - s->cx0, pdy = y - s->cy0;
        double A = cdx * cdx + cdy * cdy - dr * dr;
        double B = pdx * cdx + pdy * cdy + s->r0 * dr;
        double C = pdx * pdx + pdy * pdy - s->r0 * s->r0;
        double t;
        if (A == 0) { if (B == 0) return 0; t = 0.5 * C / B; if (s->r0 + t * dr < 0) return 0; }
        else {
            double disc = B * B - A * C;
            if (disc < 0) return 0;
            double sq = sqrt(disc), t0 = (B + sq) / A, t1 = (B - sq) / A;
            if (s->r0 + t0 * dr >= 0) t = t0; else if (s->r0 + t1 * dr >= 0) t = t1; else return 0;
        }
        if (t < 0) t = 0; if (t > 1) t = 1;
        return gradient_color(s, t);
    }
    if (s->kind == SRC_LINEAR) {
        double dx = s->cx1 - s->cx0, dy = s->cy1 - s->cy0, l = dx * dx + dy * dy;
        double t = l == 0 ? 0 : ((x - s->cx0) * dx + (y - s->cy0) * dy) / l;
        if (t < 0) t = 0; if (t > 1) t = 1;
        return gradient_color(s, t);
    }
    int64_t fxp, fyp;
    source_pixman_position(s, px, py, &fxp, &fyp);
    if (s->good) return sample_good(s, fxp, fyp);
    /* SRC_SURFACE: bilinear (7-bit weights) -- what CAIRO_FILTER_GOOD becomes for scales > .75 */
    fxp -= 0x8000; fyp -= 0x8000;
    int x0 = (int)(fxp >> 16), y0 = (int)(fyp >> 16);
    int wx = (int)((fxp >> 9) & 0x7f), wy = (int)((fyp >> 9) & 0x7f);
    uint32_t c[4];
    for (int k = 0; k < 4; k++) {
        int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        if (s->extend == 1) { xx = ((xx % s->tw) + s->tw) % s->tw; yy = ((yy % s->th) + s->th) % s->th; c[k] = s->tex[yy * s->tw + xx]; }
        else c[k] = (xx < 0 || yy < 0 || xx >= s->tw || yy >= s->th) ? 0 : s->tex[yy * s->tw + xx];
    }
    uint32_t out = 0;
    for (int sh = 0; sh < 32; sh += 8) {
        uint32_t v00 = (c[0] >> sh) & 255, v10 = (c[1] >> sh) & 255, v01 = (c[2] >> sh) & 255, v11 = (c[3] >> sh) & 255;
        uint32_t acc = v00 * (128 - wx) * (128 - wy) + v10 * wx * (128 - wy) + v01 * (128 - wx) * wy + v11 * wx * wy;
        out |= ((acc >> 14) & 255) << sh;
    }
    return out;
}

/* one coverage span [x0,x1) of row y with 8-bit coverage `cov` */
static void composite_span(swfo_ctx *c, int lerp_mode, int y, int x0, int x1, uint8_t cov)
{
    if (!cov || x0 >= x1) return;
    uint32_t *d = c->px + (size_t)y * c->w + x0;
    int n = x1 - x0;
    const source_t *s = &c->src;
    if (s->kind == SRC_SOLID) {
        uint32_t p = s->pixel;
        if (lerp_mode) {
            if (cov == 0xff) while (n--) *d++ = p;
            else while (n--) { *d = lerp8x4(p, cov, *d); d++; }
        } else {                                           /* pixman over_n_8_8888 */
            if (cov == 0xff) { if ((p >> 24) == 0xff) while (n--) *d++ = p; else while (n--) { *d = pm_over(p, *d); d++; } }
            else { uint32_t m = pm_mul_un8(p, cov); while (n--) { *d = pm_over(m, *d); d++; } }
        }
        return;
    }
    for (int i = 0; i < n; i++, d++) {
        uint32_t sp = pm_mul_un8(sample_source(s, x0 + i, y), cov);
        *d = lerp_mode ? sp : pm_over(sp, *d);             /* SRC on a clear surface / OVER */
    }
}

/* ------------------------------------------------------------------ [A.5] tor scan converter */
typedef struct { int64_t quo, rem; } qr_t;
typedef struct tedge {
    struct tedge *next, *prev;
    int ytop, height_left, dir, cell;
    qr_t x, dxdy, dxdy_full;
    int64_t dy;
} tedge_t;

static inline void qr_norm(qr_t *x, int64_t dy)
{
    if (x->rem < 0) { x->quo--; x->rem += dy; } else if (x->rem >= dy) { x->quo++; x->rem -= dy; }
}
static tedge_t *merge_sorted_edges(tedge_t *head_a, tedge_t *head_b)
{
    tedge_t *head, **next, *prev;
    int x;
    prev = head_a->prev;
    next = &head;
    if (head_a->cell <= head_b->cell) head = head_a;
    else { head = head_b; head_b->prev = prev; goto start_with_b; }
    do {
        x = head_b->cell;
        while (head_a != NULL && head_a->cell <= x) { prev = head_a; next = &head_a->next; head_a = head_a->next; }
        head_b->prev = prev; *next = head_b;
        if (head_a == NULL) return head;
start_with_b:
        x = head_a->cell;
        while (head_b != NULL && head_b->cell <= x) { prev = head_b; next = &head_b->next; head_b = head_b->next; }
        head_a->prev = prev; *next = head_a;
        if (head_b == NULL) return head;
    } while (1);
}
static tedge_t *sort_edges(tedge_t *list, unsigned level, tedge_t **head_out)
{
    tedge_t *head_other = list->next, *remaining;
    if (head_other == NULL) { *head_out = list; return NULL; }
    remaining = head_other->next;
    if (list->cell <= head_other->cell) { *head_out = list; head_other->next = NULL; }
    else { *head_out = head_other; head_other->prev = list->prev; head_other->next = list; list->prev = head_other; list->next = NULL; }
    for (unsigned i = 0; i < level && remaining; i++) {
        remaining = sort_edges(remaining, i, &head_other);
        *head_out = merge_sorted_edges(*head_out, head_other);
    }
    return remaining;
}

typedef struct {
    tedge_t head, tail;
    swfo_ctx *c;
    int xmin, xmax;                          /* pixel columns of the converter */
} active_t;

static inline void cell_add(swfo_ctx *c, int xmin, int xmax, int ix, int dch, int dua)
{
    /* cells left of xmin keep covered_height (folded into the first column) but lose area;
       cells at/after xmax are never emitted (SURVEY A.5 "converter x-range") */
    if (ix >= xmax) return;
    if (ix < xmin) { ix = xmin; dua = 0; }
    int k = ix - xmin;
    if (!c->tmark[k]) { c->tmark[k] = 1; c->touched[c->ntouched++] = k; }
    c->ch[k] += dch; c->ua[k] += dua;
}
static void cell_subspan(active_t *a, int x1, int x2)
{
    if (x1 == x2) return;
    cell_add(a->c, a->xmin, a->xmax, x1 >> 8, 1, 2 * (x1 & 255));
    cell_add(a->c, a->xmin, a->xmax, x2 >> 8, -1, -2 * (x2 & 255));
}
static inline void full_step(tedge_t *e)
{
    if (e->dy == 0) return;
    e->x.quo += e->dxdy_full.quo; e->x.rem += e->dxdy_full.rem;
    qr_norm(&e->x, e->dy);
    e->cell = (int)(e->x.quo + (e->x.rem >= e->dy / 2));
}
static void render_edge(active_t *a, tedge_t *e, int sign)
{
    qr_t x1 = e->x, x2;
    full_step(e);
    x2 = e->x;
    if (e->dy) {
        x1.quo -= e->dxdy.quo / 2; x1.rem -= e->dxdy.rem / 2; qr_norm(&x1, e->dy);
        x2.quo -= e->dxdy.quo / 2; x2.rem -= e->dxdy.rem / 2; qr_norm(&x2, e->dy);
    }
    int ix1 = (int)(x1.quo >> 8), fx1 = (int)(x1.quo & 255), ix2 = (int)(x2.quo >> 8), fx2 = (int)(x2.quo & 255);
    swfo_ctx *c = a->c;
    if (ix1 == ix2) { cell_add(c, a->xmin, a->xmax, ix1, sign * GRID_Y, sign * (fx1 + fx2) * GRID_Y); return; }
    if (ix2 < ix1) { qr_t t = x1; x1 = x2; x2 = t; int ti = ix1; ix1 = ix2; ix2 = ti; ti = fx1; fx1 = fx2; fx2 = ti; }
    int64_t dx = (x2.quo - x1.quo) * e->dy + (x2.rem - x1.rem);
    int64_t tmp = (int64_t)(ix1 + 1) * GRID_X * e->dy;
    tmp -= x1.quo * e->dy + x1.rem;
    tmp *= GRID_Y;
    qr_t y = { tmp / dx, tmp % dx };
    cell_add(c, a->xmin, a->xmax, ix1, sign * (int)y.quo, sign * (int)y.quo * (GRID_X + fx1));
    int y_last = (int)y.quo;
    if (ix1 + 1 < ix2) {
        qr_t full = { (int64_t)GRID_Y * GRID_X * e->dy / dx, (int64_t)GRID_Y * GRID_X * e->dy % dx };
        ++ix1;
        do {
            y.quo += full.quo; y.rem += full.rem;
            if (y.rem >= dx) { y.quo++; y.rem -= dx; }
            cell_add(c, a->xmin, a->xmax, ix1, sign * (int)(y.quo - y_last), sign * (int)(y.quo - y_last) * GRID_X);
            y_last = (int)y.quo;
            ++ix1;
        } while (ix1 != ix2);
    }
    cell_add(c, a->xmin, a->xmax, ix2, sign * (GRID_Y - y_last), sign * (GRID_Y - y_last) * fx2);
}
static inline void edge_dec(tedge_t *e, int h)
{
    e->height_left -= h;
    if (e->height_left == 0) { e->prev->next = e->next; e->next->prev = e->prev; }
}
static void full_row(active_t *a, unsigned mask)
{
    tedge_t *left = a->head.next;
    while (left != &a->tail) {
        tedge_t *right;
        int winding;
        edge_dec(left, GRID_Y);
        winding = left->dir;
        right = left->next;
        do {
            edge_dec(right, GRID_Y);
            winding += right->dir;
            if ((winding & mask) == 0 && right->next->cell != right->cell) break;
            full_step(right);
            right = right->next;
        } while (1);
        render_edge(a, left, +1);
        render_edge(a, right, -1);
        left = right->next;
    }
}
static void sub_row(active_t *a, unsigned mask)
{
    tedge_t *edge = a->head.next;
    int xstart = INT_MIN, prev_x = INT_MIN, winding = 0;
    while (edge != &a->tail) {
        tedge_t *next = edge->next;
        int xend = edge->cell;
        if (--edge->height_left) {
            if (edge->dy) {
                edge->x.quo += edge->dxdy.quo; edge->x.rem += edge->dxdy.rem;
                qr_norm(&edge->x, edge->dy);
                edge->cell = (int)(edge->x.quo + (edge->x.rem >= edge->dy / 2));
            }
            if (edge->cell < prev_x) {
                tedge_t *pos = edge->prev;
                pos->next = next; next->prev = pos;
                do pos = pos->prev; while (edge->cell < pos->cell);
                pos->next->prev = edge; edge->next = pos->next; edge->prev = pos; pos->next = edge;
            } else prev_x = edge->cell;
        } else { edge->prev->next = next; next->prev = edge->prev; }
        winding += edge->dir;
        if ((winding & mask) == 0) {
            if (next->cell != xend) { cell_subspan(a, xstart, xend); xstart = INT_MIN; }
        } else if (xstart == INT_MIN) xstart = xend;
        edge = next;
    }
}
static int can_do_full_row(active_t *a)
{
    int prev_x = INT_MIN, min_height = INT_MAX;
    for (tedge_t *e = a->head.next; e != &a->tail; e = e->next) if (e->height_left < min_height) min_height = e->height_left;
    if (min_height < GRID_Y) return 0;
    for (tedge_t *e = a->head.next; e != &a->tail; e = e->next) {
        int cell;
        if (e->dy) {
            qr_t x = e->x;
            x.quo += e->dxdy_full.quo; x.rem += e->dxdy_full.rem;
            qr_norm(&x, e->dy);
            cell = (int)(x.quo + (x.rem >= e->dy / 2));
        } else cell = e->cell;
        if (cell < prev_x) return 0;
        prev_x = cell;
    }
    return 1;
}
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

/* emit the accumulated cells of one pixel row as coverage spans (blit of the cell list) */
static void blit_row(swfo_ctx *c, int lerp_mode, int y, int xmin, int xmax)
{
    if (!c->ntouched) return;
    qsort(c->touched, c->ntouched, sizeof(int), cmp_int);
    int cover = 0, prev_x = xmin;
    for (int t = 0; t < c->ntouched; t++) {
        int k = c->touched[t], x = xmin + k;
        if (x > prev_x) composite_span(c, lerp_mode, y, prev_x, x, (uint8_t)(((cover) * 17 + 256) >> 9));
        cover += c->ch[k] * GRID_X * 2;
        int area = cover - c->ua[k];
        composite_span(c, lerp_mode, y, x, x + 1, (uint8_t)((area * 17 + 256) >> 9));
        prev_x = x + 1;
        c->ch[k] = 0; c->ua[k] = 0; c->tmark[k] = 0;
    }
    if (prev_x < xmax) composite_span(c, lerp_mode, y, prev_x, xmax, (uint8_t)((cover * 17 + 256) >> 9));
    c->ntouched = 0;
}

/* rasterise polygon `g` with the tor converter over pixel rect [xmin,xmax) x [ymin,ymax) */
static void tor_render(swfo_ctx *c, const polygon_t *g, int even_odd, int lerp_mode, int xmin, int ymin, int xmax, int ymax)
{
    int h = ymax - ymin;
    if (h <= 0 || xmax <= xmin) return;
    tedge_t *pool = malloc(sizeof(tedge_t) * (g->n ? g->n : 1));
    tedge_t **ybuckets = calloc(h, sizeof(tedge_t *));
    int ne = 0, gymin = ymin * GRID_Y, gymax = ymax * GRID_Y;
    for (int i = 0; i < g->n; i++) {
        const pedge_t *pe = &g->e[i];
        int ytop = (int)(((int64_t)GRID_Y * pe->top + 128) >> 8), ybot = (int)(((int64_t)GRID_Y * pe->bottom + 128) >> 8);
        if (ytop < gymin) ytop = gymin;
        if (ybot > gymax) ybot = gymax;
        if (ybot <= ytop) continue;
        tedge_t *e = &pool[ne++];
        const pt_t *p1, *p2;
        e->ytop = ytop; e->height_left = ybot - ytop;
        if (pe->p2.y > pe->p1.y) { e->dir = pe->dir; p1 = &pe->p1; p2 = &pe->p2; }
        else { e->dir = -pe->dir; p1 = &pe->p2; p2 = &pe->p1; }
        if (p2->x == p1->x) {
            e->cell = p1->x; e->x.quo = p1->x; e->x.rem = 0;
            e->dxdy.quo = e->dxdy.rem = 0; e->dxdy_full.quo = e->dxdy_full.rem = 0; e->dy = 0;
        } else {
            int64_t Ex = (int64_t)(p2->x - p1->x) * GRID_X;
            int64_t Ey = (int64_t)(p2->y - p1->y) * GRID_Y * 512;
            e->dxdy.quo = Ex * 512 / Ey; e->dxdy.rem = Ex * 512 % Ey;
            int64_t tmp = (int64_t)(2 * ytop + 1) << 8;
            tmp -= (int64_t)p1->y * GRID_Y * 2;
            tmp *= Ex;
            e->x.quo = tmp / Ey; e->x.rem = tmp % Ey;
            e->x.quo += p1->x;
            qr_norm(&e->x, Ey);
            if (e->height_left >= GRID_Y) { tmp = Ex * (2 * GRID_Y << 8); e->dxdy_full.quo = tmp / Ey; e->dxdy_full.rem = tmp % Ey; }
            else e->dxdy_full.quo = e->dxdy_full.rem = 0;
            e->cell = (int)(e->x.quo + (e->x.rem >= Ey / 2));
            e->dy = Ey;
        }
        int ix = (ytop - gymin) / GRID_Y;
        e->prev = NULL; e->next = ybuckets[ix]; ybuckets[ix] = e;     /* push front */
    }
    active_t a;
    a.c = c; a.xmin = xmin; a.xmax = xmax;
    a.head.cell = INT_MIN; a.head.prev = NULL; a.head.next = &a.tail; a.head.height_left = INT_MAX; a.head.dy = 0; a.head.dir = 0;
    a.tail.cell = INT_MAX; a.tail.prev = &a.head; a.tail.next = NULL; a.tail.height_left = INT_MAX; a.tail.dy = 0; a.tail.dir = 0;
    unsigned mask = even_odd ? 1u : ~0u;
    tedge_t *buckets[GRID_Y];
    memset(buckets, 0, sizeof(buckets));
    for (int i = 0; i < h; i++) {
        int do_full = 0, max_suby = 0;
        /* polygon_fill_buckets */
        for (tedge_t *e = ybuckets[i]; e;) {
            tedge_t *nx = e->next;
            int suby = e->ytop - (i * GRID_Y + gymin);
            if (buckets[suby]) buckets[suby]->prev = e;
            e->next = buckets[suby]; e->prev = NULL; buckets[suby] = e;
            if (suby > max_suby) max_suby = suby;
            e = nx;
        }
        if (max_suby == 0) {
            if (buckets[0]) {
                tedge_t *sorted;
                sort_edges(buckets[0], UINT_MAX, &sorted);
                a.head.next = merge_sorted_edges(a.head.next, sorted);
                buckets[0] = NULL;
            }
            if (a.head.next == &a.tail) continue;
            do_full = can_do_full_row(&a);
        }
        if (c->trace_rows) {                                /* SWFO_TRACE_ROWS: row modes and the active list, for tools/soak_case.py */
            fprintf(stderr, "row %d full %d :", ymin + i, do_full);
            for (tedge_t *e = a.head.next; e != &a.tail; e = e->next) fprintf(stderr, " [cell %d dir %d left %d]", e->cell, e->dir, e->height_left);
            fprintf(stderr, "\n");
        }
        if (do_full) full_row(&a, mask);
        else {
            for (int sub = 0; sub < GRID_Y; sub++) {
                if (buckets[sub]) {
                    tedge_t *sorted;
                    sort_edges(buckets[sub], UINT_MAX, &sorted);
                    a.head.next = merge_sorted_edges(a.head.next, sorted);
                    buckets[sub] = NULL;
                }
                sub_row(&a, mask);
            }
        }
        blit_row(c, lerp_mode, ymin + i, xmin, xmax);
    }
    free(pool); free(ybuckets);
}

/* ------------------------------------------------------------------ [A.6] rectilinear -> boxes */
typedef struct { fx_t x; fx_t top, bottom; int dir; } vedge_t;
static int cmp_fx(const void *a, const void *b) { fx_t x = *(const fx_t *)a, y = *(const fx_t *)b; return (x > y) - (x < y); }
static int cmp_vedge(const void *a, const void *b) { const vedge_t *x = a, *y = b; return (x->x > y->x) - (x->x < y->x); }

/* both painters return 1 when the operation counts as drawn (the surface is no longer clear afterwards) and 0 for
   CAIRO_INT_STATUS_NOTHING_TO_DO: the geometry's extents miss the operation's bounded rectangle (trim_extents_to_polygon /
   trim_extents_to_boxes), e.g. a stroke whose approximate extents touch the frame while its outline lies outside.  No boxes at
   all is "success" in clip_and_composite_boxes (probe: the next translucent fill then takes the OVER path). */
static int boxes_render(swfo_ctx *c, const polygon_t *g, int even_odd, int lerp_mode)
{
    /* exact area of the fill region per pixel: c = sum wx*wy over disjoint boxes, alpha = (c>>8)-(c>>16) */
    int n = g->n;
    if (!n) return 1;
    vedge_t *ve = malloc(sizeof(vedge_t) * n);
    fx_t *ys = malloc(sizeof(fx_t) * 2 * n);
    for (int i = 0; i < n; i++) {
        ve[i].x = g->e[i].p1.x; ve[i].top = g->e[i].top; ve[i].bottom = g->e[i].bottom; ve[i].dir = g->e[i].dir;
        ys[2 * i] = g->e[i].top; ys[2 * i + 1] = g->e[i].bottom;
    }
    qsort(ys, 2 * n, sizeof(fx_t), cmp_fx);
    qsort(ve, n, sizeof(vedge_t), cmp_vedge);
    int px0 = fx_floor_i(g->x1.x), px1 = fx_ceil_i(g->x2.x), py0 = fx_floor_i(g->x1.y), py1 = fx_ceil_i(g->x2.y);
    if (px0 < c->bx0) px0 = c->bx0; if (py0 < c->by0) py0 = c->by0; if (px1 > c->bx1) px1 = c->bx1; if (py1 > c->by1) py1 = c->by1;
    int bw = px1 - px0, bh = py1 - py0;
    if (bw <= 0 || bh <= 0) { free(ve); free(ys); return 0; }
    if (c->src.kind == SRC_SURFACE) source_prepare_pixman(&c->src, px0, py0, px1, py1);
    if (c->src.kind == SRC_RADIAL) source_prepare_radial(&c->src, px0, py0, px1, py1);
    uint32_t *acc = calloc((size_t)bw * bh, sizeof(uint32_t));
    unsigned mask = even_odd ? 1u : ~0u;
    for (int s = 0; s + 1 < 2 * n; s++) {
        fx_t ya = ys[s], yb = ys[s + 1];
        if (ya == yb) continue;
        int winding = 0; fx_t xstart = 0; int inside = 0;
        for (int i = 0; i < n; i++) {
            if (!(ve[i].top <= ya && ve[i].bottom >= yb)) continue;
            winding += ve[i].dir;
            int now = (winding & mask) != 0;
            if (now && !inside) { xstart = ve[i].x; inside = 1; }
            else if (!now && inside) {
                fx_t xa = xstart, xb = ve[i].x;
                inside = 0;
                if (xa == xb) continue;
                /* box (xa,ya)-(xb,yb) */
                fx_t cxa = xa < px0 * 256 ? px0 * 256 : xa, cxb = xb > px1 * 256 ? px1 * 256 : xb;
                fx_t cya = ya < py0 * 256 ? py0 * 256 : ya, cyb = yb > py1 * 256 ? py1 * 256 : yb;
                if (cxa >= cxb || cya >= cyb) continue;
                for (int py = cya >> 8; py <= (cyb - 1) >> 8; py++) {
                    int wy = (cyb < (py + 1) * 256 ? cyb : (py + 1) * 256) - (cya > py * 256 ? cya : py * 256);
                    uint32_t *row = acc + (size_t)(py - py0) * bw;
                    for (int px = cxa >> 8; px <= (cxb - 1) >> 8; px++) {
                        int wx = (cxb < (px + 1) * 256 ? cxb : (px + 1) * 256) - (cxa > px * 256 ? cxa : px * 256);
                        row[px - px0] += (uint32_t)(wx * wy);
                    }
                }
            }
        }
    }
    for (int y = 0; y < bh; y++) {
        const uint32_t *row = acc + (size_t)y * bw;
        int x = 0;
        while (x < bw) {
            uint32_t v = row[x]; int x2 = x + 1;
            while (x2 < bw && row[x2] == v) x2++;
            composite_span(c, lerp_mode, py0 + y, px0 + x, px0 + x2, (uint8_t)((v >> 8) - (v >> 16)));
            x = x2;
        }
    }
    free(acc); free(ve); free(ys);
    return 1;
}

/* ------------------------------------------------------------------ draw ops */
/* common front end: decides NOTHING_TO_DO, limits, lerp-vs-over, is_clear bookkeeping */
static int op_bounds(swfo_ctx *c, pt_t e1, pt_t e2, int *needs_limits)
{
    /* approximate extents rounded out to pixels, intersected with the surface */
    if (!(e1.x < e2.x && e1.y < e2.y)) {
        /* cairo keeps degenerate (zero-area) extents as an empty rectangle -> nothing to do */
        return 0;
    }
    int x0 = fx_floor_i(e1.x), y0 = fx_floor_i(e1.y), x1 = fx_ceil_i(e2.x), y1 = fx_ceil_i(e2.y);
    const int mw = x1 - x0, mh = y1 - y0;                  /* the mask extents */
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > c->w) x1 = c->w; if (y1 > c->h) y1 = c->h;
    if (c->src.kind == SRC_SURFACE && c->src.extend == 0) {
        /* OVER is bounded by its source: _cairo_pattern_get_extents of an EXTEND_NONE surface pattern -- the surface rectangle,
           mapped to device space; an axis the filter magnifies is padded by half a source pixel and rounded to the nearest
           pixel edge, the others are rounded out (probe-validated against libcairo 1.16.0) */
        const mat_t *pm = &c->src.inv;
        double sx0 = 0, sy0 = 0, sx1 = c->src.tw, sy1 = c->src.th;
        int round_x = 0, round_y = 0;
        if (hypot(pm->xx, pm->yx) < 1.0) { sx0 -= 0.5; sx1 += 0.5; round_x = 1; }
        if (hypot(pm->xy, pm->yy) < 1.0) { sy0 -= 0.5; sy1 += 0.5; round_y = 1; }
        mat_t im = *pm;
        if (mat_invert_cairo(&im)) {
            double bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
            for (int k = 0; k < 4; k++) {
                double x = (k & 1) ? sx1 : sx0, y = (k & 2) ? sy1 : sy0;
                mat_point(&im, &x, &y);
                if (!k || x < bx0) bx0 = x; if (!k || x > bx1) bx1 = x;
                if (!k || y < by0) by0 = y; if (!k || y > by1) by1 = y;
            }
            if (!round_x) { bx0 -= 0.5; bx1 += 0.5; }
            if (!round_y) { by0 -= 0.5; by1 += 0.5; }
            bx0 = floor(bx0 + 0.5); by0 = floor(by0 + 0.5); bx1 = floor(bx1 + 0.5); by1 = floor(by1 + 0.5);
            if (x0 < bx0) x0 = (int)bx0; if (y0 < by0) y0 = (int)by0;
            if (x1 > bx1) x1 = (int)bx1; if (y1 > by1) y1 = (int)by1;
        }
    }
    *needs_limits = mw > x1 - x0 || mh > y1 - y0;
    /* the operation is bounded by this rectangle: geometry the stroker produces outside it (a round join that fell
       through to a miter at a closing corner) is not painted */
    c->bx0 = x0; c->by0 = y0; c->bx1 = x1; c->by1 = y1;
    return x0 < x1 && y0 < y1;
}
/* cairo-pattern.c _cairo_pattern_is_clear: a transparent solid, or a gradient without a stop that is not transparent
   (_gradient_is_clear; CAIRO_COLOR_IS_CLEAR is alpha_short <= 0x00ff).  cairo-surface.c nothing_to_do then skips the OVER. */
static int source_is_clear(const source_t *s)
{
    if (s->kind == SRC_SOLID) return (s->pixel >> 24) == 0;
    if (s->kind == SRC_RADIAL || s->kind == SRC_LINEAR) {
        for (int i = 0; i < s->nstops; i++) if (s->stops[i].a * 65535.0 + 0.5 >= 256.0) return 0;
        return 1;
    }
    return 0;
}
static int source_is_opaque_solid(const source_t *s) { return s->kind == SRC_SOLID && (s->pixel >> 24) == 0xff; }

static void ensure_scratch(swfo_ctx *c)
{
    if (c->ch) return;
    int n = c->w + 2;
    c->ch = calloc(n, sizeof(int32_t)); c->ua = calloc(n, sizeof(int32_t));
    c->touched = malloc(sizeof(int) * n); c->tmark = calloc(n, 1); c->ntouched = 0;
}
static void remember_polygon(swfo_ctx *c, const polygon_t *g, int rectilinear)
{
    c->last_poly = realloc(c->last_poly, sizeof(int32_t) * 7 * (g->n ? g->n : 1));
    c->last_poly_n = g->n; c->last_poly_rect = rectilinear;
    for (int i = 0; i < g->n; i++) {
        int32_t *o = c->last_poly + 7 * i; const pedge_t *e = &g->e[i];
        o[0] = e->p1.x; o[1] = e->p1.y; o[2] = e->p2.x; o[3] = e->p2.y; o[4] = e->top; o[5] = e->bottom; o[6] = e->dir;
    }
}
static int render_polygon(swfo_ctx *c, polygon_t *g, int even_odd)
{
    ensure_scratch(c);
    if (!g->n) return 0;
    int lerp_mode = source_is_opaque_solid(&c->src) || c->is_clear;
    int xmin = fx_floor_i(g->x1.x), xmax = fx_ceil_i(g->x2.x), ymin = fx_floor_i(g->x1.y), ymax = fx_ceil_i(g->x2.y);
    if (xmin < c->bx0) xmin = c->bx0; if (ymin < c->by0) ymin = c->by0; if (xmax > c->bx1) xmax = c->bx1; if (ymax > c->by1) ymax = c->by1;
    if (xmin >= xmax || ymin >= ymax) return 0;
    if (c->src.kind == SRC_SURFACE) source_prepare_pixman(&c->src, xmin, ymin, xmax, ymax);
    if (c->src.kind == SRC_RADIAL) source_prepare_radial(&c->src, xmin, ymin, xmax, ymax);
    tor_render(c, g, even_odd, lerp_mode, xmin, ymin, xmax, ymax);
    return 1;
}

EXPORT int swfo_fill_preserve(swfo_ctx *c)
{
    path_t *p = &c->path;
    gstate_t *gs = &c->gs[c->ngs - 1];
    int needs_limits = 0;
    c->last_unsupported = 0; c->last_poly_n = 0;
    if (source_is_clear(&c->src)) return 0;               /* OVER with a clear source: no-op */
    if (!p->has_extents || !op_bounds(c, p->e1, p->e2, &needs_limits)) return 0;   /* NOTHING_TO_DO */
    polygon_t g; memset(&g, 0, sizeof(g));
    pt_t l1 = { c->bx0 * 256, c->by0 * 256 }, l2 = { c->bx1 * 256, c->by1 * 256 };   /* the limits are the unbounded rectangle */
    polygon_init(&g, needs_limits, l1, l2);
    path_fill_to_polygon(p, 0.1, &g);
    remember_polygon(c, &g, path_fill_is_rectilinear(p));
    int drawn;
    if (path_fill_is_rectilinear(p)) {
        int lerp_mode = source_is_opaque_solid(&c->src) || c->is_clear;
        drawn = boxes_render(c, &g, gs->fill_rule, lerp_mode);
    } else
        drawn = render_polygon(c, &g, gs->fill_rule);
    free(g.e);
    if (drawn) c->is_clear = 0;
    return 0;
}

EXPORT int swfo_stroke_preserve(swfo_ctx *c)
{
    path_t *p = &c->path;
    gstate_t *gs = &c->gs[c->ngs - 1];
    c->last_unsupported = 0; c->last_poly_n = 0;
    if (source_is_clear(&c->src)) return 0;
    if (!p->has_extents) return 0;
    /* approximate stroke extents: path box grown by the device-space line radius (miter: x limit) */
    mat_t inv = gs->ctm;
    if (!mat_invert(&inv)) return 0;
    double hw = gs->line_width / 2.0;
    /* _cairo_compositor_stroke: a pen that degenerates to one vertex (line width <= tolerance / 2 = 0.05 device pixels under
       the CTM) means NOTHING_TO_DO for every stroker, rectilinear included; the surface stays untouched (probe-validated) */
    if (pen_vertices_needed(0.1, hw, &gs->ctm) <= 1) return 0;
    /* _cairo_stroke_style_max_distance_from_path + _cairo_path_fixed_approximate_stroke_extents */
    double expansion = 0.5;
    if (gs->cap == 2) expansion = M_SQRT1_2;
    if (gs->join == 0 && !p->stroke_is_rect && expansion < M_SQRT2 * gs->miter_limit) expansion = M_SQRT2 * gs->miter_limit;
    expansion *= gs->line_width;
    int unity = mat_has_unity_scale(&gs->ctm);
    double gx = unity ? expansion : expansion * hypot(gs->ctm.xx, gs->ctm.xy);
    double gy = unity ? expansion : expansion * hypot(gs->ctm.yy, gs->ctm.yx);
    pt_t e1 = { p->e1.x - fx_from_double(gx), p->e1.y - fx_from_double(gy) };
    pt_t e2 = { p->e2.x + fx_from_double(gx), p->e2.y + fx_from_double(gy) };
    int needs_limits = 0;
    if (!op_bounds(c, e1, e2, &needs_limits)) return 0;
    polygon_t g; memset(&g, 0, sizeof(g));
    pt_t l1 = { 0, 0 }, l2 = { c->w * 256, c->h * 256 };
    if (p->stroke_is_rect) {
        /* rectilinear path: cairo's box stroker, when it accepts the style; painted like a rectilinear fill (A.6) */
        polygon_init(&g, 0, l1, l2);
        if (path_stroke_rectilinear(p, &gs->ctm, gs->line_width, gs->join, gs->cap, gs->miter_limit, &g)) {
            remember_polygon(c, &g, 1);
            int lerp_mode = source_is_opaque_solid(&c->src) || c->is_clear;
            if (boxes_render(c, &g, 0, lerp_mode)) c->is_clear = 0;
            free(g.e);
            return 0;
        }
        g.n = 0;
    }
    polygon_init(&g, needs_limits, l1, l2);
    stroker_t s; memset(&s, 0, sizeof(s));
    s.cw.dir = 1; s.ccw.dir = -1;
    s.g = &g; s.ctm = &gs->ctm; s.inv = &inv;
    s.ctm_identity = mat_is_identity(&inv);
    s.ctm_det_positive = mat_det(&gs->ctm) >= 0.0;
    s.half_width = hw; s.miter_limit = gs->miter_limit; s.tol = 0.1;
    s.join = gs->join; s.cap = gs->cap;
    { double t = 0.1 * 256.0; s.contour_tol = (int64_t)(t * t); }
    if (needs_limits) {                                   /* stroker bounds: the limits grown by the style's reach */
        s.has_bounds = 1;
        s.b1.x = l1.x - fx_from_double(gx); s.b1.y = l1.y - fx_from_double(gy);
        s.b2.x = l2.x + fx_from_double(gx); s.b2.y = l2.y + fx_from_double(gy);
    }
    c->last_unsupported = path_stroke_to_polygon(p, &s);
    remember_polygon(c, &g, 0);
    if (render_polygon(c, &g, 0)) c->is_clear = 0;
    free(g.e); free(s.cw.p); free(s.ccw.p);
    return c->last_unsupported;
}

/* ------------------------------------------------------------------ context API */
EXPORT swfo_ctx *swfo_create(int w, int h)
{
    swfo_ctx *c = calloc(1, sizeof(*c));
    c->w = w; c->h = h; c->px = calloc((size_t)w * h, 4); c->is_clear = 1;
    c->ngs = 1; mat_identity(&c->gs[0].ctm); mat_identity(&c->gs[0].ctm_inverse);
    c->gs[0].line_width = 2.0; c->gs[0].miter_limit = 10.0; c->gs[0].cap = 0; c->gs[0].join = 0; c->gs[0].fill_rule = 0;
    path_reset(&c->path);
    c->src.kind = SRC_SOLID; c->src.pixel = 0xff000000u;
    c->trace_rows = getenv("SWFO_TRACE_ROWS") != NULL;
    return c;
}
EXPORT void swfo_destroy(swfo_ctx *c)
{
    if (!c) return;
    free(c->px); free(c->path.ops); free(c->path.pts); free(c->src.stops); free(c->src.xpar); free(c->src.ypar); free(c->src.g_x); free(c->src.g_ramp);
    free(c->ch); free(c->ua); free(c->touched); free(c->tmark); free(c->last_poly);
    free(c);
}
EXPORT void swfo_save(swfo_ctx *c) { if (c->ngs < 64) { c->gs[c->ngs] = c->gs[c->ngs - 1]; c->ngs++; } }
EXPORT void swfo_restore(swfo_ctx *c) { if (c->ngs > 1) c->ngs--; }
EXPORT void swfo_identity_matrix(swfo_ctx *c) { mat_identity(&c->gs[c->ngs - 1].ctm); mat_identity(&c->gs[c->ngs - 1].ctm_inverse); }
/* _cairo_gstate_transform / _cairo_gstate_scale: the inverse is kept beside the CTM and updated factor by factor */
EXPORT void swfo_transform(swfo_ctx *c, double xx, double yx, double xy, double yy, double x0, double y0)
{
    gstate_t *gs = &c->gs[c->ngs - 1];
    mat_t m = { xx, yx, xy, yy, x0, y0 }, t = m;
    if (!mat_invert_cairo(&t)) return;                      /* CAIRO_STATUS_INVALID_MATRIX: the call is ignored */
    mat_multiply(&gs->ctm, &m, &gs->ctm);
    mat_multiply(&gs->ctm_inverse, &gs->ctm_inverse, &t);
}
EXPORT void swfo_scale(swfo_ctx *c, double sx, double sy)
{
    gstate_t *gs = &c->gs[c->ngs - 1];
    if (sx * sy == 0. || !isfinite(sx) || !isfinite(sy)) return;
    mat_t m = { sx, 0, 0, sy, 0, 0 }, t = { 1 / sx, 0, 0, 1 / sy, 0, 0 };
    mat_multiply(&gs->ctm, &m, &gs->ctm);
    mat_multiply(&gs->ctm_inverse, &gs->ctm_inverse, &t);
}
EXPORT void swfo_clear_all(swfo_ctx *c) { memset(c->px, 0, (size_t)c->w * c->h * 4); c->is_clear = 1; }
EXPORT void swfo_new_path(swfo_ctx *c) { path_reset(&c->path); }
EXPORT void swfo_move_to(swfo_ctx *c, double x, double y)
{
    mat_point(&c->gs[c->ngs - 1].ctm, &x, &y);
    path_move_to(&c->path, fx_from_double(x), fx_from_double(y));
}
EXPORT void swfo_line_to(swfo_ctx *c, double x, double y)
{
    mat_point(&c->gs[c->ngs - 1].ctm, &x, &y);
    path_line_to(&c->path, fx_from_double(x), fx_from_double(y));
}
EXPORT void swfo_curve_to(swfo_ctx *c, double x1, double y1, double x2, double y2, double x3, double y3)
{
    const mat_t *m = &c->gs[c->ngs - 1].ctm;
    mat_point(m, &x1, &y1); mat_point(m, &x2, &y2); mat_point(m, &x3, &y3);
    path_curve_to(&c->path, fx_from_double(x1), fx_from_double(y1), fx_from_double(x2), fx_from_double(y2),
                  fx_from_double(x3), fx_from_double(y3));
}
EXPORT void swfo_close_path(swfo_ctx *c) { path_close(&c->path); }
/* cairo_get_current_point: the 24.8-quantised point mapped back through CTM^-1; (0,0) when none */
EXPORT int swfo_get_current_point(swfo_ctx *c, double *x, double *y)
{
    if (!c->path.has_cur) { *x = 0; *y = 0; return 0; }
    mat_t inv = c->gs[c->ngs - 1].ctm;
    *x = fx_to_double(c->path.cur.x); *y = fx_to_double(c->path.cur.y);
    if (mat_invert(&inv)) mat_point(&inv, x, y);
    return 1;
}
/* Context2d::QuadraticCurveTo of node-canvas 2.6.1 */
EXPORT void swfo_quadratic_curve_to(swfo_ctx *c, double x1, double y1, double x2, double y2)
{
    double x, y;
    swfo_get_current_point(c, &x, &y);
    if (x == 0 && y == 0) { x = x1; y = y1; }
    double k = 2.0 / 3.0;
    swfo_curve_to(c, x + k * (x1 - x), y + k * (y1 - y), x2 + k * (x1 - x2), y2 + k * (y1 - y2), x2, y2);
}
EXPORT void swfo_set_source_rgba(swfo_ctx *c, double r, double g, double b, double a)
{
    c->src.kind = SRC_SOLID; c->src.pixel = color_to_pixel(r, g, b, a);
}
static void lock_pattern_matrix(swfo_ctx *c)
{
    /* the pattern space is the user space at set_source time: device -> pattern = CTM^-1 */
    c->src.inv = c->gs[c->ngs - 1].ctm_inverse;
}
EXPORT void swfo_set_source_gradient(swfo_ctx *c, int linear, double x0, double y0, double r0, double x1, double y1, double r1,
                                     int nstops, const double *offsets, const double *rgba)
{
    c->src.kind = linear ? SRC_LINEAR : SRC_RADIAL;
    c->src.cx0 = x0; c->src.cy0 = y0; c->src.r0 = r0; c->src.cx1 = x1; c->src.cy1 = y1; c->src.r1 = r1;
    free(c->src.stops);
    c->src.stops = malloc(sizeof(stop_t) * (nstops ? nstops : 1)); c->src.nstops = nstops;
    for (int i = 0; i < nstops; i++) {
        stop_t s = { offsets[i], rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3] };
        /* cairo keeps stops sorted by offset, stable */
        int j = i;
        while (j > 0 && c->src.stops[j - 1].t > s.t) { c->src.stops[j] = c->src.stops[j - 1]; j--; }
        c->src.stops[j] = s;
    }
    lock_pattern_matrix(c);
}
EXPORT void swfo_set_source_surface(swfo_ctx *c, const uint32_t *argb_premul, int tw, int th, int extend)
{
    c->src.kind = SRC_SURFACE; c->src.tex = argb_premul; c->src.tw = tw; c->src.th = th; c->src.extend = extend;
    lock_pattern_matrix(c);
    source_setup_filter(&c->src);
}
EXPORT void swfo_set_line_width(swfo_ctx *c, double w) { c->gs[c->ngs - 1].line_width = w; }
EXPORT void swfo_set_line_cap(swfo_ctx *c, int cap) { c->gs[c->ngs - 1].cap = cap; }
EXPORT void swfo_set_line_join(swfo_ctx *c, int join) { c->gs[c->ngs - 1].join = join; }
EXPORT void swfo_set_fill_rule(swfo_ctx *c, int even_odd) { c->gs[c->ngs - 1].fill_rule = even_odd; }
EXPORT const uint32_t *swfo_pixels(swfo_ctx *c) { return c->px; }
EXPORT int swfo_last_polygon(swfo_ctx *c, const int32_t **edges, int *rectilinear)
{
    *edges = c->last_poly; *rectilinear = c->last_poly_rect;
    return c->last_poly_n;
}
EXPORT int swfo_is_clear(swfo_ctx *c) { return c->is_clear; }
EXPORT uint32_t swfo_debug_sample(swfo_ctx *c, int px, int py) { return sample_source(&c->src, px, py); }   /* diagnostics */
/* diagnostics: the pixman parameters of the current radial source as last prepared (tools/pixman_probe.py) */
EXPORT int swfo_debug_radial(swfo_ctx *c, int64_t *out /* 16: pm[6], pox, poy, c1x, c1y, c1r, dx, dy, dr, n_intervals, - */, int64_t *stops_x, float *ramp)
{
    const source_t *s = &c->src;
    if (s->kind != SRC_RADIAL || !s->g_x) return 0;
    for (int i = 0; i < 6; i++) out[i] = s->pm[i / 3][i % 3];
    out[6] = s->pox; out[7] = s->poy; out[8] = s->g_c1x; out[9] = s->g_c1y; out[10] = s->g_c1r; out[11] = s->g_dx; out[12] = s->g_dy; out[13] = s->g_dr;
    out[14] = s->g_n; out[15] = 0;
    for (int i = 0; i <= s->g_n; i++) stops_x[i] = s->g_x[i];
    for (int i = 0; i < 8 * s->g_n; i++) ramp[i] = s->g_ramp[i];
    return 1;
}

/* Direct entry for timing/large scenes: fill closed polygons given in 24.8 device coordinates.
   counts[i] vertices each, colours premultiplied ARGB.  Same code path as swfo_fill_preserve. */
EXPORT void swfo_fill_polygons_fixed(swfo_ctx *c, const int32_t *xy, const int32_t *counts, const uint32_t *argb,
                                     int npoly, int even_odd)
{
    const int32_t *q = xy;
    for (int i = 0; i < npoly; i++) {
        path_reset(&c->path);
        for (int k = 0; k < counts[i]; k++, q += 2) {
            if (k == 0) path_move_to(&c->path, q[0], q[1]); else path_line_to(&c->path, q[0], q[1]);
        }
        path_close(&c->path);                              /* closed polygons */
        c->src.kind = SRC_SOLID; c->src.pixel = argb[i];
        c->gs[c->ngs - 1].fill_rule = even_odd;
        swfo_fill_preserve(c);
    }
}
