/* Diagnostic (container only): logs what libcairo hands libpixman -- the pixman transform of a pattern, the circles and stops of a
   radial gradient, the composite rectangles -- so that the oracle's restatement (SWFO_TRACE_SOURCE=1 prints its own) can be compared
   value by value.  Found the "skip the translation fix when the centre leaves 16.16" rule.

     gcc -shared -fPIC -O1 -o /tmp/pixman_log_shim.so tools/pixman_log_shim.c -ldl
     LD_PRELOAD=/tmp/pixman_log_shim.so SWFO_TRACE_SOURCE=1 python tools/soak_case.py big 300 9 2>&1 | grep -i transform

   libpixman is loaded by ctypes with local scope, so the real entry points are looked up with dlopen, not RTLD_NEXT. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { int32_t m[3][3]; } transform_t;
typedef struct { int32_t x, y; } point_t;
typedef struct { int32_t x; uint16_t r, g, b, a; } stop_t;

static void *real_of(const char *name) { return dlsym(dlopen("libpixman-1.so.0", RTLD_NOW), name); }

int pixman_image_set_transform(void *image, const transform_t *t)
{
    static int (*real)(void *, const transform_t *);
    if (!real) real = (int (*)(void *, const transform_t *))real_of("pixman_image_set_transform");
    if (t) fprintf(stderr, "pixman transform %d %d %d | %d %d %d | %d %d %d\n", t->m[0][0], t->m[0][1], t->m[0][2], t->m[1][0], t->m[1][1], t->m[1][2],
                   t->m[2][0], t->m[2][1], t->m[2][2]);
    return real(image, t);
}

void *pixman_image_create_radial_gradient(const point_t *inner, const point_t *outer, int32_t inner_radius, int32_t outer_radius, const stop_t *stops, int n)
{
    static void *(*real)(const point_t *, const point_t *, int32_t, int32_t, const stop_t *, int);
    if (!real) real = (void *(*)(const point_t *, const point_t *, int32_t, int32_t, const stop_t *, int))real_of("pixman_image_create_radial_gradient");
    fprintf(stderr, "pixman radial inner %d %d r %d outer %d %d r %d stops %d\n", inner->x, inner->y, inner_radius, outer->x, outer->y, outer_radius, n);
    for (int i = 0; i < n; i++) fprintf(stderr, "pixman  stop %d : %u %u %u %u\n", stops[i].x, stops[i].r, stops[i].g, stops[i].b, stops[i].a);
    return real(inner, outer, inner_radius, outer_radius, stops, n);
}

void pixman_image_composite32(int op, void *src, void *mask, void *dst, int32_t sx, int32_t sy, int32_t mx, int32_t my, int32_t dx, int32_t dy, int32_t w, int32_t h)
{
    static void (*real)(int, void *, void *, void *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t);
    if (!real) real = (void (*)(int, void *, void *, void *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t))real_of("pixman_image_composite32");
    if (getenv("PIXMAN_LOG_COMPOSITE")) fprintf(stderr, "pixman composite op %d src %d %d mask %d %d dst %d %d size %d %d\n", op, sx, sy, mx, my, dx, dy, w, h);
    real(op, src, mask, dst, sx, sy, mx, my, dx, dy, w, h);
}
