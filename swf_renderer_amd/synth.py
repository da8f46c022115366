"""Synthetic benchmark scenes: seeded star polygons in integer twips.

scene(seed, n_shapes, W, H) draws, per shape and in this exact order, from numpy's PCG64
`default_rng(seed)`: R, cx, cy, phi, u, colour.  Vertex k has radius R (k even) or R*u (k odd)
at angle 2*pi*k/10 + phi; coordinates are rounded to integer twips; alpha is 255; painter's order
is the shape index; CTM = scale(1/20).  S1 = scene(0xC0FFEE, 1000, 3840, 2160) (10 000 edges),
S2 = scene(0xC0FFEE, 10000, 7680, 4320) (100 000 edges).
"""
from __future__ import annotations

import math

import numpy as np

S1 = dict(seed=0xC0FFEE, n_shapes=1000, width=3840, height=2160, rmin=8.0, rmax=256.0)
S2 = dict(seed=0xC0FFEE, n_shapes=10000, width=7680, height=4320, rmin=8.0, rmax=256.0)
# libcairo 1.16.0 known answers for S1: sha256 of tight RGBA8 bytes
S1_SHA256_PREMUL = "ed7048883df0d8b3798fb408870184a092f7336d54ca2c45121b0694474aa49c"
S1_SHA256_STRAIGHT = "4eea54ae14da493e11efa03d57789f6894eaf59a773661a88309d81938730bf3"


def scene(seed, n_shapes, width, height, verts=10, rmin=8.0, rmax=256.0):
    """Returns (twips int32 [n_shapes, verts, 2], colours uint8 [n_shapes, 4])."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n_shapes, verts, 2), dtype=np.int32)
    cols = np.empty((n_shapes, 4), dtype=np.uint8)
    for i in range(n_shapes):
        R = math.exp(rng.uniform(math.log(rmin), math.log(rmax)))
        cx = rng.uniform(R, width - R)
        cy = rng.uniform(R, height - R)
        phi = rng.uniform(0, 2 * math.pi)
        u = rng.uniform(0.4, 1.0)
        for k in range(verts):
            rad = R if k % 2 == 0 else R * u
            ang = 2 * math.pi * k / verts + phi
            pts[i, k, 0] = int(np.rint(20 * (cx + rad * math.cos(ang))))
            pts[i, k, 1] = int(np.rint(20 * (cy + rad * math.sin(ang))))
        cols[i, :3] = rng.integers(0, 256, 3)
        cols[i, 3] = 255
    return pts, cols


def twips_to_fixed(pts_twips):
    """CTM scale(1/20) then Cairo's 24.8 conversion: rint(x/20*256) (ties cannot occur: 12.8*n)."""
    return np.rint(pts_twips.astype(np.float64) * (1.0 / 20.0) * 256.0).astype(np.int32)
