/*
 * Deterministic synthetic "corridor" stereo sequence:
 * analytic per-pixel ray cast of an ideal rectified rig (zero distortion, R = I,
 * T = (-B, 0, 0)) through ground plane y = +1.65 m, side walls x = +-half_width and a
 * back wall FIXED in the world at z = z_far (SURVEY 8(d) proposed a wall moving with the
 * camera; its features contradict the camera motion and wreck the reference's outlier-free
 * Umeyama fit, so the wall is world-anchored and far instead); world-anchored 3-octave value-noise
 * texture from an integer lattice hash (seed 1234).  Bench/test input generator --
 * not part of the reference and not part of the oracle.
 */
#include <math.h>
#include <stdint.h>

typedef struct {
    int w, h;
    double f, cx, cy, baseline, half_width, z_far, ground_y;
    uint32_t seed;
} vo_corridor_cfg;

static inline uint32_t lattice(int32_t ix, int32_t iy, uint32_t plane, uint32_t seed)
{
    uint32_t v = ((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ (plane * 83492791u) ^ seed;
    return (v * 2654435761u) >> 24;
}

static inline double octave(double a, double b, double cell, uint32_t plane, uint32_t seed)
{
    double u = a / cell, v = b / cell;
    double fu = floor(u), fv = floor(v);
    int32_t iu = (int32_t)fu, iv = (int32_t)fv;
    double tu = u - fu, tv = v - fv;
    double l00 = lattice(iu, iv, plane, seed), l10 = lattice(iu + 1, iv, plane, seed);
    double l01 = lattice(iu, iv + 1, plane, seed), l11 = lattice(iu + 1, iv + 1, plane, seed);
    return (l00 * (1 - tu) + l10 * tu) * (1 - tv) + (l01 * (1 - tu) + l11 * tu) * tv;
}

static inline uint8_t shade(double a, double b, uint32_t plane, uint32_t seed)
{
    double t = 0.5 * octave(a, b, 0.05, plane * 4 + 0, seed) +
               0.3 * octave(a, b, 0.25, plane * 4 + 1, seed) +
               0.2 * octave(a, b, 1.0, plane * 4 + 2, seed);
    int q = (int)floor(t + 0.5);
    return (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
}

/* camera pose of frame k: position (x, 0, z), yaw about +y */
void vo_corridor_pose(int k, double* x, double* z, double* yaw)
{
    *z = 0.25 * k;
    *x = 0.3 * sin(0.05 * k);
    *yaw = 0.01 * sin(0.07 * k);
}

static void render(const vo_corridor_cfg* c, double px, double pz, double yaw, double ox, uint8_t* img)
{
    const double cs = cos(yaw), sn = sin(yaw);
    /* optical centre: left camera at (px,0,pz), offset ox along the camera x axis */
    const double ex = px + cs * ox, ez = pz - sn * ox;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < c->h; v++)
        for (int u = 0; u < c->w; u++) {
            double dx = (u - c->cx) / c->f, dy = (v - c->cy) / c->f, dz = 1.0;
            double wx = cs * dx + sn * dz, wy = dy, wz = -sn * dx + cs * dz;
            double tbest = 1e30;
            int plane = 3;
            if (wy > 1e-12) { double t = c->ground_y / wy; if (t < tbest) { tbest = t; plane = 0; } }
            if (wx < -1e-12) { double t = (-c->half_width - ex) / wx; if (t > 0 && t < tbest) { tbest = t; plane = 1; } }
            if (wx > 1e-12) { double t = (c->half_width - ex) / wx; if (t > 0 && t < tbest) { tbest = t; plane = 2; } }
            if (wz > 1e-12) { double t = (c->z_far - ez) / wz; if (t > 0 && t < tbest) { tbest = t; plane = 3; } }
            double X = ex + tbest * wx, Y = tbest * wy, Z = ez + tbest * wz;
            uint8_t g;
            if (plane == 0) g = shade(X, Z, 0, c->seed);
            else if (plane == 1 || plane == 2) g = shade(Z, Y, (uint32_t)plane, c->seed);
            else g = shade(X, Y, 3, c->seed);
            img[(long)v * c->w + u] = g;
        }
}

void vo_corridor_render_pair(const vo_corridor_cfg* c, int k, uint8_t* left, uint8_t* right)
{
    double x, z, yaw;
    vo_corridor_pose(k, &x, &z, &yaw);
    render(c, x, z, yaw, 0.0, left);
    render(c, x, z, yaw, c->baseline, right);
}
