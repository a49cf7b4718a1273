/* nvdr_raytri.h -- the ray/triangle predicate that DEFINES a shadow-ray hit in this code base.
 *
 * In the reference the predicate lives in closed-source OptiX 7.3 (optixTrace with
 * tmin = 0, tmax = 1e16, no culling, terminate on first hit: kernel.cu:101-118).  Here it is a
 * division-free, two-sided Moeller-Trumbore test evaluated with correctly rounded IEEE-754
 * operations only, in a fixed order, so that the CPU oracle (brute force over all triangles) and
 * the gfx950 BVH traversal return bit-identical answers for the same ray and triangle.
 *
 * Triangle record: v0, e1 = v1 - v0, e2 = v2 - v0 (each rounded once, in float).
 */
#ifndef NVDR_RAYTRI_H
#define NVDR_RAYTRI_H

#include "nvdr_detmath.h"

#define NVDR_RAY_TMAX 1e16f

/* dot(a, b) = fma(ax, bx, fma(ay, by, az*bz)) */
NVDR_HD float nvdr_dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return NVDR_FMA(ax, bx, NVDR_FMA(ay, by, az * bz));
}

/* Signed (by det) barycentrics and distance.  Returns 1 when the ray o + t*d hits the triangle
 * for some t in (0, NVDR_RAY_TMAX).  On a hit, *t_num, *u_num, *v_num, *det_abs hold the numerators
 * (already multiplied by sign(det)) and |det|: t = t_num / det_abs etc. */
NVDR_HD int nvdr_ray_tri(float ox, float oy, float oz, float dx, float dy, float dz,
                         float v0x, float v0y, float v0z, float e1x, float e1y, float e1z,
                         float e2x, float e2y, float e2z,
                         float *t_num, float *u_num, float *v_num, float *det_abs)
{
    /* p = d x e2 */
    const float px = NVDR_FMA(dy, e2z, -(dz * e2y));
    const float py = NVDR_FMA(dz, e2x, -(dx * e2z));
    const float pz = NVDR_FMA(dx, e2y, -(dy * e2x));
    const float det = nvdr_dot3(e1x, e1y, e1z, px, py, pz);
    const float sgn = det < 0.0f ? -1.0f : 1.0f;
    const float adet = det * sgn;
    const float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    const float u = nvdr_dot3(tx, ty, tz, px, py, pz) * sgn;
    /* q = tv x e1 */
    const float qx = NVDR_FMA(ty, e1z, -(tz * e1y));
    const float qy = NVDR_FMA(tz, e1x, -(tx * e1z));
    const float qz = NVDR_FMA(tx, e1y, -(ty * e1x));
    const float v = nvdr_dot3(dx, dy, dz, qx, qy, qz) * sgn;
    const float t = nvdr_dot3(e2x, e2y, e2z, qx, qy, qz) * sgn;
    *t_num = t; *u_num = u; *v_num = v; *det_abs = adet;
    /* written so that any NaN fails the test */
    return (u >= 0.0f) & (v >= 0.0f) & (u + v <= adet) & (t > 0.0f) & (t < NVDR_RAY_TMAX * adet);
}

#endif /* NVDR_RAYTRI_H */
