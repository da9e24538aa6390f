// colour_diff.hpp -- per-pixel colour differences on gfx950, fp64 "reference form".
//
// These are the arbiter versions: same operation order as the reference's CPU
// functions (src/Photomosaic/ColourDifference.cpp:28-33 and :42-158), all fp64,
// and the translation unit is built with -ffp-contract=off so no a*b+c is fused.
// The reference's own CUDA device functions (CUDA/ColourDifference.cuh:24-139)
// do a few products in fp32 first and therefore differ from its CPU path in the
// last bits; the parity contract here is the CPU path.
//
// For RGB/CIE76 the result is bit-identical to the CPU: differences of
// f32-valued doubles and their squares are computed with the same IEEE
// operations and the fp64 sqrt lowering on gfx950 is correctly rounded.
// For CIEDE2000 the only deviations are the ocml vs host-libm roundings of
// atan2 / sin / cos / exp (<= 1-2 ulp each).
#pragma once

#include <hip/hip_runtime.h>

namespace mm {

enum : int { RGB_EUCLIDEAN = 0, CIE76 = 1, CIEDE2000 = 2 };

struct Px { double c0, c1, c2; };   // BGR or Lab triplet promoted f32 -> f64 (Vec3f -> Vec3d)

// ColourDifference::calculateRGBEuclidean, ColourDifference.cpp:28-33.
__device__ __forceinline__ double euclidean(const Px& p, const Px& q)
{
    const double d0 = p.c0 - q.c0, d1 = p.c1 - q.c1, d2 = p.c2 - q.c2;
    return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

__device__ __forceinline__ double seventh_power(double x) { return x * x * x * x * x * x * x; }

// Hue angle in [0, 2pi), ColourDifference.cpp:61-83.
__device__ __forceinline__ double hue_angle(double b, double a_prime, double two_pi)
{
    double h = atan2(b, a_prime);
    if (h < 0.0) h += two_pi;
    return (b == 0.0 && a_prime == 0.0) ? 0.0 : h;
}

// ColourDifference::calculateCIEDE2000, ColourDifference.cpp:42-158.
// p = main-image (cell) pixel, q = library pixel (call order of
// CPUPhotomosaicGenerator.cpp:159).
__device__ __forceinline__ double ciede2000(const Px& p, const Px& q)
{
    constexpr double kPi = 3.14159265358979323846;
    constexpr double two_pi = (360.0 * kPi) / 180.0, pi = (180.0 * kPi) / 180.0;
    constexpr double rad6 = (6.0 * kPi) / 180.0, rad25 = (25.0 * kPi) / 180.0;
    constexpr double rad30 = (30.0 * kPi) / 180.0, rad63 = (63.0 * kPi) / 180.0;
    constexpr double rad275 = (275.0 * kPi) / 180.0, rad550 = (550.0 * kPi) / 180.0;
    constexpr double twentyfive_7 = 6103515625.0;

    const double c1 = sqrt((p.c1 * p.c1) + (p.c2 * p.c2));
    const double c2 = sqrt((q.c1 * q.c1) + (q.c2 * q.c2));
    const double c_bar = (c1 + c2) / 2.0;
    const double c_bar7 = seventh_power(c_bar);
    const double g = 0.5 * (1.0 - sqrt(c_bar7 / (c_bar7 + twentyfive_7)));

    const double a1p = (1.0 + g) * p.c1;
    const double a2p = (1.0 + g) * q.c1;
    const double c1p = sqrt((a1p * a1p) + (p.c2 * p.c2));
    const double c2p = sqrt((a2p * a2p) + (q.c2 * q.c2));
    const double h1p = hue_angle(p.c2, a1p, two_pi);
    const double h2p = hue_angle(q.c2, a2p, two_pi);

    const double d_l = q.c0 - p.c0;
    const double d_c = c2p - c1p;
    const double cp_prod = c1p * c2p;
    double d_h_small = h2p - h1p;
    if (d_h_small < -pi) d_h_small += two_pi;
    else if (d_h_small > pi) d_h_small -= two_pi;
    if (cp_prod == 0.0) d_h_small = 0.0;
    const double d_h = 2.0 * sqrt(cp_prod) * sin(d_h_small / 2.0);

    const double l_bar = (p.c0 + q.c0) / 2.0;
    const double cp_bar = (c1p + c2p) / 2.0;
    const double h_sum = h1p + h2p;
    double h_bar;
    if (cp_prod == 0.0) h_bar = h_sum;
    else if (fabs(h1p - h2p) <= pi) h_bar = h_sum / 2.0;
    else if (h_sum < two_pi) h_bar = (h_sum + two_pi) / 2.0;
    else h_bar = (h_sum - two_pi) / 2.0;

    const double t = 1.0 - (0.17 * cos(h_bar - rad30))
                   + (0.24 * cos(2.0 * h_bar))
                   + (0.32 * cos((3.0 * h_bar) + rad6))
                   - (0.20 * cos((4.0 * h_bar) - rad63));

    const double d_theta = rad30 *
        exp(-((h_bar * (h_bar - rad550) + (rad275 * rad275)) / (rad25 * rad25)));

    const double cp_bar7 = seventh_power(cp_bar);
    const double r_c = 2.0 * sqrt(cp_bar7 / (cp_bar7 + twentyfive_7));
    const double s_l = 1.0 + ((0.015 * (l_bar * (l_bar - 100.0) + 2500.0))
                              / sqrt(20 + l_bar * (l_bar - 100.0) + 2500.0));
    const double s_c = 1.0 + (0.045 * cp_bar);
    const double s_h = 1.0 + (0.015 * cp_bar * t);
    const double r_t = (-sin(2.0 * d_theta)) * r_c;

    return sqrt((d_l * d_l) / (s_l * s_l)
              + (d_c * d_c) / (s_c * s_c)
              + (d_h * d_h) / (s_h * s_h)
              + (r_t * (d_c / s_c) * (d_h / s_h)));
}

template <int TYPE>
__device__ __forceinline__ double colour_diff(const Px& p, const Px& q)
{
    if constexpr (TYPE == CIEDE2000) return ciede2000(p, q);
    else return euclidean(p, q);
}

}  // namespace mm
