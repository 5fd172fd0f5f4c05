// host_math.cpp -- per-frame scalar derivations done once on the host, with the reference's operation
// order (compiled -ffp-contract=off so nothing is fused).  These are the parts of Raytracer.render /
// generateRay that do not depend on the pixel: scene constants (src/Raytracer.hs:57-65) and the look-at
// basis (linear's lookAt, src/Raytracer.hs:47), plus the catalogue record parser (src/StarMap.hs:45-75).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "bs_internal.h"

namespace bs {

namespace {
inline double quadrance(const double v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }

inline void cross(const double a[3], const double b[3], double o[3])
{
    // linear: cross (V3 a b c) (V3 d e f) = V3 (b*f-c*e) (c*d-a*f) (a*e-b*d)
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

inline void normalize(const double v[3], double o[3])
{
    // linear: normalize v = if nearZero l || nearZero (1-l) then v else fmap (/sqrt l) v ; nearZero = (<= 1e-12) . abs
    double l = quadrance(v);
    if (std::fabs(l) <= 1e-12 || std::fabs(1.0 - l) <= 1e-12) {
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {
        double s = std::sqrt(l);
        o[0] = v[0] / s; o[1] = v[1] / s; o[2] = v[2] / s;
    }
}
}  // namespace

namespace {
// toWord8 (sRGB x): Raytracer.hs:23-32 with massiv-io's toWord8 (clamp to [0,1], * 255, round half to even)
inline int srgb8_forward(double x)
{
    double y = (x < 0.0031308) ? 12.92 * x : (1 + 0.055) * std::pow(x, 1.0 / 2.4) - 0.055;
    y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);
    return (int)std::rint(255.0 * y);
}
}  // namespace

void srgb8_thresholds(double T[257])
{
    // Positive doubles are ordered like their bit patterns, and the map is monotone: bisect the patterns in (0, 2.0] for the
    // first x whose byte reaches k.  (x <= 0 maps to 0, x >= 1 to 255.)
    T[0] = -HUGE_VAL;
    T[256] = HUGE_VAL;
    auto from_bits = [](uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; };
    auto to_bits = [](double d) { uint64_t b; std::memcpy(&b, &d, 8); return b; };
    for (int k = 1; k <= 255; k++) {
        uint64_t lo = 0, hi = to_bits(2.0);  // byte(lo) = 0 < k <= 255 = byte(hi)
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (srgb8_forward(from_bits(mid)) >= k) hi = mid; else lo = mid;
        }
        T[k] = from_bits(hi);
    }
}

void host_hsi_to_rgb(double hp, double s, double i, double rgb[3], bool *ok)
{
    // massiv-io Graphics.ColorSpace: toPixelRGB (PixelHSI h' s i)  (recalled)
    const double pi = 3.141592653589793;
    double h = hp * 2 * pi;
    double is = i * s;
    double second = i - is;
    *ok = true;
    if (h >= 0 && h < 2 * pi / 3) {
        double r = i + is * std::cos(h) / std::cos(pi / 3 - h);
        double b = second;
        double g = i + 2 * is + b - r;
        rgb[0] = r; rgb[1] = g; rgb[2] = b;
    } else if (h >= 0 && h < 4 * pi / 3) {
        double g = i + is * std::cos(h - 2 * pi / 3) / std::cos(h + pi);
        double r = second;
        double b = i + 2 * is + r - g;
        rgb[0] = r; rgb[1] = g; rgb[2] = b;
    } else if (h >= 0 && h < 2 * pi) {
        double b = i + is * std::cos(h - 4 * pi / 3) / std::cos(2 * pi - pi / 3 - h);
        double g = second;
        double r = i + 2 * is + g - b;
        rgb[0] = r; rgb[1] = g; rgb[2] = b;
    } else {
        // the reference raises `error "HSI pixel is not properly scaled"`
        rgb[0] = rgb[1] = rgb[2] = std::nan("");
        *ok = false;
    }
}

// 1 / b correctly rounded, for trace_device.h div_by -- or 0 when b is no divisor div_by may be used with: not finite, outside [2^-300, 2^300],
// or with a significand of all ones (the one case in which q = RN(a y), r = a - b q, RN(q + r y) can miss the correctly rounded a / b).
static double exact_reciprocal(double b)
{
    if (!(std::fabs(b) >= 0x1p-300 && std::fabs(b) <= 0x1p300)) return 0.0;
    uint64_t bits;
    std::memcpy(&bits, &b, sizeof bits);
    if ((bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull) return 0.0;
    return 1.0 / b;
}

bool derive_params(const bs_config &c, TraceParams &p, std::string &err)
{
    if (c.width <= 0 || c.height <= 0) { err = "resolution must be positive"; return false; }
    if ((long long)c.width * c.height > (1LL << 28)) { err = "resolution too large"; return false; }
    // Inputs the reference does not survive either.  colorize has no iteration cap (src/Raytracer.hs:80-86): a NaN in the ray state makes
    // every guard of findColor false (:93-98) and the loop never ends; a non-positive stepSize never moves a ray out of [1, safeDistance];
    // lookAt == position gives every ray the velocity 0 (linear's normalize returns the zero vector unchanged).  Behind a blocking C call
    // that would be max_steps x rays steps inside one uninterruptible kernel (seconds to minutes), so such configurations are refused up
    // front.  ONLY such: what enters the ray state is the camera (position, lookAt, upVec, fov) and stepSize.  Everything else the
    // reference renders, this renders -- negative disk radii (render squares them, :61-62, so -3 is 3), infinite or NaN disk / star
    // parameters (safeDistance depends on the camera alone, :59-60, so every ray still ends; pixels come out inf / NaN as they do there).
    {
        const struct { const char *name; const double *v; int n; } fields[] = {
            {"camera.position", c.cam_pos, 3}, {"camera.lookAt", c.cam_lookat, 3}, {"camera.upVec", c.cam_up, 3}, {"camera.fov", &c.fov, 1},
            {"scene.stepSize", &c.step_size, 1}};
        for (const auto &f : fields)
            for (int i = 0; i < f.n; i++)
                if (!std::isfinite(f.v[i])) { err = std::string(f.name) + " is not finite (the reference's colorize would never terminate: src/Raytracer.hs:80-86)"; return false; }
        if (!(c.step_size > 0)) { err = "scene.stepSize must be positive (the reference's colorize would never terminate: src/Raytracer.hs:80-86)"; return false; }
        {   // linear's normalize leaves a vector with |v|^2 <= 1e-12 as it is: a view direction that short makes every ray's velocity
            // that short too (generateRay never gets a unit vector to work with), and the rays crawl: >= 1e8 steps to leave the scene
            const double dv[3] = {c.cam_lookat[0] - c.cam_pos[0], c.cam_lookat[1] - c.cam_pos[1], c.cam_lookat[2] - c.cam_pos[2]};
            if (quadrance(dv) <= 1e-12) {
                err = "camera.lookAt equals camera.position (to within 1e-6): no viewing direction -- every ray would have velocity ~0 and never terminate";
                return false;
            }
        }
    }
    std::memcpy(p.cam, c.cam_pos, sizeof p.cam);
    // linear lookAt eye center up: za = normalize (center - eye); xa = normalize (cross za up); ya = cross xa za
    double d[3] = {c.cam_lookat[0] - c.cam_pos[0], c.cam_lookat[1] - c.cam_pos[1], c.cam_lookat[2] - c.cam_pos[2]};
    double t[3];
    normalize(d, p.za);
    cross(p.za, c.cam_up, t);
    normalize(t, p.xa);
    cross(p.xa, p.za, p.ya);
    p.fov = c.fov;
    p.ss = c.supersampling ? 1 : 0;
    p.wt = p.ss ? 2 * c.width : c.width;    // Raytracer.hs:58
    p.ht = p.ss ? 2 * c.height : c.height;
    p.out_w = c.width;
    p.out_h = c.height;
    p.W = (double)p.wt;
    p.H = (double)p.ht;
    {   // trace_device.h div_by: a / b from y = RN(1 / b) in three instructions IS IEEE division unless b's significand is all ones or something
        // under- / overflows on the way: the reciprocal is handed over only for divisors in [2^-300, 2^300], and for generate_ray only if every
        // number its numerators are made of (basis, fov) is zero or at least 2^-300 (then no quotient can be subnormal)
        auto usable = [](double x) { return x == 0.0 || (std::fabs(x) >= 0x1p-300 && std::fabs(x) <= 0x1p300); };
        bool cam_ok = usable(c.fov) && c.fov != 0.0;
        for (int i = 0; i < 3; i++) cam_ok = cam_ok && usable(p.xa[i]) && usable(p.ya[i]) && usable(p.za[i]);
        p.inv_W = cam_ok ? exact_reciprocal(p.W) : 0.0;
        p.inv_H = cam_ok ? exact_reciprocal(p.H) : 0.0;
        if (p.inv_W == 0.0 || p.inv_H == 0.0) p.inv_W = p.inv_H = 0.0;
    }
    p.h = c.step_size;
    p.hh = c.step_size / 2;  // rk4: h / 2, h / 6  (:130-134)
    p.h6 = c.step_size / 6;
    p.hh2 = p.hh * p.hh;              // FAST mode only (see trace_kernel.hip rk4<true>)
    p.hhh = c.step_size * p.hh;
    p.h2_6 = c.step_size * p.h6;
    p.rcam = std::sqrt(quadrance(c.cam_pos));
    for (int i = 0; i < 3; i++) p.e1[i] = p.rcam > 0 ? c.cam_pos[i] / p.rcam : (i == 0 ? 1.0 : 0.0);
    double a = 50.0 * 50.0, b = 2 * quadrance(c.cam_pos);  // :59-60, max x y = if x <= y then y else x
    p.safe = (a <= b) ? b : a;
    p.in2 = c.disk_inner * c.disk_inner;   // :61
    p.out2 = c.disk_outer * c.disk_outer;  // :62
    p.rI = std::sqrt(p.in2);               // :107
    p.rO = std::sqrt(p.out2);              // :108
    bool ok;
    host_hsi_to_rgb(c.disk_hsi[0], c.disk_hsi[1], c.disk_hsi[2], p.disk_rgb, &ok);  // :65
    if (!ok) { err = "HSI pixel is not properly scaled (diskColor hue outside [0,360))"; return false; }
    p.disk_opacity = c.disk_opacity;
    p.star_intensity = c.star_intensity;
    p.star_saturation = c.star_saturation;
    p.star_a = std::log(2.0) / 50;  // StarMap.hs:108  a = log 2 / dynamic
    // FAST mode's guard (trace_device.h: trace_ray): a ray that needs more steps than the longest straight path through the
    // scene, N0 = (|camera| + sqrt safeDistance) / stepSize, plus one photon-sphere circumference (2 pi 1.5 ~ 9.4 -> 9.0 / stepSize
    // steps) has orbited the hole about once; every further orbit multiplies ANY rounding difference by e^(2 pi) ~ 535.  Measured
    // (scripts/fast_guard_probe.py, profiles/r02_fast_guard_probe.txt): FAST - STRICT grows 10x per 10 excess steps at h = 0.3, is
    // <= 1.2e-7 relative below 30 and reaches 1e-6 .. 2e-5 beyond; 4 .. 16 rays per million are beyond.  Those are re-traced in STRICT.
    {
        const double n0 = (p.rcam + std::sqrt(p.safe)) / c.step_size + 9.0 / c.step_size;
        p.guard_steps = (n0 > 0 && n0 < 2.0e9) ? (int32_t)std::ceil(n0) : INT32_MAX;
    }
    return true;
}

}  // namespace bs

extern "C" int bs_hsi_to_rgb(double hue, double sat, double intensity, double rgb[3])
try {
    if (!rgb) return BS_EINVAL;
    bool ok;
    bs::host_hsi_to_rgb(hue, sat, intensity, rgb, &ok);
    return ok ? BS_OK : BS_EINVAL;
} catch (...) { return bs::abi_exception("bs_hsi_to_rgb"); }

extern "C" long bs_read_ppm(const void *bytes, size_t nbytes, bs_star *out, size_t cap)
try {
    // readMap (StarMap.hs:45-58): skip 28; replicateM (remaining `div` 28) of
    //   getFloat64be ra, getFloat64be dec, getWord8 spectral, skip 1, getInt16be mag, skip 8
    if (!bytes || nbytes < 28) return BS_EINVAL;
    const unsigned char *b = static_cast<const unsigned char *>(bytes);
    size_t n = (nbytes - 28) / 28;
    auto be64 = [](const unsigned char *q) {
        uint64_t u = 0;
        for (int i = 0; i < 8; i++) u = (u << 8) | q[i];
        double dd;
        std::memcpy(&dd, &u, 8);
        return dd;
    };
    for (size_t i = 0; i < n && i < cap && out; i++) {
        const unsigned char *r = b + 28 + i * 28;
        double ra = be64(r), dec = be64(r + 8);
        bs_star &s = out[i];
        s.x = std::cos(dec) * std::cos(ra);  // raDecToCartesian (StarMap.hs:74-75)
        s.y = std::cos(dec) * std::sin(ra);
        s.z = std::sin(dec);
        s.mag = (int16_t)(((uint16_t)r[18] << 8) | r[19]);
        s._pad = 0;
        switch (r[16]) {  // starColor (StarMap.hs:64-72)
        case 'O': s.hue = 0.631; s.sat = 0.39; break;
        case 'B': s.hue = 0.628; s.sat = 0.33; break;
        case 'A': s.hue = 0.622; s.sat = 0.21; break;
        case 'F': s.hue = 0.650; s.sat = 0.03; break;
        case 'G': s.hue = 0.089; s.sat = 0.09; break;
        case 'K': s.hue = 0.094; s.sat = 0.29; break;
        case 'M': s.hue = 0.094; s.sat = 0.56; break;
        default: s.hue = 0; s.sat = 0; break;
        }
    }
    return (long)n;
} catch (...) { return bs::abi_exception("bs_read_ppm"); }
