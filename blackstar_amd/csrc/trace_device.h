// trace_device.h -- the device functions of the geodesic trace (one wavefront lane per ray), shared by the product kernels
// (trace_kernel.hip: the frame kernel and the batched starLookup) and the test-hook kernels of the debug library (debug_kernels.hip:
// per-ray records, sqrt / divide, the FP64 issue-rate probe).  Everything here is __forceinline__ / internal linkage: each
// translation unit gets its own copy, compiled with the same flags (-ffp-contract=off).
//
//
// Implements, per lane, the reference's per-pixel function (file:line into /root/reference):
//   generateRay  src/Raytracer.hs:40-51      traceRay/colorize :69-86     findColor :88-102
//   diskColor'   :104-111                    rk4 :113-134                 blend :34-37
//   starLookup   src/StarMap.hs:93-115       supersample src/ImageFilters.hs:88-97 (fused epilogue)
//
// The path is scalar FP64 ODE work: ~145 flop per RK4 step per ray against <= 24 B written per ray, so the
// bound is the FP64 VALU issue rate (v_fma_f64 / v_mul_f64 / v_add_f64 at 16 lanes/clk/SIMD), not HBM and
// not MFMA (there is no dense contraction to feed a matrix core).  The ray state lives in VGPRs for the whole
// ray; the stepping loop touches no memory at all (rare per-lane events go to LDS, see "per-lane LDS scratch").
//
// This translation unit is compiled with -ffp-contract=off: STRICT mode is one IEEE binary64 operation
// per reference operation in the reference's order (f64 sqrt and / lower to correctly rounded sequences),
// which makes step counts, fates and terminal states bit-identical to the CPU oracle.  FAST mode spells
// its FMAs explicitly.
#pragma once
#include <hip/hip_runtime.h>

#include "bs_internal.h"
#include "fast_loop_asm.h"

namespace bs {
namespace {

#ifndef BS_EXACT_SHORTCUTS
#define BS_EXACT_SHORTCUTS 1  // generate_ray: divisions by the (wave-uniform) resolution and the normalisation as short correctly-rounded sequences; 0 = the compiler's
#endif
#ifndef BS_ASM_LOOP
#define BS_ASM_LOOP 1  // the FAST stepping loop: 1 = fast_loop_asm.h, 0 = the C++ statement of the same steps (A/B knob, and the readable version; the same bits as
                       // the assembly built with -DBS_FL_SERIES=0)
#endif
constexpr int kBlock = 256;  // 4 wavefronts per workgroup; each wavefront traces one 8x8 tile of traced pixels at a time
#ifndef BS_MIN_WAVES
#define BS_MIN_WAVES 4  // __launch_bounds__ minimum waves per SIMD: 4 workgroups per CU is what the LDS budget admits
#endif

__device__ __forceinline__ double quadrance(double x, double y, double z) { return (x * x + y * y) + z * z; }

// libm calls as REAL calls.  ocml's f64 sin / cos / exp carry a large-argument (Payne-Hanek) path that is never taken
// here but costs ~30 VGPRs wherever it is inlined; the shading code sits outside the stepping loop, so a call is
// free and keeps the kernel at <= 111 VGPRs with no scratch.
__device__ __noinline__ double sin_call(double x) { return sin(x); }
__device__ __noinline__ double exp_call(double x) { return exp(x); }

// GHC.Float signum: x>0 -> 1, x<0 -> -1, otherwise x (so signum 0 = 0).
__device__ __forceinline__ double signum(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }

// signum a /= signum b for the disk guard (Raytracer.hs:96) as four compares on the lane masks instead of two selects and a compare on
// doubles (13 VALU -> 4).  For NaN-free operands it IS the reference's test (signum values are 1, -1 and +-0, and +0 == -0).  With a NaN
// operand the reference's test is true and this one may be false -- but the only consumer then computes r2ave = NaN, which fails both radius
// compares (:97): no layer either way.
__device__ __forceinline__ bool signum_differs(double a, double b) { return ((a > 0) != (b > 0)) || ((a < 0) != (b < 0)); }

// "this lane's bit is set in the wave-uniform mask m" as exec &= m (s_and_saveexec: no VALU).  Spelled ((m >> lane) & 1) it costs two v_and,
// a v_cmp_ne_u64 and two registers holding 1 << lane -- in blocks that run ~30 times per tile of 224 steps.
__device__ __forceinline__ bool in_mask(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// ---- correctly rounded f64 sqrt / divide without the range scaling -------------------------------------
// hipcc lowers f64 sqrt and '/' to exactly these FMA sequences wrapped in v_ldexp / v_div_scale /
// v_div_fixup range handling (only active for |x| < 2^-767 or extreme exponent gaps).  In the RK4 RHS the
// operands are r^2 in [~1e-3, ~1e4] and r^5, so the scaling never triggers and the bare sequences return
// the same bits -- tests/test_gpu_parity.py checks both against IEEE results on 2^20 operands.
__device__ __forceinline__ double sqrt_rn(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

__device__ __forceinline__ double div_rn(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// a / b in three instructions when y = the correctly rounded 1 / b is at hand: q = RN(a y) is within an ulp of a / b, r = a - b q is exact in
// an FMA, and RN(q + r y) is then the correctly rounded quotient (Markstein) -- i.e. the bits of IEEE division, for every b whose significand is
// not all ones.  Used where b is wave-uniform and its reciprocal comes from the host (the traced width and height: integers <= 2^16;
// tests/test_host.py checks every x / W, x < W <= 16384 and 65 M random numerators against the CPU's division: none differs).
__device__ __forceinline__ double div_by(double a, double b, double y)
{
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// v / s for three numerators and ONE per-lane divisor: the refined reciprocal of div_rn once (rcp + two Newton steps), then div_by's three
// instructions per quotient -- div_rn's own sequence, which the GPU suite checks against IEEE division (14 instead of 3 x 8 instructions; the
// compiler's '/' with its range scaling is 3 x 13).
__device__ __forceinline__ void div3_rn(const double a[3], double b, double out[3])
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = div_by(a[i], b, y);
}

// Where the bare sequences (sqrt_rn, div_rn, div_by) return IEEE's bits: nothing on the way may under- or overflow.
__device__ __forceinline__ bool mid_range(double x) { return x >= 0x1p-500 && x <= 0x1p500; }  // false for NaN

// STRICT: c = (1.5*h2) / |pos|^5 with q = quadrance pos given   (Raytracer.hs:127)
__device__ __forceinline__ double coef_strict(double h2c, double q)
{
    double n = sqrt_rn(q);      // norm pos
    double n2 = n * n;
    double n5 = (n2 * n2) * n;  // x^5 = ((x*x)*(x*x))*x  (GHC.Real (^))
    return div_rn(h2c, n5);
}

// FAST: |pos|^-5 = q^(-5/2) from v_rsq_f64 (measured seed error 2^-24.2) and the series of the exact correction:
// with y0 = rsq(q) and e = 1 - q*y0^2,  q^(-5/2) = y0^5 (1-e)^(-5/2) = y0^5 (1 + 5/2 e + 35/8 e^2 + O(e^3)),
// |e| < 2^-23 so the dropped term is < 1e-20; 7 VALU + the rsq, no sqrt, no divide.  The ray's constant factor
// -(1.5*h2) is not applied here: it is folded into the per-lane step constants of rk4_planar (PlanarK).
// BS_NEWTON=2 keeps only the linear term (35/8 e^2 ~ 6e-14 relative, one FMA cheaper) as an A/B knob.
#ifndef BS_NEWTON
#define BS_NEWTON 3
#endif
__device__ __forceinline__ double rm5_fast(double q, double c25, double c4375)
{
    double y0 = __builtin_amdgcn_rsq(q);
    double y2 = y0 * y0;
    double e = __builtin_fma(-q, y2, 1.0);
    double y4 = y2 * y2;
    double c0 = y4 * y0;
#if BS_NEWTON == 3
    double p = __builtin_fma(c4375, e, c25);
    return __builtin_fma(c0 * e, p, c0);
#else
    return __builtin_fma(c0 * e, c25, c0);
#endif
}

// Per-ray units of the FAST integrator.  x'' = k x/|x|^5 (k = -(1.5*h2) < 0) integrated with step h is the same discrete
// map, up to rounding, as X'' = kappa X/|X|^5 with unit step in the variables X = x/s, W = h*vel/s (displacement per
// step), kappa = k h^2/s^5: RK4 commutes with the rescaling of length and time.  With s = (|k| h^2/4)^(1/5), kappa = -4 and
// every step constant of the Nystrom form is a literal: p3 = p2 - c1 p, p4 = (p + W) - 2 R, new p = (p + W) - 2/3 S,
// new W = W - 2/3 T -- no per-lane constants in registers and no multiply to scale c1.  The guards compare the scaled
// |X|^2 with the per-lane thresholds lo = 1/s^2 (horizon) and hi = safeDistance/s^2.  A ray aimed at the centre has
// k = 0: s is floored at 1e-6 (|k| h^2/4 at 1e-30, an impact parameter of ~1e-14), the force term is then ~1e-30 of the
// position and the path a straight line, as it should be.
// c25 / c4375 / m23: 2.5 pinned in a VGPR pair, 4.375 and -2/3 in SGPR pairs (gfx950's VOP3 takes no literal and one
// SGPR operand; with immediates the compiler re-materialises 2.5 with two v_mov_b32 in front of every v_fmac).
struct PlanarUnits {
    double s, inv_s, lo, hi, c25, c4375, m23;
    __device__ __forceinline__ PlanarUnits(const TraceParams &P, double k) : c25(2.5), c4375(4.375), m23(-2.0 / 3.0)
    {
        // 1/s = a^(-1/5), a = |k| h^2/4: seed from the f32 log2/exp2 units (2^-21), one cubic step of the series of
        // (1-e)^(-1/5), e = 1 - a t^5 (error ~0.09 e^3 < 1e-17), no divide; s = a t^4.  a is floored at 1e-30 (s = 1e-6).
        // a = ar * 32^m with ar in [1/16, 32) (exact: powers of two), so the f32 seed never leaves its range whatever the
        // camera distance, and a^(-1/5) = ar^(-1/5) * 2^-m exactly.
        const double a = fmax(fabs(k) * P.hh2, 1e-30);
        const int m = __builtin_amdgcn_frexp_exp(a) / 5;
        const double ar = __builtin_ldexp(a, -5 * m);
        double t = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)ar));
        double t2 = t * t, t4 = t2 * t2;
        const double e = __builtin_fma(-ar, t4 * t, 1.0);
        t = __builtin_fma(t * e, __builtin_fma(0.12, e, 0.2), t);
        t2 = t * t; t4 = t2 * t2;
        inv_s = __builtin_ldexp(t, -m);
        s = __builtin_ldexp(ar * t4, m);  // a^(1/5) = ar^(1/5) * 2^m = ar * t^4 * 2^m
        lo = inv_s * inv_s;
        hi = P.safe * lo;
        asm volatile("" : "+v"(c25));
        asm volatile("" : "+s"(c4375));
        asm volatile("" : "+s"(m23));
    }
};

// STRICT: one classical RK4 step of y' = f(y), f(vel,pos) = (-(c*pos), vel)   (Raytracer.hs:113-134), the
// reference's operation order, one IEEE operation each (this TU is compiled -ffp-contract=off).
// r2 = quadrance pos on entry (carried from the previous step's findColor), r2n = quadrance newPos on exit.
// In two halves so that the stepping loop can take its ballots in between: the new POSITION needs only k1..k3
// (its k4 term is vel + a3*h), the new velocity needs the fourth force evaluation.
struct StrictMid {
    double a1[3], a2[3], a3[3], q4[3];  // the three accelerations so far and the stage-4 position
};

__device__ __forceinline__ void rk4_strict_position(const TraceParams &P, double h2c, double r2, const double v[3], const double p[3], double np[3],
                                                    double &r2n, StrictMid &M)
{
    const double h = P.h, hh = P.hh, h6 = P.h6;
    double v2[3], v3[3], v4[3], q[3];
    double c = coef_strict(h2c, r2);
#pragma unroll
    for (int i = 0; i < 3; i++) M.a1[i] = -(c * p[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) { v2[i] = v[i] + M.a1[i] * hh; q[i] = p[i] + v[i] * hh; }
    c = coef_strict(h2c, quadrance(q[0], q[1], q[2]));
#pragma unroll
    for (int i = 0; i < 3; i++) M.a2[i] = -(c * q[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) { v3[i] = v[i] + M.a2[i] * hh; q[i] = p[i] + v2[i] * hh; }
    c = coef_strict(h2c, quadrance(q[0], q[1], q[2]));
#pragma unroll
    for (int i = 0; i < 3; i++) M.a3[i] = -(c * q[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) { v4[i] = v[i] + M.a3[i] * h; M.q4[i] = p[i] + v3[i] * h; }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        // sumK = ((k1 + 2 k2) + 2 k3) + k4, position rows.  x*2 is exact, so k + x*2 rounds once either way: the FMA form
        // returns the same bits as the reference's multiply-then-add with one instruction less per term.
        double sp = __builtin_fma(v3[i], 2.0, __builtin_fma(v2[i], 2.0, v[i])) + v4[i];
        np[i] = p[i] + sp * h6;
    }
    r2n = quadrance(np[0], np[1], np[2]);
}

__device__ __forceinline__ void rk4_strict_velocity(const TraceParams &P, double h2c, const StrictMid &M, const double v[3], double nv[3])
{
    const double c = coef_strict(h2c, quadrance(M.q4[0], M.q4[1], M.q4[2]));
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double a4 = -(c * M.q4[i]);
        double sv = __builtin_fma(M.a3[i], 2.0, __builtin_fma(M.a2[i], 2.0, M.a1[i])) + a4;  // velocity rows of sumK (exact doubling, see above)
        nv[i] = v[i] + sv * P.h6;
    }
}

__device__ __forceinline__ void rk4_strict(const TraceParams &P, double h2c, double r2, const double v[3], const double p[3], double nv[3],
                                           double np[3], double &r2n)
{
    StrictMid M;
    rk4_strict_position(P, h2c, r2, v, p, np, r2n, M);
    rk4_strict_velocity(P, h2c, M, v, nv);
}

// Orbital-plane frame of a ray (FAST).  With e1 = pos/|pos| (the camera direction, wave-uniform, from the host) and
// e2 = the unit vector along the part of vel orthogonal to e1, the 3-D y coordinate (the disk plane's normal) of a point
// x e1 + y e2 is Y = n1 x + n2 y, (n1, n2) = (e1.y, e2.y).  The frame used for stepping is (e1, e2) ROTATED in the plane
// so that its first axis lies along the line in which the disk plane cuts the orbital plane:
//   f1 = cs e1 - sn e2,  f2 = sn e1 + cs e2,  (cs, sn) = (n2, n1)/m,  m = |(n1, n2)|     =>     Y = m * y'  with m > 0.
// The sign of Y is then the sign of the planar coordinate y' itself: the stepping loop needs no dot product to watch for
// disk crossings, and r2ave = (Yn r2 - Y r2n)/(Yn - Y) is unchanged by the common factor m.  A camera in the disk plane
// has n1 = 0 exactly, hence sn = 0 and y' = 0 exactly, like Y.  If the orbital plane IS the disk plane (m = 0, Y = 0 along
// the whole ray) there is never a crossing (signum 0 == signum 0, :96): `in_disk_plane`.
struct PlanarFrame {
    double f1[3], f2[3];
    double x, y, vx, vy;  // initial planar state
    double k;             // -(1.5 * h2), h2 = |pos x vel|^2
    bool in_disk_plane;
    __device__ __forceinline__ PlanarFrame(const TraceParams &P, const double v[3])
    {
        const double vr = __builtin_fma(v[2], P.e1[2], __builtin_fma(v[1], P.e1[1], v[0] * P.e1[0]));
        const double w[3] = {__builtin_fma(-vr, P.e1[0], v[0]), __builtin_fma(-vr, P.e1[1], v[1]), __builtin_fma(-vr, P.e1[2], v[2])};
        const double vt = __builtin_sqrt(quadrance(w[0], w[1], w[2]));
        const double ivt = vt > 0 ? 1.0 / vt : 0.0;  // purely radial ray: e2 = 0, the motion stays on the e1 axis
        const double e2[3] = {w[0] * ivt, w[1] * ivt, w[2] * ivt};
        const double n1 = P.e1[1], n2 = e2[1];
        const double m = __builtin_sqrt(__builtin_fma(n2, n2, n1 * n1));
        in_disk_plane = !(m > 0);
        const double im = in_disk_plane ? 0.0 : 1.0 / m;
        const double cs = in_disk_plane ? 1.0 : n2 * im, sn = n1 * im;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            f1[i] = __builtin_fma(cs, P.e1[i], -(sn * e2[i]));
            f2[i] = __builtin_fma(sn, P.e1[i], cs * e2[i]);
        }
        x = cs * P.rcam; y = sn * P.rcam;                  // pos = rcam e1
        vx = __builtin_fma(cs, vr, -(sn * vt));            // vel = vr e1 + vt e2
        vy = __builtin_fma(sn, vr, cs * vt);
        const double L = P.rcam * vt;                      // |pos x vel| in the plane
        k = -1.5 * (L * L);
    }
    __device__ __forceinline__ void to_space(double px, double py, double out[3]) const
    {
#pragma unroll
        for (int i = 0; i < 3; i++) out[i] = __builtin_fma(px, f1[i], py * f2[i]);
    }
};

// FAST: the same RK4 map evaluated in the ray's orbital plane.  f(pos) = c(|pos|) pos is rotation-covariant,
// so every RK4 stage stays in span{pos, vel}: with an orthonormal basis (e1, e2) of that plane the 6-vector
// map reduces EXACTLY (in real arithmetic) to a 4-vector one -- 2/3 of the vector work.  The stages are also
// regrouped for x'' = a(x) (the RHS does not depend on vel) and every multiply-add is fused:
//   p2 = p + (h/2) v          p3 = p2 + (h^2/4) a1        p4 = (p + h v) + (h^2/2) a2
//   np = (p + h v) + (h^2/6)(a1 + a2 + a3)                nv = v + (h/6)(a1 + 2(a2 + a3) + a4)
// Rounding differs from the reference's order at the 1e-16 level per operation (tests: <= 1e-10 on the
// terminal direction, 1e-4 relative on every pixel of the BASELINE frames).
// With c_i = |p_i|^-5 the accelerations a_i = kappa c_i p_i are never formed: the two weighted sums the update needs are
// accumulated with FMAs,  R = c2 p2 + c3 p3,   S = c1 p + R  (= (a1+a2+a3)/kappa),   T = S + R + c4 p4
// (= (a1+2a2+2a3+a4)/kappa); in the ray's own units (PlanarUnits: kappa = -4, unit step, w = displacement per step)
//   p2 = p + w/2     p3 = p2 - c1 p     p4 = (p + w) - 2 c2 p2     np = (p + w) - 2/3 S     nw = w - 2/3 T
// 22 VALU for the linear algebra of a step (29 with the a_i formed and per-lane constants).
// Split in two so that the stepping loop can take its ballots between the halves: the new POSITION needs only stages 1-3.
struct PlanarMid {
    double ux, uy, Rx, Ry, Sx, Sy;  // p4 and the partial sums stage 4 completes
};

// stages 1-3 and the position update: x, y become the new position, r2n its square
__device__ __forceinline__ void rk4_planar_position(const PlanarUnits &U, double r2, double &x, double &y, double wx, double wy, double &r2n, PlanarMid &M)
{
    const double c1 = rm5_fast(r2, U.c25, U.c4375);
    double qx = __builtin_fma(0.5, wx, x), qy = __builtin_fma(0.5, wy, y);   // p2
    const double c2 = rm5_fast(__builtin_fma(qy, qy, qx * qx), U.c25, U.c4375);
    double Rx = c2 * qx, Ry = c2 * qy;
    qx = __builtin_fma(-c1, x, qx); qy = __builtin_fma(-c1, y, qy);          // p3
    const double c3 = rm5_fast(__builtin_fma(qy, qy, qx * qx), U.c25, U.c4375);
    const double q0x = x + wx, q0y = y + wy;
    M.ux = __builtin_fma(-2.0, Rx, q0x); M.uy = __builtin_fma(-2.0, Ry, q0y);  // p4
    M.Rx = __builtin_fma(c3, qx, Rx); M.Ry = __builtin_fma(c3, qy, Ry);
    M.Sx = __builtin_fma(c1, x, M.Rx); M.Sy = __builtin_fma(c1, y, M.Ry);
    x = __builtin_fma(U.m23, M.Sx, q0x);
    y = __builtin_fma(U.m23, M.Sy, q0y);
    r2n = __builtin_fma(y, y, x * x);
}

// stage 4 and the velocity update
__device__ __forceinline__ void rk4_planar_velocity(const PlanarUnits &U, const PlanarMid &M, double &wx, double &wy)
{
    const double c4 = rm5_fast(__builtin_fma(M.uy, M.uy, M.ux * M.ux), U.c25, U.c4375);
    const double Tx = __builtin_fma(c4, M.ux, M.Sx + M.Rx), Ty = __builtin_fma(c4, M.uy, M.Sy + M.Ry);
    wx = __builtin_fma(U.m23, Tx, wx);
    wy = __builtin_fma(U.m23, Ty, wy);
}

__device__ __forceinline__ void rk4_planar(const PlanarUnits &U, double r2, double &x, double &y, double &wx, double &wy, double &r2n)
{
    PlanarMid M;
    rk4_planar_position(U, r2, x, y, wx, wy, r2n, M);
    rk4_planar_velocity(U, M, wx, wy);
}

// Per-star colour of starLookup's renderPixel (StarMap.hs:105-114), added to the running sum.  toPixelRGB (PixelHSI h s i)
// (massiv-io) with the hue's two cosines taken from the star's StarColor record (host libm, once per star):
// first = i + is*cos a / cos b, second = i - is, third = i + 2*is + second - first, (r,g,b) a rotation of (first, third, second).
__device__ __forceinline__ void add_star(const TraceParams &P, unsigned k, double d2, double &accR, double &accG, double &accB)
{
    const double two_w2 = 2 * (kStarW * kStarW);
    const int mag = P.nodes[k].mag;
    const StarColor sc = P.colors[k];
    double e = exp_call(P.star_a * (950.0 - (double)mag) - d2 / two_w2);
    double m = (1.0 <= e) ? 1.0 : e;  // min 1
    const double i = m * P.star_intensity;
    const double is = i * (P.star_saturation * sc.sat);
    const double second = i - is;
    const double first = i + is * sc.ca / sc.cb;
    const double third = i + 2 * is + second - first;
    const int sec = sc.sector;
    accR = accR + (sec == 0 ? first : (sec == 1 ? second : third));
    accG = accG + (sec == 0 ? third : (sec == 1 ? first : second));
    accB = accB + (sec == 0 ? second : (sec == 1 ? third : first));
}

// bs_internal.h grid_cell, same operations (the builder bins with it, so binning and query agree by monotonicity).
__device__ __forceinline__ int grid_cell_dev(double t)
{
    double c = (t + 1.0) * (0.5 * kGridG);
    if (!(c > 0.0)) return 0;  // also NaN
    if (c >= (double)kGridG) return kGridG - 1;
    return (int)c;
}

// starLookup (StarMap.hs:93-115) over the cube-map direction grid (bs_internal.h): the stars within the radius of nvel
// are all listed in nvel's own face, in the <= 2 x 2 cells its D-box touches; each touched row of cells is ONE contiguous
// run of entries.  Returns the number of stars within the radius; rgb = min 1 (sum of per-star colours).
// Hits (0.25 per lookup) are only RECORDED while scanning (entry index + d^2 into the lane's LDS column) and shaded
// afterwards -- exp and a divide, ~100 instructions that the wavefront would otherwise execute at every candidate at
// which any lane happens to hit.
#ifndef BS_HIT_SLOTS
#define BS_HIT_SLOTS 5
#endif
constexpr int kHitSlots = BS_HIT_SLOTS;  // per lane, in the snapshot/queue columns (free by the time the lookup runs)

__device__ __forceinline__ int star_lookup(const TraceParams &P, double *lane_col, double vx, double vy, double vz, double &R, double &G, double &B)
{
    const double r2 = kStarRadius * kStarRadius;  // kdt: distSqr p q <= radius * radius
    // linear.normalize: unchanged when |l| or |1-l| <= 1e-12
    double l = quadrance(vx, vy, vz);
    double nx = vx, ny = vy, nz = vz;
    if (!(fabs(l) <= 1e-12 || fabs(1.0 - l) <= 1e-12)) {
        double s = __builtin_sqrt(l);
        nx = vx / s; ny = vy / s; nz = vz / s;
    }
    // face of the largest |component|, gnomonic coordinates of the other two in cyclic order.  A NaN or zero vector
    // lands in some cell of face 4 and matches nothing (every d^2 compare fails).
    const double ax = fabs(nx), ay = fabs(ny), az = fabs(nz);
    const int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const double m = axis == 0 ? nx : (axis == 1 ? ny : nz);
    const double a = axis == 0 ? ny : (axis == 1 ? nz : nx);
    const double b = axis == 0 ? nz : (axis == 1 ? nx : ny);
    const int face = 2 * axis + (m < 0 ? 1 : 0);
    const double im = __builtin_amdgcn_rcp(fabs(m));  // ~2^-24 relative; kGridDelta carries 4 % of slack
    const double u = a * im, v = b * im;
    const int iu0 = grid_cell_dev(u - kGridDelta), iu1 = grid_cell_dev(u + kGridDelta);
    int iv = grid_cell_dev(v - kGridDelta);
    const int iv1 = grid_cell_dev(v + kGridDelta);
    unsigned row = (unsigned)((face * kGridG + iv) * kGridG);
    unsigned k = P.cell_start[row + iu0], e = P.cell_start[row + iu1 + 1];
    if (fabs(l) <= 1e-12) {  // not normalised above, so not a unit vector: only the stars around the origin are in reach
        k = P.cell_start[kGridCells];
        e = P.cell_start[kGridCells + 1];
        iv = iv1;
    }
    double accR = 0, accG = 0, accB = 0;
    int hits = 0;
    for (bool more = true; more;) {
        if (k < e) {
            const StarNode nd = P.nodes[k];
            double dx = nd.x - nx, dy = nd.y - ny, dz = nd.z - nz;  // qd pos nvel = quadrance (pos - nvel)
            double d2 = quadrance(dx, dy, dz);
            if (d2 <= r2) {
                if (hits < kHitSlots) {
                    lane_col[hits * kBlock] = d2;
                    lane_col[(kHitSlots + hits) * kBlock] = (double)k;
                } else {
                    add_star(P, k, d2, accR, accG, accB);  // more hits than slots: shade in place
                }
                hits++;
            }
            k++;
        } else if (iv < iv1) {  // next row of cells
            iv++;
            row += kGridG;
            k = P.cell_start[row + iu0];
            e = P.cell_start[row + iu1 + 1];
        } else {
            more = false;
        }
    }
    const int queued = hits < kHitSlots ? hits : kHitSlots;
    for (int q = 0; q < queued; q++) add_star(P, (unsigned)lane_col[(kHitSlots + q) * kBlock], lane_col[q * kBlock], accR, accG, accB);
    R = (1.0 <= accR) ? 1.0 : accR;  // fmap (min 1)
    G = (1.0 <= accG) ? 1.0 : accG;
    B = (1.0 <= accB) ? 1.0 : accB;
    return hits;
}

struct RayResult {
    double vel[3], pos[3], rgba[4];
    int steps, fate, disk_hits, star_hits;
};

// diskColor' (Raytracer.hs:104-111) blended under the accumulated colour (blend, :34-37).
__device__ __forceinline__ void shade_disk(const TraceParams &P, double r2ave, double rgba[4])
{
    const double pi = 3.141592653589793;
    double r = __builtin_sqrt(r2ave);
    double t = (P.rO - r) / (P.rO - P.rI);
    double inten = sin_call(pi * (t * t));
    double om = 1 - rgba[3];  // top + layer * (1 - top_alpha), all four channels
    rgba[0] = rgba[0] + (P.disk_rgb[0] * inten) * om;
    rgba[1] = rgba[1] + (P.disk_rgb[1] * inten) * om;
    rgba[2] = rgba[2] + (P.disk_rgb[2] * inten) * om;
    rgba[3] = rgba[3] + (inten * P.disk_opacity) * om;
}

// generateRay (Raytracer.hs:40-51); look-at basis hoisted to the host (identical arithmetic, once per frame).
__device__ __forceinline__ void generate_ray(const TraceParams &P, int yi, int xi, double v[3])
{
    // the same quotients and root as the reference's (/) and sqrt, bit for bit, in fewer instructions (BS_EXACT_SHORTCUTS=0: the compiler's own):
    // W and H are wave-uniform with host reciprocals (div_by); l = v0^2 + v1^2 + 1 lies in [1, 1 + fov^2] for every config bs_validate_config
    // lets through, far from the range where sqrt_rn / div_rn would need the scaling the compiler wraps around them.
    const bool shortcuts = BS_EXACT_SHORTCUTS && P.inv_W != 0.0;  // wave-uniform: the host found nothing tiny in the camera (host_math.cpp)
    double v0, v1;
    if (shortcuts) {
        v0 = P.fov * (div_by((double)xi, P.W, P.inv_W) - 0.5);
        v1 = div_by(P.fov * (0.5 - div_by((double)yi, P.H, P.inv_H)) * P.H, P.W, P.inv_W);
    } else {
        v0 = P.fov * ((double)xi / P.W - 0.5);
        v1 = P.fov * (0.5 - (double)yi / P.H) * P.H / P.W;
    }
    double d[3];
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = (P.xa[i] * v0 + P.ya[i] * v1) + P.za[i];  // (-za_i) * (-1) == za_i exactly
    double l = quadrance(d[0], d[1], d[2]);
    if (fabs(l) <= 1e-12 || fabs(1.0 - l) <= 1e-12) {  // linear.normalize shortcut
        v[0] = d[0]; v[1] = d[1]; v[2] = d[2];
    } else {
        if (shortcuts && mid_range(l)) {
            div3_rn(d, sqrt_rn(l), v);
        } else {
            double s = __builtin_sqrt(l);
            v[0] = d[0] / s; v[1] = d[1] / s; v[2] = d[2] / s;
        }
    }
}

// ---- per-lane LDS scratch --------------------------------------------------------------------------------
// The stepping loop is ~100% VALU-issue bound.  Any value that is live OUT of it, or conditionally modified
// inside it, costs register copies EVERY iteration (exit merges become phis in blocks all lanes run; and a
// partially masked body cannot update state in place because the compiler's liveness is per register, not
// per lane).  So: nothing leaves the loop through registers.  When a lane's guard fires it writes its
// terminal state and step count to its LDS column (a rare block placed BEFORE the RK4 body) and leaves the
// wavefront's active mask; the body itself runs UNMASKED for all 64 lanes -- a finished lane just keeps stepping, its later
// values are never looked at (f64 VALU has no slow path for the inf/NaN a captured lane can produce) -- and
// updates the state in place.  The code after the loop reloads everything from LDS.
// Layout: [word][thread] -- consecutive lanes touch consecutive 8-byte words: conflict-free.
constexpr int kDiskSlots = 4;
#ifndef BS_SNAP
#define BS_SNAP 7
#endif
constexpr int kSnapDoubles = BS_SNAP;  // STRICT: vel[3], pos[3], r2; FAST: x, y, wx, wy (in the ray's units), -, the unit of length s
constexpr int kLaneLdsDoubles = (kSnapDoubles + kDiskSlots) * kBlock;
static_assert(kSnapDoubles + kDiskSlots >= 2 * kHitSlots, "the star-hit queue reuses the lane columns");

struct LaneLds {
    double *col;  // this lane's column: col[word * kBlock]
    int *ints;    // this lane's ints: ints[0] crossing count (or kOverflow), ints[kBlock] steps, ints[2*kBlock] fate
    __device__ __forceinline__ LaneLds(double *area, int *iarea) : col(area + threadIdx.x), ints(iarea + threadIdx.x) {}
    __device__ __forceinline__ double &snap(int k) const { return col[k * kBlock]; }
    __device__ __forceinline__ double &slot(int k) const { return col[(kSnapDoubles + k) * kBlock]; }
    __device__ __forceinline__ int &count() const { return ints[0]; }
    __device__ __forceinline__ int &steps() const { return ints[kBlock]; }
    __device__ __forceinline__ int &fate() const { return ints[2 * kBlock]; }  // FAST: which guard fired (0 horizon, 1 escape, 2 neither = step cap)
};

// Disk crossings are rare (~0.2 per ray) but their shading (sqrt, divide, sin) is ~150 instructions that the
// whole wavefront would sit through each time any lane crosses.  Crossings are therefore only RECORDED in
// the stepping loop (r2ave, in order, kDiskSlots per lane) and shaded after it, when all 64 lanes do it
// together -- same arithmetic, same front-to-back order.  A ray with more crossings than slots (possible
// only when the disk reaches inside the photon sphere) is flagged and re-traced by trace_ray_simple.
constexpr int kOverflow = 1 << 20;

// findColor's disk guard (:96-98) for the step (y, r2) -> (yn, r2n).  Callers have already established that
// y*yn <= 0 (the only way signum y' /= signum y can yield a layer).  A lane whose queue overflows is flagged and
// keeps stepping (its result is discarded: trace_ray_simple redoes the ray).
// r2, r2n may be in the ray's own units of length (FAST): unit2 = s^2 brings r2ave back (1.0, exact, in STRICT).
__device__ __forceinline__ void record_crossing(const TraceParams &P, const LaneLds &lds, double y, double yn, double r2, double r2n, double unit2)
{
    if (signum_differs(yn, y)) {
        double r2ave = ((yn * r2 - y * r2n) / (yn - y)) * unit2;  // :102
        if (r2ave > P.in2 && r2ave < P.out2) {           // :97
            int n = lds.count();
            if (n >= P.disk_slots) {
                lds.count() = kOverflow;
            } else {
                lds.slot(n) = r2ave;
                lds.count() = n + 1;
            }
        }
    }
}

// The terminal `Bottom` layer of colorize (:84, :93-95) under whatever the disk left transparent.
__device__ __forceinline__ int finish_ray(const TraceParams &P, double *lane_col, int fate, const double v[3], double rgba[4])
{
    int star_hits = 0;
    if (fate == 0) {  // Bottom (PixelRGBA 0 0 0 1)
        double om = 1 - rgba[3];
        rgba[0] = rgba[0] + 0.0 * om; rgba[1] = rgba[1] + 0.0 * om; rgba[2] = rgba[2] + 0.0 * om;
        rgba[3] = rgba[3] + 1.0 * om;
    } else if (fate == 1) {  // Bottom . addAlpha 1 $ starLookup ... vel   (the PRE-step vel, :94-95)
        double sr, sg, sb;
        star_hits = star_lookup(P, lane_col, v[0], v[1], v[2], sr, sg, sb);
        double om = 1 - rgba[3];
        rgba[0] = rgba[0] + sr * om; rgba[1] = rgba[1] + sg * om; rgba[2] = rgba[2] + sb * om;
        rgba[3] = rgba[3] + 1.0 * om;
    }
    return star_hits;
}

// Plain per-lane restatement of traceRay/colorize with the disk shaded inside the loop.  Only used for the
// (rare) rays whose crossings overflow the LDS queue; same arithmetic as the fast path.
template <bool FAST>
__device__ __forceinline__ void trace_ray_simple(const TraceParams &P, int yi, int xi, double *out /* vel[3] pos[3] rgba[4] */, int *iout /* steps fate crossings */)
{
    double v[3], p[3], rgba[4] = {0, 0, 0, 0};
    generate_ray(P, yi, xi, v);
    p[0] = P.cam[0]; p[1] = P.cam[1]; p[2] = P.cam[2];
    int steps = 0, fate = 2, ncross = 0;
    if constexpr (!FAST) {
        double cx = p[1] * v[2] - p[2] * v[1], cy = p[2] * v[0] - p[0] * v[2], cz = p[0] * v[1] - p[1] * v[0];
        const double h2c = 1.5 * quadrance(cx, cy, cz);
        double r2 = quadrance(p[0], p[1], p[2]);
        while (steps < P.max_steps) {
            steps++;
            if (r2 < 1.0) { fate = 0; break; }
            if (r2 > P.safe) { fate = 1; break; }
            double nv[3], np[3], r2n;
            rk4_strict(P, h2c, r2, v, p, nv, np, r2n);
            double y = p[1], yn = np[1];
            if (P.disk_opacity != 0 && signum(yn) != signum(y)) {
                double r2ave = (yn * r2 - y * r2n) / (yn - y);
                if (r2ave > P.in2 && r2ave < P.out2) { shade_disk(P, r2ave, rgba); ncross++; }
            }
            for (int i = 0; i < 3; i++) { v[i] = nv[i]; p[i] = np[i]; }
            r2 = r2n;
        }
    } else {
        const PlanarFrame F(P, v);
        const PlanarUnits U(P, F.k);
        const double wscale = P.h * U.inv_s;
        double x = F.x * U.inv_s, y = F.y * U.inv_s, wx = F.vx * wscale, wy = F.vy * wscale, r2 = __builtin_fma(y, y, x * x);
        while (steps < P.max_steps) {
            steps++;
            if (r2 < U.lo) { fate = 0; break; }
            if (r2 > U.hi) { fate = 1; break; }
            double r2n;
            const double r2o = r2, yo = y;
            rk4_planar(U, r2o, x, y, wx, wy, r2n);
            if (P.disk_opacity != 0 && !F.in_disk_plane && signum(y) != signum(yo)) {
                double r2ave = ((y * r2o - yo * r2n) / (y - yo)) * (U.s * U.s);
                if (r2ave > P.in2 && r2ave < P.out2) { shade_disk(P, r2ave, rgba); ncross++; }
            }
            r2 = r2n;
        }
        const double vscale = U.s / P.h;
        F.to_space(wx * vscale, wy * vscale, v);
        F.to_space(x * U.s, y * U.s, p);
    }
    for (int i = 0; i < 3; i++) { out[i] = v[i]; out[3 + i] = p[i]; }
    for (int i = 0; i < 4; i++) out[6 + i] = rgba[i];
    iout[0] = steps; iout[1] = fate; iout[2] = ncross;
}

// traceRay + colorize for traced pixel (yi, xi).  `live` = this lane has a ray (tile lanes outside the image do not).
//
// Every lane of a wavefront starts its ray at iteration 0 together, so the iteration counter is a scalar
// register and a lane's step count (iterations of colorize', :80-86) is simply its value when the lane's
// guard fires.  See "per-lane LDS scratch" above for why the loop looks the way it does.
template <bool FAST>
__device__ __forceinline__ void trace_ray(const TraceParams &P, const LaneLds &lds, bool live, int yi, int xi,
                                          RayResult &res, unsigned &wave_iters)
{
    double v[3], p[3];
    generate_ray(P, yi, xi, v);
    p[0] = P.cam[0]; p[1] = P.cam[1]; p[2] = P.cam[2];
    const bool disk = P.disk_opacity != 0;
    lds.count() = 0;
    lds.steps() = 0;
    // The set of lanes still stepping is a wave-uniform 64-bit mask in scalar registers: the guards are two fresh
    // compares whose ballots are ANDed on the scalar unit, "some lane finished" is a scalar compare, and the lane-level
    // test "is this lane in the mask" is taken only inside the rare blocks, as exec &= mask (in_mask: no VALU).  (A per-lane bool costs a v_cndmask + v_cmp
    // round trip per step to turn the loop-carried mask back into a ballot.)  Lanes that are done free-run: their values
    // are never read, and a NaN state cannot enter the crossing block (y*yn <= 0 is false for NaN; the reference's
    // signum test passes NaN on to an r2ave that fails both radius compares, i.e. no layer either way).
    // "No disk" is folded into the crossing threshold (a product is never <= -inf short of overflow), held in a VGPR
    // pair: as a scalar flag it was the one value the allocator spilled and re-read (2 v_readlane) in every step.
    double cross_thr = disk ? 0.0 : -__builtin_inf();
    asm volatile("" : "+v"(cross_thr));
    unsigned long long amask = __builtin_amdgcn_ballot_w64(live);
    int it = 0;  // iterations of colorize' entered so far (wave-uniform)
    int fate_code;  // which guard ended the ray: 0 horizon, 1 escape, 2 neither (step cap)

    if constexpr (!FAST) {
        // h2 = quadrance (pos `cross` vel)   (:73)
        double cx = p[1] * v[2] - p[2] * v[1], cy = p[2] * v[0] - p[0] * v[2], cz = p[0] * v[1] - p[1] * v[0];
        const double h2c = 1.5 * quadrance(cx, cy, cz);
        double r2 = quadrance(p[0], p[1], p[2]);
        // one iteration of colorize'; returns false once no lane of the wavefront is stepping
        // `ok`, `crossed`: see the FAST branch below -- the guards of the next state and the crossing test are ballots taken as
        // soon as their operands exist and consumed by scalar branches, so a step is one basic block.
        unsigned long long ok = __builtin_amdgcn_ballot_w64(!(r2 < 1.0)) & __builtin_amdgcn_ballot_w64(!(r2 > P.safe));
        auto step = [&]() -> bool {
            // findColor guards on the PRE-step position (:93-95); the cap is ours (the reference has none)
            unsigned long long go = amask & ok;
            if (!(it < P.max_steps)) go = 0;
            if (__builtin_expect(go != amask, 0)) {  // a guard fired somewhere in the wavefront (rare, wave-uniform branch)
                if (in_mask(amask & ~go)) {  // this lane: snapshot the state fed to the terminating findColor call
#pragma unroll
                    for (int i = 0; i < 3; i++) { lds.snap(i) = v[i]; lds.snap(3 + i) = p[i]; }
                    lds.snap(6) = r2;
                    lds.steps() = it < P.max_steps ? it + 1 : it;
                }
                amask = go;
                if (go == 0) return false;
            }
            double nv[3], np[3], r2n;
            StrictMid M;
            rk4_strict_position(P, h2c, r2, v, p, np, r2n, M);
            ok = __builtin_amdgcn_ballot_w64(!(r2n < 1.0)) & __builtin_amdgcn_ballot_w64(!(r2n > P.safe));
            const unsigned long long crossed = __builtin_amdgcn_ballot_w64(p[1] * np[1] <= cross_thr);
            __builtin_amdgcn_sched_barrier(0);  // the compares issue here, a quarter of a step ahead of the scalar code that reads them
            rk4_strict_velocity(P, h2c, M, v, nv);
            if (__builtin_expect(crossed != 0, 0)) {
                asm volatile("" ::"v"(nv[0]), "v"(nv[1]), "v"(nv[2]) : "memory");  // lane test stays here, the whole step stays in front of the branch
                if (disk && in_mask(amask & crossed)) record_crossing(P, lds, p[1], np[1], r2, r2n, 1.0);
            }
#pragma unroll
            for (int i = 0; i < 3; i++) { v[i] = nv[i]; p[i] = np[i]; }
            r2 = r2n;
            ++it;
            return true;
        };
        // unrolled by two: the state ping-pongs between two register sets instead of being copied back at the latch
        if (amask != 0)  // a wavefront with no ray at all (records kernel tail) must not enter: step() only returns false on a CHANGE of amask
            while (step() && step()) {}
#pragma unroll
        for (int i = 0; i < 3; i++) { v[i] = lds.snap(i); p[i] = lds.snap(3 + i); }
        const double r2t = lds.snap(6);  // r^2 fed to the terminating findColor call
        fate_code = r2t < 1.0 ? 0 : (r2t > P.safe ? 1 : 2);
    } else {
        const PlanarFrame F(P, v);
        const PlanarUnits U(P, F.k);
        if (F.in_disk_plane) cross_thr = -__builtin_inf();
        const double wscale = P.h * U.inv_s;
        double x = F.x * U.inv_s, y = F.y * U.inv_s, wx = F.vx * wscale, wy = F.vy * wscale, r2 = __builtin_fma(y, y, x * x);
        lds.snap(5) = U.s;  // the unit of length is needed again only after the loop (and in the rare crossing block)
        // `ok`: the guards of the state about to be stepped, as a wave-uniform mask.  It is computed at the END of the step that
        // produced that state (and before the loop for the first one), so the two v_cmp are long retired when the scalar
        // unit combines them at the top of the next step; likewise the crossing compare is a ballot taken as soon as the new
        // y exists and tested with a scalar branch at the end.  A step is one basic block with two (rare) scalar exits --
        // no VALU -> SALU -> branch round trip sits on the wavefront's critical path.
        unsigned long long ok = __builtin_amdgcn_ballot_w64(!(r2 < U.lo)) & __builtin_amdgcn_ballot_w64(!(r2 > U.hi));
#if BS_ASM_LOOP
        // The steps themselves are one assembly statement (fast_loop_asm.h: the arithmetic of rk4_planar_position / rk4_planar_velocity
        // as compiled, the scalar bookkeeping by hand); it runs until something RARE happens and says what: a guard fired before a step
        // (ev 0: `go` = the lanes that go on; nothing was stepped) or a step crossed the disk plane somewhere (ev 1: yo / r2o = that step's
        // y and r^2 before it, the state is the one after it).  The rare work is C++, here; then the statement is entered again.
        if (amask != 0) {  // a wavefront with no ray at all (records kernel tail) must not enter: the loop only ends on a CHANGE of amask
            const double lo = U.lo, hi = U.hi, c25 = U.c25, c4375 = U.c4375, m23 = U.m23;
#if BS_FL_SERIES
            // stage 4's squared radius, its reciprocal and r^-5, carried from step to step (fast_loop_asm.h BS_FL_SERIES); none yet: 1 / q4 = inf
            // makes the first step's delta infinite, i.e. the step takes its own v_rsq_f64
            double q4s = 1.0, iq4s = __builtin_inf(), c4s = 0.0, c6 = -6.5625;
            asm volatile("" : "+v"(c6));
#endif
            for (;;) {
                double yo, r2o, yb, r2b, t2, t3, t4, t5, t6, t7;
#if BS_FL_SERIES == 2
                double t8;
#endif
                unsigned long long go, crossed;
                int ev;
                asm volatile(BS_FAST_LOOP_ASM
                             : [x] "+v"(x), [y] "+v"(y), [wx] "+v"(wx), [wy] "+v"(wy), [r2] "+v"(r2), [yb] "=&v"(yb), [r2b] "=&v"(r2b), [t0] "=&v"(yo),
                               [t1] "=&v"(r2o), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7),
#if BS_FL_SERIES
                               [q4] "+v"(q4s), [iq4] "+v"(iq4s), [c4] "+v"(c4s),
#endif
#if BS_FL_SERIES == 2
                               [t8] "=&v"(t8),
#endif
                               [ok] "+s"(ok), [it] "+s"(it), [go] "=&s"(go), [crossed] "=&s"(crossed), [ev] "=&s"(ev)
                             : [c25] "v"(c25), [lo] "v"(lo), [hi] "v"(hi), [thr] "v"(cross_thr), [c4375] "s"(c4375), [m23] "s"(m23),
#if BS_FL_SERIES
                               [c6] "v"(c6), [thr15] "s"(0x1p-15),
#endif
#if BS_FL_SERIES == 2
                               [c8] "s"(9.0234375), [thr12] "s"(0x1p-12),
#endif
                               [maxs] "s"(P.max_steps), [amask] "s"(amask)
                             : "vcc", "scc");
                // The statement has VGPR outputs too, and the compiler's divergence analysis then takes ALL its outputs for per-lane values:
                // amask & ~go would be computed with v_and / v_bfi and read back with v_readfirstlane.  An empty statement whose only output
                // is an SGPR (tied to its input: no instruction) says "wave-uniform".
                asm("" : "=s"(go) : "0"(go));
                asm("" : "=s"(crossed) : "0"(crossed));
                asm("" : "=s"(ev) : "0"(ev));
                asm("" : "=s"(it) : "0"(it));
                if (ev == 0) {
                    if (in_mask(amask & ~go)) {
                        lds.snap(0) = x; lds.snap(1) = y; lds.snap(2) = wx; lds.snap(3) = wy;
                        lds.fate() = r2 < U.lo ? 0 : (r2 > U.hi ? 1 : 2);  // which guard fired, decided on the very values the guards compared
                        lds.steps() = it < P.max_steps ? it + 1 : it;
                    }
                    amask = go;
                    if (go == 0) break;
                } else if (disk && in_mask(amask & crossed)) {
                    const double unit = lds.snap(5);
                    record_crossing(P, lds, yo, y, r2o, r2, unit * unit);
                }
            }
        }
#else
        auto step = [&]() -> bool {
            unsigned long long go = amask & ok;
            if (!(it < P.max_steps)) go = 0;
            if (__builtin_expect(go != amask, 0)) {
                if (in_mask(amask & ~go)) {
                    lds.snap(0) = x; lds.snap(1) = y; lds.snap(2) = wx; lds.snap(3) = wy;
                    lds.fate() = r2 < U.lo ? 0 : (r2 > U.hi ? 1 : 2);  // which guard fired, decided on the very values the guards compared
                    lds.steps() = it < P.max_steps ? it + 1 : it;
                }
                amask = go;
                if (go == 0) return false;
            }
            double r2n;
            const double r2o = r2, yo = y;
            PlanarMid M;
            rk4_planar_position(U, r2o, x, y, wx, wy, r2n, M);
            ok = __builtin_amdgcn_ballot_w64(!(r2n < U.lo)) & __builtin_amdgcn_ballot_w64(!(r2n > U.hi));
            const unsigned long long crossed = __builtin_amdgcn_ballot_w64(yo * y <= cross_thr);  // the planar y IS the disk-normal coordinate up to a positive factor (PlanarFrame)
            __builtin_amdgcn_sched_barrier(0);  // the three compares issue HERE, a quarter of a step ahead of the scalar code that reads them
            rk4_planar_velocity(U, M, wx, wy);
            if (__builtin_expect(crossed != 0, 0)) {
                // keeps the lane test in this rare block; naming the new velocity as an input keeps the whole step in front
                // of the branch (otherwise stage 4 is sunk below it and the compare -> branch latency is exposed again)
                asm volatile("" ::"v"(wx), "v"(wy) : "memory");
                if (disk && in_mask(amask & crossed)) {
                    const double unit = lds.snap(5);
                    record_crossing(P, lds, yo, y, r2o, r2n, unit * unit);
                }
            }
            r2 = r2n;
            ++it;
            return true;
        };
        if (amask != 0)  // a wavefront with no ray at all (records kernel tail) must not enter: step() only returns false on a CHANGE of amask
            while (step() && step()) {}
#endif
        // the snapshot is the PRE-step planar state of the terminating iteration (guards precede rk4), in the ray's units
        const double unit = lds.snap(5), vscale = unit / P.h;
        F.to_space(lds.snap(2) * vscale, lds.snap(3) * vscale, v);
        F.to_space(lds.snap(0) * unit, lds.snap(1) * unit, p);
        fate_code = lds.fate();
    }
    int steps = lds.steps();
    int ncross = lds.count();
    int fate = !live ? -1 : fate_code;
    double rgba[4] = {0, 0, 0, 0};  // colorize' starts from PixelRGBA 0 0 0 0 (:86)
    // FAST only: a ray that took more than P.guard_steps steps has circled the photon sphere, where every orbit multiplies any
    // rounding difference by ~535 -- its result is recomputed with STRICT arithmetic (bit-exact trajectories), so FAST's
    // deviation from the reference stays that of the well-conditioned rays.  A few rays per million; see derive_params.
    const bool guarded = FAST && live && steps > P.guard_steps;
    if (ncross < kOverflow && !guarded) {
        for (int k = 0; k < ncross; k++) shade_disk(P, lds.slot(k), rgba);  // blend the recorded layers, oldest first
    } else if (live) {  // more crossings than slots, or a guarded ray: the simple restatement redoes this ray
        double out[10];
        int iout[3];
        if (guarded) trace_ray_simple<false>(P, yi, xi, out, iout);
        else trace_ray_simple<FAST>(P, yi, xi, out, iout);
        for (int i = 0; i < 3; i++) { v[i] = out[i]; p[i] = out[3 + i]; }
        for (int i = 0; i < 4; i++) rgba[i] = out[6 + i];
        steps = iout[0]; fate = iout[1]; ncross = iout[2];
    }
    int star_hits = finish_ray(P, lds.col, fate, v, rgba);
#pragma unroll
    for (int i = 0; i < 3; i++) { res.vel[i] = v[i]; res.pos[i] = p[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) res.rgba[i] = rgba[i];
    res.steps = steps; res.fate = fate; res.disk_hits = ncross; res.star_hits = star_hits;
    wave_iters = (unsigned)it + 1u;  // iterations entered, including the one in which the last guards fired
}

// 64-bit: a wavefront's step total passes 2^32 as soon as 64 lanes x tiles x steps does (a raised step cap on capped rays)
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

}  // namespace
}  // namespace bs
