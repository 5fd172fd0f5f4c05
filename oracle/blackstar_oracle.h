/*
 * blackstar_oracle.h -- CPU restatement of blackstar's Raytracer.render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under blackstar_amd/ (the product) may
 * include, link or dlopen this.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg -- as the checker / reported baseline only.
 *
 * PARITY UNPINNED: the reference (Haskell, /root/reference) has no test suite,
 * no golden vectors, and cannot be built in this image (no ghc/stack/cabal,
 * deps un-vendored: resolver lts-13.16, stack.yaml:1).  This restatement is
 * pinned instead by (1) an independent numpy restatement (oracle/np_oracle.py),
 * (2) 50-digit mpmath evaluation of the same discrete RK4 map, (3) physics and
 * colour known-answer tests, and (4) the ONE rendered output the reference
 * repository holds, example.png (README.md:4: the default.yaml camera at 1280x720
 * from an unknown revision -- other disk colours and bloom, so no pixel golden):
 * the thin photon ring inside the shadow depends only on generateRay and on the
 * integration, and this oracle's ring sits on the reference's to +0.24 +- 0.40 px
 * at 88 of 90 angles (tests/golden/make_reference_ring.py, tests/test_oracle.py).
 * That pins the camera model and the geodesics against the reference's own
 * output; operation order and the third-party colour / normalise / in-radius
 * semantics stay unpinned -- see tests/ and DESIGN.md section "Oracle".
 *
 * Struct layouts are deliberately identical to include/blackstar_gpu.h so one
 * ctypes.Structure serves both; the code is independent.
 */
#ifndef BLACKSTAR_ORACLE_H
#define BLACKSTAR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Config as parsed (src/ConfigFile.hs:16-38): radii UN-squared, user resolution. */
typedef struct {
    double cam_pos[3], cam_lookat[3], cam_up[3], fov;
    double step_size, star_intensity, star_saturation;
    double disk_hsi[3]; /* hue already /360 (src/ConfigFile.hs:51) */
    double disk_opacity, disk_inner, disk_outer;
    int32_t width, height, supersampling;
    int32_t _pad;
} orc_config;

/* One star = one KdMap assoc after starColor' (src/StarMap.hs:25-26,61-62). */
typedef struct {
    double x, y, z, hue, sat;
    int32_t mag;
    int32_t _pad;
} orc_star;

/* Per-ray record for trajectory-level parity (not in the reference; test hook). */
typedef struct {
    double vel[3], pos[3]; /* state fed to the terminating findColor call */
    double rgba[4];        /* composited colour before dropAlpha */
    int32_t steps;         /* number of colorize' iterations (= rk4 evaluations in the reference) */
    int32_t fate;          /* 0 horizon, 1 escaped, 2 step cap */
    int32_t disk_hits;     /* number of Layer blends */
    int32_t star_hits;     /* stars within radius in the terminal lookup */
} orc_ray_record;

typedef struct {
    uint64_t rays, steps, capped, horizon, escaped, disk_hits, star_hits;
    double seconds;
    int32_t threads;
    int32_t _pad;
} orc_stats;

typedef struct orc_index orc_index;

/* Uniform-grid in-radius index over the star list (a 3-D cell grid; independent of the product's cube-map direction grid). */
orc_index *orc_index_create(const orc_star *stars, size_t n);
void orc_index_destroy(orc_index *);

/* src/Raytracer.hs:53-67 render.  out_rgb: height*width*3 interleaved RGB f64.
 * threads<=0 -> all online cores.  max_steps: safety cap (reference has none). */
int orc_render(const orc_config *cfg, const orc_index *idx, double *out_rgb, size_t out_doubles,
               int threads, int max_steps, orc_stats *stats);

/* Trace the rays of the given traced-resolution pixels (y,x pairs) and return records. */
int orc_trace_rays(const orc_config *cfg, const orc_index *idx, const int32_t *yx, size_t n_rays,
                   int max_steps, orc_ray_record *out);

/* Pieces exposed for unit tests. */
void orc_generate_ray(const orc_config *cfg, int y, int x, double vel[3], double pos[3]); /* Raytracer.hs:40-51 on cfg' */
void orc_rk4(double h, double h2, const double vel[3], const double pos[3], double nvel[3], double npos[3]); /* :113-134 */
void orc_hsi_to_rgb(double h, double s, double i, double rgb[3]); /* massiv-io toPixelRGB (PixelHSI) */
int orc_star_lookup(const orc_index *idx, double intensity, double saturation, const double vel[3], double rgb[3]); /* StarMap.hs:93-115 */
int orc_star_lookup_brute(const orc_star *stars, size_t n, double intensity, double saturation, const double vel[3], double rgb[3]);
void orc_supersample(const double *in_rgb, int h2, int w2, double *out_rgb); /* ImageFilters.hs:88-97 */
/* "next" rows: bloom (ImageFilters.hs:80-86) and writeImg's sRGB + toWord8 (Raytracer.hs:23-32) */
int orc_bloom(double strength, int divider, const double *img, int h, int w, double *out);
void orc_srgb8(const double *in, unsigned char *out, size_t n);
/* PPM catalogue record parse (src/StarMap.hs:45-75). returns number of stars written (<= cap) or -1 */
long orc_read_ppm(const unsigned char *bytes, size_t nbytes, orc_star *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
