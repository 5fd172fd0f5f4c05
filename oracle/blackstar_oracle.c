/*
 * blackstar_oracle.c -- strict FP64 CPU restatement of the hot path of
 * flannelhead/blackstar (Raytracer.render and what it calls).
 *
 * TEST INFRASTRUCTURE ONLY (see blackstar_oracle.h).  PARITY UNPINNED by the
 * reference's own tests (it has none); pinned by tests/test_oracle_*.py.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (GHC emits no FMA; every
 * operation below is one IEEE-754 binary64 operation in the reference's order).
 * Citations are file:line into /root/reference.
 */
#define _GNU_SOURCE
#include "blackstar_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ---------------------------------------------------------------- linear (third-party, recalled) */

/* linear: quadrance (V3 a b c) = a*a + b*b + c*c  (left-assoc) */
static inline double quadrance3(const double v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }

/* linear: cross (V3 a b c) (V3 d e f) = V3 (b*f-c*e) (c*d-a*f) (a*e-b*d) */
static inline void cross3(const double a[3], const double b[3], double o[3])
{
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

/* linear: normalize v = if nearZero l || nearZero (1-l) then v else fmap (/sqrt l) v ; nearZero = (<=1e-12).abs */
static inline void normalize3(const double v[3], double o[3])
{
    double l = quadrance3(v);
    if (fabs(l) <= 1e-12 || fabs(1.0 - l) <= 1e-12) {
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {
        double s = sqrt(l);
        o[0] = v[0] / s; o[1] = v[1] / s; o[2] = v[2] / s;
    }
}

/* ---------------------------------------------------------------- derived scene (Raytracer.hs:57-65) */

typedef struct {
    double cam[3], lookat[3], up[3], fov;
    double W, H;           /* traced resolution as doubles (cfg' resolution, :63) */
    int wt, ht;            /* traced resolution */
    double h, safe, in2, out2;
    double disk_rgb[3], disk_opacity;
    double star_intensity, star_saturation;
} scene_t;

void orc_hsi_to_rgb(double hp, double s, double i, double rgb[3])
{
    /* massiv-io Graphics.ColorSpace: toPixelRGB (PixelHSI h' s i), h' in [0,1) (recalled) */
    const double pi = 3.141592653589793;
    double h = hp * 2 * pi;
    double is = i * s;
    double second = i - is;
#define FIRST(a, b) (i + is * cos(a) / cos(b))
#define THIRD(v1, v2) (i + 2 * is + (v1) - (v2))
    if (h < 0) {
        rgb[0] = rgb[1] = rgb[2] = NAN; /* reference: error "HSI pixel is not properly scaled" */
    } else if (h < 2 * pi / 3) {
        double r = FIRST(h, pi / 3 - h);
        double b = second;
        double g = THIRD(b, r);
        rgb[0] = r; rgb[1] = g; rgb[2] = b;
    } else if (h < 4 * pi / 3) {
        double g = FIRST(h - 2 * pi / 3, h + pi);
        double r = second;
        double b = THIRD(r, g);
        rgb[0] = r; rgb[1] = g; rgb[2] = b;
    } else if (h < 2 * pi) {
        double b = FIRST(h - 4 * pi / 3, 2 * pi - pi / 3 - h);
        double g = second;
        double r = THIRD(g, b);
        rgb[0] = r; rgb[1] = g; rgb[2] = b;
    } else {
        rgb[0] = rgb[1] = rgb[2] = NAN;
    }
#undef FIRST
#undef THIRD
}

static void derive_scene(const orc_config *c, scene_t *s)
{
    memcpy(s->cam, c->cam_pos, sizeof s->cam);
    memcpy(s->lookat, c->cam_lookat, sizeof s->lookat);
    memcpy(s->up, c->cam_up, sizeof s->up);
    s->fov = c->fov;
    /* Raytracer.hs:58  res = if supersampling then (2*w, 2*h) else (w, h) */
    s->wt = c->supersampling ? 2 * c->width : c->width;
    s->ht = c->supersampling ? 2 * c->height : c->height;
    s->W = (double)s->wt;
    s->H = (double)s->ht;
    s->h = c->step_size;
    /* :59-60 safeDistance = max (50^2) (2 * quadrance (position cam));  max x y = if x <= y then y else x */
    double a = 50.0 * 50.0, b = 2 * quadrance3(c->cam_pos);
    s->safe = (a <= b) ? b : a;
    s->in2 = c->disk_inner * c->disk_inner;  /* :61 */
    s->out2 = c->disk_outer * c->disk_outer; /* :62 */
    orc_hsi_to_rgb(c->disk_hsi[0], c->disk_hsi[1], c->disk_hsi[2], s->disk_rgb); /* :65 */
    s->disk_opacity = c->disk_opacity;
    s->star_intensity = c->star_intensity;
    s->star_saturation = c->star_saturation;
}

/* Raytracer.hs:40-51 generateRay, evaluated per pixel exactly as the reference does. */
static void generate_ray(const scene_t *s, int yi, int xi, double vel[3], double pos[3])
{
    /* linear lookAt eye center up: za = normalize (center - eye); xa = normalize (cross za up); ya = cross xa za;
       rows of _m33 = xa, ya, -za.  (transpose m !* v)_i = (xa_i*v0 + ya_i*v1) + (-za_i)*v2 */
    double d[3] = {s->lookat[0] - s->cam[0], s->lookat[1] - s->cam[1], s->lookat[2] - s->cam[2]};
    double za[3], xa[3], ya[3], t[3];
    normalize3(d, za);
    cross3(za, s->up, t);
    normalize3(t, xa);
    cross3(xa, za, ya);
    double v0 = s->fov * ((double)xi / s->W - 0.5);
    double v1 = s->fov * (0.5 - (double)yi / s->H) * s->H / s->W;
    double v2 = -1.0;
    double dir[3];
    for (int i = 0; i < 3; i++) dir[i] = (xa[i] * v0 + ya[i] * v1) + (-za[i]) * v2;
    normalize3(dir, vel);
    pos[0] = s->cam[0]; pos[1] = s->cam[1]; pos[2] = s->cam[2];
}

void orc_generate_ray(const orc_config *cfg, int y, int x, double vel[3], double pos[3])
{
    scene_t s;
    derive_scene(cfg, &s);
    generate_ray(&s, y, x, vel, pos);
}

/* ---------------------------------------------------------------- rk4 (Raytracer.hs:113-134) */

/* f (PhotonState vel pos) = PhotonState (-1.5*h2 / (norm pos ^ 5) *^ pos) vel
 * parses as negate (((1.5*h2) / (n^5)) *^ pos); n^5 = ((n*n)*(n*n))*n (GHC.Real (^)). */
static inline void rhs(double h2, const double vel[3], const double pos[3], double kv[3], double kp[3])
{
    double n = sqrt(quadrance3(pos));
    double n2 = n * n;
    double n5 = (n2 * n2) * n;
    double c = (1.5 * h2) / n5;
    kv[0] = -(c * pos[0]); kv[1] = -(c * pos[1]); kv[2] = -(c * pos[2]);
    kp[0] = vel[0]; kp[1] = vel[1]; kp[2] = vel[2];
}

void orc_rk4(double h, double h2, const double vel[3], const double pos[3], double nvel[3], double npos[3])
{
    double hh = h / 2, h6 = h / 6;
    double k1v[3], k1p[3], k2v[3], k2p[3], k3v[3], k3p[3], k4v[3], k4p[3], sv[3], sp[3];
    rhs(h2, vel, pos, k1v, k1p);
    for (int i = 0; i < 3; i++) { sv[i] = vel[i] + k1v[i] * hh; sp[i] = pos[i] + k1p[i] * hh; } /* y `add` mul (h/2) k1 */
    rhs(h2, sv, sp, k2v, k2p);
    for (int i = 0; i < 3; i++) { sv[i] = vel[i] + k2v[i] * hh; sp[i] = pos[i] + k2p[i] * hh; }
    rhs(h2, sv, sp, k3v, k3p);
    for (int i = 0; i < 3; i++) { sv[i] = vel[i] + k3v[i] * h; sp[i] = pos[i] + k3p[i] * h; }
    rhs(h2, sv, sp, k4v, k4p);
    for (int i = 0; i < 3; i++) {
        /* sumK = ((k1 + 2*k2) + 2*k3) + k4 ; mul 2 k = k * 2 */
        double skv = ((k1v[i] + k2v[i] * 2) + k3v[i] * 2) + k4v[i];
        double skp = ((k1p[i] + k2p[i] * 2) + k3p[i] * 2) + k4p[i];
        nvel[i] = vel[i] + skv * h6;
        npos[i] = pos[i] + skp * h6;
    }
}

/* ---------------------------------------------------------------- star index (independent of the product's cube-map direction grid) */

#define GRID 256
struct orc_index {
    size_t n;
    orc_star *stars;     /* copy, original order */
    uint32_t *cell_start; /* GRID^3 + 1 */
    uint32_t *order;      /* star ids sorted by cell */
};

static inline int cell_of(double c)
{
    /* domain [-1.01, 1.01) -> GRID cells of width 2.02/256 = 0.00789 > radius 0.0015 */
    int i = (int)floor((c + 1.01) * (GRID / 2.02));
    if (i < 0) i = 0;
    if (i >= GRID) i = GRID - 1;
    return i;
}

orc_index *orc_index_create(const orc_star *stars, size_t n)
{
    orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
    if (!ix) return NULL;
    ix->n = n;
    ix->stars = (orc_star *)malloc((n ? n : 1) * sizeof(orc_star));
    ix->order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    size_t ncell = (size_t)GRID * GRID * GRID;
    ix->cell_start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t));
    if (!ix->stars || !ix->order || !ix->cell_start) { orc_index_destroy(ix); return NULL; }
    if (n) memcpy(ix->stars, stars, n * sizeof(orc_star));
    uint32_t *key = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        key[i] = ((uint32_t)cell_of(stars[i].x) * GRID + (uint32_t)cell_of(stars[i].y)) * GRID + (uint32_t)cell_of(stars[i].z);
        ix->cell_start[key[i] + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) ix->cell_start[c + 1] += ix->cell_start[c];
    uint32_t *fill = (uint32_t *)malloc(ncell * sizeof(uint32_t));
    memcpy(fill, ix->cell_start, ncell * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) ix->order[fill[key[i]]++] = (uint32_t)i; /* stable: ascending id within a cell */
    free(fill);
    free(key);
    return ix;
}

void orc_index_destroy(orc_index *ix)
{
    if (!ix) return;
    free(ix->stars); free(ix->order); free(ix->cell_start); free(ix);
}

/* ---------------------------------------------------------------- starLookup (StarMap.hs:93-115) */

static inline void star_pixel(const orc_star *st, const double nvel[3], double intensity, double saturation, double rgb[3])
{
    const double max_brightness = 950, dynamic = 50, w = 0.0005;
    double dv[3] = {st->x - nvel[0], st->y - nvel[1], st->z - nvel[2]}; /* qd pos nvel = quadrance (pos ^-^ nvel) */
    double d2 = quadrance3(dv);
    double a = log(2.0) / dynamic;
    double e = exp(a * (max_brightness - (double)st->mag) - d2 / (2 * (w * w)));
    double m = (1.0 <= e) ? 1.0 : e; /* min 1 e */
    double val = m * intensity;
    orc_hsi_to_rgb(st->hue, saturation * st->sat, val, rgb);
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* Sum order: ascending star id (the reference's order is kdt's traversal order, which only
 * affects the last ulp; StarMap.hs:115 foldl' (liftA2 (+)) (PixelRGB 0 0 0)). */
static int sum_hits(const orc_star *stars, uint32_t *hits, int nh, double intensity, double saturation,
                    const double nvel[3], double rgb[3])
{
    qsort(hits, (size_t)nh, sizeof(uint32_t), cmp_u32);
    double acc[3] = {0, 0, 0};
    for (int k = 0; k < nh; k++) {
        double c[3];
        star_pixel(&stars[hits[k]], nvel, intensity, saturation, c);
        acc[0] = acc[0] + c[0]; acc[1] = acc[1] + c[1]; acc[2] = acc[2] + c[2];
    }
    for (int i = 0; i < 3; i++) rgb[i] = (1.0 <= acc[i]) ? 1.0 : acc[i]; /* fmap (min 1) */
    return nh;
}

/* The reference folds over however many stars inRadius returns (StarMap.hs:104,115): the hit list grows on demand and is
 * NEVER truncated (round 2 capped it silently at 4096).  Running out of memory ends the process with a message: a checker
 * that quietly drops stars would be worse than none. */
typedef struct { uint32_t *v; int n, cap; uint32_t small[64]; } hitlist;

static void hits_init(hitlist *h) { h->v = h->small; h->n = 0; h->cap = (int)(sizeof h->small / sizeof h->small[0]); }

static void hits_push(hitlist *h, uint32_t id)
{
    if (h->n == h->cap) {
        int cap = h->cap * 2;
        uint32_t *nv = (uint32_t *)malloc((size_t)cap * sizeof(uint32_t));
        if (!nv) { fprintf(stderr, "blackstar_oracle: out of memory growing a star hit list to %d entries\n", cap); abort(); }
        memcpy(nv, h->v, (size_t)h->n * sizeof(uint32_t));
        if (h->v != h->small) free(h->v);
        h->v = nv; h->cap = cap;
    }
    h->v[h->n++] = id;
}

static void hits_free(hitlist *h) { if (h->v != h->small) free(h->v); }

int orc_star_lookup(const orc_index *ix, double intensity, double saturation, const double vel[3], double rgb[3])
{
    const double w = 0.0005;
    const double radius = 3 * w; /* StarMap.hs:104 */
    const double r2 = radius * radius; /* kdt inRadius: distSqr p q <= radius*radius */
    double nvel[3];
    normalize3(vel, nvel); /* :103 */
    hitlist hits;
    hits_init(&hits);
    if (ix && ix->n) {
        int lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = cell_of(nvel[a] - radius * 1.01); hi[a] = cell_of(nvel[a] + radius * 1.01); }
        for (int cx = lo[0]; cx <= hi[0]; cx++)
            for (int cy = lo[1]; cy <= hi[1]; cy++)
                for (int cz = lo[2]; cz <= hi[2]; cz++) {
                    size_t c = ((size_t)cx * GRID + cy) * GRID + cz;
                    for (uint32_t k = ix->cell_start[c]; k < ix->cell_start[c + 1]; k++) {
                        const orc_star *st = &ix->stars[ix->order[k]];
                        double dv[3] = {st->x - nvel[0], st->y - nvel[1], st->z - nvel[2]};
                        if (quadrance3(dv) <= r2) hits_push(&hits, ix->order[k]);
                    }
                }
    }
    int nh = sum_hits(ix ? ix->stars : NULL, hits.v, hits.n, intensity, saturation, nvel, rgb);
    hits_free(&hits);
    return nh;
}

int orc_star_lookup_brute(const orc_star *stars, size_t n, double intensity, double saturation, const double vel[3], double rgb[3])
{
    const double radius = 3 * 0.0005, r2 = radius * radius;
    double nvel[3];
    normalize3(vel, nvel);
    hitlist hits;
    hits_init(&hits);
    for (size_t i = 0; i < n; i++) {
        double dv[3] = {stars[i].x - nvel[0], stars[i].y - nvel[1], stars[i].z - nvel[2]};
        if (quadrance3(dv) <= r2) hits_push(&hits, (uint32_t)i);
    }
    int nh = sum_hits(stars, hits.v, hits.n, intensity, saturation, nvel, rgb);
    hits_free(&hits);
    return nh;
}

/* ---------------------------------------------------------------- colorize / findColor / blend */

static inline double signum(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); } /* GHC.Float signum */

/* Raytracer.hs:34-37  blend top layer = top_c + layer_c * (1 - top_alpha), all four channels */
static inline void blend(double top[4], const double layer[4])
{
    double ta = top[3];
    for (int c = 0; c < 4; c++) top[c] = top[c] + layer[c] * (1 - ta);
}

/* Raytracer.hs:104-111 diskColor' */
static inline void disk_color(const scene_t *s, double r, double out[4])
{
    const double pi = 3.141592653589793;
    double rI = sqrt(s->in2), rO = sqrt(s->out2);
    double t = (rO - r) / (rO - rI);
    double inten = sin(pi * (t * t));
    out[0] = s->disk_rgb[0] * inten; out[1] = s->disk_rgb[1] * inten; out[2] = s->disk_rgb[2] * inten;
    out[3] = inten * s->disk_opacity;
}

/* Raytracer.hs:69-86 traceRay + colorize, with findColor (:88-102) inlined. */
static void trace_ray(const scene_t *s, const orc_index *ix, int yi, int xi, int max_steps, orc_ray_record *rec)
{
    double vel[3], pos[3], nvel[3], npos[3], cr[3];
    generate_ray(s, yi, xi, vel, pos);
    cross3(pos, vel, cr);
    double h2 = quadrance3(cr); /* :73 */
    double rgba[4] = {0, 0, 0, 0};
    int steps = 0, fate = 2, disk_hits = 0, star_hits = 0;
    while (steps < max_steps) {
        orc_rk4(s->h, h2, vel, pos, nvel, npos); /* :81 (evaluated before the guards: Strict) */
        steps++;
        double r2 = quadrance3(pos), r2n = quadrance3(npos); /* :100-101 */
        double y = pos[1], yn = npos[1];
        double r2ave = (yn * r2 - y * r2n) / (yn - y); /* :102 */
        if (r2 < 1) { /* :93 */
            double l[4] = {0, 0, 0, 1};
            blend(rgba, l);
            fate = 0;
            break;
        } else if (r2 > s->safe) { /* :94-95 uses the OLD vel */
            double l[4];
            star_hits = orc_star_lookup(ix, s->star_intensity, s->star_saturation, vel, l);
            l[3] = 1.0;
            blend(rgba, l);
            fate = 1;
            break;
        } else if (s->disk_opacity != 0 && signum(yn) != signum(y) && r2ave > s->in2 && r2ave < s->out2) { /* :96-98 */
            double l[4];
            disk_color(s, sqrt(r2ave), l);
            blend(rgba, l);
            disk_hits++;
        }
        memcpy(vel, nvel, sizeof vel);
        memcpy(pos, npos, sizeof pos);
    }
    memcpy(rec->vel, vel, sizeof vel);
    memcpy(rec->pos, pos, sizeof pos);
    memcpy(rec->rgba, rgba, sizeof rgba);
    rec->steps = steps; rec->fate = fate; rec->disk_hits = disk_hits; rec->star_hits = star_hits;
}

int orc_trace_rays(const orc_config *cfg, const orc_index *idx, const int32_t *yx, size_t n_rays, int max_steps, orc_ray_record *out)
{
    scene_t s;
    derive_scene(cfg, &s);
    for (size_t i = 0; i < n_rays; i++) trace_ray(&s, idx, yx[2 * i], yx[2 * i + 1], max_steps, &out[i]);
    return 0;
}

/* ImageFilters.hs:88-97 supersample: 0.25 * (((p(2y,2x) + p(2y+1,2x)) + p(2y,2x+1)) + p(2y+1,2x+1)) */
void orc_supersample(const double *in, int ht, int wt, double *out)
{
    int h = ht / 2, w = wt / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                double a = in[((size_t)(2 * y) * wt + 2 * x) * 3 + c];
                double b = in[((size_t)(2 * y + 1) * wt + 2 * x) * 3 + c];
                double cc = in[((size_t)(2 * y) * wt + 2 * x + 1) * 3 + c];
                double d = in[((size_t)(2 * y + 1) * wt + 2 * x + 1) * 3 + c];
                out[((size_t)y * w + x) * 3 + c] = 0.25 * (((a + b) + cc) + d);
            }
}

/* ---------------------------------------------------------------- render (Raytracer.hs:53-67), threaded over rows */

typedef struct {
    const scene_t *s;
    const orc_index *ix;
    double *img; /* traced-resolution RGB */
    int max_steps;
    volatile int *next_row;
    orc_stats st;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const scene_t *s = j->s;
    orc_stats acc; /* thread-local accumulator: adjacent job_t structs would false-share a cache line */
    memset(&acc, 0, sizeof acc);
    for (;;) {
        int y = __sync_fetch_and_add(j->next_row, 1);
        if (y >= s->ht) break;
        for (int x = 0; x < s->wt; x++) {
            orc_ray_record r;
            trace_ray(s, j->ix, y, x, j->max_steps, &r);
            double *p = &j->img[((size_t)y * s->wt + x) * 3];
            p[0] = r.rgba[0]; p[1] = r.rgba[1]; p[2] = r.rgba[2]; /* dropAlpha */
            acc.rays++;
            acc.steps += (uint64_t)r.steps;
            acc.capped += (r.fate == 2);
            acc.horizon += (r.fate == 0);
            acc.escaped += (r.fate == 1);
            acc.disk_hits += (uint64_t)r.disk_hits;
            acc.star_hits += (uint64_t)r.star_hits;
        }
    }
    j->st = acc;
    return NULL;
}

int orc_render(const orc_config *cfg, const orc_index *idx, double *out_rgb, size_t out_doubles, int threads, int max_steps, orc_stats *stats)
{
    if (!cfg || !out_rgb || cfg->width <= 0 || cfg->height <= 0) return -1;
    if (out_doubles < (size_t)cfg->width * cfg->height * 3) return -2;
    scene_t s;
    derive_scene(cfg, &s);
    if (threads <= 0) threads = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    double *img = out_rgb;
    if (cfg->supersampling) {
        img = (double *)malloc((size_t)s.wt * s.ht * 3 * sizeof(double));
        if (!img) return -3;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    volatile int next_row = 0;
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    char *started = (char *)calloc((size_t)threads, 1);
    if (!jobs || !tid || !started) { free(jobs); free(tid); free(started); if (cfg->supersampling) free(img); return -3; }
    for (int t = 0; t < threads; t++) {
        jobs[t].s = &s; jobs[t].ix = idx; jobs[t].img = img; jobs[t].max_steps = max_steps; jobs[t].next_row = &next_row;
        if (threads == 1) worker(&jobs[t]);
        else started[t] = pthread_create(&tid[t], NULL, worker, &jobs[t]) == 0;
    }
    /* rows are pulled off one counter, so a thread that could not be started (EAGAIN under a process / thread limit) only means
     * fewer workers; with none at all the calling thread does the frame */
    if (threads > 1) {
        int any = 0;
        for (int t = 0; t < threads; t++) any |= started[t];
        if (!any) worker(&jobs[0]);
    }
    orc_stats tot;
    memset(&tot, 0, sizeof tot);
    for (int t = 0; t < threads; t++) {
        if (threads > 1 && started[t]) pthread_join(tid[t], NULL);
        tot.rays += jobs[t].st.rays; tot.steps += jobs[t].st.steps; tot.capped += jobs[t].st.capped;
        tot.horizon += jobs[t].st.horizon; tot.escaped += jobs[t].st.escaped;
        tot.disk_hits += jobs[t].st.disk_hits; tot.star_hits += jobs[t].st.star_hits;
    }
    if (cfg->supersampling) { /* :67 */
        orc_supersample(img, s.ht, s.wt, out_rgb);
        free(img);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    tot.seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    tot.threads = threads;
    if (stats) *stats = tot;
    free(jobs); free(tid); free(started);
    return 0;
}

/* ---------------------------------------------------------------- PPM catalogue (StarMap.hs:45-75) */

static double be_f64(const unsigned char *p)
{
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u = (u << 8) | p[i];
    double d;
    memcpy(&d, &u, 8);
    return d;
}

static void star_color(int ch, double *hue, double *sat)
{ /* StarMap.hs:64-72 */
    switch (ch) {
    case 'O': *hue = 0.631; *sat = 0.39; break;
    case 'B': *hue = 0.628; *sat = 0.33; break;
    case 'A': *hue = 0.622; *sat = 0.21; break;
    case 'F': *hue = 0.650; *sat = 0.03; break;
    case 'G': *hue = 0.089; *sat = 0.09; break;
    case 'K': *hue = 0.094; *sat = 0.29; break;
    case 'M': *hue = 0.094; *sat = 0.56; break;
    default: *hue = 0; *sat = 0; break;
    }
}

long orc_read_ppm(const unsigned char *b, size_t nbytes, orc_star *out, size_t cap)
{
    if (nbytes < 28) return -1; /* cereal: skip 28 fails on short input */
    size_t n = (nbytes - 28) / 28;
    if (n > cap) n = cap;
    for (size_t i = 0; i < n; i++) {
        const unsigned char *r = b + 28 + i * 28;
        double ra = be_f64(r), dec = be_f64(r + 8);
        int sp = r[16];
        int16_t mag = (int16_t)(((uint16_t)r[18] << 8) | r[19]);
        /* raDecToCartesian: V3 (cos dec*cos ra) (cos dec*sin ra) (sin dec) */
        out[i].x = cos(dec) * cos(ra);
        out[i].y = cos(dec) * sin(ra);
        out[i].z = sin(dec);
        out[i].mag = mag;
        out[i]._pad = 0;
        star_color(sp, &out[i].hue, &out[i].sat);
    }
    return (long)n;
}

/* ---------------------------------------------------------------- bloom / boxBlur (ImageFilters.hs:28-86) */

/* one sweep: n_chains chains of n samples; chain k starts at base k*chain_stride (+ channel), samples `stride` apart */
static void blur_sweep(const double *in, double *out, int n_chains, int n, long chain_stride, long stride, int r, double norm)
{
    for (int k = 0; k < n_chains; k++)
        for (int c = 0; c < 3; c++) {
            const double *src = in + (long)k * chain_stride + c;
            double *dst = out + (long)k * chain_stride + c;
            int m = r < n ? r : n;
            double s = src[0]; /* startVal = foldl1' add . map pix . take r (:59) */
            for (int i = 1; i < m; i++) s = s + src[(long)i * stride];
            for (int x = 0; x < n; x++) { /* accumulate (:61-64): newRGB = (rgb + pix (x+r)) - pix (x-r) */
                double lead = (x + r < n) ? src[(long)(x + r) * stride] : 0.0;
                double trail = (x - r >= 0) ? src[(long)(x - r) * stride] : 0.0;
                s = (s + lead) - trail;
                dst[(long)x * stride] = norm * s;
            }
        }
}

int orc_bloom(double strength, int divider, const double *img, int h, int w, double *out)
{
    int r = w / divider; /* :83 */
    if (r == 0) return -1; /* the reference crashes (foldl1' on an empty vector) */
    size_t n = (size_t)w * h * 3;
    double *a = (double *)malloc(n * sizeof(double)), *b = (double *)malloc(n * sizeof(double));
    if (!a || !b) { free(a); free(b); return -3; }
    double norm = 1 / (2 * (double)r + 1); /* :51 */
    const double *src = img;
    for (int pass = 0; pass < 3; pass++) { /* :70-76: H reads the frozen copy, V reads the H result */
        blur_sweep(src, a, h, w, (long)w * 3, 3, r, norm);
        blur_sweep(a, b, w, h, 3, (long)w * 3, r, norm);
        src = b;
    }
    for (size_t i = 0; i < n; i++) out[i] = img[i] + strength * b[i]; /* :84-86 */
    free(a); free(b);
    return 0;
}

/* writeImg's pixel map (Raytracer.hs:23-32): toWord8 . fmap sRGB */
void orc_srgb8(const double *in, unsigned char *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        double x = in[i];
        double y = (x < 0.0031308) ? 12.92 * x : (1 + 0.055) * pow(x, 1.0 / 2.4) - 0.055;
        y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);
        out[i] = (unsigned char)(int)rint(255.0 * y); /* round half to even */
    }
}
