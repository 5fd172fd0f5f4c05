"""Independent numpy FP64 restatement of blackstar's Raytracer.render hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product (blackstar_amd/).
PARITY UNPINNED by the reference's own tests (it has none, and GHC is absent);
this module exists so that two independently written restatements (this one,
vectorised over rays, and oracle/blackstar_oracle.c, scalar) can be checked
against each other, against 50-digit mpmath (oracle/mp_oracle.py) and against
physics known-answers.  It also generates tests/golden/*.npz
(tests/golden/make_golden.py).

numpy never contracts a*b+c into an FMA, so each line is one IEEE binary64
operation per element, in the reference's order.  Citations: /root/reference.

A config is a plain dict with the keys of include/blackstar_gpu.h:bs_config.
Stars are a float64 array (n, 6): x, y, z, hue, sat, mag.
"""
from __future__ import annotations

import numpy as np

PI = 3.141592653589793


# ----------------------------------------------------------------- linear (recalled semantics)
def quadrance(v):
    return (v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]


def cross(a, b):
    return np.stack(
        [a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
         a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
         a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def normalize(v):
    """linear.normalize: unchanged if |l|<=1e-12 or |1-l|<=1e-12, else v / sqrt l."""
    v = np.asarray(v, dtype=np.float64)
    l = quadrance(v)
    keep = (np.abs(l) <= 1e-12) | (np.abs(1.0 - l) <= 1e-12)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = v / np.sqrt(l)[..., None]
    return np.where(keep[..., None], v, out)


# ----------------------------------------------------------------- colour (massiv-io HSI -> RGB, recalled)
def hsi_to_rgb(hp, s, i):
    hp, s, i = np.broadcast_arrays(np.asarray(hp, np.float64), np.asarray(s, np.float64), np.asarray(i, np.float64))
    h = hp * 2 * PI
    is_ = i * s
    second = i - is_

    def first(a, b):
        return i + is_ * np.cos(a) / np.cos(b)

    def third(v1, v2):
        return i + 2 * is_ + v1 - v2

    out = np.full(h.shape + (3,), np.nan)
    m1 = (h >= 0) & (h < 2 * PI / 3)
    m2 = (h >= 2 * PI / 3) & (h < 4 * PI / 3)
    m3 = (h >= 4 * PI / 3) & (h < 2 * PI)
    r1 = first(h, PI / 3 - h); b1 = second; g1 = third(b1, r1)
    g2 = first(h - 2 * PI / 3, h + PI); r2 = second; b2 = third(r2, g2)
    b3 = first(h - 4 * PI / 3, 2 * PI - PI / 3 - h); g3 = second; r3 = third(g3, b3)
    for m, (r, g, b) in ((m1, (r1, g1, b1)), (m2, (r2, g2, b2)), (m3, (r3, g3, b3))):
        out[m, 0] = r[m]; out[m, 1] = g[m]; out[m, 2] = b[m]
    return out


# ----------------------------------------------------------------- scene derivation (Raytracer.hs:57-65)
def derive(cfg):
    ss = bool(cfg["supersampling"])
    wt = 2 * cfg["width"] if ss else cfg["width"]
    ht = 2 * cfg["height"] if ss else cfg["height"]
    cam = np.array(cfg["cam_pos"], np.float64)
    safe = max(50.0 * 50.0, 2 * float(quadrance(cam)))
    return dict(
        wt=wt, ht=ht, W=float(wt), H=float(ht), cam=cam,
        lookat=np.array(cfg["cam_lookat"], np.float64), up=np.array(cfg["cam_up"], np.float64),
        fov=float(cfg["fov"]), h=float(cfg["step_size"]), safe=safe,
        in2=cfg["disk_inner"] * cfg["disk_inner"], out2=cfg["disk_outer"] * cfg["disk_outer"],
        disk_rgb=hsi_to_rgb(*cfg["disk_hsi"]).reshape(3), opacity=float(cfg["disk_opacity"]),
        intensity=float(cfg["star_intensity"]), saturation=float(cfg["star_saturation"]))


def generate_rays(sc, ys, xs):
    """Raytracer.hs:40-51 on cfg' (traced resolution)."""
    za = normalize(sc["lookat"] - sc["cam"])
    xa = normalize(cross(za, sc["up"]))
    ya = cross(xa, za)
    xs = np.asarray(xs, np.float64)
    ys = np.asarray(ys, np.float64)
    v0 = sc["fov"] * (xs / sc["W"] - 0.5)
    v1 = sc["fov"] * (0.5 - ys / sc["H"]) * sc["H"] / sc["W"]
    v2 = -1.0
    d = np.stack([(xa[i] * v0 + ya[i] * v1) + (-za[i]) * v2 for i in range(3)], axis=-1)
    vel = normalize(d)
    pos = np.broadcast_to(sc["cam"], vel.shape).copy()
    return vel, pos


# ----------------------------------------------------------------- rk4 (Raytracer.hs:113-134)
def _f(h2, vel, pos):
    n = np.sqrt(quadrance(pos))
    n2 = n * n
    n5 = (n2 * n2) * n
    c = (1.5 * h2) / n5
    return -(c[..., None] * pos), vel


def rk4(h, h2, vel, pos):
    hh, h6 = h / 2, h / 6
    k1v, k1p = _f(h2, vel, pos)
    k2v, k2p = _f(h2, vel + k1v * hh, pos + k1p * hh)
    k3v, k3p = _f(h2, vel + k2v * hh, pos + k2p * hh)
    k4v, k4p = _f(h2, vel + k3v * h, pos + k3p * h)
    sv = ((k1v + k2v * 2) + k3v * 2) + k4v
    sp = ((k1p + k2p * 2) + k3p * 2) + k4p
    return vel + sv * h6, pos + sp * h6


# ----------------------------------------------------------------- starLookup (StarMap.hs:93-115)
def star_lookup(stars, intensity, saturation, vel, tree=None):
    """Returns (rgb (n,3), hits (n,)).  Candidate search via scipy cKDTree with a padded
    radius, then the reference's exact test qd <= (3w)^2; sum in ascending star id."""
    vel = np.atleast_2d(np.asarray(vel, np.float64))
    n = vel.shape[0]
    rgb = np.zeros((n, 3))
    hits = np.zeros(n, np.int32)
    if stars is None or len(stars) == 0:
        return rgb, hits
    w = 0.0005
    radius = 3 * w
    r2 = radius * radius
    nv = normalize(vel)
    if tree is None:
        from scipy.spatial import cKDTree
        tree = cKDTree(stars[:, :3])
    cand = tree.query_ball_point(nv, radius * 1.01)
    a = np.log(2.0) / 50
    for k, ids in enumerate(cand):
        if not ids:
            continue
        ids = np.array(sorted(ids))
        p = stars[ids, :3]
        d2 = quadrance(p - nv[k])
        sel = d2 <= r2
        ids, d2 = ids[sel], d2[sel]
        if len(ids) == 0:
            continue
        e = np.exp(a * (950 - stars[ids, 5]) - d2 / (2 * (w * w)))
        val = np.where(1.0 <= e, 1.0, e) * intensity
        c = hsi_to_rgb(stars[ids, 3], saturation * stars[ids, 4], val)
        acc = np.zeros(3)
        for row in c:
            acc = acc + row
        rgb[k] = np.where(1.0 <= acc, 1.0, acc)
        hits[k] = len(ids)
    return rgb, hits


# ----------------------------------------------------------------- colorize (Raytracer.hs:77-111)
def _signum(x):
    return np.where(x > 0, 1.0, np.where(x < 0, -1.0, x))


def trace(cfg, stars, ys, xs, max_steps=100000, tree=None):
    sc = derive(cfg)
    vel, pos = generate_rays(sc, ys, xs)
    n = vel.shape[0]
    h2 = quadrance(cross(pos, vel))
    rgba = np.zeros((n, 4))
    steps = np.zeros(n, np.int32)
    fate = np.full(n, 2, np.int32)
    disk_hits = np.zeros(n, np.int32)
    star_hits = np.zeros(n, np.int32)
    alive = np.ones(n, bool)
    rI, rO = np.sqrt(sc["in2"]), np.sqrt(sc["out2"])
    it = 0
    while alive.any() and it < max_steps:
        it += 1
        idx = np.nonzero(alive)[0]
        v, p = vel[idx], pos[idx]
        nv, np_ = rk4(sc["h"], h2[idx], v, p)
        steps[idx] += 1
        r2 = quadrance(p)
        r2n = quadrance(np_)
        y, yn = p[:, 1], np_[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            r2ave = (yn * r2 - y * r2n) / (yn - y)
        hor = r2 < 1
        esc = (~hor) & (r2 > sc["safe"])
        dsk = (~hor) & (~esc) & (sc["opacity"] != 0) & (_signum(yn) != _signum(y)) & (r2ave > sc["in2"]) & (r2ave < sc["out2"])
        # horizon: Bottom (0,0,0,1)
        if hor.any():
            j = idx[hor]
            ta = rgba[j, 3].copy()
            layer = np.array([0.0, 0.0, 0.0, 1.0])
            rgba[j] = rgba[j] + layer[None, :] * (1 - ta)[:, None]
            fate[j] = 0
        if esc.any():
            j = idx[esc]
            c, nh = star_lookup(stars, sc["intensity"], sc["saturation"], v[esc], tree)
            layer = np.concatenate([c, np.ones((len(j), 1))], axis=1)
            ta = rgba[j, 3].copy()
            rgba[j] = rgba[j] + layer * (1 - ta)[:, None]
            fate[j] = 1
            star_hits[j] = nh
        if dsk.any():
            j = idx[dsk]
            r = np.sqrt(r2ave[dsk])
            t = (rO - r) / (rO - rI)
            inten = np.sin(PI * (t * t))
            layer = np.concatenate([sc["disk_rgb"][None, :] * inten[:, None], (inten * sc["opacity"])[:, None]], axis=1)
            ta = rgba[j, 3].copy()
            rgba[j] = rgba[j] + layer * (1 - ta)[:, None]
            disk_hits[j] += 1
        done = hor | esc
        cont = idx[~done]
        vel[cont] = nv[~done]
        pos[cont] = np_[~done]
        alive[idx[done]] = False
    return dict(vel=vel, pos=pos, rgba=rgba, steps=steps, fate=fate, disk_hits=disk_hits, star_hits=star_hits, h2=h2)


def supersample(img):
    """ImageFilters.hs:88-97."""
    a = img[0::2, 0::2]; b = img[1::2, 0::2]; c = img[0::2, 1::2]; d = img[1::2, 1::2]
    return 0.25 * (((a + b) + c) + d)


def render(cfg, stars, max_steps=100000):
    sc = derive(cfg)
    ys, xs = np.mgrid[0:sc["ht"], 0:sc["wt"]]
    tree = None
    if stars is not None and len(stars):
        from scipy.spatial import cKDTree
        tree = cKDTree(stars[:, :3])
    rec = trace(cfg, stars, ys.ravel(), xs.ravel(), max_steps, tree)
    img = rec["rgba"][:, :3].reshape(sc["ht"], sc["wt"], 3)
    if cfg["supersampling"]:
        img = supersample(img)
    return img, rec


# ----------------------------------------------------------------- bloom (ImageFilters.hs:28-86), sRGB8 (Raytracer.hs:23-32)
def _sweep(img, axis, r, norm):
    """Running-sum box blur along `axis`, vectorised ACROSS chains (each chain is the reference's sequential sum)."""
    a = np.moveaxis(img, axis, 0)  # (n, chains, 3)
    n = a.shape[0]
    out = np.empty_like(a)
    m = min(r, n)
    s = a[0].copy()
    for i in range(1, m):
        s = s + a[i]
    zero = np.zeros_like(a[0])
    for x in range(n):
        lead = a[x + r] if x + r < n else zero
        trail = a[x - r] if x - r >= 0 else zero
        s = (s + lead) - trail
        out[x] = norm * s
    return np.moveaxis(out, 0, axis)


def bloom(strength, divider, img):
    h, w, _ = img.shape
    r = w // divider
    if r == 0:
        raise ValueError("radius 0")
    norm = 1 / (2 * float(r) + 1)
    b = img
    for _ in range(3):
        b = _sweep(b, 1, r, norm)  # horizontal
        b = _sweep(b, 0, r, norm)  # vertical
    return img + strength * b


def srgb8(img):
    a = 0.055
    with np.errstate(invalid="ignore"):
        y = np.where(img < 0.0031308, 12.92 * img, (1 + a) * np.power(img, 1.0 / 2.4) - a)
    return np.rint(255 * np.clip(y, 0.0, 1.0)).astype(np.uint8)
