"""Regenerates tests/golden/*.npz.  Run from the repo root:  python tests/golden/make_golden.py

The reference (Haskell) cannot run in this image and holds no golden vectors, so these fixtures come from
the INDEPENDENT numpy restatement (oracle/np_oracle.py) -- not from the C oracle and not from the product.
They are data only: inputs (config, pixel list, catalogue bytes) and expected outputs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_oracle as no  # noqa: E402
from oracle import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED_SMALL = 0x5EEDB1AC57A2 + 1


def small_catalogue():
    """2,000-star PPM-layout catalogue (SURVEY 8d recipe, seed+1), generated here independently of the product."""
    mask = (1 << 64) - 1

    def splitmix(seed, n):
        out, s = [], seed & mask
        for _ in range(n):
            s = (s + 0x9E3779B97F4A7C15) & mask
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
            out.append(z ^ (z >> 31))
        return out

    n = 2000
    u = np.array([(x >> 11) * 2.0 ** -53 for x in splitmix(SEED_SMALL, 4 * n)]).reshape(n, 4)
    dec = np.arcsin(2 * u[:, 0] - 1)
    ra = 2 * np.pi * u[:, 1]
    mag = (1200 - np.floor(700 * u[:, 2] ** 3)).astype(np.int16)
    sp = np.frombuffer(b"OBAFGKM?", np.uint8)[np.floor(8 * u[:, 3]).astype(int)]
    rec = np.zeros(n, np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]))
    rec["ra"], rec["dec"], rec["sp"], rec["mag"] = ra, dec, sp, mag
    return bytes(28) + rec.tobytes()


COLOURS = {ord("O"): (0.631, 0.39), ord("B"): (0.628, 0.33), ord("A"): (0.622, 0.21), ord("F"): (0.650, 0.03),
           ord("G"): (0.089, 0.09), ord("K"): (0.094, 0.29), ord("M"): (0.094, 0.56)}


def parse_catalogue(data):
    rec = np.frombuffer(data, np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]), offset=28)
    hs = np.array([COLOURS.get(int(c), (0.0, 0.0)) for c in rec["sp"]])
    return np.stack([np.cos(rec["dec"]) * np.cos(rec["ra"]), np.cos(rec["dec"]) * np.sin(rec["ra"]), np.sin(rec["dec"]),
                     hs[:, 0], hs[:, 1], rec["mag"].astype(np.float64)], axis=1)


def main():
    cat = small_catalogue()
    with open(os.path.join(HERE, "catalogue_2000.ppm"), "wb") as f:
        f.write(cat)
    stars = parse_catalogue(cat)
    np.savez_compressed(os.path.join(HERE, "catalogue_2000_parsed.npz"), stars=stars)

    images = {
        "c2_default_96x54_nostars": (scenes.with_res(scenes.DEFAULT, 96, 54), None),
        "c3_default_aa_96x54": (scenes.with_res(scenes.DEFAULT_AA, 96, 54), stars),
        "c4_lensing_disk_96x54": (scenes.with_res(scenes.LENSING_DISK, 96, 54), stars),
        "c5_ani_frame300_80x45": (scenes.with_res(scenes.ani_frame(300, 600), 80, 45), stars),
        "odd_default_aa_37x23": (scenes.with_res(scenes.DEFAULT_AA, 37, 23), stars),
    }
    for name, (cfg, st) in images.items():
        img, rec = no.render(cfg, st)
        np.savez_compressed(os.path.join(HERE, f"image_{name}.npz"), cfg=json.dumps(cfg), img=img,
                            total_steps=np.int64(rec["steps"].sum()), fate_counts=np.bincount(rec["fate"], minlength=3),
                            disk_hits=np.int64(rec["disk_hits"].sum()), star_hits=np.int64(rec["star_hits"].sum()))
        print(name, img.shape, int(rec["steps"].sum()))

    # per-ray traces at FULL BASELINE resolution: grid + random pixels + rays bracketing the shadow edge / disk edges
    rng = np.random.default_rng(20260927)
    trace_cfgs = {"c1": scenes.with_res(scenes.DEFAULT, 640, 480), "c2": scenes.DEFAULT, "c3": scenes.DEFAULT_AA,
                  "c4": scenes.with_res(scenes.LENSING_DISK, 3840, 2160), "c5_f0": scenes.ani_frame(0, 600),
                  "c5_f599": scenes.ani_frame(599, 600)}
    for name, cfg in trace_cfgs.items():
        sc = no.derive(cfg)
        gy, gx = np.meshgrid(np.linspace(0, sc["ht"] - 1, 12).astype(int), np.linspace(0, sc["wt"] - 1, 12).astype(int), indexing="ij")
        ys = np.concatenate([gy.ravel(), rng.integers(0, sc["ht"], 80)])
        xs = np.concatenate([gx.ravel(), rng.integers(0, sc["wt"], 80)])
        # a dense horizontal scan line through the image centre crosses the shadow edge and both disk edges
        ys = np.concatenate([ys, np.full(160, sc["ht"] // 2 + 3)])
        xs = np.concatenate([xs, np.linspace(0, sc["wt"] - 1, 160).astype(int)])
        rec = no.trace(cfg, stars, ys, xs)
        v0, p0 = no.generate_rays(sc, ys, xs)
        np.savez_compressed(os.path.join(HERE, f"trace_{name}.npz"), cfg=json.dumps(cfg), ys=ys.astype(np.int32), xs=xs.astype(np.int32),
                            vel0=v0, h2=rec["h2"], vel=rec["vel"], pos=rec["pos"], rgba=rec["rgba"], steps=rec["steps"],
                            fate=rec["fate"], disk_hits=rec["disk_hits"], star_hits=rec["star_hits"])
        print(name, len(ys), rec["steps"].mean(), np.bincount(rec["fate"]))


if __name__ == "__main__" and not {"--c1", "--extra", "--clustered"} & set(sys.argv):
    main()


# down-scaled frame of each of the reference's other six scene files (aspect kept; supersampling as in the file)
EXTRA_IMAGE_RES = {"closeup": (96, 72), "fartheraway": (96, 54), "lensing": (96, 72), "wideangle-disk": (96, 54),
                   "wideangle": (96, 51), "wideangle1": (96, 54)}


def extra_scenes():
    """Goldens for /root/reference/scenes/{closeup,fartheraway,lensing,wideangle-disk,wideangle,wideangle1}.yaml (as restated in
    oracle/scenes.py): one down-scaled image and 384 per-ray traces at the file's FULL resolution each."""
    cat = open(os.path.join(HERE, "catalogue_2000.ppm"), "rb").read()
    stars = parse_catalogue(cat)
    rng = np.random.default_rng(20260928)
    for name in scenes.EXTRA_SCENES:
        cfg_full = scenes.REFERENCE_SCENES[name]
        key = name.replace("-", "_")
        w, h = EXTRA_IMAGE_RES[name]
        cfg = scenes.with_res(cfg_full, w, h)
        img, rec = no.render(cfg, stars)
        np.savez_compressed(os.path.join(HERE, f"image_ref_{key}_{w}x{h}.npz"), cfg=json.dumps(cfg), img=img,
                            total_steps=np.int64(rec["steps"].sum()), fate_counts=np.bincount(rec["fate"], minlength=3),
                            disk_hits=np.int64(rec["disk_hits"].sum()), star_hits=np.int64(rec["star_hits"].sum()))
        print(name, img.shape, int(rec["steps"].sum()), np.bincount(rec["fate"], minlength=3), int(rec["disk_hits"].sum()))
        sc = no.derive(cfg_full)
        gy, gx = np.meshgrid(np.linspace(0, sc["ht"] - 1, 12).astype(int), np.linspace(0, sc["wt"] - 1, 12).astype(int), indexing="ij")
        ys = np.concatenate([gy.ravel(), rng.integers(0, sc["ht"], 80), np.full(160, sc["ht"] // 2 + 3)])
        xs = np.concatenate([gx.ravel(), rng.integers(0, sc["wt"], 80), np.linspace(0, sc["wt"] - 1, 160).astype(int)])
        rec = no.trace(cfg_full, stars, ys, xs)
        v0, p0 = no.generate_rays(sc, ys, xs)
        np.savez_compressed(os.path.join(HERE, f"trace_ref_{key}.npz"), cfg=json.dumps(cfg_full), ys=ys.astype(np.int32), xs=xs.astype(np.int32),
                            vel0=v0, h2=rec["h2"], vel=rec["vel"], pos=rec["pos"], rgba=rec["rgba"], steps=rec["steps"],
                            fate=rec["fate"], disk_hits=rec["disk_hits"], star_hits=rec["star_hits"])
        print("  trace", len(ys), rec["steps"].mean(), np.bincount(rec["fate"]), int(rec["disk_hits"].sum()))


if __name__ == "__main__" and "--extra" in sys.argv:
    extra_scenes()


def c1_summary():
    """BASELINE configs[0]: default.yaml at 640x480, no supersampling, full frame -- summary statistics and a 24x32 grid of
    sampled pixels (the full image is 7 MB; the per-ray traces of trace_c1.npz cover individual rays)."""
    cat = open(os.path.join(HERE, "catalogue_2000.ppm"), "rb").read()
    stars = parse_catalogue(cat)
    cfg = scenes.with_res(scenes.DEFAULT, 640, 480)
    img, rec = no.render(cfg, stars)
    ys, xs = np.meshgrid(np.arange(10, 480, 20), np.arange(10, 640, 20), indexing="ij")
    np.savez_compressed(os.path.join(HERE, "summary_c1_default_640x480.npz"), cfg=json.dumps(cfg), total_steps=np.int64(rec["steps"].sum()),
                        fate_counts=np.bincount(rec["fate"], minlength=3), disk_hits=np.int64(rec["disk_hits"].sum()),
                        star_hits=np.int64(rec["star_hits"].sum()), ys=ys.ravel(), xs=xs.ravel(), samples=img[ys.ravel(), xs.ravel()],
                        channel_sums=img.reshape(-1, 3).sum(axis=0), max_steps=np.int64(rec["steps"].max()))
    print("c1", img.shape, int(rec["steps"].sum()), rec["steps"].mean())


if __name__ == "__main__" and "--c1" in sys.argv:
    c1_summary()


def _ppm_records(xyz, mag, sp):
    xyz = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
    rec = np.zeros(len(xyz), np.dtype([("ra", ">f8"), ("dec", ">f8"), ("sp", "u1"), ("skip", "u1"), ("mag", ">i2"), ("pad", "u1", 8)]))
    rec["ra"] = np.mod(np.arctan2(xyz[:, 1], xyz[:, 0]), 2 * np.pi)
    rec["dec"] = np.arcsin(np.clip(xyz[:, 2], -1, 1))
    rec["sp"], rec["mag"] = sp, mag
    return rec.tobytes()


def clustered():
    """A NON-uniform sky (round 3): the 2,000-star catalogue + 48 clusters of 5..40 stars, each inside 0.001 rad of the direction
    in which one ray of the default-aa camera at 96x54 leaves the scene (so the rendered frame itself holds pixels whose
    starLookup sums 6, 7, 12, ... 40 stars) + a band at 10x the mean density.  starLookup folds over every star inRadius returns
    (src/StarMap.hs:104,115); the fixtures pin lookups with up to 40+ hits.  Written: catalogue_clustered.ppm, the image golden,
    the per-ray golden of the cluster rays and their neighbours, and a batch of plain starLookup queries."""
    rng = np.random.default_rng(20260929)
    base = open(os.path.join(HERE, "catalogue_2000.ppm"), "rb").read()
    cfg = scenes.with_res(scenes.DEFAULT_AA, 96, 54)
    sc = no.derive(cfg)
    ys, xs = np.mgrid[0:sc["ht"], 0:sc["wt"]]
    ys, xs = ys.ravel(), xs.ravel()
    rec0 = no.trace(cfg, None, ys, xs)
    esc = np.nonzero(rec0["fate"] == 1)[0]
    pick = np.sort(rng.choice(esc, 48, replace=False))
    centres = no.normalize(rec0["vel"][pick])
    sizes = np.concatenate([[5, 6, 7, 12, 40, 40, 33, 21], rng.integers(6, 41, 40)])
    c = np.repeat(centres, sizes, axis=0)
    ref = np.where(np.abs(c[:, 2:3]) < 0.9, [[0.0, 0.0, 1.0]], [[1.0, 0.0, 0.0]])
    e1 = np.cross(c, ref); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(c, e1)
    rho, phi = 0.001 * np.sqrt(rng.random((len(c), 1))), 2 * np.pi * rng.random((len(c), 1))
    members = c + rho * (np.cos(phi) * e1 + np.sin(phi) * e2)
    cl = _ppm_records(members, rng.integers(1150, 1450, len(c)).astype(np.int16), np.frombuffer(b"OBAFGKM?", np.uint8)[rng.integers(0, 8, len(c))])
    # band: +-0.035 rad about the great circle whose pole is (0.3, -0.5, 0.81), filled to 10x the mean density of the 2,000 stars
    pole = np.array([0.3, -0.5, 0.81]); pole /= np.linalg.norm(pole)
    b1 = np.cross(pole, [0, 0, 1.0]); b1 /= np.linalg.norm(b1); b2 = np.cross(pole, b1)
    nb = int(round(9 * 2000 / (4 * np.pi) * 2 * np.pi * 2 * np.sin(0.035)))
    sb, lam = np.sin(0.035) * (2 * rng.random((nb, 1)) - 1), 2 * np.pi * rng.random((nb, 1))
    band = np.sqrt(1 - sb * sb) * (np.cos(lam) * b1 + np.sin(lam) * b2) + sb * pole
    bd = _ppm_records(band, (1200 - np.floor(700 * rng.random(nb) ** 3)).astype(np.int16), np.frombuffer(b"OBAFGKM?", np.uint8)[rng.integers(0, 8, nb)])
    cat = base + cl + bd
    with open(os.path.join(HERE, "catalogue_clustered.ppm"), "wb") as f:
        f.write(cat)
    stars = parse_catalogue(cat)
    print("clustered catalogue:", len(stars), "stars,", len(c), "in clusters,", nb, "in the band")

    img, rec = no.render(cfg, stars)
    np.savez_compressed(os.path.join(HERE, "image_clustered_default_aa_96x54.npz"), cfg=json.dumps(cfg), img=img,
                        total_steps=np.int64(rec["steps"].sum()), fate_counts=np.bincount(rec["fate"], minlength=3),
                        disk_hits=np.int64(rec["disk_hits"].sum()), star_hits=np.int64(rec["star_hits"].sum()),
                        max_star_hits=np.int64(rec["star_hits"].max()), star_hits_hist=np.bincount(rec["star_hits"], minlength=64))
    print("image", img.shape, "star hits/ray histogram:", np.bincount(rec["star_hits"]))
    assert rec["star_hits"].max() >= 40 and (rec["star_hits"] == 6).any() and (rec["star_hits"] == 5).any()

    # per-ray golden: the 48 cluster rays, their right-hand neighbours, and 96 random rays
    sel = np.unique(np.concatenate([pick, np.minimum(pick + 1, len(ys) - 1), rng.choice(len(ys), 96, replace=False)]))
    v0, p0 = no.generate_rays(sc, ys[sel], xs[sel])
    np.savez_compressed(os.path.join(HERE, "trace_clustered.npz"), cfg=json.dumps(cfg), ys=ys[sel].astype(np.int32), xs=xs[sel].astype(np.int32),
                        vel0=v0, h2=rec["h2"][sel], vel=rec["vel"][sel], pos=rec["pos"][sel], rgba=rec["rgba"][sel], steps=rec["steps"][sel],
                        fate=rec["fate"][sel], disk_hits=rec["disk_hits"][sel], star_hits=rec["star_hits"][sel])

    # plain starLookup queries: cluster centres (all members in reach), members (part of the cluster in reach), points 0.0005..0.003
    # away from centres (clusters partly / just out of reach), band directions, random directions; un-normalised like `vel` is
    q = [centres * rng.uniform(0.5, 3, (48, 1)), members[rng.choice(len(members), 400, replace=False)],
         c[rng.choice(len(c), 400)] + rng.normal(scale=0.0012, size=(400, 3)), band[rng.choice(nb, 300)] + rng.normal(scale=4e-4, size=(300, 3)),
         rng.normal(size=(352, 3))]
    dirs = np.concatenate(q)
    rgb, hits = no.star_lookup(stars, 0.4, 1.5, dirs)
    np.savez_compressed(os.path.join(HERE, "lookup_clustered.npz"), dirs=dirs, rgb=rgb, hits=hits, intensity=0.4, saturation=1.5)
    print("lookup golden:", len(dirs), "queries, hits histogram:", np.bincount(hits))
    assert hits.max() >= 40 and (hits >= 6).sum() > 300


if __name__ == "__main__" and "--clustered" in sys.argv:
    clustered()
