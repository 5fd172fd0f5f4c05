// blackstar_gpu.hpp -- C++ host mirror of the reference's interface for the hot path, header-only, over the
// C ABI of blackstar_gpu.h.  The reference is compiled Haskell and its toolchain is absent here, so this is the
// compiled-language host side: same names and argument meaning as the reference's modules --
//   ConfigFile: Config{scene, camera}   (src/ConfigFile.hs:16-38, defaults :66-79)
//   StarMap:    StarTree, readMapFromFile (src/StarMap.hs:25-26, 77-80)
//   Raytracer:  render :: Config -> StarTree -> Image   (src/Raytracer.hs:53)
//   ImageFilters: bloom (src/ImageFilters.hs:80)
// Errors surface as std::runtime_error carrying bs_last_error() (the reference's Either String / error calls).
#pragma once

#include <array>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "blackstar_gpu.h"

namespace blackstar {

using V3 = std::array<double, 3>;

struct Camera {  // src/ConfigFile.hs:34-38
    V3 position{}, lookAt{}, upVec{};
    double fov = 1.0;
};

struct Scene {  // src/ConfigFile.hs:20-32 with the defaults of :66-79
    double safeDistance = 0;  // never read from YAML; render derives it (src/Raytracer.hs:59-60)
    double stepSize = 0.3, bloomStrength = 0.4;
    int bloomDivider = 25;
    double starIntensity = 0.7, starSaturation = 0.7;
    V3 diskColor{0.16, 0.1, 0.95};  // PixelHSI, hue in [0,1)
    double diskOpacity = 0, diskInner = 3, diskOuter = 12;
    std::pair<int, int> resolution{1280, 720};
    bool supersampling = false;
};

struct Config {  // src/ConfigFile.hs:16-18
    Scene scene;
    Camera camera;
    bs_config to_bs_config() const
    {
        bs_config c{};
        for (int i = 0; i < 3; i++) {
            c.cam_pos[i] = camera.position[i];
            c.cam_lookat[i] = camera.lookAt[i];
            c.cam_up[i] = camera.upVec[i];
            c.disk_hsi[i] = scene.diskColor[i];
        }
        c.fov = camera.fov;
        c.step_size = scene.stepSize;
        c.star_intensity = scene.starIntensity;
        c.star_saturation = scene.starSaturation;
        c.disk_opacity = scene.diskOpacity;
        c.disk_inner = scene.diskInner;  // as parsed: un-squared
        c.disk_outer = scene.diskOuter;
        c.width = scene.resolution.first;
        c.height = scene.resolution.second;
        c.supersampling = scene.supersampling ? 1 : 0;
        return c;
    }
};

// Image U RGB Double: h x w interleaved RGB f64, linear light.
struct Image {
    int width = 0, height = 0;
    std::vector<double> rgb;
    double &at(int y, int x, int c) { return rgb[((size_t)y * width + x) * 3 + c]; }
};

using Star = bs_star;  // (V3 position, (mag, hue, sat)) after starColor'

// readMapFromFile (+ starColor'): PPM catalogue file -> stars
inline std::vector<Star> readMapFromFile(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> buf;
    unsigned char tmp[1 << 16];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    long cnt = bs_read_ppm(buf.data(), buf.size(), nullptr, 0);
    if (cnt < 0) throw std::runtime_error("too few bytes");
    std::vector<Star> stars((size_t)cnt);
    bs_read_ppm(buf.data(), buf.size(), stars.data(), stars.size());
    return stars;
}

// The StarTree argument of render: the star set resident on one GPU (built once, reused for every scene).
class StarTree {
public:
    explicit StarTree(const std::vector<Star> &stars, int device = 0) : ctx_(nullptr)
    {
        if (bs_abi_version() != BS_ABI_VERSION)  // a stale libblackstar_gpu.so under the same name: struct layouts may differ
            throw std::runtime_error("libblackstar_gpu has ABI version " + std::to_string(bs_abi_version()) + ", this header is version " +
                                     std::to_string(BS_ABI_VERSION));
        ctx_ = bs_create(device, stars.data(), stars.size());
        if (!ctx_) throw std::runtime_error(std::string("bs_create: ") + bs_last_error());
    }
    // The arithmetic a render of cfg would be traced with (FAST contexts trace stepSize > 0.5 and paths of more than
    // BS_FAST_MAX_EXPECTED_STEPS expected steps per ray in STRICT): bs_effective_mode
    int effectiveMode(const bs_config &c) const { return bs_effective_mode(ctx_, &c); }
    // What this tree's writer did in the last renderToFiles call (bs_files_stats), and the NUMA node of its GPU (bs_numa_node; -1 unknown)
    bs_files_stats_t filesStats() const
    {
        bs_files_stats_t st;
        if (bs_files_stats(ctx_, &st)) throw std::runtime_error(std::string("bs_files_stats: ") + bs_last_error());
        return st;
    }
    int numaNode() const { return bs_numa_node(ctx_); }
    StarTree(const StarTree &) = delete;
    StarTree &operator=(const StarTree &) = delete;
    ~StarTree() { bs_destroy(ctx_); }
    bs_ctx *handle() const { return ctx_; }

private:
    bs_ctx *ctx_;
};

inline StarTree buildStarTree(const std::vector<Star> &stars, int device = 0) { return StarTree(stars, device); }

// render :: Config -> StarTree -> Image U RGB Double
inline Image render(const Config &cfg, const StarTree &tree)
{
    bs_config c = cfg.to_bs_config();
    if (c.width <= 0 || c.height <= 0) throw std::runtime_error("bs_render: resolution must be positive");
    Image img;
    img.width = c.width;
    img.height = c.height;
    img.rgb.resize((size_t)c.width * c.height * 3);
    if (bs_render(tree.handle(), &c, img.rgb.data(), img.rgb.size())) throw std::runtime_error(std::string("bs_render: ") + bs_last_error());
    return img;
}

// Output rows [row0, row1) of render(cfg): one band of a frame split by rows over several StarTrees / GPUs (bs_render_rows).
inline Image renderRows(const Config &cfg, const StarTree &tree, int row0, int row1)
{
    bs_config c = cfg.to_bs_config();
    if (c.width <= 0 || row0 < 0 || row1 > c.height || row0 >= row1) throw std::runtime_error("bs_render_rows: bad row band");
    Image img;
    img.width = c.width;
    img.height = row1 - row0;
    img.rgb.resize((size_t)c.width * img.height * 3);
    if (bs_render_rows(tree.handle(), &c, row0, row1, img.rgb.data(), img.rgb.size()))
        throw std::runtime_error(std::string("bs_render_rows: ") + bs_last_error());
    return img;
}

// The directory batch mode (app/Main.hs:68-77): scene i on trees[i % trees.size()] (one StarTree per GPU), bs_render_batch.
inline std::vector<Image> renderBatch(const std::vector<Config> &cfgs, const std::vector<const StarTree *> &trees)
{
    if (trees.empty()) throw std::runtime_error("bs_render_batch: no StarTree");
    std::vector<bs_config> cs;
    std::vector<Image> imgs(cfgs.size());
    std::vector<double *> outs;
    std::vector<bs_ctx *> ctxs;
    for (const StarTree *t : trees) ctxs.push_back(t->handle());
    for (size_t i = 0; i < cfgs.size(); i++) {
        cs.push_back(cfgs[i].to_bs_config());
        if (cs[i].width <= 0 || cs[i].height <= 0) throw std::runtime_error("bs_render_batch: resolution must be positive");
        imgs[i].width = cs[i].width;
        imgs[i].height = cs[i].height;
        imgs[i].rgb.resize((size_t)cs[i].width * cs[i].height * 3);
        outs.push_back(imgs[i].rgb.data());
    }
    if (!cfgs.empty() && bs_render_batch(ctxs.data(), (int)ctxs.size(), cs.data(), (int)cs.size(), outs.data()))
        throw std::runtime_error(std::string("bs_render_batch: ") + bs_last_error());
    return imgs;
}

// bloom :: Double -> Int -> Image U RGB Double -> IO (Image U RGB Double)
inline Image bloom(double strength, int divider, const Image &img, const StarTree &tree)
{
    Image out = img;
    if (bs_bloom(tree.handle(), img.rgb.data(), out.rgb.data(), img.width, img.height, strength, divider))
        throw std::runtime_error(std::string("bs_bloom: ") + bs_last_error());
    return out;
}

// the pixel map of writeImg: toWord8 . fmap sRGB
inline std::vector<unsigned char> toSRGB8(const Image &img, const StarTree &tree)
{
    std::vector<unsigned char> out(img.rgb.size());
    if (bs_srgb8(tree.handle(), img.rgb.data(), out.data(), out.size())) throw std::runtime_error(std::string("bs_srgb8: ") + bs_last_error());
    return out;
}

// writeImg :: FilePath -> Image U RGB Double -> IO ()  (src/Raytracer.hs:29-32) minus the write: the bytes of the PNG file, its pixel
// map AND its encoder on the device (massiv-io's writeImage spends 0.1-0.25 s of a host core on a 1080p frame)
inline std::vector<unsigned char> encodeImg(const Image &img, const StarTree &tree)
{
    const std::vector<unsigned char> rgb8 = toSRGB8(img, tree);
    size_t cap = 0, n = 0;
    if (bs_png_bound(img.width, img.height, &cap)) throw std::runtime_error(std::string("bs_png_bound: ") + bs_last_error());
    std::vector<unsigned char> file(cap);
    if (bs_encode_png(tree.handle(), rgb8.data(), img.width, img.height, file.data(), cap, &n))
        throw std::runtime_error(std::string("bs_encode_png: ") + bs_last_error());
    file.resize(n);
    return file;
}

// doRender (app/Main.hs:105-123) in one call: render, bloom when scene.bloomStrength /= 0, writeImg's pixel map and file format;
// returns the bytes to write
inline std::vector<unsigned char> renderPng(const Config &cfg, const StarTree &tree)
{
    const bs_config c = cfg.to_bs_config();
    size_t cap = 0, n = 0;
    if (bs_png_bound(c.width, c.height, &cap)) throw std::runtime_error(std::string("bs_png_bound: ") + bs_last_error());
    std::vector<unsigned char> file(cap);
    if (bs_render_png(tree.handle(), &c, cfg.scene.bloomStrength, cfg.scene.bloomDivider, file.data(), cap, &n))
        throw std::runtime_error(std::string("bs_render_png: ") + bs_last_error());
    file.resize(n);
    return file;
}

// doStart's loop over doRender (app/Main.hs:68-77, :105-123) including writeImg's write: scene i is rendered on trees[i % trees.size()]
// and its PNG file written to paths[i] by the library (per tree: a rolling pipeline of frames in flight, its own ring of page-locked file
// buffers and its own writer thread, on the GPU's NUMA node) -- one call for a directory of scenes
inline void renderToFiles(const std::vector<Config> &cfgs, const std::vector<const StarTree *> &trees, const std::vector<std::string> &paths)
{
    if (cfgs.size() != paths.size()) throw std::runtime_error("renderToFiles: one path per scene");
    std::vector<bs_ctx *> ctxs;
    for (const StarTree *t : trees) ctxs.push_back(t->handle());
    std::vector<bs_config> cs;
    std::vector<double> strengths;
    std::vector<int> dividers;
    std::vector<const char *> ps;
    for (size_t i = 0; i < cfgs.size(); i++) {
        cs.push_back(cfgs[i].to_bs_config());
        strengths.push_back(cfgs[i].scene.bloomStrength);
        dividers.push_back(cfgs[i].scene.bloomDivider);
        ps.push_back(paths[i].c_str());
    }
    if (bs_render_png_files(ctxs.data(), (int)ctxs.size(), cs.data(), (int)cs.size(), strengths.data(), dividers.data(), ps.data(), 0))
        throw std::runtime_error(std::string("bs_render_png_files: ") + bs_last_error());
}

}  // namespace blackstar
