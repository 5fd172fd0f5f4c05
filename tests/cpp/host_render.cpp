// Test driver for include/blackstar_gpu.hpp: plain C++ (g++, no HIP headers, no Python, no torch) linked against
// libblackstar_gpu.so.  Usage: host_render CATALOGUE.ppm OUT.f64 [bloom]   -- renders the default-aa camera at
// 96x54 (the golden fixture image_c3_default_aa_96x54) through blackstar::render and dumps raw doubles.
#include <cstdio>
#include <cstring>
#include <string>

#include "blackstar_gpu.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: host_render CATALOGUE.ppm OUT.f64 [bloom]\n"); return 2; }
    try {
        using namespace blackstar;
        Config cfg;  // scenes/default-aa.yaml (reference scenes/default-aa.yaml:1-16) at 96x54
        cfg.camera = Camera{{0, 1, -20}, {2, 0, 0}, {-0.2, 1, 0}, 1.5};
        cfg.scene.resolution = {96, 54};
        cfg.scene.bloomStrength = 0.15;
        cfg.scene.starIntensity = 0.4;
        cfg.scene.starSaturation = 1.5;
        cfg.scene.diskColor = {180.0 / 360, 0.1, 1.05};
        cfg.scene.diskOpacity = 0.95;
        cfg.scene.diskInner = 1.8;
        cfg.scene.diskOuter = 13;
        cfg.scene.supersampling = true;
        StarTree tree = buildStarTree(readMapFromFile(argv[1]), 0);
        bs_set_mode(tree.handle(), BS_MODE_STRICT);
        Image img = render(cfg, tree);
        {   // row bands and the batch entry reproduce the frame bit for bit
            Image top = renderRows(cfg, tree, 0, 19), bottom = renderRows(cfg, tree, 19, 54);
            std::vector<double> cat = top.rgb;
            cat.insert(cat.end(), bottom.rgb.begin(), bottom.rgb.end());
            if (cat != img.rgb) return 6;
            std::vector<Image> two = renderBatch({cfg, cfg}, {&tree});
            if (two.size() != 2 || two[0].rgb != img.rgb || two[1].rgb != img.rgb) return 7;
        }
        if (argc > 3 && !std::strcmp(argv[3], "bloom")) {
            img = bloom(cfg.scene.bloomStrength, cfg.scene.bloomDivider, img, tree);
            // writeImg's file both ways: encodeImg of the bloomed image = renderPng of the scene (OUT.f64.png: checked by the caller)
            const std::vector<unsigned char> file = encodeImg(img, tree);
            if (file != renderPng(cfg, tree)) return 8;
            FILE *g = std::fopen((std::string(argv[2]) + ".png").c_str(), "wb");
            if (!g) return 3;
            std::fwrite(file.data(), 1, file.size(), g);
            std::fclose(g);
            // ... and the batch loop that writes the files itself: three scenes -> OUT.f64.0.png .. .2.png, each the same bytes
            std::vector<std::string> paths;
            for (int k = 0; k < 3; k++) paths.push_back(std::string(argv[2]) + "." + std::to_string(k) + ".png");
            renderToFiles({cfg, cfg, cfg}, {&tree}, paths);
            {   // the context's own writer wrote them (bs_files_stats): three files, their bytes, from a ring of at least four buffers
                const bs_files_stats_t st = tree.filesStats();
                if (st.files != 3 || st.bytes != 3 * file.size() || st.writer_threads != 1 || st.ring < 4 || st.numa_node_gpu != tree.numaNode()) return 11;
            }
            for (const std::string &p : paths) {
                FILE *r = std::fopen(p.c_str(), "rb");
                if (!r) return 9;
                std::vector<unsigned char> back(file.size() + 1);
                const size_t n = std::fread(back.data(), 1, back.size(), r);
                std::fclose(r);
                back.resize(n);
                if (back != file) return 10;
            }
        }
        FILE *f = std::fopen(argv[2], "wb");
        if (!f) return 3;
        std::fwrite(img.rgb.data(), sizeof(double), img.rgb.size(), f);
        std::fclose(f);
        // error behaviour: a bad hue is rejected with the reference's message, not rendered
        cfg.scene.diskColor[0] = 1.0;
        try { render(cfg, tree); return 4; } catch (const std::runtime_error &e) { if (!std::strstr(e.what(), "not properly scaled")) return 5; }
        std::printf("ok %dx%d\n", img.width, img.height);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
