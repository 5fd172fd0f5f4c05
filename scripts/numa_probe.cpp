// numa_probe.cpp -- what the host of a GPU box looks like to bs_render_png_files' per-context writers: NUMA node of the GPU, its CPUs,
// where hipHostMalloc pages land (move_pages query), one-thread and 8-thread write(2) rates into a directory.
//   hipcc -O2 scripts/numa_probe.cpp -o /tmp/numa_probe -lpthread && /tmp/numa_probe [dir]
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>
#include <fcntl.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <fstream>

static std::string slurp(const std::string &p) { std::ifstream f(p); std::string s; std::getline(f, s); return s; }
static int node_of(void *p) {
    void *pages[1] = {p}; int status[1] = {-99};
    long rc = syscall(SYS_move_pages, 0, 1UL, pages, nullptr, status, 0);
    return rc == 0 ? status[0] : -1000 - (int)rc;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    int n = 0; hipGetDeviceCount(&n);
    printf("devices %d  online cpus %ld  hw_concurrency %u\n", n, sysconf(_SC_NPROCESSORS_ONLN), std::thread::hardware_concurrency());
    cpu_set_t cs; CPU_ZERO(&cs); sched_getaffinity(0, sizeof cs, &cs); printf("allowed cpus %d\n", CPU_COUNT(&cs));
    printf("nodes online: %s  possible: %s\n", slurp("/sys/devices/system/node/online").c_str(), slurp("/sys/devices/system/node/possible").c_str());
    for (int d = 0; d < n; d++) {
        char bdf[64] = {0}; hipDeviceGetPCIBusId(bdf, sizeof bdf, d);
        for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
        std::string node = slurp(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
        printf("device %d bdf %s numa_node '%s' local_cpulist '%s'\n", d, bdf, node.c_str(), slurp(std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist").c_str());
        if (!node.empty() && atoi(node.c_str()) >= 0) printf("  node cpulist %s\n", slurp("/sys/devices/system/node/node" + node + "/cpulist").c_str());
    }
    hipSetDevice(0);
    for (unsigned flags : {0u, (unsigned)hipHostMallocPortable, (unsigned)hipHostMallocNumaUser}) {
        void *p = nullptr; hipError_t e = hipHostMalloc(&p, 8 << 20, flags);
        if (e != hipSuccess) { printf("hipHostMalloc flags %u failed\n", flags); continue; }
        memset(p, 1, 8 << 20);
        printf("hipHostMalloc flags %u: first page node %d, last page node %d\n", flags, node_of(p), node_of((char *)p + (8 << 20) - 4096));
        hipHostFree(p);
    }
    { void *p = malloc(8 << 20); memset(p, 1, 8 << 20); printf("malloc: page node %d (sanity)\n", node_of((char *)(((uintptr_t)p + 4095) & ~4095ul))); free(p); }
    // write rates: files of 2.36 MB like a C3 PNG
    const size_t bytes = 2360000; std::vector<unsigned char> buf(bytes, 0x5a);
    auto writer = [&](int id, int files, double *secs) {
        double t0 = now();
        for (int i = 0; i < files; i++) {
            char path[512]; snprintf(path, sizeof path, "%s/numa_probe_%d_%d.bin", dir, id, i % 16);
            FILE *f = fopen(path, "wb"); if (!f) { perror(path); return; } fwrite(buf.data(), 1, bytes, f); fclose(f);
        }
        *secs = now() - t0;
    };
    for (int threads : {1, 2, 4, 8}) {
        std::vector<std::thread> th; std::vector<double> secs(threads, 0);
        const int files = 200; double t0 = now();
        for (int t = 0; t < threads; t++) th.emplace_back(writer, t, files, &secs[t]);
        for (auto &t : th) t.join();
        double wall = now() - t0;
        printf("%d writer thread(s): %.2f GB/s aggregate, %.0f files/s (dir %s)\n", threads, threads * files * bytes / wall / 1e9, threads * files / wall, dir);
    }
    for (int t = 0; t < 8; t++) for (int i = 0; i < 16; i++) { char path[512]; snprintf(path, sizeof path, "%s/numa_probe_%d_%d.bin", dir, t, i); unlink(path); }
    return 0;
}
