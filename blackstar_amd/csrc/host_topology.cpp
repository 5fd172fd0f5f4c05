// host_topology.cpp -- where a context's GPU sits in the HOST: its NUMA node (sysfs), that node's CPUs, thread affinity for the host threads
// that drive the context, and the node a page of host memory lives on.  Linux only, no libnuma: sysfs + sched_setaffinity + move_pages(2).
// The reference has one backend -- the host's own cores (blackstar.cabal:47) -- and no such concern; with 8 GPUs under two sockets a
// context's host thread, its file writer and its page-locked buffers belong on the socket the GPU hangs off (app/Main.hs:68-77 is the
// loop those threads replace).
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "bs_context.h"

namespace bs {

namespace {

bool read_line(const std::string &path, std::string *out)
{
    std::ifstream f(path);
    if (!f) return false;
    std::getline(f, *out);
    return true;
}

// "0-15,64-79" -> cpu set; false on anything else
bool parse_cpulist(const std::string &s, cpu_set_t *set)
{
    CPU_ZERO(set);
    const char *p = s.c_str();
    bool any = false;
    while (*p) {
        while (*p == ',' || std::isspace((unsigned char)*p)) p++;
        if (!*p) break;
        char *end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            b = std::strtol(p + 1, &end, 10);
            if (end == p + 1 || b < a) return false;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET((int)c, set);
            any = true;
        }
    }
    return any;
}

}  // namespace

// Fills ctx->numa_node (-1: unknown, or a host without NUMA information for the device) and ctx->numa_cpus (the node's CPUs this process may
// run on; empty when unknown).  Called once, from bs_create, after the device is known.  Never fails: no information = no binding.
void probe_host_topology(bs_ctx *ctx)
{
    ctx->numa_node = -1;
    ctx->numa_bind = false;
    ctx->numa_confined = false;
    CPU_ZERO(&ctx->numa_cpus);
    if (const char *m = std::getenv("BLACKSTAR_NUMA_BIND"))
        if (std::atoi(m) == 0) return;   // A/B switch: leave every thread where the caller's scheduler puts it
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (char *c = bdf; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    std::string node_s, list;
    if (!read_line(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node", &node_s)) return;
    const int node = std::atoi(node_s.c_str());
    if (node < 0) return;   // (-1: a single-node host, or firmware that does not say)
    if (!read_line("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", &list)) return;
    cpu_set_t node_cpus, allowed;
    if (!parse_cpulist(list, &node_cpus)) return;
    ctx->numa_node = node;
    // only CPUs this process is allowed on (a container's cpuset may exclude the node altogether: then nothing is bound)
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    CPU_AND(&ctx->numa_cpus, &node_cpus, &allowed);
    ctx->numa_confined = CPU_COUNT(&ctx->numa_cpus) > 0 && CPU_EQUAL(&ctx->numa_cpus, &allowed);   // the process may only run on the node anyway (e.g. bound by its launcher)
    ctx->numa_bind = CPU_COUNT(&ctx->numa_cpus) > 0 && !ctx->numa_confined;
}

// The NUMA node the page at p lives on (move_pages(2) with a null node list only queries), or -1: not resident, not a page the kernel
// accounts per node (device-file mappings), or no NUMA support.
int numa_node_of_page(const void *p)
{
    void *pages[1] = {reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(4095))};
    int status[1] = {-1};
#ifdef SYS_move_pages
    if (syscall(SYS_move_pages, 0, 1UL, pages, nullptr, status, 0) != 0) return -1;
#endif
    return status[0] >= 0 ? status[0] : -1;
}

NumaBind::NumaBind(const bs_ctx *ctx)
{
    if (!ctx || !ctx->numa_bind) return;
    if (sched_getaffinity(0, sizeof saved_, &saved_) != 0) return;   // (0 = the calling THREAD: sched_* act on thread ids)
    bound_ = sched_setaffinity(0, sizeof ctx->numa_cpus, &ctx->numa_cpus) == 0;
}

NumaBind::~NumaBind()
{
    if (bound_) (void)sched_setaffinity(0, sizeof saved_, &saved_);
}

}  // namespace bs
