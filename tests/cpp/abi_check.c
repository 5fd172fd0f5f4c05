/* include/blackstar_gpu.h must be a valid C99 header (the Haskell FFI binds a C ABI): compile with gcc -std=c99 -pedantic,
 * check struct layouts against the byte offsets INTEGRATION.md's shim pokes, call the host-only entry points. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "blackstar_gpu_debug.h" /* (includes blackstar_gpu.h; the hooks header must be C99 too -- only its struct is used here) */

int main(void)
{
    double rgb[3];
    bs_star s[2];
    unsigned char cat[28 + 28];
    if (sizeof(bs_config) != 168 || offsetof(bs_config, width) != 152 || offsetof(bs_config, fov) != 72 ||
        offsetof(bs_config, disk_hsi) != 104 || offsetof(bs_config, disk_opacity) != 128)
        return 10;
    if (sizeof(bs_star) != 48 || offsetof(bs_star, mag) != 40) return 11;
    if (sizeof(bs_ray_record) != 96) return 12;
    if (sizeof(bs_stats_t) != 88 || offsetof(bs_stats_t, kernel_ms) != 64 || offsetof(bs_stats_t, effective_mode) != 80 ||
        offsetof(bs_stats_t, zero_copy) != 84)
        return 19;
    if (sizeof(bs_files_stats_t) != 64 || offsetof(bs_files_stats_t, wall_ms) != 16 || offsetof(bs_files_stats_t, ring) != 40 ||
        offsetof(bs_files_stats_t, threads_bound) != 56)
        return 24;
    if (bs_numa_node(NULL) != -1 || bs_host_page_node(NULL) != -1 || bs_files_stats(NULL, NULL) != BS_EINVAL) return 25;
    if (BS_ABI_VERSION != 5) return 20;
    if (bs_abi_version() != BS_ABI_VERSION) return 13;
    if (bs_hsi_to_rgb(0.5, 0.1, 1.05, rgb) != BS_OK || rgb[0] < 0.944 || rgb[0] > 0.946) return 14;
    if (bs_hsi_to_rgb(1.0, 0.1, 1.05, rgb) != BS_EINVAL) return 15;
    memset(cat, 0, sizeof cat);
    cat[28 + 16] = 'G';
    if (bs_read_ppm(cat, sizeof cat, s, 2) != 1 || s[0].hue != 0.089 || s[0].x != 1.0) return 16; /* ra = dec = 0 -> (1,0,0) */
    if (bs_read_ppm(cat, 10, s, 2) != BS_EINVAL) return 17;
    if (bs_create(-1, NULL, 0) != NULL || strstr(bs_last_error(), "no CPU backend") == NULL) return 18;
    {   /* bs_validate_config: host-only; what every render entry point checks before touching the GPU */
        bs_config c;
        memset(&c, 0, sizeof c);
        c.cam_pos[2] = -20; c.cam_up[1] = 1; c.fov = 1.5; c.step_size = 0.3; c.disk_inner = 3; c.disk_outer = 12; c.width = 8; c.height = 8;
        if (bs_validate_config(&c) != BS_OK) return 21;
        c.step_size = 0;
        if (bs_validate_config(&c) != BS_EINVAL || strstr(bs_last_error(), "stepSize") == NULL) return 22;
        c.step_size = 0.3; c.cam_pos[0] = 0.0 / c.cam_pos[0]; /* 0/0 = NaN without <math.h> */
        if (bs_validate_config(&c) != BS_EINVAL || strstr(bs_last_error(), "camera.position") == NULL) return 23;
    }
    printf("abi ok\n");
    return 0;
}
