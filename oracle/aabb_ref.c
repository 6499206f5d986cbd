/* aabb_ref.c — plain-C restatement of the reference's two box-test kernels. TEST INFRASTRUCTURE
 * ONLY (parity oracle for lidf_ray_aabb_* / lidf_pcl_aabb_*); never linked into the product.
 *
 * Follows extensions/ray_aabb/ray_aabb_cuda_kernel.cu:24-88 and
 * extensions/pcl_aabb/pcl_aabb_cuda_kernel.cu:23-44 statement by statement (one CUDA thread =
 * one (voxel, ray) iteration of the loops below). Pinned by source: both .cu files are in the
 * reference tree; there is no nvcc here, so the reference kernels themselves cannot be run.
 * Build: make -C oracle  (gcc -O2 -ffp-contract=off: no fused multiply-add may be introduced).
 */
#include <math.h>
#include <stdint.h>

void ray_aabb_ref(const float* ray_dir, const float* voxel_bound, const int32_t* ray_bid,
                  const int32_t* voxel_bid, int64_t ray_num, int64_t voxel_num, int32_t* mask,
                  float* dist) {
    for (int64_t voxel_idx = 0; voxel_idx < voxel_num; ++voxel_idx) {
        for (int64_t ray_idx = 0; ray_idx < ray_num; ++ray_idx) {
            if (ray_bid[ray_idx] != voxel_bid[voxel_idx]) continue;
            float tmin_max, tmax_min;
            float bmin, bmax, tmin, tmax;
            /* x: `1 / (dir + 1e-12)` is evaluated in double and rounded to float (cu:32) */
            float inv = 1 / (ray_dir[ray_idx * 3 + 0] + 1e-12);
            if (inv >= 0) { bmin = voxel_bound[voxel_idx * 6 + 0]; bmax = voxel_bound[voxel_idx * 6 + 3]; }
            else          { bmin = voxel_bound[voxel_idx * 6 + 3]; bmax = voxel_bound[voxel_idx * 6 + 0]; }
            tmin_max = bmin * inv;
            tmax_min = bmax * inv;
            /* y */
            inv = 1 / (ray_dir[ray_idx * 3 + 1] + 1e-12);
            if (inv >= 0) { bmin = voxel_bound[voxel_idx * 6 + 1]; bmax = voxel_bound[voxel_idx * 6 + 4]; }
            else          { bmin = voxel_bound[voxel_idx * 6 + 4]; bmax = voxel_bound[voxel_idx * 6 + 1]; }
            tmin = bmin * inv;
            tmax = bmax * inv;
            if ((tmin_max > tmax) || (tmax_min < tmin)) continue;
            tmin_max = fmaxf(tmin_max, tmin);
            tmax_min = fminf(tmax_min, tmax);
            /* z */
            inv = 1 / (ray_dir[ray_idx * 3 + 2] + 1e-12);
            if (inv >= 0) { bmin = voxel_bound[voxel_idx * 6 + 2]; bmax = voxel_bound[voxel_idx * 6 + 5]; }
            else          { bmin = voxel_bound[voxel_idx * 6 + 5]; bmax = voxel_bound[voxel_idx * 6 + 2]; }
            tmin = bmin * inv;
            tmax = bmax * inv;
            if ((tmin_max > tmax) || (tmax_min < tmin)) continue;
            tmin_max = fmaxf(tmin_max, tmin);
            tmax_min = fminf(tmax_min, tmax);
            mask[voxel_idx * ray_num + ray_idx] = 1;
            dist[voxel_idx * ray_num * 2 + ray_idx * 2 + 0] = tmin_max;
            dist[voxel_idx * ray_num * 2 + ray_idx * 2 + 1] = tmax_min;
        }
    }
}

void pcl_aabb_ref(const float* pcl_pos, const float* voxel_bound, const int32_t* pcl_bid,
                  const int32_t* voxel_bid, int64_t pcl_num, int64_t voxel_num, int32_t* mask) {
    for (int64_t v = 0; v < voxel_num; ++v) {
        for (int64_t i = 0; i < pcl_num; ++i) {
            if (pcl_bid[i] != voxel_bid[v]) continue;
            float x = pcl_pos[i * 3 + 0];
            if ((x < voxel_bound[v * 6 + 0]) || (x > voxel_bound[v * 6 + 3])) continue;
            float y = pcl_pos[i * 3 + 1];
            if ((y < voxel_bound[v * 6 + 1]) || (y > voxel_bound[v * 6 + 4])) continue;
            float z = pcl_pos[i * 3 + 2];
            if ((z < voxel_bound[v * 6 + 2]) || (z > voxel_bound[v * 6 + 5])) continue;
            mask[v * pcl_num + i] = 1;
        }
    }
}
