/*
 * oracle/voxel_pool_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C sequential restatement of the reference voxel-pooling CUDA kernel
 *   /root/reference/open_loop_training/ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-36
 * (the reference has NO CPU implementation of this op: voxel_pooling_forward.cpp:10-11
 * CHECK_CUDA).  The .cu cannot be compiled in this image (needs nvcc / the CUDA runtime
 * headers it relies on implicitly); see DESIGN.md "oracle".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Semantics restated, line by line:
 *   cu:15-17  one work item per point, pt_idx in [0, B*Np)
 *   cu:20     batch_idx = pt_idx / num_points
 *   cu:21-23  x,y,z = geom_xyz[pt_idx*3 + {0,1,2}]
 *   cu:25-27  skip if any coordinate is outside [0, num_voxel_*)
 *   cu:28-30  pos_memo[pt_idx*3 + {0,1,2}] = (batch_idx, y, x)
 *   cu:31-35  for every channel: out[((b*Y + y)*X + x)*C + c] += in[pt_idx*C + c]
 * The CUDA kernel's sum order is nondeterministic (fp32 atomics); this restatement sums in
 * point order, accumulating in double when acc64 != 0 so the oracle is the better-rounded side.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_voxel_pool_fwd(int batch_size, int num_points, int num_channels, int num_voxel_x,
                          int num_voxel_y, int num_voxel_z, const int32_t* geom_xyz,
                          const float* input_features, float* output_features, int32_t* pos_memo,
                          int acc64) {
    const long long total = (long long)batch_size * num_points;
    const long long ncell = (long long)batch_size * num_voxel_y * num_voxel_x;
    double* acc = NULL;
    if (acc64) {
        acc = (double*)calloc((size_t)(ncell * num_channels), sizeof(double));
        if (!acc) return -1;
        for (long long i = 0; i < ncell * num_channels; ++i) acc[i] = output_features[i];
    }
    for (long long pt_idx = 0; pt_idx < total; ++pt_idx) {
        const int batch_idx = (int)(pt_idx / num_points);
        const int x = geom_xyz[pt_idx * 3];
        const int y = geom_xyz[pt_idx * 3 + 1];
        const int z = geom_xyz[pt_idx * 3 + 2];
        if (x < 0 || x >= num_voxel_x || y < 0 || y >= num_voxel_y || z < 0 || z >= num_voxel_z)
            continue;
        if (pos_memo) {
            pos_memo[pt_idx * 3] = batch_idx;
            pos_memo[pt_idx * 3 + 1] = y;
            pos_memo[pt_idx * 3 + 2] = x;
        }
        const long long o =
            ((long long)batch_idx * num_voxel_y * num_voxel_x + (long long)y * num_voxel_x + x) *
            num_channels;
        const float* src = input_features + pt_idx * num_channels;
        if (acc64) {
            for (int c = 0; c < num_channels; ++c) acc[o + c] += (double)src[c];
        } else {
            for (int c = 0; c < num_channels; ++c) output_features[o + c] += src[c];
        }
    }
    if (acc64) {
        for (long long i = 0; i < ncell * num_channels; ++i) output_features[i] = (float)acc[i];
        free(acc);
    }
    return 1; /* the reference wrapper returns 1 (voxel_pooling_forward.cpp:36) */
}

/* VoxelPooling.backward (ops/voxel_pooling/voxel_pooling.py:57-69): gather. grad_out is [B,Y,X,C]. */
int oracle_voxel_pool_bwd(int batch_size, int num_points, int num_channels, int num_voxel_x,
                          int num_voxel_y, const int32_t* pos_memo, const float* grad_out,
                          float* grad_in) {
    const long long total = (long long)batch_size * num_points;
    for (long long p = 0; p < total; ++p) {
        float* dst = grad_in + p * num_channels;
        const int b = pos_memo[p * 3];
        if (b == -1) {
            memset(dst, 0, sizeof(float) * (size_t)num_channels);
            continue;
        }
        const int y = pos_memo[p * 3 + 1], x = pos_memo[p * 3 + 2];
        const float* src =
            grad_out + (((long long)b * num_voxel_y + y) * num_voxel_x + x) * num_channels;
        memcpy(dst, src, sizeof(float) * (size_t)num_channels);
    }
    return 1;
}
