// LiDAR branch kernels: hard voxelisation + mean VFE + sparse 3-D convolution.
//
// Replaces, for LidarNet.forward (backbones/lidarnet.py:87-96):
//   mmcv Voxelization (hard, deterministic)  -> tt_lidar_voxelize   (stable radix sort by voxel id,
//                                               run heads by scan, first <=max_points points per voxel
//                                               IN POINT ORDER, mean over them = HardSimpleVFE)
//   spconv SubMConv3d / SparseConv3d          -> tt_sp_hash_build / tt_sp_strided_outputs /
//                                               tt_sp_rulebook, then tt_conv2d_fwd in gather mode
//                                               (MFMA gathered GEMM, fused BN1d + ReLU + residual)
//   SparseConvTensor.dense() + view           -> tt_sp_to_dense (channel index c*D + z, channel-last)
// No host synchronisation: active-row counts stay in device memory; kernels are launched over an
// upper bound and exit early.
#include <hipcub/hipcub.hpp>

#include "tt_common.h"

namespace tt {

constexpr unsigned long long kInvalidKey = ~0ull;
constexpr unsigned kEmpty = 0xFFFFFFFFu;

struct Dims3 { int z, y, x; };

__global__ void lidar_keys_kernel(const float* __restrict__ pts, int B, int Np, int nfeat, float x0, float y0,
                                  float z0, float vx, float vy, float vz, int gx, int gy, int gz, int zlimit,
                                  unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Np) return;
    const int b = (int)(i / Np);
    const float* p = pts + i * nfeat;
    // mmcv hard voxelize: c = floor((p - range_min) / voxel_size); drop if outside the grid
    const int cx = (int)floorf((p[0] - x0) / vx);
    const int cy = (int)floorf((p[1] - y0) / vy);
    const int cz = (int)floorf((p[2] - z0) / vz);
    const bool ok = cx >= 0 && cx < gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz && cz < zlimit;
    keys[i] = ok ? ((((unsigned long long)b * gz + cz) * gy + cy) * gx + cx) : kInvalidKey;
    vals[i] = (unsigned)i;
}

__global__ void mark_heads_kernel(const unsigned long long* __restrict__ keys, long long n, int* __restrict__ flags,
                                  int* __restrict__ n_valid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const bool valid = k != kInvalidKey;
    flags[i] = (valid && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
    if (valid && (i == n - 1 || keys[i + 1] == kInvalidKey)) *n_valid = (int)(i + 1);
}

__global__ void voxel_heads_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ flags,
                                   const int* __restrict__ scan, long long n, int gx, int gy, int gz,
                                   int* __restrict__ coords, int* __restrict__ first, int* __restrict__ num_voxels) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) {
        const int v = scan[i];
        unsigned long long k = keys[i];
        const int x = (int)(k % gx); k /= gx;
        const int y = (int)(k % gy); k /= gy;
        const int z = (int)(k % gz); k /= gz;
        coords[v * 4 + 0] = (int)k;
        coords[v * 4 + 1] = z;
        coords[v * 4 + 2] = y;
        coords[v * 4 + 3] = x;
        first[v] = (int)i;
    }
    if (i == n - 1) *num_voxels = scan[i] + flags[i];
}

__global__ void vfe_mean_kernel(const float* __restrict__ pts, const unsigned* __restrict__ vals,
                                const int* __restrict__ first, const int* __restrict__ num_voxels,
                                const int* __restrict__ n_valid, int nfeat, int max_points, long long max_rows,
                                float* __restrict__ feats) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long v = t / nfeat;
    const int f = (int)(t % nfeat);
    const int M = *num_voxels;
    if (v >= M || v >= max_rows) return;
    const int s = first[v];
    const int e = (v + 1 < M) ? first[v + 1] : *n_valid;
    const int cnt = min(e - s, max_points);
    float sum = 0.f;
    for (int j = 0; j < cnt; ++j) sum += pts[(long long)vals[s + j] * nfeat + f];
    feats[v * nfeat + f] = sum / (float)cnt;     // HardSimpleVFE: sum / num_points
}

// ----------------------------------------------------------------------------- hash of active sites
__device__ __forceinline__ unsigned hash_u32(unsigned k) {
    k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
    return k;
}

__device__ __forceinline__ unsigned lin_key(int b, int z, int y, int x, Dims3 d) {
    return (unsigned)((((long long)b * d.z + z) * d.y + y) * d.x + x);
}

__device__ __forceinline__ int hash_find(const unsigned* __restrict__ hk, const int* __restrict__ hv, unsigned mask,
                                         unsigned key) {
    unsigned s = hash_u32(key) & mask;
    while (true) {
        const unsigned k = hk[s];
        if (k == key) return hv[s];
        if (k == kEmpty) return -1;
        s = (s + 1) & mask;
    }
}

__global__ void hash_build_kernel(const int* __restrict__ coords, const int* __restrict__ num_rows, long long max_rows,
                                  Dims3 d, unsigned* __restrict__ hk, int* __restrict__ hv, unsigned mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *num_rows || i >= max_rows) return;
    const unsigned key = lin_key(coords[i * 4], coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3], d);
    unsigned s = hash_u32(key) & mask;
    while (true) {
        const unsigned old = atomicCAS(&hk[s], kEmpty, key);
        if (old == kEmpty || old == key) { hv[s] = (int)i; return; }
        s = (s + 1) & mask;
    }
}

struct ConvGeom { int kz, ky, kx, sz, sy, sx, pz, py, px; };

// SparseConv3d active outputs: o is active iff some input i and tap k satisfy i = o*s - p + k
__global__ void strided_outputs_kernel(const int* __restrict__ in_coords, const int* __restrict__ in_rows,
                                       long long max_in, ConvGeom g, Dims3 od, unsigned* __restrict__ hk,
                                       int* __restrict__ hv, unsigned mask, int* __restrict__ out_coords,
                                       int* __restrict__ out_rows, long long max_out) {
    const int KV = g.kz * g.ky * g.kx;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = t / KV;
    const int k = (int)(t % KV);
    if (i >= *in_rows || i >= max_in) return;
    const int kz = k / (g.ky * g.kx), ky = (k / g.kx) % g.ky, kx = k % g.kx;
    const int nz = in_coords[i * 4 + 1] + g.pz - kz;
    const int ny = in_coords[i * 4 + 2] + g.py - ky;
    const int nx = in_coords[i * 4 + 3] + g.px - kx;
    if (nz < 0 || ny < 0 || nx < 0 || nz % g.sz || ny % g.sy || nx % g.sx) return;
    const int oz = nz / g.sz, oy = ny / g.sy, ox = nx / g.sx;
    if (oz >= od.z || oy >= od.y || ox >= od.x) return;
    const int b = in_coords[i * 4];
    const unsigned key = lin_key(b, oz, oy, ox, od);
    unsigned s = hash_u32(key) & mask;
    while (true) {
        // ~27 candidates map to each output site: probe with a plain load first so that only the
        // first arrival pays for an atomic (a stale miss just falls through to the CAS)
        const unsigned cur = __hip_atomic_load(&hk[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return;
        if (cur != kEmpty) { s = (s + 1) & mask; continue; }
        const unsigned old = atomicCAS(&hk[s], kEmpty, key);
        if (old == key) return;
        if (old == kEmpty) {
            const int row = atomicAdd(out_rows, 1);
            if (row < max_out) {
                hv[s] = row;
                out_coords[row * 4 + 0] = b;
                out_coords[row * 4 + 1] = oz;
                out_coords[row * 4 + 2] = oy;
                out_coords[row * 4 + 3] = ox;
            }
            return;
        }
        s = (s + 1) & mask;
    }
}

// rulebook: nbr[o][k] = input row at (o*s - p + k) or -1
__global__ void rulebook_kernel(const int* __restrict__ out_coords, const int* __restrict__ out_rows, long long max_out,
                                ConvGeom g, Dims3 id, const unsigned* __restrict__ hk, const int* __restrict__ hv,
                                unsigned mask, int* __restrict__ nbr) {
    const int KV = g.kz * g.ky * g.kx;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long o = t / KV;
    const int k = (int)(t % KV);
    if (o >= *out_rows || o >= max_out) return;
    const int kz = k / (g.ky * g.kx), ky = (k / g.kx) % g.ky, kx = k % g.kx;
    const int z = out_coords[o * 4 + 1] * g.sz - g.pz + kz;
    const int y = out_coords[o * 4 + 2] * g.sy - g.py + ky;
    const int x = out_coords[o * 4 + 3] * g.sx - g.px + kx;
    int r = -1;
    if (z >= 0 && z < id.z && y >= 0 && y < id.y && x >= 0 && x < id.x)
        r = hash_find(hk, hv, mask, lin_key(out_coords[o * 4], z, y, x, id));
    nbr[o * KV + k] = r;
}

template <typename T>
__global__ void sp_to_dense_kernel(const T* __restrict__ f, const int* __restrict__ coords,
                                   const int* __restrict__ rows_n, long long max_rows, int C, Dims3 d,
                                   T* __restrict__ dense) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = t / C;
    const int c = (int)(t % C);
    if (r >= *rows_n || r >= max_rows) return;
    const int b = coords[r * 4], z = coords[r * 4 + 1], y = coords[r * 4 + 2], x = coords[r * 4 + 3];
    dense[(((long long)b * d.y + y) * d.x + x) * ((long long)C * d.z) + (long long)c * d.z + z] = f[r * C + c];
}

}  // namespace tt

using namespace tt;

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

extern "C" long long tt_lidar_voxelize_workspace_bytes(long long num_points_total) {
    size_t sort_bytes = 0, scan_bytes = 0;
    const int n = (int)num_points_total;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (unsigned long long*)nullptr,
                                       (unsigned long long*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, n);
    size_t tot = 0;
    tot += 2 * align256(sizeof(unsigned long long) * n);   // keys in/out
    tot += 2 * align256(sizeof(unsigned) * n);             // vals in/out
    tot += 2 * align256(sizeof(int) * n);                  // flags, scan
    tot += align256(sizeof(int) * (n + 1));                // first
    tot += align256(sizeof(int) * 4);                      // n_valid
    tot += align256(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    return (long long)tot;
}

extern "C" int tt_lidar_voxelize(const float* points, int B, int Np, int nfeat, const float* pc_range_lo,
                                 const float* voxel_size, const int* grid_xyz, int z_limit, int max_points,
                                 void* workspace, long long workspace_bytes, float* voxel_feats, int* coords,
                                 int* num_voxels, void* stream) {
    TT_REQUIRE(points && pc_range_lo && voxel_size && grid_xyz && workspace && voxel_feats && coords && num_voxels,
               "tt_lidar_voxelize: null");
    const long long n = (long long)B * Np;
    TT_REQUIRE(n > 0 && n < (1ll << 30), "tt_lidar_voxelize: bad point count");
    TT_REQUIRE(workspace_bytes >= tt_lidar_voxelize_workspace_bytes(n), "tt_lidar_voxelize: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    auto take = [&](size_t bytes) { char* p = w; w += align256(bytes); return p; };
    auto* keys_in = (unsigned long long*)take(sizeof(unsigned long long) * n);
    auto* keys_out = (unsigned long long*)take(sizeof(unsigned long long) * n);
    auto* vals_in = (unsigned*)take(sizeof(unsigned) * n);
    auto* vals_out = (unsigned*)take(sizeof(unsigned) * n);
    auto* flags = (int*)take(sizeof(int) * n);
    auto* scan = (int*)take(sizeof(int) * n);
    auto* first = (int*)take(sizeof(int) * (n + 1));
    auto* n_valid = (int*)take(sizeof(int) * 4);
    void* tmp = w;
    size_t tmp_bytes = (size_t)(workspace_bytes - (w - (char*)workspace));
    (void)hipMemsetAsync(n_valid, 0, sizeof(int), st);
    (void)hipMemsetAsync(num_voxels, 0, sizeof(int), st);
    const unsigned blocks = (unsigned)div_up(n, 256);
    hipLaunchKernelGGL(lidar_keys_kernel, dim3(blocks), dim3(256), 0, st, points, B, Np, nfeat, pc_range_lo[0],
                       pc_range_lo[1], pc_range_lo[2], voxel_size[0], voxel_size[1], voxel_size[2], grid_xyz[0],
                       grid_xyz[1], grid_xyz[2], z_limit, keys_in, vals_in);
    size_t sb = tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(tmp, sb, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 64, st) !=
        hipSuccess) {
        set_error("tt_lidar_voxelize: radix sort failed");
        return -2;
    }
    hipLaunchKernelGGL(mark_heads_kernel, dim3(blocks), dim3(256), 0, st, keys_out, n, flags, n_valid);
    sb = tmp_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(tmp, sb, flags, scan, (int)n, st) != hipSuccess) {
        set_error("tt_lidar_voxelize: scan failed");
        return -2;
    }
    hipLaunchKernelGGL(voxel_heads_kernel, dim3(blocks), dim3(256), 0, st, keys_out, flags, scan, n, grid_xyz[0],
                       grid_xyz[1], grid_xyz[2], coords, first, num_voxels);
    hipLaunchKernelGGL(vfe_mean_kernel, dim3((unsigned)div_up(n * nfeat, 256)), dim3(256), 0, st, points, vals_out,
                       first, num_voxels, n_valid, nfeat, max_points, n, voxel_feats);
    return check_launch("tt_lidar_voxelize");
}

extern "C" int tt_sp_hash_build(const int* coords, const int* num_rows, long long max_rows, const int* dims_zyx,
                                unsigned* hash_keys, int* hash_vals, long long hash_size, void* stream) {
    TT_REQUIRE(coords && num_rows && dims_zyx && hash_keys && hash_vals, "tt_sp_hash_build: null");
    TT_REQUIRE(hash_size > 0 && (hash_size & (hash_size - 1)) == 0, "tt_sp_hash_build: hash_size must be 2^k");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(hash_keys, 0xFF, sizeof(unsigned) * hash_size, st);
    Dims3 d{dims_zyx[0], dims_zyx[1], dims_zyx[2]};
    hipLaunchKernelGGL(hash_build_kernel, dim3((unsigned)div_up(max_rows, 256)), dim3(256), 0, st, coords, num_rows,
                       max_rows, d, hash_keys, hash_vals, (unsigned)(hash_size - 1));
    return check_launch("tt_sp_hash_build");
}

static ConvGeom geom_of(const int* g) { return ConvGeom{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8]}; }

extern "C" int tt_sp_strided_outputs(const int* in_coords, const int* in_rows, long long max_in,
                                     const int* kernel_stride_pad, const int* out_dims_zyx, unsigned* hash_keys,
                                     int* hash_vals, long long hash_size, int* out_coords, int* out_rows,
                                     long long max_out, void* stream) {
    TT_REQUIRE(in_coords && in_rows && kernel_stride_pad && out_dims_zyx && hash_keys && hash_vals && out_coords &&
                   out_rows, "tt_sp_strided_outputs: null");
    TT_REQUIRE(hash_size > 0 && (hash_size & (hash_size - 1)) == 0, "tt_sp_strided_outputs: hash_size must be 2^k");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(hash_keys, 0xFF, sizeof(unsigned) * hash_size, st);
    (void)hipMemsetAsync(out_rows, 0, sizeof(int), st);
    ConvGeom g = geom_of(kernel_stride_pad);
    Dims3 od{out_dims_zyx[0], out_dims_zyx[1], out_dims_zyx[2]};
    const long long t = max_in * g.kz * g.ky * g.kx;
    hipLaunchKernelGGL(strided_outputs_kernel, dim3((unsigned)div_up(t, 256)), dim3(256), 0, st, in_coords, in_rows,
                       max_in, g, od, hash_keys, hash_vals, (unsigned)(hash_size - 1), out_coords, out_rows, max_out);
    return check_launch("tt_sp_strided_outputs");
}

extern "C" int tt_sp_rulebook(const int* out_coords, const int* out_rows, long long max_out,
                              const int* kernel_stride_pad, const int* in_dims_zyx, const unsigned* hash_keys,
                              const int* hash_vals, long long hash_size, int* nbr, void* stream) {
    TT_REQUIRE(out_coords && out_rows && kernel_stride_pad && in_dims_zyx && hash_keys && hash_vals && nbr,
               "tt_sp_rulebook: null");
    ConvGeom g = geom_of(kernel_stride_pad);
    Dims3 id{in_dims_zyx[0], in_dims_zyx[1], in_dims_zyx[2]};
    const long long t = max_out * g.kz * g.ky * g.kx;
    hipLaunchKernelGGL(rulebook_kernel, dim3((unsigned)div_up(t, 256)), dim3(256), 0, (hipStream_t)stream, out_coords,
                       out_rows, max_out, g, id, hash_keys, hash_vals, (unsigned)(hash_size - 1), nbr);
    return check_launch("tt_sp_rulebook");
}

extern "C" int tt_sp_to_dense(const void* feats, const int* coords, const int* num_rows, long long max_rows, int C,
                              const int* dims_zyx, void* dense, int dtype, void* stream) {
    TT_REQUIRE(feats && coords && num_rows && dims_zyx && dense, "tt_sp_to_dense: null");
    Dims3 d{dims_zyx[0], dims_zyx[1], dims_zyx[2]};
    const dim3 grid((unsigned)div_up(max_rows * C, 256));
    if (dtype == TT_F32)
        hipLaunchKernelGGL(sp_to_dense_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)feats,
                           coords, num_rows, max_rows, C, d, (float*)dense);
    else
        hipLaunchKernelGGL(sp_to_dense_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)feats, coords, num_rows, max_rows, C, d, (uint16_t*)dense);
    return check_launch("tt_sp_to_dense");
}
