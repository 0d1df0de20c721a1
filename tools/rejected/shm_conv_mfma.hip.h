// Step 1 + 2 (fp64) with the accumulation X += g W on the fp64 matrix cores.
//
// conv_normalize_kernel<double, 4> is bound by vector-ALU issue: 25.8 VALU instructions per (node, source) pair at 92 % VALU-busy
// (profiles/r02_sq_counters_conv.txt).  Three of them are the accumulate FMAs  X_c += g W_c (c = x, y, z) -- the only part of the pair
// evaluation with a matrix shape:  X[node][c] += G[node][source] W[source][c].  v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4) B(4x16)
// with A[i][k] in lane 16 k + i and B[k][j] in lane 16 k + j, so the mapping
//        lane = 16 * (source % 4) + (node column of the wave),      A = g of that (column, source),      B = W[source][j] (j < 3, else 0)
// feeds the matrix pipe straight from the registers the VALU leaves g in: no cross-lane movement, no extra VALU work.  The four
// lane quarters of a wave therefore evaluate four consecutive sources for the same 16 node columns (x 4 nodes in z per lane, which
// still share dx^2 + dy^2), instead of one source for 64 columns; the pair count per wave instruction is unchanged.  Only three of
// the sixteen output columns carry data -- the matrix pipe is otherwise idle in this kernel and runs beside the VALU (4 MFMAs of 64
// cycles against ~380 VALU cycles per step), so its inefficiency costs nothing; what is saved is 12 of the ~107 VALU issue slots of
// a step.  Arithmetic: the same fp64 products and sums as the VALU version, accumulated in a different (fixed) order.
// Tile = 8 x 8 x 4 nodes per workgroup pass (wave w owns rows 2w, 2w+1 of the 8 x 8 columns); everything else -- Morton clusters
// through LDS, skip / far classification, the fp32 branch for clusters below e^-25, the table exponential -- as in
// conv_normalize_kernel (shm_kernels.hip.h), whose comments carry the reference citations (signed_heat_grid_solver.cpp:48-65, :157-174).
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

constexpr int kConvMfmaTileZ = 4;
constexpr int kConvRec = 8;   // doubles per source record in LDS: x y z wx wy wz 0 0 (the zero pad feeds the unused B columns)

__global__ __launch_bounds__(kBlock) void conv_normalize_mfma_kernel(ConvParams P, const double* __restrict__ src /* [S][6] */, const float* __restrict__ src32,
                                                                     const float* __restrict__ clusters, const double* __restrict__ exp_tab_g,
                                                                     double* __restrict__ Y0, double* __restrict__ Y1, double* __restrict__ Y2) {
    constexpr int NPT = kConvMfmaTileZ;
    __shared__ double tile[kSrcTile * kConvRec];   // also the accumulator transpose buffer at the end of a tile
    __shared__ float tile32[kSrcTile * 6];
    __shared__ float red[kBlock / kWave], redw[kBlock / kWave];
    __shared__ double exp_tab[1 << YukawaMath<double>::kExpTabBits];
    for (int a = threadIdx.x; a < (1 << YukawaMath<double>::kExpTabBits); a += kBlock) exp_tab[a] = exp_tab_g[a];
    const int n = P.n;
    const size_t plane = (size_t)n * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, quarter = lane >> 4;
    for (int bt = blockIdx.x; bt < P.n_tiles; bt += gridDim.x) {
        __syncthreads();  // LDS reuse between consecutive tiles of this workgroup
        const int tz = bt / (P.tiles_x * P.tiles_y), trem = bt - tz * (P.tiles_x * P.tiles_y);
        const int ty = trem / P.tiles_x, tx = trem - ty * P.tiles_x;
        const int i0 = tx * kConvTile, j0 = ty * kConvTile, kk0 = P.kk_begin + tz * NPT;
        const int li = i0 + (col & 7), lj = j0 + 2 * wave + (col >> 3);
        const int ci = min(li, n - 1), cj = min(lj, n - 1);
        // indicesToNodePosition: (i,j,k)*cellSize + bboxMin, evaluated in double like the reference
        const double px = ci * P.cell + P.bbox_min[0], py = cj * P.cell + P.bbox_min[1];
        const float qx = (float)px, qy = (float)py;
        double pz[NPT];
        float qz[NPT], fx[NPT], fy[NPT], fz[NPT];
        gj_f64x4 acc[NPT];
#pragma unroll
        for (int e = 0; e < NPT; e++) {
            const int kk = min(kk0 + e, P.kk_end - 1);
            const double z = (P.k0 + kk - 1) * P.cell + P.bbox_min[2];
            pz[e] = z;
            qz[e] = (float)z;
            fx[e] = fy[e] = fz[e] = 0.f;
            acc[e] = gj_f64x4{0., 0., 0., 0.};
        }
        // tile centre / circumscribed radius, then the workgroup-wide minimum distance from the centre to the sources
        constexpr double kHalfZ = 0.5 * (NPT - 1);
        const float cx = (float)((i0 + 3.5) * P.cell + P.bbox_min[0]), cy = (float)((j0 + 3.5) * P.cell + P.bbox_min[1]);
        const float cz = (float)((P.k0 + kk0 - 1 + kHalfZ) * P.cell + P.bbox_min[2]);
        const float rt = (float)(sqrt(3.5 * 3.5 * 2 + kHalfZ * kHalfZ) * P.cell) * 1.000001f;
        float dmin = 3.0e38f, wnear = 0.f;   // nearest source and |A N|^2 of it (ties go to the larger weight)
        for (int s = threadIdx.x; s < P.S; s += kBlock) {
            const float dx = cx - (float)src[(size_t)s * 6], dy = cy - (float)src[(size_t)s * 6 + 1], dz = cz - (float)src[(size_t)s * 6 + 2];
            const float wx = (float)src[(size_t)s * 6 + 3], wy = (float)src[(size_t)s * 6 + 4], wz = (float)src[(size_t)s * 6 + 5];
            const float d2 = dx * dx + dy * dy + dz * dz, w2 = wx * wx + wy * wy + wz * wz;
            if (d2 < dmin || (d2 == dmin && w2 > wnear)) {
                dmin = d2;
                wnear = w2;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(dmin, off, kWave), ow = __shfl_xor(wnear, off, kWave);
            if (od < dmin || (od == dmin && ow > wnear)) {
                dmin = od;
                wnear = ow;
            }
        }
        if (lane == 0) {
            red[wave] = dmin;
            redw[wave] = wnear;
        }
        __syncthreads();
        dmin = red[0];
        wnear = redw[0];
#pragma unroll
        for (int a = 1; a < kBlock / kWave; a++)
            if (red[a] < dmin || (red[a] == dmin && redw[a] > wnear)) {
                dmin = red[a];
                wnear = redw[a];
            }
        dmin = sqrtf(dmin);
        const float ln_anear = 0.5f * __logf(fmaxf(wnear, 1e-37f)) - 1e-5f;   // rounded down
        const float r_hi = dmin * 1.000001f + rt;                   // every node of the tile has a source at most this far
        const float d0t = fmaxf(0.f, dmin * 0.999999f - rt);        // no source is closer than this to any node of the tile
        const float lamf = (float)P.lambda;

        for (int c0 = 0; c0 < P.n_clusters; c0 += kConvChunk) {
            const int ncl = min(kConvChunk, P.n_clusters - c0);
            const int cnt = ncl * kConvCluster;
            float gaps[kConvChunk];
            bool skip[kConvChunk];
            bool any = false;
#pragma unroll
            for (int c = 0; c < kConvChunk; c++) {
                const int cc = min(c0 + c, P.n_clusters - 1);
                const float* rec = clusters + (size_t)cc * kConvClusterRec;
                const float gdx = cx - rec[0], gdy = cy - rec[1], gdz = cz - rec[2];
                gaps[c] = sqrtf(gdx * gdx + gdy * gdy + gdz * gdz) * 0.999999f - rt - rec[3] - r_hi;
                skip[c] = gaps[c] > (P.skip_base + rec[4] - ln_anear) * P.inv_lambda;
                any = any || (c < ncl && !skip[c]);
            }
            if (!any) continue;
            __syncthreads();
            for (int a = threadIdx.x; a < cnt; a += kBlock) {   // one source per thread: 6 values in, an 8-value record out
                const size_t g = ((size_t)c0 * kConvCluster + a) * 6;
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    tile[a * kConvRec + b] = src[g + b];
                    tile32[a * 6 + b] = src32[g + b];
                }
                tile[a * kConvRec + 6] = 0.;
                tile[a * kConvRec + 7] = 0.;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < kConvChunk; c++) {
                if (c >= ncl) break;
                if (skip[c]) continue;
                const bool far = gaps[c] > P.far_gap;
                if (far) {
                    // below e^-25 of the tile's dominant term: scalar fp32 with the tile's exponent offset (see conv_normalize_kernel);
                    // each lane takes the sources of its quarter, the quarters are summed at the end of the tile
#pragma unroll 2
                    for (int t = 0; t < kConvCluster / 4; t++) {
                        const int s = c * kConvCluster + 4 * t + quarter;
                        const float sz = tile32[6 * s + 2];
                        const float wx = tile32[6 * s + 3], wy = tile32[6 * s + 4], wz = tile32[6 * s + 5];
                        const float dx = qx - tile32[6 * s], dy = qy - tile32[6 * s + 1];
                        const float dxy2 = dx * dx + dy * dy;
#pragma unroll
                        for (int e = 0; e < NPT; e++) {
                            const float dz = qz[e] - sz;
                            float r, rinv;
                            YukawaMath<float>::rsqrt_and_sqrt(dxy2 + dz * dz, rinv, r);
                            const float g = YukawaMath<float>::exp_neg(-lamf * (r - d0t)) * rinv;
                            fx[e] += wx * g; fy[e] += wy * g; fz[e] += wz * g;
                        }
                    }
                } else {
#pragma unroll 2
                    for (int t = 0; t < kConvCluster / 4; t++) {
                        const double* rec = tile + (size_t)(c * kConvCluster + 4 * t + quarter) * kConvRec;
                        const double sz = rec[2];
                        const double bw = rec[3 + min(col, 3)];          // B[k = quarter][j = col]: W_j of this quarter's source, 0 beyond j = 2
                        const double dx = px - rec[0], dy = py - rec[1];
                        const double dxy2 = dx * dx + dy * dy;
#pragma unroll
                        for (int e = 0; e < NPT; e++) {
                            const double dz = pz[e] - sz;
                            const double g = YukawaMath<double>::yukawa(dxy2 + dz * dz, P.cexp, exp_tab);   // r = 0 -> NaN, like exp(0)/0
                            acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(g, bw, acc[e], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- end of tile: far-cluster sums over the four quarters, accumulators out of the matrix layout, normalise, store
        const double e0 = YukawaMath<double>::exp_neg(-P.lambda * (double)d0t);
#pragma unroll
        for (int e = 0; e < NPT; e++) {
            fx[e] += __shfl_xor(fx[e], 16, kWave); fy[e] += __shfl_xor(fy[e], 16, kWave); fz[e] += __shfl_xor(fz[e], 16, kWave);
            fx[e] += __shfl_xor(fx[e], 32, kWave); fy[e] += __shfl_xor(fy[e], 32, kWave); fz[e] += __shfl_xor(fz[e], 32, kWave);
        }
        __syncthreads();   // every wave is done reading `tile`
        double* xs = tile + (size_t)wave * (NPT * 16 * 4);   // [e][column i][j], j padded to 4
        if (col < 3) {
            // D[i][j]: lane (j = col, l4 = quarter), register r holds row i = quarter + 4 r
#pragma unroll
            for (int e = 0; e < NPT; e++)
#pragma unroll
                for (int r = 0; r < 4; r++) xs[(e * 16 + quarter + 4 * r) * 4 + col] = acc[e][r];
        }
        __syncthreads();
        {
            // lane -> node (column col, z-node e = quarter)
            const int e = quarter;
            const float ffx = e == 0 ? fx[0] : e == 1 ? fx[1] : e == 2 ? fx[2] : fx[3];
            const float ffy = e == 0 ? fy[0] : e == 1 ? fy[1] : e == 2 ? fy[2] : fy[3];
            const float ffz = e == 0 ? fz[0] : e == 1 ? fz[1] : e == 2 ? fz[2] : fz[3];
            const double x0 = xs[(e * 16 + col) * 4 + 0] + (double)ffx * e0;
            const double x1 = xs[(e * 16 + col) * 4 + 1] + (double)ffy * e0;
            const double x2 = xs[(e * 16 + col) * 4 + 2] + (double)ffz * e0;
            const int kk = kk0 + e;
            if (li < n && lj < n && kk < P.kk_end) {
                const double nrm = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
                const size_t v = (size_t)kk * plane + (size_t)lj * n + li;
                Y0[v] = x0 / nrm;  // 0/0 -> NaN exactly like X /= X.norm() (:61)
                Y1[v] = x1 / nrm;
                Y2[v] = x2 / nrm;
            }
        }
    }  // tile loop
}

}  // namespace shm
