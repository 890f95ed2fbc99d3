// kge_rescal.hip - RESCAL (models/pytorch/score_fun.py:378-449): every relation row is a
// [rel_dim x ent_dim] matrix M (rel_dim == ent_dim == D, general_models.py:232-236) and
//     positive score   p = h . (M t)                                  (edge_func, :387-394)
//     pos-side vector  a = M x,  x = tail (head mode) or head (tail mode)   (create_neg, :428-447)
//     negative score   n_ij = a_i . neg_j                              -> the dot-product GEMM kernels
// The per-edge work is matrix-vector products over D*D floats (640 KB at D = 400): pure HBM streaming.
// One workgroup per edge makes ONE pass over M and produces up to four products at once
// (M y1, M y2 by rows; M^T z1, M^T z2 by columns), which covers the forward (M t, M h) and the backward
// (M^T h, M^T GA).  The relation gradient dp h t^T + GA x^T (+ regulariser) is never materialised on the
// fused path: the relation update kernel rebuilds it per element from the rank-1 factors.
#include "kge_common.hpp"
#include <type_traits>

using namespace kge;

static inline int check_launch_r() { return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; }

// ---------------------------------------------------------------------------------------------
// one pass over M_i (i = edge): out_r1 = M y1, out_r2 = M y2 (row dots), out_c1 = M^T z1, out_c2 = M^T z2
// (column sums); any product may be absent (null).  NCH = ceil(D / 64) column chunks per lane.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(KGE_BLOCK) void rescal_matvec_kernel(RescalMatvecArgs a) {
    __shared__ float colsum[2][KGE_WAVES_PER_BLOCK][NCH * 64];
    __shared__ float pred[KGE_WAVES_PER_BLOCK];
    const int i = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D = a.D;                                // rows of M
    const int Dc = a.Dc > 0 ? a.Dc : a.D;             // columns of M (RESCAL: square; TransR projections: D_e x D_r)
    const float *M = a.rel + (a.ridx ? a.ridx[i] : (int64_t)i) * (int64_t)D * Dc;
    auto vec = [&](const float *base, const int64_t *idx, int len) -> const float * {
        return base ? base + (idx ? idx[i] : (int64_t)i) * (int64_t)len : nullptr;
    };
    const float *y1 = vec(a.y1, a.y1idx, Dc), *y2 = vec(a.y2, a.y2idx, Dc);      // column-space vectors
    const float *z1 = vec(a.z1, a.z1idx, D), *z2 = vec(a.z2, a.z2idx, D);        // row-space vectors
    const float *pd = vec(a.pd, a.pdidx, D);
    float pacc = 0.f;
    float y1v[NCH], y2v[NCH], c1[NCH], c2[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int b = lane + 64 * k;
        y1v[k] = (y1 && b < Dc) ? y1[b] : 0.f;
        y2v[k] = (y2 && b < Dc) ? y2[b] : 0.f;
        c1[k] = 0.f; c2[k] = 0.f;
    }
    for (int r = wave; r < D; r += KGE_WAVES_PER_BLOCK) {
        const float *row = M + (int64_t)r * Dc;
        const float zz1 = z1 ? z1[r] : 0.f, zz2 = z2 ? z2[r] : 0.f;
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int b = lane + 64 * k;
            const float m = b < Dc ? row[b] : 0.f;
            d1 = fmaf(m, y1v[k], d1);
            d2 = fmaf(m, y2v[k], d2);
            c1[k] = fmaf(m, zz1, c1[k]);
            c2[k] = fmaf(m, zz2, c2[k]);
        }
        if (a.r1 || a.r2 || a.p) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
            if (lane == 0) {
                if (a.r1) a.r1[(int64_t)i * D + r] = d1;
                if (a.r2) a.r2[(int64_t)i * D + r] = d2;
                if (pd) pacc = fmaf(pd[r], d1, pacc);
            }
        }
    }
    if (a.p) {               // p = pd . (M y1): lane 0 of every wavefront holds its rows' part
        if (lane == 0) pred[wave] = pacc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sp = 0.f;
#pragma unroll
            for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) sp += pred[w];
            a.p[i] = sp;
        }
    }
    if (a.c1 || a.c2) {      // column sums: add the four wavefronts' partials in a fixed order
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            colsum[0][wave][lane + 64 * k] = c1[k];
            colsum[1][wave][lane + 64 * k] = c2[k];
        }
        __syncthreads();
        for (int b = threadIdx.x; b < Dc; b += KGE_BLOCK) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) { s1 += colsum[0][w][b]; s2 += colsum[1][w][b]; }
            if (a.c1) a.c1[(int64_t)i * Dc + b] = s1;
            if (a.c2) a.c2[(int64_t)i * Dc + b] = s2;
        }
    }
}

// the same pass with 16-byte accesses (columns % 4 == 0): a lane owns the columns 4 lane + 256 k .. + 3
template <int NC4>
__global__ __launch_bounds__(KGE_BLOCK) void rescal_matvec4_kernel(RescalMatvecArgs a) {
    __shared__ float colsum[2][KGE_WAVES_PER_BLOCK][NC4 * 256];
    __shared__ float pred[KGE_WAVES_PER_BLOCK];
    const int i = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D = a.D, Dc = a.Dc > 0 ? a.Dc : a.D;
    const float *M = a.rel + (a.ridx ? a.ridx[i] : (int64_t)i) * (int64_t)D * Dc;
    auto vec = [&](const float *base, const int64_t *idx, int len) -> const float * {
        return base ? base + (idx ? idx[i] : (int64_t)i) * (int64_t)len : nullptr;
    };
    const float *y1 = vec(a.y1, a.y1idx, Dc), *y2 = vec(a.y2, a.y2idx, Dc);
    const float *z1 = vec(a.z1, a.z1idx, D), *z2 = vec(a.z2, a.z2idx, D);
    const float *pd = vec(a.pd, a.pdidx, D);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float pacc = 0.f;
    float4 y1v[NC4], y2v[NC4], c1[NC4], c2[NC4];
#pragma unroll
    for (int k = 0; k < NC4; ++k) {
        const int b = 4 * lane + 256 * k;
        y1v[k] = (y1 && b < Dc) ? *reinterpret_cast<const float4 *>(y1 + b) : zero4;
        y2v[k] = (y2 && b < Dc) ? *reinterpret_cast<const float4 *>(y2 + b) : zero4;
        c1[k] = zero4; c2[k] = zero4;
    }
    const bool rowdots = a.r1 || a.r2 || a.p;
    // U rows of this wavefront per iteration, all requested before the first is used (a wavefront walks D / 4 rows one after the
    // other: with one row in flight the pass is a chain of round trips).  Rows past the matrix are clamped and dropped.
    constexpr int U = NC4 == 1 ? 8 : (NC4 == 2 ? 4 : 2);
    for (int rbase = wave; rbase < D; rbase += U * KGE_WAVES_PER_BLOCK) {
        float4 m[U][NC4];
        float zz1[U], zz2[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = rbase + j * KGE_WAVES_PER_BLOCK, rc = min(r, D - 1);
            const float *row = M + (int64_t)rc * Dc;
#pragma unroll
            for (int k = 0; k < NC4; ++k) {
                const int b = 4 * lane + 256 * k;
                m[j][k] = (b < Dc && r < D) ? *reinterpret_cast<const float4 *>(row + b) : zero4;
            }
            zz1[j] = (z1 && r < D) ? z1[rc] : 0.f;
            zz2[j] = (z2 && r < D) ? z2[rc] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = rbase + j * KGE_WAVES_PER_BLOCK;
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int k = 0; k < NC4; ++k) {
                const float4 mm = m[j][k];
                d1 = fmaf(mm.w, y1v[k].w, fmaf(mm.z, y1v[k].z, fmaf(mm.y, y1v[k].y, fmaf(mm.x, y1v[k].x, d1))));
                d2 = fmaf(mm.w, y2v[k].w, fmaf(mm.z, y2v[k].z, fmaf(mm.y, y2v[k].y, fmaf(mm.x, y2v[k].x, d2))));
                c1[k].x = fmaf(mm.x, zz1[j], c1[k].x); c1[k].y = fmaf(mm.y, zz1[j], c1[k].y);
                c1[k].z = fmaf(mm.z, zz1[j], c1[k].z); c1[k].w = fmaf(mm.w, zz1[j], c1[k].w);
                c2[k].x = fmaf(mm.x, zz2[j], c2[k].x); c2[k].y = fmaf(mm.y, zz2[j], c2[k].y);
                c2[k].z = fmaf(mm.z, zz2[j], c2[k].z); c2[k].w = fmaf(mm.w, zz2[j], c2[k].w);
            }
            if (rowdots) {
                d1 = wave_sum(d1); d2 = wave_sum(d2);
                if (lane == 0 && r < D) {
                    if (a.r1) a.r1[(int64_t)i * D + r] = d1;
                    if (a.r2) a.r2[(int64_t)i * D + r] = d2;
                    if (pd) pacc = fmaf(pd[r], d1, pacc);
                }
            }
        }
    }
    if (a.p) {
        if (lane == 0) pred[wave] = pacc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sp = 0.f;
#pragma unroll
            for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) sp += pred[w];
            a.p[i] = sp;
        }
    }
    if (a.c1 || a.c2) {
#pragma unroll
        for (int k = 0; k < NC4; ++k) {
            *reinterpret_cast<float4 *>(&colsum[0][wave][4 * lane + 256 * k]) = c1[k];
            *reinterpret_cast<float4 *>(&colsum[1][wave][4 * lane + 256 * k]) = c2[k];
        }
        __syncthreads();
        for (int b = threadIdx.x; b < Dc; b += KGE_BLOCK) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) { s1 += colsum[0][w][b]; s2 += colsum[1][w][b]; }
            if (a.c1) a.c1[(int64_t)i * Dc + b] = s1;
            if (a.c2) a.c2[(int64_t)i * Dc + b] = s2;
        }
    }
}

int launch_rescal_matvec(const RescalMatvecArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    const dim3 g(a.B), b(KGE_BLOCK);
    const int cols = a.Dc > 0 ? a.Dc : a.D;
    if (cols % 4 == 0 && cols <= 1024) {
        if (cols <= 256) hipLaunchKernelGGL(rescal_matvec4_kernel<1>, g, b, 0, s, a);
        else if (cols <= 512) hipLaunchKernelGGL(rescal_matvec4_kernel<2>, g, b, 0, s, a);
        else hipLaunchKernelGGL(rescal_matvec4_kernel<4>, g, b, 0, s, a);
        return check_launch_r();
    }
    if (cols <= 256) hipLaunchKernelGGL(rescal_matvec_kernel<4>, g, b, 0, s, a);
    else if (cols <= 512) hipLaunchKernelGGL(rescal_matvec_kernel<8>, g, b, 0, s, a);
    else if (cols <= 1024) hipLaunchKernelGGL(rescal_matvec_kernel<16>, g, b, 0, s, a);
    else return KGE_ERR_ARG;
    return check_launch_r();
}

// out_i = s1_i * u1_i (+ u2_i)   (vector combine: GH = dp * (M t) + M^T GA ...), one wavefront per edge
__global__ __launch_bounds__(KGE_BLOCK) void rescal_axpy_kernel(const float *s1, const float *u1, const float *u2, int B,
                                                                int D, float *out, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= B) return;
    const int lane = threadIdx.x & 63;
    const float c = alpha * (s1 ? s1[i] : 1.f);
    for (int b = lane; b < D; b += 64)
        out[i * (int64_t)D + b] = c * u1[i * (int64_t)D + b] + (u2 ? u2[i * (int64_t)D + b] : 0.f);
}

int launch_rescal_axpy(const float *s1, const float *u1, const float *u2, int B, int D, float *out, hipStream_t s,
                       float alpha) {
    if (B == 0) return KGE_OK;
    hipLaunchKernelGGL(rescal_axpy_kernel, dim3((B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s,
                       s1, u1, u2, B, D, out, alpha);
    return check_launch_r();
}

// materialised per-edge relation gradient (drop-in / debugging path only):
// G_i[a][b] = c_i * u_i[a] * v_i[b] (+ G_i[a][b] if accumulate) (+ regulariser of M_i if reg)
__global__ __launch_bounds__(KGE_BLOCK) void rescal_outer_kernel(RescalOuterArgs a) {
    const int i = blockIdx.x;
    const int D = a.D;
    const float c = a.c ? a.c[i] : 1.f;
    const float *u = a.u + (a.uidx ? a.uidx[i] : (int64_t)i) * (int64_t)D;
    const float *v = a.v + (a.vidx ? a.vidx[i] : (int64_t)i) * (int64_t)D;
    const float *M = a.rel ? a.rel + (a.ridx ? a.ridx[i] : (int64_t)i) * (int64_t)D * D : nullptr;
    float *G = a.G + (int64_t)i * D * D;
    const int64_t n = (int64_t)D * D;
    for (int64_t e = threadIdx.x; e < n; e += KGE_BLOCK) {
        const int r = (int)(e / D), b = (int)(e - (int64_t)r * D);
        float g = c * u[r] * v[b];
        if (a.accumulate) g += G[e];
        if (M) g += reg_grad(M[e], a.reg_coef, a.reg_norm);
        G[e] = g;
    }
}

int launch_rescal_outer(const RescalOuterArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    hipLaunchKernelGGL(rescal_outer_kernel, dim3(a.B), dim3(KGE_BLOCK), 0, s, a);
    return check_launch_r();
}

// ---------------------------------------------------------------------------------------------
// fused relation update.  Per edge e the traced-row gradient is g_e = dp_e h_e t_e^T + GA_e x_e^T + R
// (R = regulariser gradient of the current M, one copy per traced row like the reference,
// general_models.py:572-576) and Adagrad (tensor_models.py:330-361) needs S_u = sum_{e in E_u} mean(g_e^2)
// BEFORE the row changes.  Three launches, all deterministic (fixed summation orders, no atomics):
//   1. rescal_edge_sq:   mean(g_e^2) per edge.  Without regulariser the rank-2 matrix has the closed form
//        |g|^2 = dp^2 |h|^2 |t|^2 + |GA|^2 |x|^2 + 2 dp (h.GA)(t.x)      (one wavefront per edge, O(D));
//      with regulariser: RESCAL_RB row blocks per edge sum (dp h_r t_b + GA_r x_b + R_rb)^2 over the matrix.
//   2. rescal_rel_state: one thread per unique relation: S_u in plan order, state += S_u, 1/std -> scratch.
//   3. rescal_apply:     (unique relation, row block) workgroups: M -= lr * (sum_e g_e) / std, the gradient
//      rebuilt per element from the rank-1 factors (never materialised).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(KGE_BLOCK) void rescal_edge_sq_kernel(RescalUpdateArgs a) {      // no regulariser
    const int64_t e = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (e >= a.B) return;
    const int lane = threadIdx.x & 63, D = a.D;
    const float *h = a.ent + a.hidx[e] * (int64_t)D, *t = a.ent + a.tidx[e] * (int64_t)D, *x = a.neg_head ? t : h;
    const float *ga = a.GA + e * (int64_t)D;
    float hh = 0.f, tt = 0.f, gg = 0.f, xx = 0.f, hg = 0.f, tx = 0.f;
    for (int b = lane; b < D; b += 64) {
        const float hv = h[b], tv = t[b], gv = ga[b], xv = x[b];
        hh = fmaf(hv, hv, hh); tt = fmaf(tv, tv, tt); gg = fmaf(gv, gv, gg);
        xx = fmaf(xv, xv, xx); hg = fmaf(hv, gv, hg); tx = fmaf(tv, xv, tx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        hh += __shfl_xor(hh, o, 64); tt += __shfl_xor(tt, o, 64); gg += __shfl_xor(gg, o, 64);
        xx += __shfl_xor(xx, o, 64); hg += __shfl_xor(hg, o, 64); tx += __shfl_xor(tx, o, 64);
    }
    if (lane == 0) {
        const float dp = a.dpos[e];
        a.gs[e * RESCAL_RB] = fmaxf(dp * dp * hh * tt + gg * xx + 2.f * dp * hg * tx, 0.f) / ((float)D * (float)D);
    }
}

// with regulariser, closed form (round 6): g = dp h t^T + GA x^T + R  ->
//   sum g^2 = [dp^2 |h|^2 |t|^2 + |GA|^2 |x|^2 + 2 dp (h.GA)(t.x)] + 2 [dp h.(R t) + GA.(R x)] + sum R^2
// with R t / R x / sum R^2 left by the forward pass over the matrices (rescal_rel_fwd_kernel<.., REG>): no pass over M here
// (rescal_edge_sq_reg_kernel below read every traced matrix once more: 328 us of the FB15k recipe's 923-us step)
__global__ __launch_bounds__(KGE_BLOCK) void rescal_edge_sq_regx_kernel(RescalUpdateArgs a) {
    const int64_t e = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (e >= a.B) return;
    const int lane = threadIdx.x & 63, D = a.D;
    const float *h = a.ent + a.hidx[e] * (int64_t)D, *t = a.ent + a.tidx[e] * (int64_t)D, *x = a.neg_head ? t : h;
    const float *ga = a.GA + e * (int64_t)D;
    const float *pv = a.PV + e * (int64_t)D, *px = a.neg_head ? pv : a.PW + e * (int64_t)D;
    float hh = 0.f, tt = 0.f, gg = 0.f, xx = 0.f, hg = 0.f, tx = 0.f, hp = 0.f, gp = 0.f;
    for (int b = lane; b < D; b += 64) {
        const float hv = h[b], tv = t[b], gv = ga[b], xv = x[b];
        hh = fmaf(hv, hv, hh); tt = fmaf(tv, tv, tt); gg = fmaf(gv, gv, gg);
        xx = fmaf(xv, xv, xx); hg = fmaf(hv, gv, hg); tx = fmaf(tv, xv, tx);
        hp = fmaf(hv, pv[b], hp); gp = fmaf(gv, px[b], gp);
    }
    float rr = lane < RESCAL_RBN ? a.rho[e * RESCAL_RBN + lane] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        hh += __shfl_xor(hh, o, 64); tt += __shfl_xor(tt, o, 64); gg += __shfl_xor(gg, o, 64);
        xx += __shfl_xor(xx, o, 64); hg += __shfl_xor(hg, o, 64); tx += __shfl_xor(tx, o, 64);
        hp += __shfl_xor(hp, o, 64); gp += __shfl_xor(gp, o, 64); rr += __shfl_xor(rr, o, 64);
    }
    if (lane == 0) {
        const float dp = a.dpos[e];
        const float ss = dp * dp * hh * tt + gg * xx + 2.f * dp * hg * tx + 2.f * (dp * hp + gp) + rr;
        a.gs[e * RESCAL_RB] = fmaxf(ss, 0.f) / ((float)D * (float)D);
    }
}

__global__ __launch_bounds__(KGE_BLOCK) void rescal_edge_sq_reg_kernel(RescalUpdateArgs a) {  // with regulariser, one pass over M per edge
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    const int e = blockIdx.x / RESCAL_RB, rb = blockIdx.x % RESCAL_RB;
    const int D = a.D;
    const int rows = (D + RESCAL_RB - 1) / RESCAL_RB, r0 = rb * rows, r1 = min(D, r0 + rows);
    const float *h = a.ent + a.hidx[e] * (int64_t)D, *t = a.ent + a.tidx[e] * (int64_t)D, *x = a.neg_head ? t : h;
    const float *ga = a.GA + (int64_t)e * D;
    const float *M = a.rel + a.rel_ids[e] * (int64_t)D * D;
    const float dp = a.dpos[e];
    float ss = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float dh = dp * h[r], gr = ga[r];
        for (int b = threadIdx.x; b < D; b += KGE_BLOCK) {
            const float g = dh * t[b] + gr * x[b] + reg_grad(M[(int64_t)r * D + b], a.reg_coef, a.reg_norm);
            ss = fmaf(g, g, ss);
        }
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) a.gs[(int64_t)e * RESCAL_RB + rb] = ss / ((float)D * (float)D);
}

__global__ void rescal_rel_state_kernel(RescalUpdateArgs a, int nparts) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int64_t id = a.ur_id[u];
    float S = 0.f;
    for (int q = a.ur_ptr[u]; q < a.ur_ptr[u + 1]; ++q) {
        const int64_t e = a.ur_edge[q];
        float g = 0.f;
        for (int k = 0; k < nparts; ++k) g += a.gs[e * RESCAL_RB + k];
        S += g;
    }
    const float sN = a.rel_state[id] + S;
    a.rel_state[id] = sN;
    a.inv_std[u] = 1.f / (sqrtf(sN) + a.eps);
}

// NCOL = ceil(D / 256): every thread owns columns b = tid + 256 j of all rows of the block, so the t / x
// operands of an edge are loaded once per row from L1 and the per-row factors (dp h_r, GA_r) are uniform
template <int NCOL>
__global__ __launch_bounds__(KGE_BLOCK) void rescal_apply_kernel(RescalUpdateArgs a) {
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    const int u = blockIdx.x / RESCAL_RB, rb = blockIdx.x % RESCAL_RB;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int D = a.D;
    const int rows = (D + RESCAL_RB - 1) / RESCAL_RB, r0 = rb * rows, r1 = min(D, r0 + rows);
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    float *M = a.rel + a.ur_id[u] * (int64_t)D * D;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    const float cnt = (float)(e1 - e0), step = -a.lr * a.inv_std[u];
    // per-edge metadata does not depend on the row: fetch it once (the chain ur_edge -> ids -> rows would
    // otherwise be three dependent loads in every row iteration)
    constexpr int NEC = 32;
    __shared__ int64_t s_ho[NEC], s_to[NEC], s_go[NEC];
    __shared__ float s_dp[NEC];
    const int nec = min(NEC, e1 - e0);
    if ((int)threadIdx.x < nec) {
        const int64_t e = a.ur_edge[e0 + threadIdx.x];
        s_ho[threadIdx.x] = a.hidx[e] * (int64_t)D;
        s_to[threadIdx.x] = a.tidx[e] * (int64_t)D;
        s_go[threadIdx.x] = e * (int64_t)D;
        s_dp[threadIdx.x] = a.dpos[e];
    }
    __syncthreads();
    float rv = 0.f;
#pragma unroll 2
    for (int r = r0; r < r1; ++r) {
        float m[NCOL], g[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int b = threadIdx.x + KGE_BLOCK * j;
            m[j] = b < D ? M[(int64_t)r * D + b] : 0.f;
            g[j] = 0.f;
            if (reg) { g[j] = cnt * reg_grad(m[j], a.reg_coef, a.reg_norm); rv += reg_val(m[j], a.reg_norm); }
        }
        for (int q = 0; q < e1 - e0; ++q) {
            int64_t ho, to, go; float dp;
            if (q < NEC) { ho = s_ho[q]; to = s_to[q]; go = s_go[q]; dp = s_dp[q]; }
            else {
                const int64_t e = a.ur_edge[e0 + q];
                ho = a.hidx[e] * (int64_t)D; to = a.tidx[e] * (int64_t)D; go = e * (int64_t)D; dp = a.dpos[e];
            }
            const int64_t xo = a.neg_head ? to : ho;
            const float dh = dp * a.ent[ho + r], gr = a.GA[go + r];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int b = threadIdx.x + KGE_BLOCK * j;
                if (b < D) g[j] = fmaf(dh, a.ent[to + b], fmaf(gr, a.ent[xo + b], g[j]));
            }
        }
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int b = threadIdx.x + KGE_BLOCK * j;
            if (b < D) M[(int64_t)r * D + b] = fmaf(step, g[j], m[j]);
        }
    }
    if (a.reg_part) {        // regularisation value of the traced copies: cnt * coef * sum |M|^p, per row block
        const float tot = reg ? block_sum(rv, red) : 0.f;
        if (threadIdx.x == 0) a.reg_part[(int64_t)u * RESCAL_RB + rb] = a.reg_coef * tot * cnt;
    }
}

// 16-byte variant (D % 4 == 0): TPR = power of two >= D/4 threads cover one row with one float4 each, the
// workgroup walks 256 / TPR rows per iteration; two iterations are in flight (unroll) for memory parallelism
__global__ __launch_bounds__(KGE_BLOCK) void rescal_apply_vec_kernel(RescalUpdateArgs a, int tpr) {
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    const int u = blockIdx.x / RESCAL_RB, rb = blockIdx.x % RESCAL_RB;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int D = a.D;
    const int rows = (D + RESCAL_RB - 1) / RESCAL_RB, r0 = rb * rows, r1 = min(D, r0 + rows);
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    float *M = a.rel + a.ur_id[u] * (int64_t)D * D;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    const float cnt = (float)(e1 - e0), step = -a.lr * a.inv_std[u];
    constexpr int NEC = 32;
    __shared__ int64_t s_ho[NEC], s_to[NEC], s_go[NEC];
    __shared__ float s_dp[NEC];
    const int nec = min(NEC, e1 - e0);
    if ((int)threadIdx.x < nec) {
        const int64_t e = a.ur_edge[e0 + threadIdx.x];
        s_ho[threadIdx.x] = a.hidx[e] * (int64_t)D;
        s_to[threadIdx.x] = a.tidx[e] * (int64_t)D;
        s_go[threadIdx.x] = e * (int64_t)D;
        s_dp[threadIdx.x] = a.dpos[e];
    }
    __syncthreads();
    const int rpi = KGE_BLOCK / tpr;                       // rows per iteration
    const int rsub = threadIdx.x / tpr, b = (threadIdx.x % tpr) * 4;
    const bool colok = b < D;
    float rv = 0.f;
#pragma unroll 2
    for (int rr = r0; rr < r1; rr += rpi) {
        const int r = rr + rsub;
        if (r >= r1 || !colok) continue;
        float4 m = *reinterpret_cast<const float4 *>(M + (int64_t)r * D + b);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (reg) {
            g.x = cnt * reg_grad(m.x, a.reg_coef, a.reg_norm); g.y = cnt * reg_grad(m.y, a.reg_coef, a.reg_norm);
            g.z = cnt * reg_grad(m.z, a.reg_coef, a.reg_norm); g.w = cnt * reg_grad(m.w, a.reg_coef, a.reg_norm);
            rv += reg_val(m.x, a.reg_norm) + reg_val(m.y, a.reg_norm) + reg_val(m.z, a.reg_norm) + reg_val(m.w, a.reg_norm);
        }
        for (int q = 0; q < e1 - e0; ++q) {
            int64_t ho, to, go; float dp;
            if (q < NEC) { ho = s_ho[q]; to = s_to[q]; go = s_go[q]; dp = s_dp[q]; }
            else {
                const int64_t e = a.ur_edge[e0 + q];
                ho = a.hidx[e] * (int64_t)D; to = a.tidx[e] * (int64_t)D; go = e * (int64_t)D; dp = a.dpos[e];
            }
            const int64_t xo = a.neg_head ? to : ho;
            const float dh = dp * a.ent[ho + r], gr = a.GA[go + r];
            const float4 tv = *reinterpret_cast<const float4 *>(a.ent + to + b);
            const float4 xv = *reinterpret_cast<const float4 *>(a.ent + xo + b);
            g.x = fmaf(dh, tv.x, fmaf(gr, xv.x, g.x)); g.y = fmaf(dh, tv.y, fmaf(gr, xv.y, g.y));
            g.z = fmaf(dh, tv.z, fmaf(gr, xv.z, g.z)); g.w = fmaf(dh, tv.w, fmaf(gr, xv.w, g.w));
        }
        m.x = fmaf(step, g.x, m.x); m.y = fmaf(step, g.y, m.y); m.z = fmaf(step, g.z, m.z); m.w = fmaf(step, g.w, m.w);
        *reinterpret_cast<float4 *>(M + (int64_t)r * D + b) = m;
    }
    if (a.reg_part) {
        const float tot = reg ? block_sum(rv, red) : 0.f;
        if (threadIdx.x == 0) a.reg_part[(int64_t)u * RESCAL_RB + rb] = a.reg_coef * tot * cnt;
    }
}

// ---------------------------------------------------------------------------------------------
// fused step, per UNIQUE relation (the batch plan's ur_id / ur_ptr / ur_edge; D % 4 == 0): a relation matrix is streamed from
// HBM once in the forward and once in the backward + update, however many edges of the batch carry the relation - the per-edge
// kernel above streams it once per edge and the stand-alone Adagrad pass a third time.
// Workgroup = (unique relation, row block of RESCAL_RBN); a wavefront owns whole rows, a lane the columns 4 lane + 256 k .. + 3
// (16-byte accesses); the edges of the relation are taken EG at a time with their vectors in registers (further groups re-read
// the row block from L2).
// ---------------------------------------------------------------------------------------------
#define RESCAL_RMAX ((1024 + RESCAL_RBN - 1) / RESCAL_RBN)     // rows of one block at the widest supported matrix
#ifndef RESCAL_UNR
#define RESCAL_UNR 2                 // rows in flight per wavefront in the backward + update pass
#endif
#ifndef RESCAL_EG
#define RESCAL_EG 1                  // edges of a relation whose vectors are held in registers per pass (D <= 512)
#endif

__device__ __forceinline__ float4 ld4z(const float *p, bool ok) {
    return ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b, float acc) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}
__device__ __forceinline__ void axpy4(float4 &acc, const float4 &m, float s) {
    acc.x = fmaf(m.x, s, acc.x); acc.y = fmaf(m.y, s, acc.y); acc.z = fmaf(m.z, s, acc.z); acc.w = fmaf(m.w, s, acc.w);
}

__device__ __forceinline__ float4 reg_grad4(const float4 &m, float coef, int q) {
    if (q == 3)                                        // (wave-uniform; the default norm: plain multiplies instead of exp2 / log)
        return make_float4(reg_grad3(m.x, coef), reg_grad3(m.y, coef), reg_grad3(m.z, coef), reg_grad3(m.w, coef));
    return make_float4(reg_grad(m.x, coef, q), reg_grad(m.y, coef, q), reg_grad(m.z, coef, q), reg_grad(m.w, coef, q));
}

// REG (round 6): the pass also applies the regulariser's gradient R = reg_grad(M) to the edges' vectors - R t (and R h with W2) per row,
// sum(R^2) per row block - so that the update needs no pass of its own for the traced rows' mean squares (rescal_edge_sq_regx_kernel)
template <int NC4, int EG, bool W2, bool REG>
__global__ __launch_bounds__(KGE_BLOCK) void rescal_rel_fwd_kernel(RescalRelFwdArgs a) {
    __shared__ float s_h[EG][RESCAL_RMAX];
    __shared__ float s_p[EG][KGE_WAVES_PER_BLOCK];
    __shared__ float s_r[KGE_WAVES_PER_BLOCK];
    float rsum = 0.f;                                  // this lane's share of sum(R^2) over the row block (first edge group only)
    const int u = blockIdx.x / RESCAL_RBN, rb = blockIdx.x % RESCAL_RBN;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int D = a.D, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = (D + RESCAL_RBN - 1) / RESCAL_RBN, r0 = rb * rows, r1 = min(D, r0 + rows);
    const float *M = a.rel + a.ur_id[u] * (int64_t)D * D;
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    for (int q0 = e0; q0 < e1; q0 += EG) {
        int64_t e[EG], ho[EG];
        bool on[EG];
        float4 yt[EG][NC4], yh[EG][NC4];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            on[g] = q0 + g < e1;
            e[g] = a.ur_edge[on[g] ? q0 + g : q0];
            ho[g] = a.hidx[e[g]] * (int64_t)D;
            const float *tp = a.ent + a.tidx[e[g]] * (int64_t)D, *hp = a.ent + ho[g];
#pragma unroll
            for (int k = 0; k < NC4; ++k) {
                const int b = 4 * lane + 256 * k;
                yt[g][k] = ld4z(tp + b, b < D);
                yh[g][k] = ld4z(hp + b, W2 && b < D);
            }
        }
        __syncthreads();                               // the previous group's readers of s_h are done
#pragma unroll
        for (int g = 0; g < EG; ++g)
            for (int r = threadIdx.x; r < r1 - r0; r += KGE_BLOCK) s_h[g][r] = a.ent[ho[g] + r0 + r];
        __syncthreads();
        float pa[EG];
#pragma unroll
        for (int g = 0; g < EG; ++g) pa[g] = 0.f;
        // two rows per iteration by hand (the wavefront reductions keep the compiler from unrolling a loop of unknown length)
        for (int r = r0 + wave; r < r1; r += 2 * KGE_WAVES_PER_BLOCK) {
            const int rB = r + KGE_WAVES_PER_BLOCK;
            const bool hasB = rB < r1;
            const float *rowA = M + (int64_t)r * D, *rowB = M + (int64_t)(hasB ? rB : r) * D;
            float4 mA[NC4], mB[NC4];
#pragma unroll
            for (int k = 0; k < NC4; ++k) { const int b = 4 * lane + 256 * k; mA[k] = ld4z(rowA + b, b < D); }
#pragma unroll
            for (int k = 0; k < NC4; ++k) { const int b = 4 * lane + 256 * k; mB[k] = ld4z(rowB + b, b < D); }
            float4 pA[REG ? NC4 : 1], pB[REG ? NC4 : 1];
            if constexpr (REG) {
#pragma unroll
                for (int k = 0; k < NC4; ++k) { pA[k] = reg_grad4(mA[k], a.reg_coef, a.reg_norm); pB[k] = reg_grad4(mB[k], a.reg_coef, a.reg_norm); }
                if (q0 == e0) {
#pragma unroll
                    for (int k = 0; k < NC4; ++k) { rsum = dot4(pA[k], pA[k], rsum); if (hasB) rsum = dot4(pB[k], pB[k], rsum); }
                }
            }
#pragma unroll
            for (int g = 0; g < EG; ++g) {
                float d1A = 0.f, d2A = 0.f, d1B = 0.f, d2B = 0.f, d3A = 0.f, d4A = 0.f, d3B = 0.f, d4B = 0.f;
#pragma unroll
                for (int k = 0; k < NC4; ++k) {
                    d1A = dot4(mA[k], yt[g][k], d1A); d1B = dot4(mB[k], yt[g][k], d1B);
                    if (W2) { d2A = dot4(mA[k], yh[g][k], d2A); d2B = dot4(mB[k], yh[g][k], d2B); }
                    if constexpr (REG) {
                        d3A = dot4(pA[k], yt[g][k], d3A); d3B = dot4(pB[k], yt[g][k], d3B);
                        if (W2) { d4A = dot4(pA[k], yh[g][k], d4A); d4B = dot4(pB[k], yh[g][k], d4B); }
                    }
                }
                d1A = wave_sum(d1A); d1B = wave_sum(d1B);
                if (W2) { d2A = wave_sum(d2A); d2B = wave_sum(d2B); }
                if constexpr (REG) {
                    d3A = wave_sum(d3A); d3B = wave_sum(d3B);
                    if (W2) { d4A = wave_sum(d4A); d4B = wave_sum(d4B); }
                }
                if (lane == 0 && on[g]) {
                    a.V[e[g] * D + r] = d1A;
                    if (W2) a.W[e[g] * D + r] = d2A;
                    if constexpr (REG) { a.PV[e[g] * D + r] = d3A; if (W2) a.PW[e[g] * D + r] = d4A; }
                    pa[g] = fmaf(s_h[g][r - r0], d1A, pa[g]);
                    if (hasB) {
                        a.V[e[g] * D + rB] = d1B;
                        if (W2) a.W[e[g] * D + rB] = d2B;
                        if constexpr (REG) { a.PV[e[g] * D + rB] = d3B; if (W2) a.PW[e[g] * D + rB] = d4B; }
                        pa[g] = fmaf(s_h[g][rB - r0], d1B, pa[g]);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < EG; ++g) if (lane == 0) s_p[g][wave] = pa[g];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < EG; ++g)
            if ((int)threadIdx.x == g && on[g]) {
                float sp = 0.f;
#pragma unroll
                for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) sp += s_p[g][w];
                a.ppart[e[g] * RESCAL_RBN + rb] = sp;
            }
    }
    if constexpr (REG) {                               // sum(R^2) of this row block: the same value for every edge of the relation
        rsum = wave_sum(rsum);
        __syncthreads();
        if (lane == 0) s_r[wave] = rsum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) tot += s_r[w];
        for (int q = e0 + (int)threadIdx.x; q < e1; q += KGE_BLOCK) a.rho[(int64_t)a.ur_edge[q] * RESCAL_RBN + rb] = tot;
    }
}

__global__ void rescal_psum_kernel(RescalRelFwdArgs a) {       // p = the row blocks' parts in a fixed order
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B) return;
    float sp = 0.f;
#pragma unroll
    for (int k = 0; k < RESCAL_RBN; ++k) sp += a.ppart[(int64_t)i * RESCAL_RBN + k];
    a.P[i] = sp;
}

int launch_rescal_rel_fwd(const RescalRelFwdArgs &a, hipStream_t s) {
    if (a.B == 0 || a.UR == 0) return KGE_OK;
    if (a.D % 4 != 0 || a.D > 1024) return KGE_ERR_ARG;
    const dim3 g(a.UR * RESCAL_RBN), b(KGE_BLOCK);
    const bool w2 = a.W != nullptr;
    const bool rg = a.PV && a.rho && a.reg_coef > 0.f && a.reg_norm > 0 && (!w2 || a.PW);
#define RF(NC4, EG) do { if (w2 && rg) hipLaunchKernelGGL((rescal_rel_fwd_kernel<NC4, EG, true, true>), g, b, 0, s, a); \
                         else if (w2) hipLaunchKernelGGL((rescal_rel_fwd_kernel<NC4, EG, true, false>), g, b, 0, s, a); \
                         else if (rg) hipLaunchKernelGGL((rescal_rel_fwd_kernel<NC4, EG, false, true>), g, b, 0, s, a); \
                         else hipLaunchKernelGGL((rescal_rel_fwd_kernel<NC4, EG, false, false>), g, b, 0, s, a); } while (0)
    if (a.D <= 256) RF(1, RESCAL_EG);
    else if (a.D <= 512) RF(2, RESCAL_EG);
    else RF(4, 1);
#undef RF
    hipLaunchKernelGGL(rescal_psum_kernel, dim3((a.B + 255) / 256), dim3(256), 0, s, a);
    return check_launch_r();
}

// backward + Adagrad in ONE pass over the row block: the column products M^T h_e and M^T GA_e of every edge of the relation
// (parts per row block, summed by rescal_combine) from the rows as loaded, then the rows' update
//   M -= lr / std * sum_e (dp_e h_e t_e^T + GA_e x_e^T (+ R))      (edges in plan order, like rescal_apply)
// with the last EG edges' t / x vectors in registers and the earlier ones (relations with more than EG edges) read through L1.
template <int NC4, int EG>
__global__ __launch_bounds__(KGE_BLOCK) void rescal_rel_bwd_apply_kernel(RescalUpdateArgs a) {
    __shared__ float s_h[EG][RESCAL_RMAX], s_g[EG][RESCAL_RMAX];
    __shared__ float s_c[KGE_WAVES_PER_BLOCK][NC4 * 256];
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    const int u = blockIdx.x / RESCAL_RBN, rb = blockIdx.x % RESCAL_RBN;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int D = a.D, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = (D + RESCAL_RBN - 1) / RESCAL_RBN, r0 = rb * rows, r1 = min(D, r0 + rows);
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    float *M = a.rel + a.ur_id[u] * (int64_t)D * D;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    const float cnt = (float)(e1 - e0), step = -a.lr * a.inv_std[u];
    const int ngr = (e1 - e0 + EG - 1) / EG;
    float rv = 0.f;
    for (int j = 0; j < ngr; ++j) {
        const bool last = j == ngr - 1;
        const int qb = e0 + j * EG;
        int64_t e[EG], ho[EG], go[EG];
        bool on[EG];
        float dp[EG];
        float4 tv[EG][NC4], xv[EG][NC4];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            on[g] = qb + g < e1;
            e[g] = a.ur_edge[on[g] ? qb + g : qb];
            ho[g] = a.hidx[e[g]] * (int64_t)D;
            go[g] = e[g] * (int64_t)D;
            dp[g] = on[g] ? a.dpos[e[g]] : 0.f;
            const int64_t to = a.tidx[e[g]] * (int64_t)D, xo = a.neg_head ? to : ho[g];
#pragma unroll
            for (int k = 0; k < NC4; ++k) {
                const int b = 4 * lane + 256 * k;
                tv[g][k] = ld4z(a.ent + to + b, last && on[g] && b < D);
                xv[g][k] = ld4z(a.ent + xo + b, last && on[g] && b < D);
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < EG; ++g)
            for (int r = threadIdx.x; r < r1 - r0; r += KGE_BLOCK) {
                s_h[g][r] = on[g] ? a.ent[ho[g] + r0 + r] : 0.f;
                s_g[g][r] = on[g] ? a.GA[go[g] + r0 + r] : 0.f;
            }
        __syncthreads();
        float4 c1[EG][NC4], c2[EG][NC4];
#pragma unroll
        for (int g = 0; g < EG; ++g)
#pragma unroll
            for (int k = 0; k < NC4; ++k) { c1[g][k] = make_float4(0.f, 0.f, 0.f, 0.f); c2[g][k] = c1[g][k]; }
        // one row: column products from the row as loaded, then (last group) its update.  EARLIER: the relation has edges before
        // the register group - a variable-length inner loop, kept out of the common instance so that its row loop unrolls
        auto do_row = [&](int r, auto earlier) {
            constexpr bool EARLIER = decltype(earlier)::value;
            float *row = M + (int64_t)r * D;
            float4 m[NC4];
            float hr[EG], gr[EG];
#pragma unroll
            for (int k = 0; k < NC4; ++k) { const int b = 4 * lane + 256 * k; m[k] = ld4z(row + b, b < D); }
#pragma unroll
            for (int g = 0; g < EG; ++g) {
                hr[g] = s_h[g][r - r0]; gr[g] = s_g[g][r - r0];
#pragma unroll
                for (int k = 0; k < NC4; ++k) { axpy4(c1[g][k], m[k], hr[g]); axpy4(c2[g][k], m[k], gr[g]); }
            }
            if (last) {
                float4 gk[NC4];
#pragma unroll
                for (int k = 0; k < NC4; ++k) {
                    gk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (reg) {
                        gk[k].x = cnt * reg_grad(m[k].x, a.reg_coef, a.reg_norm); gk[k].y = cnt * reg_grad(m[k].y, a.reg_coef, a.reg_norm);
                        gk[k].z = cnt * reg_grad(m[k].z, a.reg_coef, a.reg_norm); gk[k].w = cnt * reg_grad(m[k].w, a.reg_coef, a.reg_norm);
                        rv += reg_val(m[k].x, a.reg_norm) + reg_val(m[k].y, a.reg_norm) + reg_val(m[k].z, a.reg_norm) +
                              reg_val(m[k].w, a.reg_norm);
                    }
                }
                if constexpr (EARLIER)
                    for (int q = e0; q < qb; ++q) {        // edges before the register group (> EG edges of one relation)
                        const int64_t eq = a.ur_edge[q];
                        const int64_t hq = a.hidx[eq] * (int64_t)D, tq = a.tidx[eq] * (int64_t)D, xq = a.neg_head ? tq : hq;
                        const float dh = a.dpos[eq] * a.ent[hq + r], gq = a.GA[eq * (int64_t)D + r];
#pragma unroll
                        for (int k = 0; k < NC4; ++k) {
                            const int b = 4 * lane + 256 * k;
                            const float4 t4 = ld4z(a.ent + tq + b, b < D), x4 = ld4z(a.ent + xq + b, b < D);
                            gk[k].x = fmaf(dh, t4.x, fmaf(gq, x4.x, gk[k].x)); gk[k].y = fmaf(dh, t4.y, fmaf(gq, x4.y, gk[k].y));
                            gk[k].z = fmaf(dh, t4.z, fmaf(gq, x4.z, gk[k].z)); gk[k].w = fmaf(dh, t4.w, fmaf(gq, x4.w, gk[k].w));
                        }
                    }
#pragma unroll
                for (int g = 0; g < EG; ++g) {
                    const float dh = dp[g] * hr[g];        // (an absent edge of the group: dp = 0, GA = 0, vectors = 0)
#pragma unroll
                    for (int k = 0; k < NC4; ++k) {
                        gk[k].x = fmaf(dh, tv[g][k].x, fmaf(gr[g], xv[g][k].x, gk[k].x));
                        gk[k].y = fmaf(dh, tv[g][k].y, fmaf(gr[g], xv[g][k].y, gk[k].y));
                        gk[k].z = fmaf(dh, tv[g][k].z, fmaf(gr[g], xv[g][k].z, gk[k].z));
                        gk[k].w = fmaf(dh, tv[g][k].w, fmaf(gr[g], xv[g][k].w, gk[k].w));
                    }
                }
#pragma unroll
                for (int k = 0; k < NC4; ++k) {
                    const int b = 4 * lane + 256 * k;
                    if (b < D)
                        *reinterpret_cast<float4 *>(row + b) = make_float4(fmaf(step, gk[k].x, m[k].x), fmaf(step, gk[k].y, m[k].y),
                                                                           fmaf(step, gk[k].z, m[k].z), fmaf(step, gk[k].w, m[k].w));
                }
            }
        };
        if (qb > e0) {
            for (int r = r0 + wave; r < r1; r += KGE_WAVES_PER_BLOCK) do_row(r, std::true_type{});
        } else {
#pragma unroll RESCAL_UNR
            for (int r = r0 + wave; r < r1; r += KGE_WAVES_PER_BLOCK) do_row(r, std::false_type{});
        }
        // column products of this row block: the four wavefronts' sums in a fixed order
#pragma unroll
        for (int g = 0; g < EG; ++g)
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NC4; ++k)
                    *reinterpret_cast<float4 *>(&s_c[wave][4 * lane + 256 * k]) = which ? c2[g][k] : c1[g][k];
                __syncthreads();
                if (on[g]) {
                    float *out = (which ? a.c2p : a.c1p) + (e[g] * RESCAL_RBN + rb) * (int64_t)D;
                    for (int b = threadIdx.x; b < D; b += KGE_BLOCK) {
                        float sc = 0.f;
#pragma unroll
                        for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) sc += s_c[w][b];
                        out[b] = sc;
                    }
                }
            }
    }
    if (a.reg_part) {
        const float tot = reg ? block_sum(rv, red) : 0.f;
        if (threadIdx.x == 0) a.reg_part[(int64_t)u * RESCAL_RBN + rb] = a.reg_coef * tot * cnt;
    }
}

__global__ __launch_bounds__(KGE_BLOCK) void rescal_combine_kernel(RescalCombineArgs a) {     // one wavefront per edge
    const int64_t i = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= a.B) return;
    const int lane = threadIdx.x & 63, D = a.D;
    const float dp = a.dpos[i];
    const float *p1 = a.c1p + i * RESCAL_RBN * (int64_t)D, *p2 = a.c2p + i * RESCAL_RBN * (int64_t)D;
    for (int b = lane; b < D; b += 64) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < RESCAL_RBN; ++k) { s1 += p1[(int64_t)k * D + b]; s2 += p2[(int64_t)k * D + b]; }
        a.GH[i * D + b] = dp * a.V[i * D + b] + (a.neg_head ? 0.f : s2);
        a.GT[i * D + b] = dp * s1 + (a.neg_head ? s2 : 0.f);
    }
}

int launch_rescal_combine(const RescalCombineArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    hipLaunchKernelGGL(rescal_combine_kernel, dim3((a.B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s, a);
    return check_launch_r();
}

__global__ void rescal_reg_finalize_kernel(RescalUpdateArgs a) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    float v = 0.f;
    const int np = (a.c1p && a.c2p) ? RESCAL_RBN : RESCAL_RB;      // parts per relation as the pass that wrote them laid them out
    for (int k = 0; k < np; ++k) v += a.reg_part[(int64_t)u * np + k];
    if (a.reg_rel) a.reg_rel[u] = v;
    if (a.acc) {
        float *slot = &a.acc[3 * KGE_ACC_SLOTS + (int)((u + a.UE) & (KGE_ACC_SLOTS - 1))];
        if (a.UE + a.UR <= KGE_ACC_SLOTS) *slot += v; else atomicAdd(slot, v);
    }
}

int launch_rescal_update_rel(const RescalUpdateArgs &a, hipStream_t s) {
    if (a.UR == 0 || a.B == 0) return KGE_OK;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    const bool regx = reg && a.PV && a.rho && (a.neg_head || a.PW);      // the forward pass left the regulariser's products
    static_assert(RESCAL_RBN <= 64, "rescal_edge_sq_regx_kernel: one lane per row-block part of sum(R^2)");
    if (regx) hipLaunchKernelGGL(rescal_edge_sq_regx_kernel, dim3((a.B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s, a);
    else if (reg) hipLaunchKernelGGL(rescal_edge_sq_reg_kernel, dim3(a.B * RESCAL_RB), dim3(KGE_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(rescal_edge_sq_kernel, dim3((a.B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s, a);
    hipLaunchKernelGGL(rescal_rel_state_kernel, dim3((a.UR + 255) / 256), dim3(256), 0, s, a, (reg && !regx) ? RESCAL_RB : 1);
    const dim3 ga(a.UR * RESCAL_RB), ba(KGE_BLOCK), gn(a.UR * RESCAL_RBN);
    if (a.c1p && a.c2p) {                                  // backward products + update in one pass per unique relation
        if (a.D % 4 != 0 || a.D > 1024) return KGE_ERR_ARG;
        if (a.D <= 256) hipLaunchKernelGGL((rescal_rel_bwd_apply_kernel<1, RESCAL_EG>), gn, ba, 0, s, a);
        else if (a.D <= 512) hipLaunchKernelGGL((rescal_rel_bwd_apply_kernel<2, RESCAL_EG>), gn, ba, 0, s, a);
        else hipLaunchKernelGGL((rescal_rel_bwd_apply_kernel<4, 1>), gn, ba, 0, s, a);
    } else if (a.D % 4 == 0) {
        int tpr = 64;                                      // >= one wavefront per row: the row factors stay uniform
        while (tpr * 4 < a.D) tpr *= 2;
        hipLaunchKernelGGL(rescal_apply_vec_kernel, ga, ba, 0, s, a, tpr);
    } else if (a.D <= 256) hipLaunchKernelGGL(rescal_apply_kernel<1>, ga, ba, 0, s, a);
    else if (a.D <= 512) hipLaunchKernelGGL(rescal_apply_kernel<2>, ga, ba, 0, s, a);
    else hipLaunchKernelGGL(rescal_apply_kernel<4>, ga, ba, 0, s, a);
    if (a.reg_part && (a.reg_rel || a.acc))
        hipLaunchKernelGGL(rescal_reg_finalize_kernel, dim3((a.UR + 255) / 256), dim3(256), 0, s, a);
    return check_launch_r();
}
