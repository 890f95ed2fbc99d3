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
    const int D = a.D;
    const float *M = a.rel + (a.ridx ? a.ridx[i] : (int64_t)i) * (int64_t)D * D;
    auto vec = [&](const float *base, const int64_t *idx) -> const float * {
        return base ? base + (idx ? idx[i] : (int64_t)i) * (int64_t)D : nullptr;
    };
    const float *y1 = vec(a.y1, a.y1idx), *y2 = vec(a.y2, a.y2idx);
    const float *z1 = vec(a.z1, a.z1idx), *z2 = vec(a.z2, a.z2idx);
    const float *pd = vec(a.pd, a.pdidx);
    float pacc = 0.f;
    float y1v[NCH], y2v[NCH], c1[NCH], c2[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int b = lane + 64 * k;
        y1v[k] = (y1 && b < D) ? y1[b] : 0.f;
        y2v[k] = (y2 && b < D) ? y2[b] : 0.f;
        c1[k] = 0.f; c2[k] = 0.f;
    }
    for (int r = wave; r < D; r += KGE_WAVES_PER_BLOCK) {
        const float *row = M + (int64_t)r * D;
        const float zz1 = z1 ? z1[r] : 0.f, zz2 = z2 ? z2[r] : 0.f;
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int b = lane + 64 * k;
            const float m = b < D ? row[b] : 0.f;
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
        for (int b = threadIdx.x; b < D; b += KGE_BLOCK) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) { s1 += colsum[0][w][b]; s2 += colsum[1][w][b]; }
            if (a.c1) a.c1[(int64_t)i * D + b] = s1;
            if (a.c2) a.c2[(int64_t)i * D + b] = s2;
        }
    }
}

int launch_rescal_matvec(const RescalMatvecArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    const dim3 g(a.B), b(KGE_BLOCK);
    if (a.D <= 256) hipLaunchKernelGGL(rescal_matvec_kernel<4>, g, b, 0, s, a);
    else if (a.D <= 512) hipLaunchKernelGGL(rescal_matvec_kernel<8>, g, b, 0, s, a);
    else if (a.D <= 1024) hipLaunchKernelGGL(rescal_matvec_kernel<16>, g, b, 0, s, a);
    else return KGE_ERR_ARG;
    return check_launch_r();
}

// out_i = s1_i * u1_i (+ u2_i)   (vector combine: GH = dp * (M t) + M^T GA ...), one wavefront per edge
__global__ __launch_bounds__(KGE_BLOCK) void rescal_axpy_kernel(const float *s1, const float *u1, const float *u2, int B,
                                                                int D, float *out) {
    const int64_t i = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= B) return;
    const int lane = threadIdx.x & 63;
    const float c = s1 ? s1[i] : 1.f;
    for (int b = lane; b < D; b += 64)
        out[i * (int64_t)D + b] = c * u1[i * (int64_t)D + b] + (u2 ? u2[i * (int64_t)D + b] : 0.f);
}

int launch_rescal_axpy(const float *s1, const float *u1, const float *u2, int B, int D, float *out, hipStream_t s) {
    if (B == 0) return KGE_OK;
    hipLaunchKernelGGL(rescal_axpy_kernel, dim3((B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s,
                       s1, u1, u2, B, D, out);
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
// fused relation update: one workgroup per unique relation u with edges E_u (plan lists ur_ptr / ur_edge).
// Per edge e the traced-row gradient is g_e = dp_e h_e t_e^T + GA_e x_e^T + R  (R = regulariser gradient of
// the current M, one copy per traced row like the reference, general_models.py:572-576) and Adagrad
// (tensor_models.py:330-361) needs  S = sum_e mean(g_e^2)  BEFORE the row changes:
//   pass 1 (per edge, over all D*D elements): sum of squares -> S, deterministic block reduction;
//   pass 2: M -= lr * sum_e g_e / (sqrt(state + S) + eps).
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

__global__ __launch_bounds__(KGE_BLOCK) void rescal_update_rel_kernel(RescalUpdateArgs a) {
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    const int u = blockIdx.x;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int D = a.D;
    const int64_t n = (int64_t)D * D;
    const int64_t id = a.ur_id[u];
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    float *M = a.rel + id * n;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    auto row_of = [&](const int64_t *idx, int e) { return a.ent + idx[e] * (int64_t)D; };
    // pass 1: S = sum_e mean(g_e^2); regulariser value of the traced copies
    float S = 0.f, rv = 0.f;
    for (int q = e0; q < e1; ++q) {
        const int e = a.ur_edge[q];
        const float dp = a.dpos[e];
        const float *h = row_of(a.hidx, e), *t = row_of(a.tidx, e), *x = a.neg_head ? t : h;
        const float *ga = a.GA + (int64_t)e * D;
        float ss = 0.f;
        for (int64_t k = threadIdx.x; k < n; k += KGE_BLOCK) {
            const int r = (int)(k / D), b = (int)(k - (int64_t)r * D);
            float g = dp * h[r] * t[b] + ga[r] * x[b];
            if (reg) {
                const float m = M[k];
                g += reg_grad(m, a.reg_coef, a.reg_norm);
                if (q == e0) rv += reg_val(m, a.reg_norm);
            }
            ss = fmaf(g, g, ss);
        }
        S += block_sum(ss, red) / (float)n;
    }
    const float sN = a.rel_state[id] + S;
    const float sd = sqrtf(sN) + a.eps;
    // pass 2: apply the summed gradient
    const float cnt = (float)(e1 - e0);
    for (int64_t k = threadIdx.x; k < n; k += KGE_BLOCK) {
        const int r = (int)(k / D), b = (int)(k - (int64_t)r * D);
        const float m = M[k];
        float g = reg ? cnt * reg_grad(m, a.reg_coef, a.reg_norm) : 0.f;
        for (int q = e0; q < e1; ++q) {
            const int e = a.ur_edge[q];
            const float *h = row_of(a.hidx, e), *t = row_of(a.tidx, e), *x = a.neg_head ? t : h;
            g += a.dpos[e] * h[r] * t[b] + a.GA[(int64_t)e * D + r] * x[b];
        }
        M[k] = m + (-a.lr * g) / sd;
    }
    if (reg && (a.reg_rel || a.acc)) {
        const float tot = block_sum(rv, red);
        if (threadIdx.x == 0) {
            const float val = a.reg_coef * tot * cnt;
            if (a.reg_rel) a.reg_rel[u] = val;
            if (a.acc) {
                float *slot = &a.acc[3 * KGE_ACC_SLOTS + (int)((u + a.UE) & (KGE_ACC_SLOTS - 1))];
                if (a.UE + a.UR <= KGE_ACC_SLOTS) *slot += val; else atomicAdd(slot, val);
            }
        }
    } else if (a.reg_rel && threadIdx.x == 0) a.reg_rel[u] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) a.rel_state[id] = sN;
}

int launch_rescal_update_rel(const RescalUpdateArgs &a, hipStream_t s) {
    if (a.UR == 0) return KGE_OK;
    hipLaunchKernelGGL(rescal_update_rel_kernel, dim3(a.UR), dim3(KGE_BLOCK), 0, s, a);
    return check_launch_r();
}
