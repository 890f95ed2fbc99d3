// kge_neg_pair.hip - chunked negative scoring for the score functions that have NO GEMM form:
// TransE_l1 (th.cdist p=1, models/pytorch/score_fun.py:36-38) and RotatE (complex modulus of the
// broadcast difference, score_fun.py:526-531, 548-552) - plus a generic path for the GEMM models
// when their shapes are not 16-byte aligned (and a validation aid, KGE_FLAG_FORCE_PAIRWISE).
//
// These are VALU-bound pairwise reductions  n_ij = gamma - sum_k f(a_ik, b_jk)  and their
// gradients  GA_ik = sum_j W_ij * d n_ij/d a_ik ,  GN_jk = sum_i W_ij * d n_ij/d b_jk .
// Structure: LDS-tiled like an SGEMM micro-kernel; a workgroup (256 threads) owns a 32x32 output
// tile, each thread a 2x2 register micro-tile, operands staged through LDS transposed ([k][row])
// so that the inner loop reads float2 per operand; the reference instead materialises the
// [C,chunk,N,D] difference tensor (839 MB at the Freebase config).
#include "kge_common.hpp"

using namespace kge;

static inline int check_launch_p() {
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

#define PT 32          // tile edge (rows and cols of the output tile)
#define PK 32          // k-slab
#define PLD (PT + 2)   // padded leading dimension (keeps float2 alignment, breaks bank aliasing)

// pair functors ---------------------------------------------------------------------------
// REAL models accumulate f(a,b); CPLX models (RotatE) take (are, aim, bre, bim).
template <int MODEL> struct PairF;
template <> struct PairF<KGE_TRANSE_L1> {
    static constexpr bool cplx = false;
    __device__ static float acc(float a, float b) { return fabsf(a - b); }
    __device__ static float fin(float s, float gamma) { return gamma - s; }
    // d n / d a  (d n / d b is the negative)
    __device__ static float da(float a, float b) { return -sgnf(a - b); }
};
template <> struct PairF<KGE_TRANSE_L2> {   // direct form; W is pre-divided by the distance
    static constexpr bool cplx = false;
    __device__ static float acc(float a, float b) { const float u = a - b; return u * u; }
    __device__ static float fin(float s, float gamma) { return gamma - sqrtf(fmaxf(s, 1e-30f)); }
    __device__ static float da(float a, float b) { return -(a - b); }
};
template <> struct PairF<KGE_DISTMULT> {    // also ComplEx: plain dot product over the full row
    static constexpr bool cplx = false;
    __device__ static float acc(float a, float b) { return a * b; }
    __device__ static float fin(float s, float) { return s; }
};
template <> struct PairF<KGE_ROTATE> {
    static constexpr bool cplx = true;
    __device__ static float fin(float s, float gamma) { return gamma - s; }
};

// stage a [PT rows x PK k] slab of `rows` (row stride ld, starting at column k0) into LDS as
// dst[k][row]; rows beyond nrows / columns beyond kmax are zero-filled.
__device__ __forceinline__ void stage_rows(float (*dst)[PLD], const float *base, const int64_t *idx,
                                           int64_t row0, int nrows, int ld, int k0, int kmax,
                                           int col_off) {
    // 256 threads: thread t loads row t/8, columns (t%8)*4 .. +3
    const int t = threadIdx.x;
    const int rr = t >> 3, kk = (t & 7) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (rr < nrows) {
        const float *p = row_ptr(base, idx, row0 + rr, ld) + col_off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + kk + e;
            if (k < kmax) v[e] = p[k];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[kk + e][rr] = v[e];
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_pair_kernel(NegArgs a, int ti, int tj) {
    using F = PairF<MODEL>;
    __shared__ __attribute__((aligned(16))) float As[F::cplx ? 2 : 1][PK][PLD];
    __shared__ __attribute__((aligned(16))) float Bs[F::cplx ? 2 : 1][PK][PLD];
    const int tile = blockIdx.x;
    const int jt = tile % tj, it = (tile / tj) % ti, c = tile / (tj * ti);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int D = a.d_e;
    const int K = F::cplx ? D / 2 : D;
    const int i0 = it * PT, j0 = jt * PT;
    const int ni = min(PT, a.chunk - i0), nj = min(PT, a.N - j0);
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < K; k0 += PK) {
        __syncthreads();
        stage_rows(As[0], a.A, nullptr, (int64_t)c * a.chunk + i0, ni, D, k0, K, 0);
        stage_rows(Bs[0], a.nbase, a.nidx, (int64_t)c * a.N + j0, nj, D, k0, K, 0);
        if (F::cplx) {
            stage_rows(As[F::cplx ? 1 : 0], a.A, nullptr, (int64_t)c * a.chunk + i0, ni, D, k0, K, K);
            stage_rows(Bs[F::cplx ? 1 : 0], a.nbase, a.nidx, (int64_t)c * a.N + j0, nj, D, k0, K, K);
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < PK; ++k) {
            const float2 av = *reinterpret_cast<const float2 *>(&As[0][k][2 * ty]);
            const float2 bv = *reinterpret_cast<const float2 *>(&Bs[0][k][2 * tx]);
            if constexpr (F::cplx) {
                const float2 ai = *reinterpret_cast<const float2 *>(&As[1][k][2 * ty]);
                const float2 bi = *reinterpret_cast<const float2 *>(&Bs[1][k][2 * tx]);
                float dr, di;
                dr = av.x - bv.x; di = ai.x - bi.x; acc[0][0] += sqrtf(dr * dr + di * di);
                dr = av.x - bv.y; di = ai.x - bi.y; acc[0][1] += sqrtf(dr * dr + di * di);
                dr = av.y - bv.x; di = ai.y - bi.x; acc[1][0] += sqrtf(dr * dr + di * di);
                dr = av.y - bv.y; di = ai.y - bi.y; acc[1][1] += sqrtf(dr * dr + di * di);
            } else {
                acc[0][0] += F::acc(av.x, bv.x);
                acc[0][1] += F::acc(av.x, bv.y);
                acc[1][0] += F::acc(av.y, bv.x);
                acc[1][1] += F::acc(av.y, bv.y);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int i = i0 + 2 * ty + r, j = j0 + 2 * tx + s;
            if (i < a.chunk && j < a.N)
            {
                float v = F::fin(acc[r][s], a.gamma);
                if (a.clampv > 0.f) v = fminf(fmaxf(v, -a.clampv), a.clampv);
                a.S[((int64_t)c * a.chunk + i) * a.N + j] = v;
            }
        }
}

int launch_neg_fwd_pair(const NegArgs &a, hipStream_t s) {
    if (neg_bcast_supported(a.model, a.d_e)) return launch_neg_fwd_bcast(a, s);    // fast path (kge_neg_bcast.hip)
    const int ti = (a.chunk + PT - 1) / PT, tj = (a.N + PT - 1) / PT;
    const int nb = a.C * ti * tj;
    if (nb == 0) return KGE_OK;
    switch (a.model) {
        case KGE_TRANSE_L1:
            hipLaunchKernelGGL(neg_fwd_pair_kernel<KGE_TRANSE_L1>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj); break;
        case KGE_TRANSE_L2:
            hipLaunchKernelGGL(neg_fwd_pair_kernel<KGE_TRANSE_L2>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj); break;
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_SIMPLE: case KGE_RESCAL:
            hipLaunchKernelGGL(neg_fwd_pair_kernel<KGE_DISTMULT>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj); break;
        case KGE_ROTATE:
            hipLaunchKernelGGL(neg_fwd_pair_kernel<KGE_ROTATE>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj); break;
        default: return KGE_ERR_ARG;
    }
    return check_launch_p();
}

// ------------------------------------------------------------------------------------------
// backward.  OUT[r,k] = sum_s W(r,s) * psi(x_r[k], y_s[k])
//   GA: r = positive i, s = negative j, x = a, y = b, psi = d n/d a
//   GN: r = negative j, s = positive i, x = b, y = a, psi = d n/d b
// A workgroup owns a [32 rows x 32 k] output tile and loops over s in slabs of 32; W and y are
// staged through LDS, x lives in registers.
// ------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(KGE_BLOCK) void neg_bwd_pair_kernel(NegArgs a, int ti, int tj, int tk) {
    using F = PairF<MODEL>;
    constexpr bool DOT = (MODEL == KGE_DISTMULT);
    __shared__ __attribute__((aligned(16))) float Ws[PK][PLD];               // [s][r]
    __shared__ __attribute__((aligned(16))) float Ys[F::cplx ? 2 : 1][PK][PLD];   // [s][k]
    const int nGA = a.C * ti * tk;
    int tile = blockIdx.x;
    const bool isGA = tile < nGA;
    if (!isGA) tile -= nGA;
    const int tr = isGA ? ti : tj;
    const int kt = tile % tk, rt = (tile / tk) % tr, c = tile / (tk * tr);
    const int D = a.d_e;
    const int K = F::cplx ? D / 2 : D;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int R = isGA ? a.chunk : a.N;          // rows of this product (per chunk)
    const int S = isGA ? a.N : a.chunk;          // reduction length
    const int r0 = rt * PT, k0 = kt * PK;
    const float *Wc = a.W + (int64_t)c * a.chunk * a.N;
    // x micro-tile (2 rows x 2 k)
    float xr[2][2], xi[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = min(r0 + 2 * ty + r, R - 1);
        const float *xp = isGA ? a.A + ((int64_t)c * a.chunk + row) * D
                               : row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + row, D);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = k0 + 2 * tx + s;
            xr[r][s] = (k < K) ? xp[k] : 0.f;
            xi[r][s] = (F::cplx && k < K) ? xp[K + k] : 0.f;
        }
    }
    float outr[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, outi[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float wsum[2] = {0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += PK) {
        __syncthreads();
        {   // stage W tile: Ws[s][r] = W(r0 + r, s0 + s)
            const int t = threadIdx.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lin = t + e * KGE_BLOCK;      // 0..1023
                int rr, ss;
                if (isGA) { rr = lin >> 5; ss = lin & 31; }   // contiguous along s (= j)
                else      { ss = lin >> 5; rr = lin & 31; }   // contiguous along r (= j)
                const int gr = r0 + rr, gs = s0 + ss;
                float v = 0.f;
                if (gr < R && gs < S) {
                    const int i = isGA ? gr : gs, j = isGA ? gs : gr;
                    v = Wc[(int64_t)i * a.N + j];
                }
                Ws[ss][rr] = v;
            }
        }
        {   // stage y slab transposed to [s][k]: thread t -> row s = t/8, k = (t%8)*4..+3
            const int t = threadIdx.x;
            const int ss = t >> 3, kk = (t & 7) * 4;
            const int gs = s0 + ss;
            float vr[4] = {0.f, 0.f, 0.f, 0.f}, vi[4] = {0.f, 0.f, 0.f, 0.f};
            if (gs < S) {
                const float *yp = isGA ? row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + gs, D)
                                       : a.A + ((int64_t)c * a.chunk + gs) * D;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + kk + e;
                    if (k < K) { vr[e] = yp[k]; if (F::cplx) vi[e] = yp[K + k]; }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Ys[0][ss][kk + e] = vr[e];
                if (F::cplx) Ys[F::cplx ? 1 : 0][ss][kk + e] = vi[e];
            }
        }
        __syncthreads();
        const int smax = min(PK, S - s0);
        for (int s = 0; s < smax; ++s) {
            const float2 wv = *reinterpret_cast<const float2 *>(&Ws[s][2 * ty]);
            const float2 yv = *reinterpret_cast<const float2 *>(&Ys[0][s][2 * tx]);
            const float w2[2] = {wv.x, wv.y}, y2[2] = {yv.x, yv.y};
            if constexpr (F::cplx) {
                const float2 yiv = *reinterpret_cast<const float2 *>(&Ys[1][s][2 * tx]);
                const float yi2[2] = {yiv.x, yiv.y};
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // difference is always (a - b); x is a for GA and b for GN
                        const float dr = isGA ? xr[r][q] - y2[q] : y2[q] - xr[r][q];
                        const float di = isGA ? xi[r][q] - yi2[q] : yi2[q] - xi[r][q];
                        const float mm = sqrtf(dr * dr + di * di);
                        const float iv = mm > 0.f ? w2[r] / mm : 0.f;
                        // d n/d a = -(a-b)/|a-b| ; d n/d b = +(a-b)/|a-b|
                        outr[r][q] += (isGA ? -dr : dr) * iv;
                        outi[r][q] += (isGA ? -di : di) * iv;
                    }
            } else if constexpr (DOT) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) outr[r][q] += w2[r] * y2[q];
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // F::da is d n/d a evaluated at (a, b); for GN the sign flips
                        const float g = isGA ? F::da(xr[r][q], y2[q]) : -F::da(y2[q], xr[r][q]);
                        outr[r][q] += w2[r] * g;
                    }
            }
            wsum[0] += w2[0]; wsum[1] += w2[1];
        }
    }
    (void)wsum;
    const bool reg = (!isGA) && a.reg_coef > 0.f && a.reg_norm > 0;
    float *O = isGA ? a.GA : a.GN;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r0 + 2 * ty + r;
        if (row >= R) continue;
        const int64_t grow = (int64_t)c * R + row;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = k0 + 2 * tx + q;
            if (k >= K) continue;
            float vr = outr[r][q], vi = outi[r][q];
            if (reg) {
                vr += reg_grad(xr[r][q], a.reg_coef, a.reg_norm);
                if (F::cplx) vi += reg_grad(xi[r][q], a.reg_coef, a.reg_norm);
            }
            O[grow * D + k] = vr;
            if (F::cplx) O[grow * D + K + k] = vi;
        }
    }
}

int launch_neg_bwd_pair(const NegArgs &a, hipStream_t s) {
    if (neg_bcast_supported(a.model, a.d_e)) return launch_neg_bwd_bcast(a, s);
    const int ti = (a.chunk + PT - 1) / PT, tj = (a.N + PT - 1) / PT;
    const int K = (a.model == KGE_ROTATE) ? a.d_e / 2 : a.d_e;
    const int tk = (K + PK - 1) / PK;
    const int nb = a.C * (ti + tj) * tk;
    if (nb == 0) return KGE_OK;
    switch (a.model) {
        case KGE_TRANSE_L1:
            hipLaunchKernelGGL(neg_bwd_pair_kernel<KGE_TRANSE_L1>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj, tk); break;
        case KGE_TRANSE_L2:
            hipLaunchKernelGGL(neg_bwd_pair_kernel<KGE_TRANSE_L2>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj, tk); break;
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_SIMPLE: case KGE_RESCAL:
            hipLaunchKernelGGL(neg_bwd_pair_kernel<KGE_DISTMULT>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj, tk); break;
        case KGE_ROTATE:
            hipLaunchKernelGGL(neg_bwd_pair_kernel<KGE_ROTATE>, dim3(nb), dim3(KGE_BLOCK), 0, s, a, ti, tj, tk); break;
        default: return KGE_ERR_ARG;
    }
    return check_launch_p();
}
