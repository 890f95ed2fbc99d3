// kge_transr.hip - TransR (models/pytorch/score_fun.py:110-220): every relation owns a projection matrix
// P [ent_dim x rel_dim] (a THIRD table, projection_emb) and
//     positive score   p    = gamma - |h P + r - t P|_1                                 (edge_func, :122-127)
//     negative score   n_ij = gamma - |e_j P_i - (x_i P_i - r_i)|_1                     (create_neg, :199-219;
//                             x = tail in head mode, head in tail mode - BOTH closures subtract r)
// i.e. every negative of a chunk is projected with every positive's matrix: B batched [N x D_e] x [D_e x D_r]
// products per step (32 GMAC at the FB15k shape) - real GEMM work, done here by one LDS-tiled fp32-MFMA tile
// routine (64 x 64 output tile per workgroup, v_mfma_f32_16x16x4_f32) shared by the three products:
//     forward   Y_i  = Neg_c  P_i              -> L1 epilogue (scores) + sign(Y - q) as int8 for the backward
//     backward  GN_c = sum_i dY_i P_i^T         (K runs over the chunk's positives x D_r)
//               GP_i = Neg_c^T dY_i + x_i (x) dq_i   (per-edge gradient of the projection matrix, materialised:
//                       it is full rank and Adagrad needs its mean square per traced row)
// with dY_ij = -W_ij sign(Y_ij - q_i), dq_i = -sum_j dY_ij.  The reference materialises Y [B, N, D_r] fp32
// (320 MB at the FB15k shape); here only the sign byte survives the forward (80 MB).
#include "kge_common.hpp"

using namespace kge;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define TR_T 64          // output tile edge
#define TR_K 16          // reduction slab
#define TR_LD (TR_T + 4)

static inline int check_launch_t() { return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; }

// Operand loads may return RAW data (sign bytes + a scale) that is converted to fp32 only when it is written to LDS, one slab
// later: a conversion inside the load would make the wavefront wait for the load before the slab's MFMAs instead of under them
// (seen in the ISA as a vmcnt(0) right after the prefetch; profiles/r05_transr_rescal.txt).
struct Sgn4 { int w; float s; };                 // four sign bytes (one aligned 32-bit word) times a scale
struct Sgn1 { signed char z; float s; };
__device__ __forceinline__ float4 cvt_op(const float4 &v) { return v; }
__device__ __forceinline__ float cvt_op(float v) { return v; }
__device__ __forceinline__ float4 cvt_op(const Sgn4 &v) {
    return make_float4(v.s * (float)(signed char)(v.w & 0xff), v.s * (float)(signed char)((v.w >> 8) & 0xff),
                       v.s * (float)(signed char)((v.w >> 16) & 0xff), v.s * (float)(signed char)(v.w >> 24));
}
__device__ __forceinline__ float cvt_op(const Sgn1 &v) { return v.s * (float)v.z; }

__device__ __forceinline__ float block_sum_t(float v, float *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) s += red[w];
    return s;
}

// one K sweep of a 64 x 64 tile: acc[ct][r] = C[wave*16 + 4*(lane/16) + r][ct*16 + lane%16].
// loadA(row, k0, kk) / loadB(k0, kk, col) return the (zero-padded) operand elements at reduction index k0 + kk (k0 = the slab's
// first index, uniform over the workgroup: index arithmetic on it stays on the scalar unit); AKF / BKF say whether consecutive
// threads should walk the reduction index (operand rows contiguous along k) or the tile row / column.
// Software pipeline: the global loads of slab s+1 are issued BEFORE the MFMAs of slab s and land in the other LDS
// buffer after them - one barrier per slab, load latency under the matrix work (the single-buffered version
// exposed a full global-load round trip per 16-deep slab: 50 TFLOP/s).
// NCT: the 16-column blocks of the tile this wavefront multiplies (0: none - its rows, or the whole tile, hold no real data; it
// still takes part in the loads and barriers).  A 200-wide operand fills 13 of the 16 blocks of its four tiles.
template <bool AKF, bool BKF, int NCT, class LA, class LB>
__device__ __forceinline__ void tile_sweep_n(f32x4 (&acc)[4], int Ktot, LA loadA, LB loadB, float (*As)[TR_K][TR_LD],
                                             float (*Bs)[TR_K][TR_LD]) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    decltype(loadA(0, 0, 0)) av[4];
    decltype(loadB(0, 0, 0)) bv[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lin = t + KGE_BLOCK * e;
            const int ar = AKF ? lin >> 4 : lin & 63, ak = AKF ? lin & 15 : lin >> 6;
            const int bc = BKF ? lin >> 4 : lin & 63, bk = BKF ? lin & 15 : lin >> 6;
            av[e] = loadA(ar, k0, ak);
            bv[e] = loadB(k0, bk, bc);
        }
    };
    if (Ktot > 0) gload(0);                    // (an empty reduction must not touch the operands at all)
    int buf = 0;
    __syncthreads();                           // the caller's previous sweep has left the LDS buffers
    for (int k0 = 0; k0 < Ktot; k0 += TR_K, buf ^= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lin = t + KGE_BLOCK * e;
            const int ar = AKF ? lin >> 4 : lin & 63, ak = AKF ? lin & 15 : lin >> 6;
            const int bc = BKF ? lin >> 4 : lin & 63, bk = BKF ? lin & 15 : lin >> 6;
            As[buf][ak][ar] = cvt_op(av[e]);
            Bs[buf][bk][bc] = cvt_op(bv[e]);
        }
        __syncthreads();
        if (k0 + TR_K < Ktot) gload(k0 + TR_K);
        if constexpr (NCT > 0) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float a = As[buf][4 * s4 + q][wave * 16 + m];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = MFMA16(a, Bs[buf][4 * s4 + q][ct * 16 + m], acc[ct]);
            }
        }
    }
}

// Same sweep with 16-byte operand loads (D_e % 4 == 0 and D_r % 4 == 0): every thread fetches ONE float4 (or four
// sign bytes as one 32-bit word) per operand and slab instead of four scalars.  AROW / BROW: the four elements are
// consecutive tile rows (columns) at one k - a float4 store into LDS; otherwise they are consecutive k of one row.
template <bool AROW, bool BROW, int NCT, class LA, class LB>
__device__ __forceinline__ void tile_sweep4_n(f32x4 (&acc)[4], int Ktot, LA loadA4, LB loadB4, float (*As)[TR_K][TR_LD],
                                              float (*Bs)[TR_K][TR_LD]) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    const int ai = AROW ? (t & 15) * 4 : t >> 2, ak = AROW ? t >> 4 : (t & 3) * 4;
    const int bi = BROW ? (t & 15) * 4 : t >> 2, bk = BROW ? t >> 4 : (t & 3) * 4;
    decltype(loadA4(0, 0, 0)) ra{};
    decltype(loadB4(0, 0, 0)) rb{};
    if (Ktot > 0) { ra = loadA4(ai, 0, ak); rb = loadB4(0, bk, bi); }
    int buf = 0;
    __syncthreads();
    for (int k0 = 0; k0 < Ktot; k0 += TR_K, buf ^= 1) {
        const float4 av = cvt_op(ra), bv = cvt_op(rb);
        if constexpr (AROW) *reinterpret_cast<float4 *>(&As[buf][ak][ai]) = av;
        else { As[buf][ak][ai] = av.x; As[buf][ak + 1][ai] = av.y; As[buf][ak + 2][ai] = av.z; As[buf][ak + 3][ai] = av.w; }
        if constexpr (BROW) *reinterpret_cast<float4 *>(&Bs[buf][bk][bi]) = bv;
        else { Bs[buf][bk][bi] = bv.x; Bs[buf][bk + 1][bi] = bv.y; Bs[buf][bk + 2][bi] = bv.z; Bs[buf][bk + 3][bi] = bv.w; }
        __syncthreads();
        if (k0 + TR_K < Ktot) { ra = loadA4(ai, k0 + TR_K, ak); rb = loadB4(k0 + TR_K, bk, bi); }
        if constexpr (NCT > 0) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float a = As[buf][4 * s4 + q][wave * 16 + m];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = MFMA16(a, Bs[buf][4 * s4 + q][ct * 16 + m], acc[ct]);
            }
        }
    }
}
// the sweep for `nct` real column blocks (wavefront-uniform; every instance runs the same loads and barriers)
template <bool AKF, bool BKF, class LA, class LB>
__device__ __forceinline__ void tile_sweep(f32x4 (&acc)[4], int Ktot, LA loadA, LB loadB, float (*As)[TR_K][TR_LD],
                                           float (*Bs)[TR_K][TR_LD], int nct = 4) {
    switch (nct) {
    case 0: tile_sweep_n<AKF, BKF, 0>(acc, Ktot, loadA, loadB, As, Bs); break;
    case 1: tile_sweep_n<AKF, BKF, 1>(acc, Ktot, loadA, loadB, As, Bs); break;
    case 2: tile_sweep_n<AKF, BKF, 2>(acc, Ktot, loadA, loadB, As, Bs); break;
    case 3: tile_sweep_n<AKF, BKF, 3>(acc, Ktot, loadA, loadB, As, Bs); break;
    default: tile_sweep_n<AKF, BKF, 4>(acc, Ktot, loadA, loadB, As, Bs); break;
    }
}
template <bool AROW, bool BROW, class LA, class LB>
__device__ __forceinline__ void tile_sweep4(f32x4 (&acc)[4], int Ktot, LA loadA4, LB loadB4, float (*As)[TR_K][TR_LD],
                                            float (*Bs)[TR_K][TR_LD], int nct = 4) {
    switch (nct) {
    case 0: tile_sweep4_n<AROW, BROW, 0>(acc, Ktot, loadA4, loadB4, As, Bs); break;
    case 1: tile_sweep4_n<AROW, BROW, 1>(acc, Ktot, loadA4, loadB4, As, Bs); break;
    case 2: tile_sweep4_n<AROW, BROW, 2>(acc, Ktot, loadA4, loadB4, As, Bs); break;
    case 3: tile_sweep4_n<AROW, BROW, 3>(acc, Ktot, loadA4, loadB4, As, Bs); break;
    default: tile_sweep4_n<AROW, BROW, 4>(acc, Ktot, loadA4, loadB4, As, Bs); break;
    }
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------------------------
// positive score, sign vector, q = x P - r   (one wavefront per edge; hp / tp from the projection pass)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KGE_BLOCK) void transr_pos_kernel(TransRArgs a) {
    const int64_t i = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= a.B) return;
    const int lane = threadIdx.x & 63, Dr = a.Dr;
    const float *hp = a.HP + i * Dr, *tp = a.TP + i * Dr, *r = a.rel + a.rel_ids[i] * (int64_t)Dr;
    float s = 0.f;
    for (int d = lane; d < Dr; d += 64) {
        const float u = hp[d] + r[d] - tp[d];
        s += fabsf(u);
        a.SG[i * Dr + d] = (u > 0.f) ? 1.f : ((u < 0.f) ? -1.f : 0.f);
        a.Q[i * Dr + d] = (a.neg_head ? tp[d] : hp[d]) - r[d];
    }
    s = wave_sum(s);
    if (lane == 0) a.P[i] = a.gamma - s;
}

// ---------------------------------------------------------------------------------------------
// forward: workgroup = (positive i, block of 64 negatives); loops the D_r column tiles
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(KGE_BLOCK) void transr_fwd_kernel(TransRArgs a, int nJB) {
    __shared__ float As[2][TR_K][TR_LD], Bs[2][TR_K][TR_LD];
    __shared__ int64_t rowoff[TR_T];
    const int i = blockIdx.x / nJB, j0 = (blockIdx.x % nJB) * TR_T;
    const int c = i / a.chunk, De = a.De, Dr = a.Dr, N = a.N;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    if (t < TR_T) rowoff[t] = (j0 + t < N) ? a.neg_ids[(int64_t)c * N + j0 + t] * (int64_t)De : -1;
    __syncthreads();
    const float *Pi = a.proj + a.rel_ids[i] * (int64_t)De * Dr;
    const float *Qi = a.Q + (int64_t)i * Dr;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int dr0 = 0; dr0 < Dr; dr0 += TR_T) {
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nct = min(4, (Dr - dr0 + 15) / 16);
        const bool won = j0 + wave * 16 < N;
        if constexpr (VEC)
            tile_sweep4<false, true>(acc, De,
                [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; const int64_t o = rowoff[row];
                                      return (o >= 0 && k < De) ? *reinterpret_cast<const float4 *>(a.ent + o + k) : f4zero(); },
                [&](int k0_, int kk_, int col) { const int k = k0_ + kk_; return (k < De && dr0 + col < Dr) ? *reinterpret_cast<const float4 *>(Pi + (int64_t)k * Dr + dr0 + col)
                                                                        : f4zero(); }, As, Bs, won ? nct : 0);
        else
        tile_sweep<true, false>(acc, De,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; const int64_t o = rowoff[row]; return (o >= 0 && k < De) ? a.ent[o + k] : 0.f; },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_; return (k < De && dr0 + col < Dr) ? Pi[(int64_t)k * Dr + dr0 + col] : 0.f; }, As, Bs, won ? nct : 0);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int col = dr0 + ct * 16 + m;
            const float qv = col < Dr ? Qi[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + wave * 16 + 4 * q + r;
                const float v = acc[ct][r] - qv;
                if (col < Dr && j < N) {
                    d[r] += fabsf(v);
                    if (a.Z) a.Z[((int64_t)i * N + j) * Dr + col] = (signed char)((v > 0.f) - (v < 0.f));
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) d[r] += __shfl_xor(d[r], o, 64);
        const int j = j0 + wave * 16 + 4 * q + r;
        if (m == 0 && j < N) a.S[(int64_t)i * N + j] = a.gamma - d[r];
    }
}

// dq_i = -sum_j dY_ij = sum_j W_ij sign(Y_ij - q_i).  One workgroup per positive: the wavefronts split the
// negatives, every lane owns 4 consecutive columns (one 32-bit load = 4 sign bytes), partial sums are added in
// a fixed order through LDS.
__global__ __launch_bounds__(KGE_BLOCK) void transr_dq_kernel(TransRArgs a) {
    __shared__ float part[KGE_WAVES_PER_BLOCK][1024];
    const int64_t i = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, Dr = a.Dr, N = a.N;
    const bool vec = (Dr & 3) == 0;
    for (int d0 = 0; d0 < Dr; d0 += 256) {
        const int d = d0 + lane * 4;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (d < Dr) {
#pragma unroll 4
            for (int j = wave; j < N; j += KGE_WAVES_PER_BLOCK) {
                const float w = a.S[i * N + j];
                const signed char *z = a.Z + (i * N + j) * Dr + d;
                if (vec) {
                    const int pk = *reinterpret_cast<const int *>(z);
                    s0 = fmaf(w, (float)(signed char)(pk & 0xff), s0);
                    s1 = fmaf(w, (float)(signed char)((pk >> 8) & 0xff), s1);
                    s2 = fmaf(w, (float)(signed char)((pk >> 16) & 0xff), s2);
                    s3 = fmaf(w, (float)(signed char)((pk >> 24) & 0xff), s3);
                } else {
                    s0 = fmaf(w, (float)z[0], s0);
                    if (d + 1 < Dr) s1 = fmaf(w, (float)z[1], s1);
                    if (d + 2 < Dr) s2 = fmaf(w, (float)z[2], s2);
                    if (d + 3 < Dr) s3 = fmaf(w, (float)z[3], s3);
                }
            }
        }
        __syncthreads();
        if (d < Dr) { part[wave][d] = s0; if (d + 1 < 1024) part[wave][d + 1] = s1; if (d + 2 < 1024) part[wave][d + 2] = s2;
                      if (d + 3 < 1024) part[wave][d + 3] = s3; }
        __syncthreads();
        for (int k = threadIdx.x; k < 256 && d0 + k < Dr; k += KGE_BLOCK) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KGE_WAVES_PER_BLOCK; ++w) v += part[w][d0 + k];
            a.DQ[i * Dr + d0 + k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GN_c[j][de] = sum_{i in chunk} sum_dr dY_i[j][dr] P_i[de][dr]    workgroup = (chunk, 64 negatives, 64 de)
// ---------------------------------------------------------------------------------------------
// The chunk's positives are split into a.nG groups (split-K): with 4-5 chunks there are only ~64 output tiles,
// far too few workgroups; every group writes its own partial tile and transr_gn_reduce_kernel adds the groups
// in a fixed order (deterministic, no atomics).
template <bool VEC>
__global__ __launch_bounds__(KGE_BLOCK) void transr_gn_kernel(TransRArgs a, int nJB, int nEB) {
    __shared__ float As[2][TR_K][TR_LD], Bs[2][TR_K][TR_LD];
    extern __shared__ char gn_dyn[];             // per positive of the group: its projection matrix's offset, -W_ij of the 64 negatives
    const int g = blockIdx.x % a.nG;
    const int blk = blockIdx.x / a.nG;
    const int eb = blk % nEB, jb = (blk / nEB) % nJB, c = blk / (nEB * nJB);
    const int j0 = jb * TR_T, de0 = eb * TR_T, De = a.De, Dr = a.Dr, N = a.N, chunk = a.chunk;
    const int DrP = (Dr + TR_K - 1) / TR_K * TR_K;              // every positive contributes whole slabs
    const int ipg = (chunk + a.nG - 1) / a.nG, i0 = g * ipg, i1 = min(chunk, i0 + ipg);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    // (the sweep's operand loads would otherwise each wait for a dependent id / weight load of their own, once per slab)
    int64_t *s_pb = reinterpret_cast<int64_t *>(gn_dyn);
    float *s_w = reinterpret_cast<float *>(s_pb + ipg);
    for (int k = t; k < i1 - i0; k += KGE_BLOCK) s_pb[k] = a.rel_ids[(int64_t)c * chunk + i0 + k] * (int64_t)De * Dr;
    for (int k = t; k < (i1 - i0) * TR_T; k += KGE_BLOCK) {
        const int il = k / TR_T, j = j0 + (k % TR_T);
        s_w[k] = j < N ? -a.S[((int64_t)c * chunk + i0 + il) * N + j] : 0.f;
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nct = min(4, (De - de0 + 15) / 16);
    const bool won = j0 + wave * 16 < N;
    if constexpr (VEC)
        tile_sweep4<false, false>(acc, max(0, i1 - i0) * DrP,
            [&](int row, int k0_, int kk_) {          // 4 consecutive d_r of one (positive, negative): one word of sign bytes
                const int il = k0_ / DrP, dr = k0_ % DrP + kk_, j = j0 + row;
                if (j >= N || dr >= Dr) return Sgn4{0, 0.f};
                const int64_t ij = ((int64_t)c * chunk + i0 + il) * N + j;
                return Sgn4{*reinterpret_cast<const int *>(a.Z + ij * Dr + dr), s_w[il * TR_T + row]};
            },
            [&](int k0_, int kk_, int col) {          // P_i[de][dr .. dr+3]
                const int il = k0_ / DrP, dr = k0_ % DrP + kk_, de = de0 + col;
                if (de >= De || dr >= Dr) return f4zero();
                return *reinterpret_cast<const float4 *>(a.proj + s_pb[il] + (int64_t)de * Dr + dr);
            }, As, Bs, won ? nct : 0);
    else
    tile_sweep<true, true>(acc, max(0, i1 - i0) * DrP,
        [&](int row, int k0_, int kk_) {
            const int il = k0_ / DrP, dr = k0_ % DrP + kk_, j = j0 + row;
            if (j >= N || dr >= Dr) return Sgn1{0, 0.f};
            const int64_t ij = ((int64_t)c * chunk + i0 + il) * N + j;
            return Sgn1{a.Z[ij * Dr + dr], s_w[il * TR_T + row]};
        },
        [&](int k0_, int kk_, int col) {
            const int il = k0_ / DrP, dr = k0_ % DrP + kk_, de = de0 + col;
            if (de >= De || dr >= Dr) return 0.f;
            return a.proj[s_pb[il] + (int64_t)de * Dr + dr];
        }, As, Bs, won ? nct : 0);
    float *out = a.GNp + (int64_t)g * a.C * N * De;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int de = de0 + ct * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wave * 16 + 4 * q + r;
            if (de < De && j < N) out[((int64_t)c * N + j) * De + de] = acc[ct][r];
        }
    }
}

// GN = sum over the groups (fixed order) + regulariser of the negative row copy
__global__ __launch_bounds__(KGE_BLOCK) void transr_gn_reduce_kernel(TransRArgs a) {
    const int64_t row = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t rows = (int64_t)a.C * a.N;
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, De = a.De;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0 && !a.nd_chunk;      // (neg_deg_sample: added by the update kernel, sampled rows only)
    const float *x = a.ent + a.neg_ids[row] * (int64_t)De;
    for (int d = lane; d < De; d += 64) {
        float v = 0.f;
        int g = 0;
        for (; g + 8 <= a.nG; g += 8) {               // eight groups' partials in flight (up to 64 groups; same order of additions)
            float pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pv[k] = a.GNp[((int64_t)(g + k) * rows + row) * De + d];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += pv[k];
        }
        for (; g < a.nG; ++g) v += a.GNp[((int64_t)g * rows + row) * De + d];
        if (reg) v += reg_grad(x[d], a.reg_coef, a.reg_norm);
        a.GN[row * De + d] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// GP_i[de][dr] = sum_j Neg_j[de] dY_ij[dr] + x_i[de] dq_i[dr]     workgroup = (positive, 64 de, 64 dr)
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(KGE_BLOCK) void transr_gp_kernel(TransRArgs a, int nEB, int nRB) {
    __shared__ float As[2][TR_K][TR_LD], Bs[2][TR_K][TR_LD];
    extern __shared__ char gp_dyn[];             // per negative of the chunk: its row's offset in the entity table, -W_ij
    const int rb = blockIdx.x % nRB, eb = (blockIdx.x / nRB) % nEB, i = blockIdx.x / (nRB * nEB);
    const int de0 = eb * TR_T, dr0 = rb * TR_T, De = a.De, Dr = a.Dr, N = a.N;
    const int c = i / a.chunk;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    int64_t *s_off = reinterpret_cast<int64_t *>(gp_dyn);
    float *s_w = reinterpret_cast<float *>(s_off + N);
    for (int k = t; k < N; k += KGE_BLOCK) {
        s_off[k] = a.neg_ids[(int64_t)c * N + k] * (int64_t)De;
        s_w[k] = -a.S[(int64_t)i * N + k];
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nct = min(4, (Dr - dr0 + 15) / 16);
    const bool won = de0 + wave * 16 < De;
    if constexpr (VEC)
        tile_sweep4<true, true>(acc, N,
            [&](int row, int k0_, int kk_) {          // Neg_k[de0 + row .. +3]
                const int k = k0_ + kk_;
                return (k < N && de0 + row < De) ? *reinterpret_cast<const float4 *>(a.ent + s_off[k] + de0 + row) : f4zero();
            },
            [&](int k0_, int kk_, int col) {          // dY_ik[dr0 + col .. +3]: one word of sign bytes
                const int k = k0_ + kk_;
                if (k >= N || dr0 + col >= Dr) return Sgn4{0, 0.f};
                return Sgn4{*reinterpret_cast<const int *>(a.Z + ((int64_t)i * N + k) * Dr + dr0 + col), s_w[k]};
            }, As, Bs, won ? nct : 0);
    else
    tile_sweep<false, false>(acc, N,
        [&](int row, int k0_, int kk_) {
            const int k = k0_ + kk_;
            return (k < N && de0 + row < De) ? a.ent[s_off[k] + de0 + row] : 0.f;
        },
        [&](int k0_, int kk_, int col) {
            const int k = k0_ + kk_;
            if (k >= N || dr0 + col >= Dr) return Sgn1{0, 0.f};
            return Sgn1{a.Z[((int64_t)i * N + k) * Dr + dr0 + col], s_w[k]};
        }, As, Bs, won ? nct : 0);
    const float *x = a.ent + (a.neg_head ? a.t_gid[i] : a.h_gid[i]) * (int64_t)De;
    const float *dq = a.DQ + (int64_t)i * Dr;
    float *G = a.GP + (int64_t)i * De * Dr;
    float ss = 0.f;                               // Adagrad needs mean(GP_i^2): summed here, tile by tile, instead of re-reading GP
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int dr = dr0 + ct * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int de = de0 + wave * 16 + 4 * q + r;
            if (de < De && dr < Dr) {
                const float v = acc[ct][r] + x[de] * dq[dr];
                G[(int64_t)de * Dr + dr] = v;
                ss = fmaf(v, v, ss);
            }
        }
    }
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    ss = block_sum_t(ss, red);
    if (t == 0) a.gs1p[((int64_t)i * nEB + eb) * nRB + rb] = ss;
}

#include "kge_transr_wide.hpp"

// relation-vector gradient per edge: GR_i = -dp_i s_i - dq_i (+ regulariser of the traced copy)
__global__ __launch_bounds__(KGE_BLOCK) void transr_gr_kernel(TransRArgs a) {
    const int64_t i = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= a.B) return;
    const int lane = threadIdx.x & 63, Dr = a.Dr;
    const float dp = a.dpos[i];
    const float *r = a.rel + a.rel_ids[i] * (int64_t)Dr;
    const bool reg = a.reg_coef > 0.f && a.reg_norm > 0;
    for (int d = lane; d < Dr; d += 64) {
        float g = -dp * a.SG[i * Dr + d] - a.DQ[i * Dr + d];
        if (reg) g += reg_grad(r[d], a.reg_coef, a.reg_norm);
        a.GR[i * Dr + d] = g;
    }
}

// ---------------------------------------------------------------------------------------------
// projection-table Adagrad (score_func.update -> projection_emb.update, score_fun.py:173-174): two traces per
// step over the same relation ids - trace 0 from prepare() (G0_e = dp_e (t_e - h_e) (x) s_e, rank 1), trace 1
// from the neg-prepare closure (G1_e = GP_e above).  Like tensor_models.py:330-361 per trace:
//   st0 = st + sum_e mean(G0_e^2), P -= lr sum_e G0_e / (sqrt(st0) + eps);  st1 = st0 + sum_e mean(G1_e^2), ...
// both gradients were taken before any update, so one pass applies both.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(KGE_BLOCK) void transr_proj_sq_kernel(TransRArgs a, int ntiles) {      // one wavefront per edge
    const int64_t e = (int64_t)blockIdx.x * KGE_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (e >= a.B) return;
    const int lane = threadIdx.x & 63, De = a.De, Dr = a.Dr;
    const float n = (float)De * (float)Dr;
    const float *h = a.ent + a.h_gid[e] * (int64_t)De, *t = a.ent + a.t_gid[e] * (int64_t)De;
    float dd = 0.f, sg = 0.f, s1 = 0.f;
    for (int k = lane; k < De; k += 64) { const float u = t[k] - h[k]; dd = fmaf(u, u, dd); }
    for (int k = lane; k < Dr; k += 64) { const float v = a.SG[e * Dr + k]; sg = fmaf(v, v, sg); }
    for (int k = lane; k < ntiles; k += 64) s1 += a.gs1p[e * ntiles + k];      // (tile sums of GP_e^2 from transr_gp_kernel)
    dd = wave_sum(dd); sg = wave_sum(sg); s1 = wave_sum(s1);
    if (lane == 0) {
        const float dp = a.dpos[e];
        a.gs0[e] = dp * dp * dd * sg / n;
        a.gs1[e] = s1 / n;
    }
}

__global__ void transr_proj_state_kernel(TransRArgs a) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int64_t id = a.ur_id[u];
    float S0 = 0.f, S1 = 0.f;
    for (int q = a.ur_ptr[u]; q < a.ur_ptr[u + 1]; ++q) { const int e = a.ur_edge[q]; S0 += a.gs0[e]; S1 += a.gs1[e]; }
    const float st0 = a.proj_state[id] + S0, st1 = st0 + S1;
    a.proj_state[id] = st1;
    a.k0[u] = -a.lr / (sqrtf(st0) + a.eps);
    a.k1[u] = -a.lr / (sqrtf(st1) + a.eps);
}

#define TR_RB 8      // row blocks per projection matrix in the apply kernel
__global__ __launch_bounds__(KGE_BLOCK) void transr_proj_apply_kernel(TransRArgs a) {
    const int u = blockIdx.x / TR_RB, rb = blockIdx.x % TR_RB;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int De = a.De, Dr = a.Dr;
    const int rows = (De + TR_RB - 1) / TR_RB, r0 = rb * rows, r1 = min(De, r0 + rows);
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    float *Pm = a.proj + a.ur_id[u] * (int64_t)De * Dr;
    const float k0 = a.k0[u], k1 = a.k1[u];
    for (int r = r0; r < r1; ++r) {
        for (int b = threadIdx.x; b < Dr; b += KGE_BLOCK) {
            float g0 = 0.f, g1 = 0.f;
            for (int q = e0; q < e1; ++q) {
                const int64_t e = a.ur_edge[q];
                const float th = a.ent[a.t_gid[e] * (int64_t)De + r] - a.ent[a.h_gid[e] * (int64_t)De + r];
                g0 = fmaf(a.dpos[e] * th, a.SG[e * Dr + b], g0);
                g1 += a.GP[e * De * Dr + (int64_t)r * Dr + b];
            }
            Pm[(int64_t)r * Dr + b] += k0 * g0 + k1 * g1;
        }
    }
}

// 16-byte variant (D_r % 4 == 0): TPR = power of two >= D_r / 4 threads cover one row with one float4 each, the workgroup walks
// 256 / TPR rows per iteration
__global__ __launch_bounds__(KGE_BLOCK) void transr_proj_apply_vec_kernel(TransRArgs a, int tpr) {
    const int u = blockIdx.x / TR_RB, rb = blockIdx.x % TR_RB;
    if (u >= (a.counts_dev ? a.counts_dev[1] : a.UR)) return;
    const int De = a.De, Dr = a.Dr;
    const int rows = (De + TR_RB - 1) / TR_RB, r0 = rb * rows, r1 = min(De, r0 + rows);
    const int e0 = a.ur_ptr[u], e1 = a.ur_ptr[u + 1];
    float *Pm = a.proj + a.ur_id[u] * (int64_t)De * Dr;
    const float k0 = a.k0[u], k1 = a.k1[u];
    constexpr int NEC = 32;                        // per-edge metadata does not depend on the row: fetched once
    __shared__ int64_t s_ho[NEC], s_to[NEC];
    __shared__ float s_dp[NEC];
    __shared__ int s_e[NEC];
    const int nec = min(NEC, e1 - e0);
    if ((int)threadIdx.x < nec) {
        const int e = a.ur_edge[e0 + threadIdx.x];
        s_e[threadIdx.x] = e;
        s_ho[threadIdx.x] = a.h_gid[e] * (int64_t)De;
        s_to[threadIdx.x] = a.t_gid[e] * (int64_t)De;
        s_dp[threadIdx.x] = a.dpos[e];
    }
    __syncthreads();
    const int rpi = KGE_BLOCK / tpr, rsub = threadIdx.x / tpr, b = (threadIdx.x % tpr) * 4;
    if (b >= Dr) return;
#pragma unroll 2
    for (int rr = r0; rr < r1; rr += rpi) {
        const int r = rr + rsub;
        if (r >= r1) continue;
        float4 p = *reinterpret_cast<const float4 *>(Pm + (int64_t)r * Dr + b);
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        for (int q = 0; q < e1 - e0; ++q) {
            int64_t e, ho, to; float dp;
            if (q < NEC) { e = s_e[q]; ho = s_ho[q]; to = s_to[q]; dp = s_dp[q]; }
            else { e = a.ur_edge[e0 + q]; ho = a.h_gid[e] * (int64_t)De; to = a.t_gid[e] * (int64_t)De; dp = a.dpos[e]; }
            const float th = dp * (a.ent[to + r] - a.ent[ho + r]);
            const float4 sg = *reinterpret_cast<const float4 *>(a.SG + e * Dr + b);
            const float4 gp = *reinterpret_cast<const float4 *>(a.GP + e * De * Dr + (int64_t)r * Dr + b);
            g0.x = fmaf(th, sg.x, g0.x); g0.y = fmaf(th, sg.y, g0.y); g0.z = fmaf(th, sg.z, g0.z); g0.w = fmaf(th, sg.w, g0.w);
            g1.x += gp.x; g1.y += gp.y; g1.z += gp.z; g1.w += gp.w;
        }
        p.x += k0 * g0.x + k1 * g1.x; p.y += k0 * g0.y + k1 * g1.y; p.z += k0 * g0.z + k1 * g1.z; p.w += k0 * g0.w + k1 * g1.w;
        *reinterpret_cast<float4 *>(Pm + (int64_t)r * Dr + b) = p;
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
int launch_transr_pos(const TransRArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    hipLaunchKernelGGL(transr_pos_kernel, dim3((a.B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s, a);
    return check_launch_t();
}
// the 128 x 208-tile kernels: operand widths, and their per-workgroup id / weight tables beside 45 KB of tiles
static bool transr_use_wide(int De, int Dr, int chunk, int N) {
    const int nG = chunk < TRANSR_GN_GROUPS_WIDE ? chunk : TRANSR_GN_GROUPS_WIDE, ipg = nG > 0 ? (chunk + nG - 1) / nG : 0;
    return transr_wide_supported(De, Dr) && (size_t)N * 12 + TW_C * 4 <= 16 * 1024 && (size_t)ipg * (8 + TW_R * 4) <= 16 * 1024;
}
int transr_gn_groups(int De, int Dr, int chunk, int N) {
    const int cap = transr_use_wide(De, Dr, chunk, N) ? TRANSR_GN_GROUPS_WIDE : TRANSR_GN_GROUPS;
    return chunk < cap ? chunk : cap;
}
int launch_transr_fwd(const TransRArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    if (transr_use_wide(a.De, a.Dr, a.chunk, a.N)) {
        const int nJW = (a.N + TW_R - 1) / TW_R;
        hipLaunchKernelGGL(transr_fwd_wide_kernel, dim3(a.B * nJW), dim3(KGE_BLOCK), 0, s, a, nJW);
        return check_launch_t();
    }
    const int nJB = (a.N + TR_T - 1) / TR_T;
    if (a.De % 4 == 0 && a.Dr % 4 == 0) hipLaunchKernelGGL(transr_fwd_kernel<true>, dim3(a.B * nJB), dim3(KGE_BLOCK), 0, s, a, nJB);
    else hipLaunchKernelGGL(transr_fwd_kernel<false>, dim3(a.B * nJB), dim3(KGE_BLOCK), 0, s, a, nJB);
    return check_launch_t();
}
int launch_transr_bwd(const TransRArgs &a, hipStream_t s) {
    if (a.B == 0) return KGE_OK;
    const int nJB = (a.N + TR_T - 1) / TR_T, nEB = (a.De + TR_T - 1) / TR_T, nRB = (a.Dr + TR_T - 1) / TR_T;
    const dim3 gw((a.B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), b(KGE_BLOCK);
    if (transr_use_wide(a.De, a.Dr, a.chunk, a.N)) {      // (dq and GR come out of the projection-gradient kernel's own sweep)
        const int nJW = (a.N + TW_R - 1) / TW_R, nEW = (a.De + TW_R - 1) / TW_R, ipgw = (a.chunk + a.nG - 1) / a.nG;
        hipLaunchKernelGGL(transr_gn_wide_kernel, dim3(a.C * nJW * a.nG), b, (size_t)ipgw * (sizeof(int64_t) + TW_R * sizeof(float)), s, a, nJW);
        hipLaunchKernelGGL(transr_gn_reduce_kernel, dim3(((int64_t)a.C * a.N + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), b, 0, s, a);
        #ifndef TW_GP_PER_EDGE
        const int gp_blocks = a.B * nEW;
#else
        const int gp_blocks = a.B;
#endif
        hipLaunchKernelGGL(transr_gp_wide_kernel, dim3(gp_blocks), b, (size_t)a.N * (sizeof(int64_t) + sizeof(float)) + TW_C * sizeof(float), s, a, nEW);
        return check_launch_t();
    }
    hipLaunchKernelGGL(transr_dq_kernel, dim3(a.B), b, 0, s, a);
    const bool vec = a.De % 4 == 0 && a.Dr % 4 == 0;      // 16-byte operand loads (4 sign bytes per word)
    const int ipg = (a.chunk + a.nG - 1) / a.nG;
    const size_t gn_lds = (size_t)ipg * (sizeof(int64_t) + TR_T * sizeof(float)), gp_lds = (size_t)a.N * (sizeof(int64_t) + sizeof(float));
    if (gn_lds > 44 * 1024 || gp_lds > 44 * 1024) return KGE_ERR_ARG;      // (64 KB of LDS per workgroup: chunk <= 2 700, N <= 3 700)
    if (vec) hipLaunchKernelGGL(transr_gn_kernel<true>, dim3(a.C * nJB * nEB * a.nG), b, gn_lds, s, a, nJB, nEB);
    else hipLaunchKernelGGL(transr_gn_kernel<false>, dim3(a.C * nJB * nEB * a.nG), b, gn_lds, s, a, nJB, nEB);
    hipLaunchKernelGGL(transr_gn_reduce_kernel, dim3(((int64_t)a.C * a.N + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), b, 0, s, a);
    if (vec) hipLaunchKernelGGL(transr_gp_kernel<true>, dim3(a.B * nEB * nRB), b, gp_lds, s, a, nEB, nRB);
    else hipLaunchKernelGGL(transr_gp_kernel<false>, dim3(a.B * nEB * nRB), b, gp_lds, s, a, nEB, nRB);
    hipLaunchKernelGGL(transr_gr_kernel, gw, b, 0, s, a);
    return check_launch_t();
}
int launch_transr_proj_update(const TransRArgs &a, hipStream_t s) {
    if (a.B == 0 || a.UR == 0) return KGE_OK;
    const int ntiles = transr_use_wide(a.De, a.Dr, a.chunk, a.N) ? (a.De + TW_R - 1) / TW_R
                                                                  : ((a.De + TR_T - 1) / TR_T) * ((a.Dr + TR_T - 1) / TR_T);
    hipLaunchKernelGGL(transr_proj_sq_kernel, dim3((a.B + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK), dim3(KGE_BLOCK), 0, s, a, ntiles);
    hipLaunchKernelGGL(transr_proj_state_kernel, dim3((a.UR + 255) / 256), dim3(256), 0, s, a);
    if (a.Dr % 4 == 0 && a.Dr <= 1024) {
        int tpr = 16;
        while (tpr * 4 < a.Dr) tpr *= 2;
        hipLaunchKernelGGL(transr_proj_apply_vec_kernel, dim3(a.UR * TR_RB), dim3(KGE_BLOCK), 0, s, a, tpr);
    } else
        hipLaunchKernelGGL(transr_proj_apply_kernel, dim3(a.UR * TR_RB), dim3(KGE_BLOCK), 0, s, a);
    return check_launch_t();
}

// =============================================================================================
// per-op (drop-in) route: the projections of TransRScore.prepare / create_neg_prepare (score_fun.py:131-166) and their
// autograd as stand-alone operators on DENSE gathered operands - the same tile routine as the fused kernels above, plain
// fp32 operands and plain epilogues (the reference's decomposition materialises Y [C, chunk, N, D_r]; so does this route)
// =============================================================================================
struct PnArgs {
    int B, C, chunk, N, De, Dr, accumulate;
    const float *proj;             // [B, De * Dr] gathered projection rows (one per positive edge)
    const float *neg;              // [C * N, De]
    const float *x;                // [B, De]
    const float *gy;               // [B, Dr]
    float *Y;                      // fwd out / bwd in [B, N, Dr]
    float *gneg, *gproj, *gx;
};

// Y_i = Neg_c P_i        workgroup = (positive i, 64 negatives, 64 columns of D_r)
template <bool VEC>
__global__ __launch_bounds__(KGE_BLOCK) void transr_pn_fwd_kernel(PnArgs a, int nJB, int nRB) {
    __shared__ float As[2][TR_K][TR_LD], Bs[2][TR_K][TR_LD];
    const int rb = blockIdx.x % nRB, jb = (blockIdx.x / nRB) % nJB, i = blockIdx.x / (nRB * nJB);
    const int c = i / a.chunk, De = a.De, Dr = a.Dr, N = a.N, j0 = jb * TR_T, dr0 = rb * TR_T;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    const float *Pi = a.proj + (int64_t)i * De * Dr, *Nc = a.neg + (int64_t)c * N * De;
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC)
        tile_sweep4<false, true>(acc, De,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; return (j0 + row < N && k < De) ? *reinterpret_cast<const float4 *>(Nc + (int64_t)(j0 + row) * De + k) : f4zero(); },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_; return (k < De && dr0 + col < Dr) ? *reinterpret_cast<const float4 *>(Pi + (int64_t)k * Dr + dr0 + col) : f4zero(); },
            As, Bs);
    else
        tile_sweep<true, false>(acc, De,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; return (j0 + row < N && k < De) ? Nc[(int64_t)(j0 + row) * De + k] : 0.f; },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_; return (k < De && dr0 + col < Dr) ? Pi[(int64_t)k * Dr + dr0 + col] : 0.f; }, As, Bs);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int col = dr0 + ct * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wave * 16 + 4 * q + r;
            if (col < Dr && j < N) a.Y[((int64_t)i * N + j) * Dr + col] = acc[ct][r];
        }
    }
}

// gNeg_c[j][de] = sum_{i in chunk} sum_dr gY_i[j][dr] P_i[de][dr]     workgroup = (chunk, 64 negatives, 64 de)
template <bool VEC>
__global__ __launch_bounds__(KGE_BLOCK) void transr_pn_bwd_neg_kernel(PnArgs a, int nJB, int nEB) {
    __shared__ float As[2][TR_K][TR_LD], Bs[2][TR_K][TR_LD];
    const int eb = blockIdx.x % nEB, jb = (blockIdx.x / nEB) % nJB, c = blockIdx.x / (nEB * nJB);
    const int j0 = jb * TR_T, de0 = eb * TR_T, De = a.De, Dr = a.Dr, N = a.N, chunk = a.chunk;
    const int DrP = (Dr + TR_K - 1) / TR_K * TR_K;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC)
        tile_sweep4<false, false>(acc, chunk * DrP,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_;
                const int il = k0_ / DrP, dr = k0_ % DrP + kk_, j = j0 + row;
                if (j >= N || dr >= Dr) return f4zero();
                return *reinterpret_cast<const float4 *>(a.Y + (((int64_t)c * chunk + il) * N + j) * Dr + dr);
            },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_;
                const int il = k0_ / DrP, dr = k0_ % DrP + kk_, de = de0 + col;
                if (de >= De || dr >= Dr) return f4zero();
                return *reinterpret_cast<const float4 *>(a.proj + ((int64_t)c * chunk + il) * De * Dr + (int64_t)de * Dr + dr);
            }, As, Bs);
    else
        tile_sweep<true, true>(acc, chunk * DrP,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_;
                const int il = k0_ / DrP, dr = k0_ % DrP + kk_, j = j0 + row;
                return (j >= N || dr >= Dr) ? 0.f : a.Y[(((int64_t)c * chunk + il) * N + j) * Dr + dr];
            },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_;
                const int il = k0_ / DrP, dr = k0_ % DrP + kk_, de = de0 + col;
                return (de >= De || dr >= Dr) ? 0.f : a.proj[((int64_t)c * chunk + il) * De * Dr + (int64_t)de * Dr + dr];
            }, As, Bs);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int de = de0 + ct * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + wave * 16 + 4 * q + r;
            if (de < De && j < N) a.gneg[((int64_t)c * N + j) * De + de] = acc[ct][r];
        }
    }
}

// gP_i[de][dr] (+)= sum_j Neg_j[de] gY_ij[dr]     workgroup = (positive, 64 de, 64 dr)
template <bool VEC>
__global__ __launch_bounds__(KGE_BLOCK) void transr_pn_bwd_proj_kernel(PnArgs a, int nEB, int nRB) {
    __shared__ float As[2][TR_K][TR_LD], Bs[2][TR_K][TR_LD];
    const int rb = blockIdx.x % nRB, eb = (blockIdx.x / nRB) % nEB, i = blockIdx.x / (nRB * nEB);
    const int de0 = eb * TR_T, dr0 = rb * TR_T, De = a.De, Dr = a.Dr, N = a.N, c = i / a.chunk;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    const float *Nc = a.neg + (int64_t)c * N * De, *Yi = a.Y + (int64_t)i * N * Dr;
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nct = min(4, (Dr - dr0 + 15) / 16);
    const bool won = de0 + wave * 16 < De;
    if constexpr (VEC)
        tile_sweep4<true, true>(acc, N,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; return (k < N && de0 + row < De) ? *reinterpret_cast<const float4 *>(Nc + (int64_t)k * De + de0 + row) : f4zero(); },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_; return (k < N && dr0 + col < Dr) ? *reinterpret_cast<const float4 *>(Yi + (int64_t)k * Dr + dr0 + col) : f4zero(); },
            As, Bs, won ? nct : 0);
    else
        tile_sweep<false, false>(acc, N,
            [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; return (k < N && de0 + row < De) ? Nc[(int64_t)k * De + de0 + row] : 0.f; },
            [&](int k0_, int kk_, int col) { const int k = k0_ + kk_; return (k < N && dr0 + col < Dr) ? Yi[(int64_t)k * Dr + dr0 + col] : 0.f; }, As, Bs, won ? nct : 0);
    float *G = a.gproj + (int64_t)i * De * Dr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int dr = dr0 + ct * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int de = de0 + wave * 16 + 4 * q + r;
            if (de < De && dr < Dr) {
                const int64_t o = (int64_t)de * Dr + dr;
                G[o] = a.accumulate ? G[o] + acc[ct][r] : acc[ct][r];
            }
        }
    }
}

// one workgroup per edge: y_i = x_i P_i (fwd) / gx_i = P_i gy_i and gP_i (+)= x_i (x) gy_i (bwd)
__global__ __launch_bounds__(KGE_BLOCK) void transr_pv_fwd_kernel(PnArgs a) {
    const int i = blockIdx.x, De = a.De, Dr = a.Dr;
    const float *P = a.proj + (int64_t)i * De * Dr, *x = a.x + (int64_t)i * De;
    for (int dr = threadIdx.x; dr < Dr; dr += KGE_BLOCK) {
        float s = 0.f;
        for (int de = 0; de < De; ++de) s = fmaf(x[de], P[(int64_t)de * Dr + dr], s);
        a.Y[(int64_t)i * Dr + dr] = s;
    }
}
__global__ __launch_bounds__(KGE_BLOCK) void transr_pv_bwd_kernel(PnArgs a) {
    const int i = blockIdx.x, De = a.De, Dr = a.Dr;
    const float *P = a.proj + (int64_t)i * De * Dr, *x = a.x + (int64_t)i * De, *gy = a.gy + (int64_t)i * Dr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (a.gx) {
        for (int de = wave; de < De; de += KGE_WAVES_PER_BLOCK) {       // one wavefront per row of P: coalesced along d_r
            float s = 0.f;
            for (int dr = lane; dr < Dr; dr += 64) s = fmaf(P[(int64_t)de * Dr + dr], gy[dr], s);
            s = wave_sum(s);
            if (lane == 0) a.gx[(int64_t)i * De + de] = s;
        }
    }
    if (a.gproj) {
        float *G = a.gproj + (int64_t)i * De * Dr;
        for (int o = threadIdx.x; o < De * Dr; o += KGE_BLOCK) {
            const float v = x[o / Dr] * gy[o % Dr];
            G[o] = a.accumulate ? G[o] + v : v;
        }
    }
}

extern "C" {

int kge_transr_project(const float *proj, const float *x, int64_t B, int d_e, int d_r, float *out, void *stream) {
    if (B < 0 || d_e <= 0 || d_r <= 0 || (B && (!proj || !x || !out))) return KGE_ERR_ARG;
    if (B == 0) return KGE_OK;
    PnArgs a{}; a.B = (int)B; a.De = d_e; a.Dr = d_r; a.proj = proj; a.x = x; a.Y = out;
    hipLaunchKernelGGL(transr_pv_fwd_kernel, dim3((unsigned)B), dim3(KGE_BLOCK), 0, (hipStream_t)stream, a);
    return check_launch_t();
}

int kge_transr_project_bwd(const float *proj, const float *x, const float *gy, int64_t B, int d_e, int d_r, float *gx,
                           float *gproj, int accumulate, void *stream) {
    if (B < 0 || d_e <= 0 || d_r <= 0 || (B && (!proj || !x || !gy))) return KGE_ERR_ARG;
    if (B == 0 || (!gx && !gproj)) return KGE_OK;
    PnArgs a{}; a.B = (int)B; a.De = d_e; a.Dr = d_r; a.proj = proj; a.x = x; a.gy = gy; a.gx = gx; a.gproj = gproj;
    a.accumulate = accumulate;
    hipLaunchKernelGGL(transr_pv_bwd_kernel, dim3((unsigned)B), dim3(KGE_BLOCK), 0, (hipStream_t)stream, a);
    return check_launch_t();
}

int kge_transr_project_neg(const float *proj, const float *neg, int C, int chunk, int N, int d_e, int d_r, float *Y, void *stream) {
    if (C < 0 || chunk <= 0 || N <= 0 || d_e <= 0 || d_r <= 0 || (C && (!proj || !neg || !Y))) return KGE_ERR_ARG;
    if (C == 0) return KGE_OK;
    PnArgs a{}; a.B = C * chunk; a.C = C; a.chunk = chunk; a.N = N; a.De = d_e; a.Dr = d_r; a.proj = proj; a.neg = neg; a.Y = Y;
    const int nJB = (N + TR_T - 1) / TR_T, nRB = (d_r + TR_T - 1) / TR_T;
    const dim3 g((unsigned)(a.B * nJB * nRB)), b(KGE_BLOCK);
    if (d_e % 4 == 0 && d_r % 4 == 0) hipLaunchKernelGGL(transr_pn_fwd_kernel<true>, g, b, 0, (hipStream_t)stream, a, nJB, nRB);
    else hipLaunchKernelGGL(transr_pn_fwd_kernel<false>, g, b, 0, (hipStream_t)stream, a, nJB, nRB);
    return check_launch_t();
}

int kge_transr_project_neg_bwd(const float *proj, const float *neg, const float *gY, int C, int chunk, int N, int d_e, int d_r,
                               float *gneg, float *gproj, int accumulate, void *stream) {
    if (C < 0 || chunk <= 0 || N <= 0 || d_e <= 0 || d_r <= 0 || (C && (!proj || !neg || !gY))) return KGE_ERR_ARG;
    if (C == 0) return KGE_OK;
    PnArgs a{}; a.B = C * chunk; a.C = C; a.chunk = chunk; a.N = N; a.De = d_e; a.Dr = d_r; a.proj = proj; a.neg = neg;
    a.Y = const_cast<float *>(gY); a.gneg = gneg; a.gproj = gproj; a.accumulate = accumulate;
    const int nJB = (N + TR_T - 1) / TR_T, nEB = (d_e + TR_T - 1) / TR_T, nRB = (d_r + TR_T - 1) / TR_T;
    const bool vec = d_e % 4 == 0 && d_r % 4 == 0;
    const dim3 b(KGE_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    if (gneg) {
        if (vec) hipLaunchKernelGGL(transr_pn_bwd_neg_kernel<true>, dim3((unsigned)(C * nJB * nEB)), b, 0, s, a, nJB, nEB);
        else hipLaunchKernelGGL(transr_pn_bwd_neg_kernel<false>, dim3((unsigned)(C * nJB * nEB)), b, 0, s, a, nJB, nEB);
    }
    if (gproj) {
        if (vec) hipLaunchKernelGGL(transr_pn_bwd_proj_kernel<true>, dim3((unsigned)(a.B * nEB * nRB)), b, 0, s, a, nEB, nRB);
        else hipLaunchKernelGGL(transr_pn_bwd_proj_kernel<false>, dim3((unsigned)(a.B * nEB * nRB)), b, 0, s, a, nEB, nRB);
    }
    return check_launch_t();
}

}  // extern "C"
