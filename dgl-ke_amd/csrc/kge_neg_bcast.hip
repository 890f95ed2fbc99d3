// kge_neg_bcast.hip - fast path of the pairwise (non-GEMM) negative scoring: TransE_l1 (th.cdist p=1,
// models/pytorch/score_fun.py:36-38) and RotatE (complex modulus of the broadcast difference,
// score_fun.py:526-531, 548-552), also the direct L2 / dot forms when the GEMM path is disabled.
//
//     n_ij = gamma - sum_k f(a_ik, b_jk)      GA_ik = sum_j W_ij dn_ij/da_ik      GN_jk = sum_i W_ij dn_ij/db_jk
//
// These are VALU-bound (no matrix form).  Layout: "lane = row of one operand (x), the other operand
// (y) is wave-uniform".  Every lane keeps a slab of ITS x row in VGPRs.  The y rows are fetched with
// vector loads and broadcast inside the wavefront: the backward kernel lays them out "lane & 3 = element"
// (one VGPR = four operands replicated in every quad of lanes, picked by a DPP quad_perm modifier that
// is folded into the consuming VALU instruction or costs one v_mov_dpp), the forward kernel "lane =
// element" (one VGPR = 64 operands, v_readlane_b32).  No LDS, no barriers, no scalar-cache latency,
// prefetch one row / sub-slab ahead, branch-free main loops.
// Measured alternatives on MI355X (profiles/r01_microbench_mi355x.txt): s_load_dwordx8 for the uniform
// rows (SGPR double buffers spill; scalar loads can only be waited for with lgkmcnt(0)): 3x slower for
// TransE_l1; "one VGPR = 64 operands" + v_readlane_b32 broadcasts: ~16 cycles per broadcast.
// The LDS-tiled kernels of kge_neg_pair.hip stay as the generic fallback for other row widths.
#include <utility>
#include "kge_common.hpp"

using namespace kge;

#define SB_KB 16                                 // real models: reduction elements per forward sub-slab
#define SB_KC 8                                  // RotatE: complex columns per sub-slab
#ifndef SB_KWR
#define SB_KWR 8                                 // backward, real models: output columns per wavefront
#endif
#ifndef SB_RWR
#define SB_RWR 4                                 // forward, real models: uniform rows per wavefront
#endif

static inline int check_launch_b() { return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH; }

__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32 (1 ulp)
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }     // v_rsq_f32
// operand held by lane L of v, broadcast to the whole wavefront (L is a compile-time constant)
template <int L> __device__ __forceinline__ float bcast(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L));
}
// operand held by lane (4*(l/4) + Q) of v, broadcast inside every quad of lanes: a DPP quad_perm modifier
// that the compiler folds into the consuming VALU instruction (no issue slot of its own; v_readlane, in
// contrast, measured ~16 cycles per broadcast on MI355X)
template <int Q> __device__ __forceinline__ float quad(float v) {
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xF, 0xF, true));
}
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

bool neg_bcast_supported(int model, int d_e) {
    if (model == KGE_ROTATE) return d_e > 0 && d_e % 2 == 0 && (d_e / 2) % SB_KC == 0;
    return d_e > 0 && d_e % SB_KB == 0;
}

// ---------------------------------------------------------------------------------------------
// shared inner step of the forward kernel: NE (complex) elements of the lane's own row x against RW
// uniform rows whose operands sit in lanes [lb, lb+NE) of yr / yi.  LB >= 0: compile-time lane
// indices (main loop); LB < 0: the lane base is the run-time (wave-uniform) value lbr (tail).
// ---------------------------------------------------------------------------------------------
template <int MODEL, int RW, int NE, int LB>
__device__ __forceinline__ void fwd_step(const float (&xc)[16], const float (&yr)[RW], const float (&yi)[RW],
                                         float (&acc)[RW], int lbr) {
    constexpr bool CPLX = MODEL == KGE_ROTATE;
    static_for<NE>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            float y0, y1 = 0.f;
            if constexpr (LB >= 0) {
                y0 = bcast<(LB >= 0 ? LB : 0) + e>(yr[r]);
                if constexpr (CPLX) y1 = bcast<(LB >= 0 ? LB : 0) + e>(yi[r]);
            } else {
                y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yr[r]), lbr + e));
                if constexpr (CPLX) y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yi[r]), lbr + e));
            }
            if constexpr (CPLX) {
                const float dr = y0 - xc[e], di = y1 - xc[8 + e];
                acc[r] += fast_sqrt(fmaf(di, di, dr * dr));
            } else if constexpr (MODEL == KGE_TRANSE_L1) {
                acc[r] += fabsf(y0 - xc[e]);
            } else if constexpr (MODEL == KGE_TRANSE_L2) {
                const float u = y0 - xc[e];
                acc[r] = fmaf(u, u, acc[r]);
            } else {
                acc[r] = fmaf(y0, xc[e], acc[r]);
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------
// forward: lanes = negatives j (score rows are then written coalesced), uniform = pos-side rows a_i.
// One task (= one wavefront) = (chunk, strip of 64 negatives, RW positives).  The uniform rows are
// fetched "lane = element" - yr[r] = a_{i0+r}[64*t + lane], ONE VGPR = 64 operands - and broadcast with
// v_readlane_b32 (measured: fewer, wider loads beat the quad layout of the backward kernel here, 14.8
// vs 21.8 us for TransE_l1); inside a 64-element block, sub-slabs of 16 (real) / 8 complex elements of
// the lane's own row are double-buffered in VGPRs.  The main loop has NO conditionals (loads inside
// branches make the compiler wait for everything at the joins): prefetch addresses are clamped
// instead, and the partial last block runs in a separate tail loop.
// ---------------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(KGE_BLOCK) void neg_fwd_bcast_kernel(NegArgs a, int ns, int ng) {
    constexpr bool CPLX = MODEL == KGE_ROTATE;
    constexpr int RW = CPLX ? 2 : SB_RWR;                        // uniform rows per wavefront
    constexpr int NE = CPLX ? SB_KC : SB_KB;                     // (complex) elements per sub-slab
    constexpr int NSUB = 64 / NE;                                // sub-slabs per 64-element block
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // tasks are numbered with the row group fastest and a workgroup takes 4 consecutive ones, so that
    // its wavefronts (almost always) share the strip - the x rows then hit in L1 - while the number of
    // workgroups carries no padding: with ~1 workgroup per CU a few extra workgroups double the
    // makespan (260 instead of 250 at cfg-T).
    const int task = blockIdx.x * KGE_WAVES_PER_BLOCK + wave;
    if (task >= a.C * ns * ng) return;                           // wave-uniform
    const int g = task % ng, st = (task / ng) % ns, c = task / (ng * ns);
    const int D = a.d_e, K = CPLX ? D / 2 : D;
    const int i0 = g * RW;
    const int j = st * 64 + lane;
    const float *x = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + min(j, a.N - 1), D);
    const float *y[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) y[r] = a.A + ((int64_t)c * a.chunk + min(i0 + r, a.chunk - 1)) * D;
    float acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = 0.f;

    auto loady = [&](float (&yr)[RW], float (&yi)[RW], int kb) {     // block of 64 elements: lane = element
        const int k = min(kb + lane, K - 1);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            yr[r] = y[r][k];
            yi[r] = CPLX ? y[r][K + k] : 0.f;
        }
    };
    // the lane's own row, one sub-slab: real models 16 values; RotatE [0..7] = re, [8..15] = im
    auto loadx = [&](float (&xr)[16], int k0) {
        k0 = min(k0, K - NE);                                    // prefetch past the end re-reads the last slab
#pragma unroll
        for (int e = 0; e < NE; e += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(x + k0 + e);
            xr[e] = t.x; xr[e + 1] = t.y; xr[e + 2] = t.z; xr[e + 3] = t.w;
            if constexpr (CPLX) {
                const float4 u = *reinterpret_cast<const float4 *>(x + K + k0 + e);
                xr[8 + e] = u.x; xr[9 + e] = u.y; xr[10 + e] = u.z; xr[11 + e] = u.w;
            }
        }
    };
    float yr[RW], yi[RW], ynr[RW], yni[RW], xa[16], xb[16];
    loady(yr, yi, 0);
    loadx(xa, 0);
    int kb = 0;
    for (; kb + 64 <= K; kb += 64) {                             // full blocks, branch-free
        loady(ynr, yni, kb + 64);                                // next block, one block ahead
        static_for<NSUB>([&](auto subc) {
            constexpr int SUB = decltype(subc)::value;
            float(&xc)[16] = (SUB & 1) ? xb : xa;
            float(&xn)[16] = (SUB & 1) ? xa : xb;
            loadx(xn, kb + (SUB + 1) * NE);                      // next sub-slab, one ahead
            fwd_step<MODEL, RW, NE, SUB * NE>(xc, yr, yi, acc, 0);
        });
#pragma unroll
        for (int r = 0; r < RW; ++r) { yr[r] = ynr[r]; yi[r] = yni[r]; }
    }
    for (int k0 = kb; k0 < K; k0 += NE) {                        // partial last block (K % 64 elements)
        fwd_step<MODEL, RW, NE, -1>(xa, yr, yi, acc, k0 - kb);
        loadx(xa, k0 + NE);
    }
    if (j < a.N) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (i0 + r < a.chunk) {
                float v = acc[r];
                if (MODEL == KGE_TRANSE_L1 || CPLX) v = a.gamma - v;
                else if (MODEL == KGE_TRANSE_L2) v = a.gamma - sqrtf(fmaxf(v, 1e-30f));
                else if (a.clampv > 0.f) v = fminf(fmaxf(v, -a.clampv), a.clampv);
                a.S[((int64_t)c * a.chunk + i0 + r) * a.N + j] = v;
            }
        }
    }
}

template <int MODEL> static int fwd_launch(const NegArgs &a, hipStream_t s) {
    constexpr int RW = MODEL == KGE_ROTATE ? 2 : SB_RWR;
    const int ns = (a.N + 63) / 64, ng = (a.chunk + RW - 1) / RW;
    const int64_t nb = ((int64_t)a.C * ns * ng + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    if (nb == 0) return KGE_OK;
    hipLaunchKernelGGL(neg_fwd_bcast_kernel<MODEL>, dim3((unsigned)nb), dim3(KGE_BLOCK), 0, s, a, ns, ng);
    return check_launch_b();
}

int launch_neg_fwd_bcast(const NegArgs &a, hipStream_t s) {
    switch (a.model) {
        case KGE_TRANSE_L1: return fwd_launch<KGE_TRANSE_L1>(a, s);
        case KGE_TRANSE_L2: return fwd_launch<KGE_TRANSE_L2>(a, s);
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_SIMPLE: case KGE_RESCAL: return fwd_launch<KGE_DISTMULT>(a, s);
        case KGE_ROTATE: return fwd_launch<KGE_ROTATE>(a, s);
    }
    return KGE_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// backward.  OUT[r, k] = sum_s W(r, s) * psi(x_r[k], y_s[k])
//   GA: r = positive i (lanes), s = negative j (uniform), x = a, y = b, psi = dn/da
//   GN: r = negative j (lanes), s = positive i (uniform), x = b, y = a, psi = dn/db
// One task (= one wavefront) = (product, chunk, strip of 64 rows, slab of 16 output columns - 8
// complex columns for RotatE); it walks the reduction side in groups of GS = 64 / slab-width rows:
// ONE coalesced load (lane l -> row s0 + l / KW, column k0 + l % KW) brings the operands of a whole
// group into one VGPR (two for RotatE: re, im).  W(r, s): GA lanes read their own row (a float4 per 4
// columns when VECW), GN lanes read one coalesced element per s.  Loads are branch-free (clamped
// addresses, masks), groups are prefetched one ahead.
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool GA, bool VECW>
__device__ __forceinline__ void bwd_task(const NegArgs &a, int c, int st, int k0, int lane) {
    constexpr bool CPLX = MODEL == KGE_ROTATE;
    constexpr int KW = CPLX ? SB_KC : SB_KWR;                   // (complex) columns per wavefront
    constexpr int GS = 4;                                        // reduction rows per weight group
    const int D = a.d_e, K = CPLX ? D / 2 : D;
    const int R = GA ? a.chunk : a.N, S = GA ? a.N : a.chunk;
    const int r = st * 64 + lane, rc = min(r, R - 1);
    const float *xrow = GA ? a.A + ((int64_t)c * a.chunk + rc) * D
                           : row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + rc, D);
    float xr[KW], xi[KW], outr[KW], outi[KW];
#pragma unroll
    for (int e = 0; e < KW; e += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(xrow + k0 + e);
        xr[e] = t.x; xr[e + 1] = t.y; xr[e + 2] = t.z; xr[e + 3] = t.w;
        if constexpr (CPLX) {
            const float4 u = *reinterpret_cast<const float4 *>(xrow + K + k0 + e);
            xi[e] = u.x; xi[e + 1] = u.y; xi[e + 2] = u.z; xi[e + 3] = u.w;
        } else {
            xi[e] = xi[e + 1] = xi[e + 2] = xi[e + 3] = 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < KW; ++e) { outr[e] = 0.f; outi[e] = 0.f; }
    const float *Wc = a.W + (int64_t)c * a.chunk * a.N;
    constexpr int NQ = KW / 4;                                   // quad registers per reduction row (and half)
    const int lq = lane & 3;
    // operands of ONE reduction row s: register t holds y_s[k0 + 4t + (lane & 3)] - every quad of lanes has
    // the same four values, so element e is a quad_perm broadcast of register e / 4 (rows beyond S are
    // clamped, their W is 0)
    auto loady = [&](float (&yr)[NQ], float (&yi)[NQ], int s) {
        s = min(s, S - 1);
        const float *y = (GA ? row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + s, D)
                             : a.A + ((int64_t)c * a.chunk + s) * D) + k0 + lq;
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            yr[t] = y[4 * t];
            yi[t] = CPLX ? y[K + 4 * t] : 0.f;
        }
    };
    // W(r, s0 .. s0+GS-1) of one group of GS rows; beyond S: 0
    auto loadw = [&](float (&w)[GS], int s0) {
        if constexpr (GA && VECW) {                              // S % GS == 0 and 16-byte aligned rows
            const float *wp = Wc + (int64_t)rc * a.N + min(s0, S - GS);
#pragma unroll
            for (int q = 0; q < GS; q += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(wp + q);
                w[q] = t.x; w[q + 1] = t.y; w[q + 2] = t.z; w[q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const int s = min(s0 + q, S - 1);
                const float v = GA ? Wc[(int64_t)rc * a.N + s] : Wc[(int64_t)s * a.N + rc];
                w[q] = s0 + q < S ? v : 0.f;
            }
        }
    };
    // sign convention: the difference is always (a - b); x is a for GA and b for GN and in both products
    // the contribution is  -w * dir(x - y)  (GN: +dir(a - b) = -dir(b - a)).
    auto row = [&](const float (&yr)[NQ], const float (&yi)[NQ], float w) {
        static_for<KW>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const float y0 = quad<e & 3>(yr[e >> 2]);
            if constexpr (CPLX) {
                const float y1 = quad<e & 3>(yi[e >> 2]);
                const float dr = xr[e] - y0, di = xi[e] - y1;
                const float m2 = fmaf(di, di, fmaf(dr, dr, 1e-30f));     // + tiny: zero difference -> zero gradient
                const float iv = -w * fast_rsq(m2);
                outr[e] = fmaf(dr, iv, outr[e]);
                outi[e] = fmaf(di, iv, outi[e]);
            } else if constexpr (MODEL == KGE_TRANSE_L1) {
                // sign(d) = med3(d * 2^126, -1, 1): exact for every normal d, 0 at d == 0, and free of
                // VCC round trips (cmp + cndmask cost 8 issue slots per element)
                const float sg = __builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(xr[e] - y0, 126), -1.f, 1.f);
                outr[e] = fmaf(-w, sg, outr[e]);
            } else if constexpr (MODEL == KGE_TRANSE_L2) {
                outr[e] = fmaf(-w, xr[e] - y0, outr[e]);          // W is pre-divided by the distance
            } else {
                outr[e] = fmaf(w, y0, outr[e]);
            }
        });
    };
    // rows are processed in pairs with the operand registers double-buffered (row s+1 is requested before
    // row s is consumed); the weights of a group of GS rows are requested one group ahead
    float ya[NQ], yai[NQ], yb[NQ], ybi[NQ], wc[GS], wn[GS];
    loady(ya, yai, 0);
    loadw(wc, 0);
    for (int s0 = 0; s0 < S; s0 += GS) {
        loadw(wn, s0 + GS);
#pragma unroll
        for (int q = 0; q < GS; q += 2) {
            loady(yb, ybi, s0 + q + 1);
            row(ya, yai, wc[q]);
            loady(ya, yai, s0 + q + 2);
            row(yb, ybi, wc[q + 1]);
        }
#pragma unroll
        for (int t = 0; t < GS; ++t) wc[t] = wn[t];
    }
    if (r >= R) return;
    const bool reg = (!GA) && a.reg_coef > 0.f && a.reg_norm > 0;
    float *O = (GA ? a.GA : a.GN) + ((int64_t)c * R + r) * D;
#pragma unroll
    for (int e = 0; e < KW; e += 4) {
        float4 o;
        float *ov = reinterpret_cast<float *>(&o);
#pragma unroll
        for (int q = 0; q < 4; ++q) ov[q] = outr[e + q] + (reg ? reg_grad(xr[e + q], a.reg_coef, a.reg_norm) : 0.f);
        *reinterpret_cast<float4 *>(O + k0 + e) = o;
        if constexpr (CPLX) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ov[q] = outi[e + q] + (reg ? reg_grad(xi[e + q], a.reg_coef, a.reg_norm) : 0.f);
            *reinterpret_cast<float4 *>(O + K + k0 + e) = o;
        }
    }
}

template <int MODEL, bool VECW>
__global__ __launch_bounds__(KGE_BLOCK) void neg_bwd_bcast_kernel(NegArgs a, int nsA, int nsN) {
    constexpr int KW = MODEL == KGE_ROTATE ? SB_KC : SB_KWR;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // tasks: slab fastest; a workgroup takes 4 consecutive tasks (see the forward kernel)
    const int K = MODEL == KGE_ROTATE ? a.d_e / 2 : a.d_e;
    const int nks = K / KW;
    const int nGA = a.C * nsA * nks;
    int task = blockIdx.x * KGE_WAVES_PER_BLOCK + wave;
    if (task >= a.C * (nsA + nsN) * nks) return;                 // wave-uniform
    if (task < nGA) {
        bwd_task<MODEL, true, VECW>(a, task / (nks * nsA), (task / nks) % nsA, (task % nks) * KW, lane);
    } else {
        task -= nGA;
        bwd_task<MODEL, false, VECW>(a, task / (nks * nsN), (task / nks) % nsN, (task % nks) * KW, lane);
    }
}

template <int MODEL> static int bwd_launch(const NegArgs &a, hipStream_t s) {
    constexpr int KW = MODEL == KGE_ROTATE ? SB_KC : SB_KWR;
    constexpr int GS = 4;
    const int K = MODEL == KGE_ROTATE ? a.d_e / 2 : a.d_e;
    const int nsA = (a.chunk + 63) / 64, nsN = (a.N + 63) / 64;
    const int64_t nb = ((int64_t)a.C * (nsA + nsN) * (K / KW) + KGE_WAVES_PER_BLOCK - 1) / KGE_WAVES_PER_BLOCK;
    if (nb == 0) return KGE_OK;
    const bool vecw = a.N % 4 == 0 && a.N % GS == 0 && a.N >= GS;
    if (vecw) hipLaunchKernelGGL((neg_bwd_bcast_kernel<MODEL, true>), dim3((unsigned)nb), dim3(KGE_BLOCK), 0, s, a, nsA, nsN);
    else hipLaunchKernelGGL((neg_bwd_bcast_kernel<MODEL, false>), dim3((unsigned)nb), dim3(KGE_BLOCK), 0, s, a, nsA, nsN);
    return check_launch_b();
}

int launch_neg_bwd_bcast(const NegArgs &a, hipStream_t s) {
    switch (a.model) {
        case KGE_TRANSE_L1: return bwd_launch<KGE_TRANSE_L1>(a, s);
        case KGE_TRANSE_L2: return bwd_launch<KGE_TRANSE_L2>(a, s);
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_SIMPLE: case KGE_RESCAL: return bwd_launch<KGE_DISTMULT>(a, s);
        case KGE_ROTATE: return bwd_launch<KGE_ROTATE>(a, s);
    }
    return KGE_ERR_ARG;
}
