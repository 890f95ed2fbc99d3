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
// element" (one coalesced load = 64 operands, broadcast back out of wavefront-private LDS rows).  No
// barriers, no scalar-cache latency, prefetch one row / sub-slab ahead, branch-free main loops.
// The default backward is the shared-pair kernel further down (lane = column).
// Measured alternatives on MI355X (profiles/r01_microbench_mi355x.txt): s_load_dwordx8 for the uniform
// rows (SGPR double buffers spill; scalar loads can only be waited for with lgkmcnt(0)): 3x slower for
// TransE_l1; "one VGPR = 64 operands" + v_readlane_b32 broadcasts: ~16 cycles per broadcast.
// The LDS-tiled kernels of kge_neg_pair.hip stay as the generic fallback for other row widths.
#include <utility>
#include "kge_common.hpp"
#include "kge_edge_fwd_body.hpp"

using namespace kge;
KGE_TL_DEFINE(bcast)

#define SB_KB 16                                 // real models: reduction elements per forward sub-slab
#define SB_KC 8                                  // RotatE: complex columns per sub-slab
#ifndef SB_KWR
#define SB_KWR 8                                 // backward, real models: output columns per wavefront
#endif
#ifndef SB_RWR
#define SB_RWR 4                                 // forward, real models: uniform rows per wavefront
#endif
#ifndef SB_RWC
#define SB_RWC 4                                 // forward, RotatE: uniform rows per wavefront (2: 19.4 us, 4: 16.4 us)
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
typedef float v2f __attribute__((ext_vector_type(2)));   // two fp32 in an even-aligned VGPR pair: v_pk_add/mul/fma_f32

template <int MODEL, int RW, int NE, int LB>
__device__ __forceinline__ void fwd_step(const float (&xc)[16], const float *ly, v2f (&acc)[RW], int lbr) {
    constexpr bool CPLX = MODEL == KGE_ROTATE;
    // two reduction elements (e, e+1) per step in packed fp32 math.  The uniform operands are read back from the
    // wavefront's LDS rows with ONE address for all lanes (a broadcast read: ds_read_b64, no VALU slot, unlike the
    // two v_readlane_b32 + hazard nops it replaces), the lane's own operands are adjacent VGPRs of the float4
    // loads, the accumulator is a pair as well (even / odd elements, summed at the end)
    static_for<NE / 2>([&](auto ec) {
        constexpr int e = 2 * decltype(ec)::value;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float *yp = ly + r * 64 + (LB >= 0 ? LB : lbr) + e;
#ifdef PROBE_FWD_NOY            // tuning probes (wrong results): timing without one ingredient of the loop
            (void)yp;
            const v2f y0 = {xc[(e + 2 * r) & 7], xc[(e + 2 * r + 1) & 7]};
            v2f y1 = {xc[8 + ((e + 2 * r) & 7)], xc[8 + ((e + 2 * r + 1) & 7)]};
#else
            const v2f y0 = *reinterpret_cast<const v2f *>(yp);
            v2f y1 = {0.f, 0.f};
            if constexpr (CPLX) y1 = *reinterpret_cast<const v2f *>(yp + RW * 64);
#endif
            const v2f x0 = {xc[e], xc[e + 1]};
            if constexpr (CPLX) {
                const v2f x1 = {xc[8 + e], xc[9 + e]};
                const v2f dr = y0 - x0, di = y1 - x1;
                const v2f m2 = __builtin_elementwise_fma(di, di, dr * dr);
#ifdef PROBE_FWD_NOSQRT
                acc[r] += m2;
#else
                acc[r] += (v2f){fast_sqrt(m2.x), fast_sqrt(m2.y)};
#endif
            } else if constexpr (MODEL == KGE_TRANSE_L1) {
                // one packed subtraction, then |.| as the free source modifier of two scalar additions (written
                // as asm: the vectoriser otherwise re-packs the additions and pays two v_and for the |.|)
                const v2f u = y0 - x0;
                float ax = acc[r].x, ay = acc[r].y;
                asm("v_add_f32 %0, |%1|, %0" : "+v"(ax) : "v"(u.x));
                asm("v_add_f32 %0, |%1|, %0" : "+v"(ay) : "v"(u.y));
                acc[r] = (v2f){ax, ay};
            } else if constexpr (MODEL == KGE_TRANSE_L2) {
                const v2f u = y0 - x0;
                acc[r] = __builtin_elementwise_fma(u, u, acc[r]);
            } else {
                acc[r] = __builtin_elementwise_fma(y0, x0, acc[r]);
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------
// forward: lanes = negatives j (score rows are then written coalesced), uniform = pos-side rows a_i.
// One task (= one wavefront) = (chunk, strip of 64 negatives, RW positives).  The uniform rows are
// fetched "lane = element" - yr[r] = a_{i0+r}[64*t + lane], ONE coalesced load = 64 operands - parked in the
// wavefront's own LDS rows and read back with one address for all lanes (broadcast ds_read, 4 operands per
// instruction); v_readlane_b32 broadcasts out of the register measured the same (13.4 vs 13.3 us TransE_l1,
// 19.5 vs 19.4 us RotatE) at twice the issue slots.  Inside a 64-element block, sub-slabs of 16 (real) /
// 8 complex elements of the lane's own row are double-buffered in VGPRs.  The main loop has NO conditionals
// (loads inside branches make the compiler wait for everything at the joins): prefetch addresses are clamped
// instead, and the partial last block runs in a separate tail loop.
// PMC (TransE_l1, cfg-T, per wavefront): 2 659 VALU + 414 LDS + 176 VMEM instructions; 29 % of the wave
// cycles issue VALU, 35 % wait on memory counters, 26 % wait for instruction issue.  The per-lane row loads
// (64 cache lines per instruction) are the RotatE bound (probe with one line per instruction: 19.5 -> 12.2 us)
// but not the TransE_l1 one (13.4 -> 13.0 us).
// ---------------------------------------------------------------------------------------------
// OTF (round 3, TransE_l1 in the merged first launch): the uniform rows are the pos-side vectors a_i = x_i + asign * r_i built on
// the fly from the table rows (x = head / tail through xidx, r through ridx: two coalesced loads and one fma per 64 elements of
// a uniform row instead of one load) - the kernel then does not wait for edge_fwd, which runs as the other half of the launch
// Round 3, occupancy: the kernel is bound by the instruction issue of its wavefronts (tools/valu_rate_probe.hip: one wavefront
// issues one VALU instruction per ~5.2 cycles whatever the kind, a SIMD with four resident wavefronts one per 2.2 (plain fp32) /
// 3.2 (packed) / 6.3 (v_sqrt) cycles - the RotatE mix costs 21.6 cycles per complex element and SIMD with one wavefront, 14.6 with
// four) and a task per wavefront left ONE wavefront per SIMD at the recipes' shapes (1 000 tasks on 1 024 SIMDs).  A task (chunk,
// strip of 64 negatives, RW positives) is therefore split over SB_KS wavefronts along the reduction (contiguous runs of
// sub-slabs); a workgroup = SB_TPB tasks x SB_KS runs = 16 wavefronts, four per SIMD (<= 128 VGPRs), the partial sums meet in
// LDS in fixed order (run 0 + run 1 + ...): deterministic, one barrier per workgroup.
// RotatE (arithmetic-heavy: a square root per complex element) gains from it; the real-valued models do not (TransE_l1: the
// per-lane row loads bound the kernel whatever the occupancy) and keep one wavefront per task.
#ifndef SB_KS_C
#define SB_KS_C 4                                // RotatE: wavefronts per task (runs of the reduction)
#endif
#ifndef SB_TPB_C
#define SB_TPB_C 4                               // RotatE: tasks per workgroup (consecutive: same strip, the x rows hit in L1)
#endif
#ifndef SB_KS_R
#define SB_KS_R 1                                // real-valued models
#endif
#ifndef SB_TPB_R
#define SB_TPB_R 4
#endif
template <int MODEL> struct FwdShape {
    static constexpr bool CPLX = MODEL == KGE_ROTATE;
    static constexpr int RW = CPLX ? SB_RWC : SB_RWR, KS = CPLX ? SB_KS_C : SB_KS_R, TPB = CPLX ? SB_TPB_C : SB_TPB_R;
    static constexpr int WAVES = KS * TPB, BLOCK = 64 * WAVES;
};

template <int MODEL, bool OTF>
__device__ __forceinline__ void neg_fwd_bcast_body(const NegArgs &a, int ns, int ng, int bid) {
    constexpr bool CPLX = MODEL == KGE_ROTATE;
    constexpr int RW = FwdShape<MODEL>::RW;                      // uniform rows per wavefront
    constexpr int SB_KS = FwdShape<MODEL>::KS, SB_TPB = FwdShape<MODEL>::TPB;
    constexpr int SB_FWD_WAVES = FwdShape<MODEL>::WAVES, SB_FWD_BLOCK = FwdShape<MODEL>::BLOCK;
    constexpr int NE = CPLX ? SB_KC : SB_KB;                     // (complex) elements per sub-slab
    constexpr int NSUB = 64 / NE;                                // sub-slabs per 64-element block
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // tasks are numbered with the row group fastest and a workgroup takes SB_TPB consecutive ones, so that
    // its wavefronts (almost always) share the strip - the x rows then hit in L1 - while the number of
    // workgroups carries no padding.  Wavefront = (run kw of the reduction, task tw): the SB_TPB wavefronts of a run are adjacent.
    const int tw = wave % SB_TPB, kw = wave / SB_TPB;
    const int ntask = a.C * ns * ng;
    const int task = min(bid * SB_TPB + tw, ntask - 1);          // (a padding wavefront repeats the last task and stores nothing)
    const int g = task % ng, st = (task / ng) % ns, c = task / (ng * ns);
    const int D = a.d_e, K = CPLX ? D / 2 : D;
    const int nsl = K / NE;                                      // sub-slabs of a row (K % NE == 0: neg_bcast_supported)
    const int k_lo = (kw * nsl / SB_KS) * NE, k_hi = ((kw + 1) * nsl / SB_KS) * NE;   // this wavefront's run [k_lo, k_hi)
    const int i0 = g * RW;
    const int j = st * 64 + lane;
    const float *x = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + min(j, a.N - 1), D);
    const float *y[RW], *yrel[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int64_t ie = (int64_t)c * a.chunk + min(i0 + r, a.chunk - 1);
        if constexpr (OTF) {
            y[r] = a.xbase + a.xidx[ie] * (int64_t)D;           // (wave-uniform ids: scalar loads)
            yrel[r] = a.rbase + a.ridx[ie] * (int64_t)D;
        } else {
            y[r] = a.A + ie * D;
            yrel[r] = y[r];
        }
    }
    const float asg = a.asign;
    v2f acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = (v2f){0.f, 0.f};

    auto loady = [&](float (&yr)[RW], float (&yi)[RW], int kb) {     // block of 64 elements: lane = element
        const int k = min(kb + lane, K - 1);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if constexpr (OTF) yr[r] = fmaf(asg, yrel[r][k], y[r][k]);      // = edge_fwd's x +/- r, bit for bit
            else yr[r] = y[r][k];
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
    // the uniform rows of the current 64-element block, wavefront-private in LDS: [RW][64] (+ [RW][64] imaginary).
    // Written "lane = element" straight from the coalesced loads, read back as broadcasts.  One wavefront's LDS
    // instructions execute in order: no barrier between its writes and its reads.
    __shared__ __attribute__((aligned(16))) float ylds[SB_FWD_WAVES][2 * RW * 64];
    __shared__ float part[SB_FWD_WAVES][RW][64];                 // partial sums of the runs
    float *ly = ylds[wave];
    auto puty = [&](const float (&yr_)[RW], const float (&yi_)[RW]) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            ly[r * 64 + lane] = yr_[r];
            if constexpr (CPLX) ly[(RW + r) * 64 + lane] = yi_[r];
        }
    };
    // the lane's own row: sub-slabs double-buffered in VGPRs, one ahead.  (A ring of four buffers requested three
    // sub-slabs ahead was SLOWER, 13.3 -> 17.9 us for TransE_l1: these per-lane row loads touch 64 cache lines per
    // instruction and more of them in flight only lengthen the queue in front of the texture addresser.)
    float yr[RW], yi[RW], ynr[RW], yni[RW], xa[16], xb[16];
    loady(yr, yi, k_lo);
    loadx(xa, k_lo);
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(1, 0);           // ids, row pointers and the first operands have arrived
#endif
    if constexpr (SB_KS == 1) {
        // one wavefront per task (real-valued models): full blocks unrolled with compile-time lane bases, branch-free (round 2's
        // loop: at one wavefront per SIMD the rolled form below costs TransE_l1 6 us per launch)
        int kb = 0;
        for (; kb + 64 <= K; kb += 64) {
            loady(ynr, yni, kb + 64);                            // next block, one block ahead
            puty(yr, yi);
            static_for<NSUB>([&](auto subc) {
                constexpr int SUB = decltype(subc)::value;
                float(&xc)[16] = (SUB & 1) ? xb : xa;
                float(&xn)[16] = (SUB & 1) ? xa : xb;
                loadx(xn, kb + (SUB + 1) * NE);                  // next sub-slab, one ahead
                fwd_step<MODEL, RW, NE, SUB * NE>(xc, ly, acc, 0);
            });
#pragma unroll
            for (int r = 0; r < RW; ++r) { yr[r] = ynr[r]; yi[r] = yni[r]; }
        }
        if (kb < K) puty(yr, yi);
        for (int k0 = kb; k0 < K; k0 += NE) {                    // partial last block (K % 64 elements)
            fwd_step<MODEL, RW, NE, -1>(xa, ly, acc, k0 - kb);
            loadx(xa, k0 + NE);
        }
    } else {
    // blocks of 64 elements (the last one of a run partial), sub-slabs with a run-time lane base in a ROLLED loop (the unrolled
    // block body above needs 180 VGPRs for RotatE; four wavefronts per SIMD have 128), the two register buffers alternating,
    // every load unconditional (clamped) and ahead of the arithmetic that precedes its use
    for (int kb = k_lo; kb < k_hi; kb += 64) {
        const int kend = min(kb + 64, k_hi);
        loady(ynr, yni, kb + 64);                                // next block, one block ahead
        puty(yr, yi);
#pragma unroll 1
        for (int k0 = kb; k0 < kend; k0 += 2 * NE) {
            const bool two = k0 + NE < kend;                     // (only the last block of a run can hold an odd number)
            loadx(xb, k0 + NE);
            fwd_step<MODEL, RW, NE, -1>(xa, ly, acc, k0 - kb);
            loadx(xa, two ? k0 + 2 * NE : k0 + NE);
            if (two) fwd_step<MODEL, RW, NE, -1>(xb, ly, acc, k0 + NE - kb);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) { yr[r] = ynr[r]; yi[r] = yni[r]; }
    }
    }
    if constexpr (SB_KS == 1) {                                 // one wavefront per task: its sums are final
        if (bid * SB_TPB + tw < ntask && j < a.N) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (i0 + r < a.chunk) {
                    float v = acc[r].x + acc[r].y;
                    if (MODEL == KGE_TRANSE_L1 || CPLX) v = a.gamma - v;
                    else if (MODEL == KGE_TRANSE_L2) v = a.gamma - sqrtf(fmaxf(v, 1e-30f));
                    else if (a.clampv > 0.f) v = fminf(fmaxf(v, -a.clampv), a.clampv);
                    a.S[((int64_t)c * a.chunk + i0 + r) * a.N + j] = v;
                }
            }
        }
        return;
    }
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(1, 1);           // this wavefront's run is done
#endif
#pragma unroll
    for (int r = 0; r < RW; ++r) part[wave][r][lane] = acc[r].x + acc[r].y;
    __syncthreads();
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(1, 2);           // every run of the workgroup is done
#endif
    // one output per thread and pass: (task tw2, row r, negative l); the runs are added in run order
    for (int o = threadIdx.x; o < SB_TPB * RW * 64; o += SB_FWD_BLOCK) {
        const int l = o & 63, r = (o >> 6) % RW, tw2 = (o >> 6) / RW;
        const int t2 = bid * SB_TPB + tw2;
        if (t2 >= ntask) continue;
        const int g2 = t2 % ng, st2 = (t2 / ng) % ns, c2 = t2 / (ng * ns);
        const int i2 = g2 * RW + r, j2 = st2 * 64 + l;
        if (j2 >= a.N || i2 >= a.chunk) continue;
        float v = part[tw2][r][l];
#pragma unroll
        for (int q = 1; q < SB_KS; ++q) v += part[q * SB_TPB + tw2][r][l];
        if (MODEL == KGE_TRANSE_L1 || CPLX) v = a.gamma - v;
        else if (MODEL == KGE_TRANSE_L2) v = a.gamma - sqrtf(fmaxf(v, 1e-30f));
        else if (a.clampv > 0.f) v = fminf(fmaxf(v, -a.clampv), a.clampv);
        a.S[((int64_t)c2 * a.chunk + i2) * a.N + j2] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// RotatE forward, round 4: the lanes' OWN rows (the 64 negatives of the strip) through LDS.
// PMC of the body above at the FB15k recipe (profiles/r04_rotate_pmc.txt): 7.9 M L1 accesses and 1.4 M L1->L2 line requests per
// launch (~180 MB: 10 TB/s out of the L2s over the kernel's 17 us) for 26 MB of distinct x data - a per-lane row load touches 64
// lines for 16 bytes each, the line comes back for the next sub-slab and for the three other tasks of the workgroup, and the 64 KB
// of lines a workgroup has open do not fit the 32-KB L1; the VALU is busy 40 % of the launch.
// Here the SB_TPB tasks of a run group (same strip, same run of the reduction) share ONE staged image of the strip's rows:
// 16 complex columns per stage, fetched in 64-byte segments (4 lanes per segment: 16 lines per load instruction instead of 64),
// one ds_write_b128 per thread and row half, read back "lane = row" with ds_read_b128 (row stride 36 dwords = 4 x odd:
// conflict-free, MI355X_MICROARCH.md LDS table).  Two buffers per run group, the next stage's global loads issued before the
// stage's arithmetic and written after it, one workgroup barrier per stage.  Same sums in the same order as the body above
// (sub-slab by sub-slab, even / odd accumulators, runs added in run order): bit-identical scores.
// ---------------------------------------------------------------------------------------------
#ifndef SB_XLDS
#define SB_XLDS 1
#endif
#define SB_XST 16                                // complex columns per stage
#define SB_XRS 36                                // dwords per staged row: 16 re | 16 im | 4 pad
__device__ __forceinline__ void neg_fwd_rot_xlds_body(const NegArgs &a, int ns, int ng, int bid) {
    typedef FwdShape<KGE_ROTATE> FS;
    constexpr int RW = FS::RW, SB_KS = FS::KS, SB_TPB = FS::TPB, WAVES = FS::WAVES, BLOCK = FS::BLOCK, NE = SB_KC;
    static_assert(SB_TPB * 64 == 256 && SB_XST == 2 * NE, "staging map: 256 threads per run group, two sub-slabs per stage");
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int tw = wave % SB_TPB, kw = wave / SB_TPB;
    const int ntask = a.C * ns * ng;
    const int task = min(bid * SB_TPB + tw, ntask - 1);
    const int g = task % ng, st = (task / ng) % ns, c = task / (ng * ns);
    // the strip whose rows this run group stages: the strip of the workgroup's FIRST task (a workgroup whose tasks straddle two
    // strips - ng % SB_TPB != 0 - is sent to the per-lane body by the launcher)
    const int D = a.d_e, K = D / 2;
    const int nsl = K / NE;
    const int k_lo = (kw * nsl / SB_KS) * NE, k_hi = ((kw + 1) * nsl / SB_KS) * NE;
    const int lmax = (nsl + SB_KS - 1) / SB_KS;                  // sub-slabs of the longest run
    const int nst = (lmax + 1) / 2;                              // stages every wavefront walks (same barrier count for all)
    const int i0 = g * RW;
    const int j = st * 64 + lane;
    const float *y[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) y[r] = a.A + ((int64_t)c * a.chunk + min(i0 + r, a.chunk - 1)) * D;
    v2f acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = (v2f){0.f, 0.f};

    __shared__ __attribute__((aligned(16))) float ylds[WAVES][2 * RW * 64];
    __shared__ __attribute__((aligned(16))) float xs[SB_KS][2][64 * SB_XRS];
    __shared__ float part[WAVES][RW][64];
    float *ly = ylds[wave];
    auto loady = [&](float (&yr)[RW], float (&yi)[RW], int kb) {
        const int k = min(kb + lane, K - 1);
#pragma unroll
        for (int r = 0; r < RW; ++r) { yr[r] = y[r][k]; yi[r] = y[r][K + k]; }
    };
    auto puty = [&](const float (&yr_)[RW], const float (&yi_)[RW]) {
#pragma unroll
        for (int r = 0; r < RW; ++r) { ly[r * 64 + lane] = yr_[r]; ly[(RW + r) * 64 + lane] = yi_[r]; }
    };
    // staging map of the run group's 256 threads: thread tg -> rows tg / 8 and tg / 8 + 32, piece tg % 8 of the row's 32 staged
    // floats (pieces 0-3: re columns, 4-7: im columns); 8 consecutive lanes write 32 consecutive dwords
    const int tg = tw * 64 + lane, srow = tg >> 3, sp = tg & 7, scomp = sp >> 2, sq4 = (sp & 3) * 4;
    const float *xg0 = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + min(st * 64 + srow, a.N - 1), D) + scomp * K;
    const float *xg1 = row_ptr(a.nbase, a.nidx, (int64_t)c * a.N + min(st * 64 + srow + 32, a.N - 1), D) + scomp * K;
    float *xw = &xs[kw][0][srow * SB_XRS + sp * 4];
    const float *xr_ = &xs[kw][0][lane * SB_XRS];
    float4 sx0, sx1;
    auto gx = [&](int k0) {                                      // global -> registers (columns clamped: in bounds, unused beyond k_hi)
        const int kc = min(k0 + sq4, K - 4);
        sx0 = *reinterpret_cast<const float4 *>(xg0 + kc);
        sx1 = *reinterpret_cast<const float4 *>(xg1 + kc);
    };
    auto sxw = [&](int bf) {                                     // registers -> LDS buffer bf
        *reinterpret_cast<float4 *>(xw + bf * (64 * SB_XRS)) = sx0;
        *reinterpret_cast<float4 *>(xw + bf * (64 * SB_XRS) + 32 * SB_XRS) = sx1;
    };
    auto xread = [&](float (&xc)[16], int bf, int sub) {         // this lane's row: 8 re + 8 im columns of sub-slab `sub` of the stage
        const float *p = xr_ + bf * (64 * SB_XRS) + sub * NE;
        const float4 t0 = *reinterpret_cast<const float4 *>(p), t1 = *reinterpret_cast<const float4 *>(p + 4);
        const float4 u0 = *reinterpret_cast<const float4 *>(p + 16), u1 = *reinterpret_cast<const float4 *>(p + 20);
        xc[0] = t0.x; xc[1] = t0.y; xc[2] = t0.z; xc[3] = t0.w; xc[4] = t1.x; xc[5] = t1.y; xc[6] = t1.z; xc[7] = t1.w;
        xc[8] = u0.x; xc[9] = u0.y; xc[10] = u0.z; xc[11] = u0.w; xc[12] = u1.x; xc[13] = u1.y; xc[14] = u1.z; xc[15] = u1.w;
    };
    float yr[RW], yi[RW], ynr[RW], yni[RW], xa[16], xb[16];
    gx(k_lo);
    loady(yr, yi, k_lo);
    sxw(0);
    __syncthreads();
    int bf = 0;
    for (int s0 = 0; s0 < nst; s0 += 4) {                        // blocks of 64 columns = 4 stages
        const int kb = k_lo + s0 * SB_XST;
        loady(ynr, yni, kb + 64);                                // next block, one block ahead
        puty(yr, yi);
        const int send = min(4, nst - s0);
#pragma unroll 1
        for (int s = 0; s < send; ++s) {
            const int k0 = kb + s * SB_XST;
#ifdef PROBE_FWD_NOX            // tuning probe (wrong results): no staging, no barrier, x read once
            if (s0 == 0 && s == 0) { xread(xa, bf, 0); xread(xb, bf, 1); }
            if (k0 < k_hi) fwd_step<KGE_ROTATE, RW, NE, -1>(xa, ly, acc, k0 - kb);
            if (k0 + NE < k_hi) fwd_step<KGE_ROTATE, RW, NE, -1>(xb, ly, acc, k0 + NE - kb);
#else
            gx(k0 + SB_XST);                                     // next stage (past the run: a harmless clamped re-read)
            xread(xa, bf, 0);
            xread(xb, bf, 1);
            if (k0 < k_hi) fwd_step<KGE_ROTATE, RW, NE, -1>(xa, ly, acc, k0 - kb);            // (wave-uniform)
            if (k0 + NE < k_hi) fwd_step<KGE_ROTATE, RW, NE, -1>(xb, ly, acc, k0 + NE - kb);
            sxw(bf ^ 1);
            __syncthreads();
            bf ^= 1;
#endif
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) { yr[r] = ynr[r]; yi[r] = yni[r]; }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) part[wave][r][lane] = acc[r].x + acc[r].y;
    __syncthreads();
    for (int o = threadIdx.x; o < SB_TPB * RW * 64; o += BLOCK) {
        const int l = o & 63, r = (o >> 6) % RW, tw2 = (o >> 6) / RW;
        const int t2 = bid * SB_TPB + tw2;
        if (t2 >= ntask) continue;
        const int g2 = t2 % ng, st2 = (t2 / ng) % ns, c2 = t2 / (ng * ns);
        const int i2 = g2 * RW + r, j2 = st2 * 64 + l;
        if (j2 >= a.N || i2 >= a.chunk) continue;
        float v = part[tw2][r][l];
#pragma unroll
        for (int q = 1; q < SB_KS; ++q) v += part[q * SB_TPB + tw2][r][l];
        a.S[((int64_t)c2 * a.chunk + i2) * a.N + j2] = a.gamma - v;
    }
    (void)j;
}

// XCD-aware block order (round 6).  Hardware block b runs on XCD b % 8 and every XCD has its own L2.  Consecutive LOGICAL workgroups
// share operands - the forward's 16 workgroups of a (chunk, strip) read the same 64 negative rows (205 KB at cfg-R), the backward's
// workgroups of a (chunk, column slab) the same slab of A and of the negatives - and in hardware order they sat on eight different
// XCDs: every L2 fetched every strip (PMC FETCH_SIZE of the forward 29 MB for 6.6 MB of distinct operands at cfg-R,
// profiles/r05_rotate_wide_pmc_FETCH_SIZE.txt).  xcd_remap hands an XCD a contiguous range of logical ids.  -DSB_NO_XCD: hardware order.
__device__ __forceinline__ int sb_block(int b, int nb) {
#ifdef SB_NO_XCD
    (void)nb; return b;
#else
    return kge::xcd_remap(b, nb);
#endif
}
template <int MODEL>
__global__ __launch_bounds__(FwdShape<MODEL>::BLOCK) void neg_fwd_bcast_kernel(NegArgs a, int ns, int ng) {
    KGE_TL(1);
    neg_fwd_bcast_body<MODEL, false>(a, ns, ng, sb_block((int)blockIdx.x, (int)gridDim.x));
}
__global__ __launch_bounds__(FwdShape<KGE_ROTATE>::BLOCK) void neg_fwd_rot_xlds_kernel(NegArgs a, int ns, int ng) {
    KGE_TL(1);
    neg_fwd_rot_xlds_body(a, ns, ng, sb_block((int)blockIdx.x, (int)gridDim.x));
}

// TransE_l1, strict step: forward pairwise tasks (first nbF workgroups) + the edge-forward rows of the SAME step (the rest: one
// wavefront per row, SB_FWD_WAVES rows per workgroup - edge_fwd_body counts in workgroups of KGE_WAVES_PER_BLOCK wavefronts)
template <bool LEAN>
__global__ __launch_bounds__(FwdShape<KGE_TRANSE_L1>::BLOCK) void neg_fwd_bcast_edge_kernel(NegArgs a, int ns, int ng, int nbF, EdgeFwdArgs e) {
    static_assert(FwdShape<KGE_TRANSE_L1>::WAVES % KGE_WAVES_PER_BLOCK == 0, "edge_fwd_body counts workgroups of KGE_WAVES_PER_BLOCK wavefronts");
    if ((int)blockIdx.x < nbF) neg_fwd_bcast_body<KGE_TRANSE_L1, true>(a, ns, ng, sb_block((int)blockIdx.x, nbF));
    else edge_fwd_body<KGE_TRANSE_L1, 4, LEAN>(e, ((int)blockIdx.x - nbF) * (FwdShape<KGE_TRANSE_L1>::WAVES / KGE_WAVES_PER_BLOCK));
}

bool neg_fwd_bcast_with_edge_supported(int model, int d_e, int d_r) {
    return model == KGE_TRANSE_L1 && d_r == d_e && d_e % 4 == 0 && neg_bcast_supported(model, d_e);
}

int launch_neg_fwd_bcast_with_edge(const NegArgs &a, const EdgeFwdArgs &e, hipStream_t s) {
    if (!neg_fwd_bcast_with_edge_supported(a.model, a.d_e, e.d_r) || e.model != a.model || !a.xbase || !a.xidx || !a.rbase ||
        !a.ridx || e.src.em.n || e.src.rm.n || e.nd_own)
        return KGE_ERR_ARG;
    typedef FwdShape<KGE_TRANSE_L1> FS;
    constexpr int RW = FS::RW, SB_TPB = FS::TPB, SB_FWD_WAVES = FS::WAVES, SB_FWD_BLOCK = FS::BLOCK;
    const int ns = (a.N + 63) / 64, ng = (a.chunk + RW - 1) / RW;
    const int nbF = (int)(((int64_t)a.C * ns * ng + SB_TPB - 1) / SB_TPB);
    const bool negjob = e.bsq || e.Bn;
    const int64_t waves = (int64_t)e.B + (negjob ? e.n_neg : 0);
    EdgeFwdArgs ee = e;
    if (!negjob) ee.n_neg = 0;
    const int nbP = (int)((waves + SB_FWD_WAVES - 1) / SB_FWD_WAVES);
    if (nbF == 0) return KGE_ERR_ARG;
    const bool lean = e.lp.genre == KGE_LOSS_LOGSIGMOID && !e.row_pos && !e.Hc;
    if (lean) hipLaunchKernelGGL(neg_fwd_bcast_edge_kernel<true>, dim3(nbF + nbP), dim3(SB_FWD_BLOCK), 0, s, a, ns, ng, nbF, ee);
    else hipLaunchKernelGGL(neg_fwd_bcast_edge_kernel<false>, dim3(nbF + nbP), dim3(SB_FWD_BLOCK), 0, s, a, ns, ng, nbF, ee);
    return check_launch_b();
}

template <int MODEL> static int fwd_launch(const NegArgs &a, hipStream_t s) {
    constexpr int RW = FwdShape<MODEL>::RW, SB_TPB = FwdShape<MODEL>::TPB, SB_FWD_BLOCK = FwdShape<MODEL>::BLOCK;
    const int ns = (a.N + 63) / 64, ng = (a.chunk + RW - 1) / RW;
    const int64_t nb = ((int64_t)a.C * ns * ng + SB_TPB - 1) / SB_TPB;
    if (nb == 0) return KGE_OK;
    if constexpr (MODEL == KGE_ROTATE && SB_XLDS) {
        // staged-row instance: the tasks of a workgroup share their strip (ng % SB_TPB == 0) and a stage of 16 columns exists
        if (ng % SB_TPB == 0 && a.d_e / 2 >= SB_XST && (a.d_e / 2) % 4 == 0) {
            hipLaunchKernelGGL(neg_fwd_rot_xlds_kernel, dim3((unsigned)nb), dim3(SB_FWD_BLOCK), 0, s, a, ns, ng);
            return check_launch_b();
        }
    }
    hipLaunchKernelGGL(neg_fwd_bcast_kernel<MODEL>, dim3((unsigned)nb), dim3(SB_FWD_BLOCK), 0, s, a, ns, ng);
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

// ---------------------------------------------------------------------------------------------
// backward, "lane = column" layout (TransE_l1 and RotatE): ONE evaluation of every pair (i, j) feeds BOTH
// products.  The kernels above compute GA and GN in separate passes because a lane owns an output row and the
// other operand is broadcast - every pair difference, modulus and reciprocal root is evaluated twice.  Here a
// lane owns two adjacent (complex) columns instead:
//     lane = (sg, kk):  columns k0 + 2 kk, k0 + 2 kk + 1  (packed fp32 pairs: v_pk_add/mul/fma_f32),
//                       sg = lane / 16 selects one of the 4 negatives of a quad  s = 4 q + sg.
// A wavefront keeps RT positive rows x_n (and their GA accumulators) in VGPRs and streams the quads of
// negatives (operands staged through LDS, see the kernel): y = b_s[cols], W(r0 + n, s) comes from ONE register per quad
// (lane (sg, kk) holds W(r0 + kk, 4 q + sg)) through the DPP row_newbcast:n modifier (lane n of every row of
// 16 lanes).  Per pair and per two columns:  d = x - y, m2 = |d|^2, rsq, iv = w rsq, GA_n -= iv d, GN_s += iv d.
// GA needs the sum over the 4 lane rows at the very end (two cross-row shuffles per value); GN is complete
// over the wavefront's RT rows after every quad and is summed over the 4 wavefronts of the workgroup (= 4 row
// blocks of the same chunk and column slab) in LDS, in fixed order, every LC_GQ quads; what remains is one
// partial per workgroup row group (chunk / (4 RT), 7 for chunk = 200), added up - with the regulariser of the
// negative rows - by gn_reduce_kernel.  Deterministic (no atomics).
// MI355X, per launch (profiles/r01_microbench_mi355x.txt): TransE_l1 cfg-T 40.0 -> 24.3 + 5.1 us (reduce),
// RotatE (B 1024, N 256, 200 complex columns) 58.8 -> 27.9 + 5.4 us; arithmetic alone (no loads, no LDS) 13 / 10 us.
// ---------------------------------------------------------------------------------------------
#define LC_CW 32                                 // (complex) columns per wavefront: 16 lanes x 2
#ifndef LC_GQ_CPLX
#define LC_GQ_CPLX 4                             // quads of negatives per group (staging + LDS reduction round), RotatE
#endif
#define LC_GQ_REAL 8                             // ... TransE_l1; both sized so that TWO workgroups fit the LDS of a CU
#define LC_RTMAX 20                              // most positive rows a wavefront keeps in registers (RotatE: 16)

__device__ __forceinline__ float4 zero4b() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// the GN partials / GA parts leave write-through (kge_common.hpp KGE_ST_NEXT): 13 MB (FB15k shape) - 26 MB (cfg-R) of lines left dirty in the XCDs' L2s
// were written back at the end of the kernel, in front of the next launch (gap 3.1 us instead of 1.2-1.3, tools/timeline.py)
__device__ __forceinline__ void st_v2(float *p, v2f x) {
#ifdef KGE_PLAIN_INTERMEDIATES
    *reinterpret_cast<v2f *>(p) = x;
#else
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
#endif
}
template <int L> __device__ __forceinline__ float rowb(float v) {    // lane L of every row of 16 lanes
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + L, 0xF, 0xF, true));
}

// workgroups per (chunk, slab) and rows per wavefront: at least as many workgroups as keep every wavefront at
// <= LC_RTMAX rows; more (down to 8 rows per wavefront) while the launch still fits TWO workgroups per CU (the
// kernel is bound by the instruction issue of its wavefronts: measured 31 -> 24 us for TransE_l1 at cfg-T)
// TransE_l1 (measured at cfg-T, us/step): its backward is bound by LDS traffic per pair (operands and GN partials through LDS
// per quad) rather than by VALU issue - more rows per wavefront amortise it, split negatives restore the workgroup count:
// 8 rows x 455 workgroups 58.3; 8 rows, 2 splits 58.6; 10-12 rows x 325 workgroups x 3 splits 55.9 (taken); 13-16 rows x 260 x 3
// 59.6; 9 rows x 390 x 2 64.7; 10-12 rows unsplit 64.0
#ifndef LC_RPW_MIN_REAL
#define LC_RPW_MIN_REAL 10                       // TransE_l1: fewest rows a wavefront keeps when the launch can afford more workgroups
#endif
#ifndef LC_RPW_MIN_CPLX
#define LC_RPW_MIN_CPLX 8
#endif
#ifndef LC_NO_SPLIT_REAL
#define LC_SPLIT_REAL 1
#endif
// Wavefronts per workgroup (round 6).  The GN partials are one per WORKGROUP row group: a workgroup of 4 wavefronts x 8 rows sums 32
// positives in LDS and cfg-R (chunk 256, D_e 800) left 8 partials per negative row - 26 MB written write-through and read back by the
// reduction, the largest single item of that step's fabric traffic (profiles/r05_rotate_wide_pmc_*).  8 wavefronts per workgroup (64
// positives meet in LDS) halve them at the same rows per wavefront, the same occupancy (two workgroups of 8 instead of four of 4 per
// CU) and the same order of additions inside a wavefront; round 4 measured the launch times a wash (backward +0.5 us, reduction -0.4)
// and dropped it, round 6 takes it for the bytes.  Wide RotatE rows only: at the FB15k shapes the partials are small.
#ifndef LC_WPB8_MIN_DE
#define LC_WPB8_MIN_DE 512
#endif
static inline int lc_wpb(int model, int d_e) {
#ifdef LC_NO_WPB8
    (void)model; (void)d_e; return 4;
#else
    return (model == KGE_ROTATE && d_e >= LC_WPB8_MIN_DE) ? 8 : 4;
#endif
}
static inline void lc_shape(int model, int C, int chunk, int d_e, int &nslab, int &nrw, int &rpw) {
    const int K = model == KGE_ROTATE ? d_e / 2 : d_e;
    const int wpb = lc_wpb(model, d_e);
    nslab = (K + LC_CW - 1) / LC_CW;
    const int rtmax = model == KGE_ROTATE ? 16 : LC_RTMAX;
    nrw = 1;
    while ((chunk + wpb * nrw - 1) / (wpb * nrw) > rtmax) ++nrw;
    const int rpw_min = model == KGE_ROTATE ? LC_RPW_MIN_CPLX : LC_RPW_MIN_REAL;
    // (one more row group only while it still SHORTENS the wavefronts' row blocks: chunk = 256 went to nrw = 9 with 8 rows per
    //  wavefront - the ninth group held no rows and ran the whole arithmetic on zero weights, 11 % of the launch; round 4)
    auto rows = [&](int n) { return (chunk + wpb * n - 1) / (wpb * n); };
    for (int n = nrw + 1; (int64_t)C * nslab * n <= 2048 / wpb && rows(n) >= rpw_min; ++n)
        if (rows(n) < rows(nrw)) nrw = n;
    rpw = rows(nrw);
}
bool neg_bwd_lc_supported(int model, int d_e) {
    if (model == KGE_ROTATE) return d_e % 4 == 0 && neg_bcast_supported(model, d_e);
    return model == KGE_TRANSE_L1 && d_e % 2 == 0 && neg_bcast_supported(model, d_e);
}
size_t neg_bwd_lc_partial_floats(int model, int C, int chunk, int N, int d_e) {
    int nslab, nrw, rpw;
    lc_shape(model, C, chunk, d_e, nslab, nrw, rpw);
    return (size_t)nrw * C * N * d_e;
}

// LC_WPE: wavefronts per SIMD the register allocation aims at where the instantiation fits without spilling (RotatE: 8 rows per
// wavefront, TransE_l1: up to 16); LC_RED_SINGLE: one buffer of GN partials and a second barrier per group (32 instead of 48 KB
// of LDS).  Together: four workgroups per CU - every workgroup of the split RotatE launch resident at once (60.9 vs 61.7 us/step)
#ifndef LC_NO_OCC4
#define LC_WPE 4
#define LC_RED_SINGLE 1
#endif
#ifdef LC_WPE
#define LC_OCC __attribute__((amdgpu_waves_per_eu((RT <= (MODEL == KGE_ROTATE ? 8 : 16)) ? LC_WPE : 1, (RT <= (MODEL == KGE_ROTATE ? 8 : 16)) ? LC_WPE : 8)))
#else
#define LC_OCC
#endif
#ifndef LC_WG_CAP
#define LC_WG_CAP 1024                           // negatives are split while the launch stays within this many workgroups
#endif
template <int MODEL, int RT, int WPB = KGE_WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WPB) LC_OCC void neg_bwd_lc_kernel(NegArgs a, int nslab, int nrw, int rpw) {
    KGE_TL(3);
    constexpr int LC_BLOCK = 64 * WPB;                           // threads of the workgroup (WPB wavefronts: lc_wpb)
    constexpr bool CPLX = MODEL == KGE_ROTATE;
    constexpr int NV = CPLX ? 4 : 2;                             // floats per lane in a GN partial
    constexpr int LC_GQ = CPLX ? LC_GQ_CPLX : LC_GQ_REAL, LC_SG = 4 * LC_GQ;   // quads / negatives per group
#ifdef LC_RED_SINGLE
    constexpr int NRED = 1;                                      // one buffer of GN partials + a second barrier per group: 32 KB of LDS, 4 workgroups per CU
#else
    constexpr int NRED = 2;                                      // two alternating buffers
#endif
    __shared__ __attribute__((aligned(16))) float red[NRED * LC_GQ * WPB * 64 * NV];   // GN partials
    __shared__ __attribute__((aligned(16))) float stage[2 * LC_SG * LC_CW * (CPLX ? 2 : 1) + 2 * WPB * LC_SG * (RT > 16 ? 32 : 16)];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, sg = lane >> 4, kk = lane & 15;
    // a.ga_parts > 1: the quads of the chunk's negatives are split over that many workgroups per (chunk, slab, row group) - three
    // or four wavefronts per SIMD instead of one (the kernel is bound by the instruction issue of its wavefronts,
    // tools/valu_rate_probe.hip); every negative still has ONE workgroup per row group (GN partials unchanged), GA leaves as
    // ga_parts partial sums that the consumer adds
    const int nq = (a.N + 3) >> 2;
    const int ngr = (nq + LC_GQ - 1) / LC_GQ;                    // groups of quads of a column
    int nsp = max(a.ga_parts, 1), sp, blk;                       // this workgroup: part sp of nsp of column blk
    int g_lo, g_hi;                                              // ... = groups [g_lo, g_hi)
    if (a.lc_P > 0) {
        // balanced split (NegArgs::lc_P): sorted index i of this block id (rounds of 256 ids, odd rounds reversed), then the
        // i-th heaviest workgroup: classes in descending size - big / small parts of the lc_P-part columns, then of the others
        // (round 6: inside a round of 256 block ids the XCD of id k - k % 8 - takes the 32 consecutive sorted indices
        //  [32 (k % 8), 32 (k % 8) + 32): parts of the same (chunk, slab) columns share an L2; the heaviest-first dealing over the
        //  rounds is unchanged per CU slot)
        const int b = (int)blockIdx.x, rnd = b >> 8;
#ifdef SB_NO_XCD
        const int k = b & 255;
#else
        const int k = ((b & 7) << 5) | ((b & 255) >> 3);
#endif
        int t = (rnd << 8) + ((rnd & 1) ? 255 - k : k);
        const int P = a.lc_P, nB = a.lc_nB, ncol = a.C * nslab * nrw, nA = ncol - nB;
        const int remB = ngr % P, remA = ngr % (P + 1);
        const int n1 = nB * remB, n2 = nB * (P - remB), n3 = nA * remA;
        int j;
        if (t < n1) { blk = t / remB; j = t % remB; nsp = P; }
        else if (t < n1 + n2) { t -= n1; blk = t / (P - remB); j = remB + t % (P - remB); nsp = P; }
        else if (t < n1 + n2 + n3) { t -= n1 + n2; blk = nB + t / remA; j = t % remA; nsp = P + 1; }
        else { t -= n1 + n2 + n3; blk = nB + t / (P + 1 - remA); j = remA + t % (P + 1 - remA); nsp = P + 1; }
        sp = j;
        const int base = ngr / nsp, rem = ngr % nsp;             // the first `rem` parts hold one group more
        g_lo = j * base + min(j, rem); g_hi = g_lo + base + (j < rem ? 1 : 0);
    } else {
        const int lb = sb_block((int)blockIdx.x, (int)gridDim.x);
        sp = lb % nsp; blk = lb / nsp;
        g_lo = sp * ngr / nsp; g_hi = (sp + 1) * ngr / nsp;
    }
    const int rw = blk % nrw, slab = (blk / nrw) % nslab, c = blk / (nrw * nslab);
    const int D = a.d_e, K = CPLX ? D / 2 : D, N = a.N, chunk = a.chunk;
    const int col = slab * LC_CW + 2 * kk;
    const bool colok = col < K;                                  // K is even: both columns or none
    const int colc = colok ? col : 0;
    const int r0 = (rw * WPB + wave) * rpw;                      // first positive row of this wavefront
    const int rend = min(r0 + rpw, chunk);                       // rows [r0, rend)

    v2f xr[RT], xi[RT], gr[RT], gi[RT];
#pragma unroll
    for (int n = 0; n < RT; ++n) {
        const float *xp = a.A + ((int64_t)c * chunk + min(r0 + n, chunk - 1)) * D + colc;
        xr[n] = *reinterpret_cast<const v2f *>(xp);
        xi[n] = CPLX ? *reinterpret_cast<const v2f *>(xp + K) : (v2f){0.f, 0.f};
        gr[n] = (v2f){0.f, 0.f};
        gi[n] = (v2f){0.f, 0.f};
    }
    const float *Wc = a.W + (int64_t)c * chunk * N;
    const int q_lo = g_lo * LC_GQ, q_hi = min(g_hi * LC_GQ, nq);
    // ---- operand staging through LDS, one group of LC_GQ quads (= LC_SG negatives) at a time ----------------
    // The 4 wavefronts of the workgroup need the SAME negative rows (their slab columns) and each its own block
    // of W.  The global loads of group i+1 are issued BEFORE the arithmetic of group i and written to the other
    // LDS buffer after it: their latency (an L2 miss is ~1.5 k cycles, a quad only ~0.3-0.6 k cycles of
    // arithmetic, one wavefront per SIMD) is covered by a whole group of work; inside a group the operands of a
    // quad come from LDS, requested one quad ahead.  (Register slots filled straight from global memory either
    // stall - the compiler drains every outstanding load at the loop head - or, unrolled far enough to avoid
    // that, blow up the register allocation.)
    constexpr int YF = LC_CW * (CPLX ? 2 : 1);                   // floats per negative in the y buffer
    constexpr int RTW = RT > 16 ? 32 : 16;                       // W rows per wavefront in LDS
    float *ybuf = stage;                                         // [2][LC_SG][YF]
    float *wbuf = stage + 2 * LC_SG * YF;                        // [2][4 wavefronts][LC_SG][RTW]  (negative-major)
    const int tid = threadIdx.x;
    const int ys = tid >> 3, yp4 = (tid & 7) * 4;                // y staging: negative ys (< LC_SG), 4 columns from yp4
    const bool ystg = ys < LC_SG;
    const int ycol = slab * LC_CW + yp4;
    const int ycolc = ycol < K ? ycol : 0;
    constexpr int NP = LC_SG / 4, RPP = 64 / NP, NPASS = RTW / RPP;   // W staging: float4 parts per row, rows per pass
    const int wrow = lane / NP, wp4 = (lane % NP) * 4;           // rows wrow (+RPP, ...), 4 negatives from wp4
    float4 sy0 = zero4b(), sy1 = zero4b(), sw[NPASS];
    auto gload = [&](int qb) {                                   // global -> registers (group starting at quad qb)
        if (ystg) {
            const int s = min(4 * qb + ys, N - 1);
            const float *yp = a.nbase + ((int64_t)c * N + s) * D + ycolc;
            sy0 = *reinterpret_cast<const float4 *>(yp);
            if constexpr (CPLX) sy1 = *reinterpret_cast<const float4 *>(yp + K);
        }
        const int s4 = min(4 * qb + wp4, N - 4);                 // N % 4 == 0: whole float4 or (masked later) a re-read
#pragma unroll
        for (int t = 0; t < NPASS; ++t)
            sw[t] = *reinterpret_cast<const float4 *>(Wc + (int64_t)min(r0 + wrow + RPP * t, chunk - 1) * N + s4);
    };
    // registers -> LDS buffer bf.  W is zeroed HERE for rows outside this wavefront's block and negatives beyond
    // N (the values arrived a whole group ago, the selects cost nothing): the arithmetic needs no masks at all.
    auto lstore = [&](int bf, int qb) {
        if (ystg) {
            float *yb = ybuf + (bf * LC_SG + ys) * YF;
            *reinterpret_cast<float4 *>(yb + yp4) = sy0;
            if constexpr (CPLX) *reinterpret_cast<float4 *>(yb + LC_CW + yp4) = sy1;
        }
        float *wb = wbuf + ((bf * WPB + wave) * LC_SG + wp4) * RTW + wrow;
        const int s4 = 4 * qb + wp4;
        const bool sval = qb < nq && s4 + 3 < N;                 // N % 4 == 0: a float4 of negatives is all in or all out
#pragma unroll
        for (int t = 0; t < NPASS; ++t) {
            const bool ok = sval && wrow + RPP * t < RT && r0 + wrow + RPP * t < rend;
            wb[RPP * t] = ok ? sw[t].x : 0.f; wb[RTW + RPP * t] = ok ? sw[t].y : 0.f;
            wb[2 * RTW + RPP * t] = ok ? sw[t].z : 0.f; wb[3 * RTW + RPP * t] = ok ? sw[t].w : 0.f;
        }
    };
    // operands of one quad out of LDS buffer bf: y = b_s[cols] (s = quad g, lane row sg), raw W values
    auto lread = [&](int bf, int g, v2f &yr_, v2f &yi_, float &w0_, float &w1_) {
#ifdef PROBE_BWD_NOLREAD        // tuning probe (wrong results): operands of a quad without the LDS round
        yr_ = xr[g & 3] * 0.5f; yi_ = xi[g & 3] * 0.5f; w0_ = xr[0].x; w1_ = 0.f; (void)bf;
        return;
#endif
        const float *yb = ybuf + (bf * LC_SG + 4 * g + sg) * YF + 2 * kk;
        yr_ = *reinterpret_cast<const v2f *>(yb);
        yi_ = CPLX ? *reinterpret_cast<const v2f *>(yb + LC_CW) : (v2f){0.f, 0.f};
        const float *wb = wbuf + ((bf * WPB + wave) * LC_SG + 4 * g + sg) * RTW + kk;
        w0_ = wb[0];
        w1_ = RT > 16 ? wb[16] : 0.f;
    };
    // one quad: RT rows of this wavefront against the 4 negatives of the quad; GN partial -> LDS
    auto quad = [&](int g, float *redb, const v2f &yr_, const v2f &yi_, float wa, float wb) {
        v2f nr = {0.f, 0.f}, ni = {0.f, 0.f};
        static_for<RT>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            const float w = n < 16 ? rowb<(n & 15)>(wa) : rowb<(n & 15)>(wb);
            if constexpr (CPLX) {
                const v2f dr = xr[n] - yr_, di = xi[n] - yi_;
                const v2f m2 = __builtin_elementwise_fma(di, di, __builtin_elementwise_fma(dr, dr, (v2f){1e-30f, 1e-30f}));
#ifdef PROBE_BWD_NORSQ
                const v2f iv = m2 * w;
#else
                const v2f iv = (v2f){fast_rsq(m2.x), fast_rsq(m2.y)} * w;   // + tiny: zero difference -> zero gradient
#endif
                gr[n] = __builtin_elementwise_fma(dr, -iv, gr[n]);
                gi[n] = __builtin_elementwise_fma(di, -iv, gi[n]);
                nr = __builtin_elementwise_fma(dr, iv, nr);
                ni = __builtin_elementwise_fma(di, iv, ni);
            } else {
                const v2f d = xr[n] - yr_;
                // sign(d) = med3(d * 2^126, -1, 1): exact for every normal d, 0 at d == 0
                const v2f sgn = {__builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(d.x, 126), -1.f, 1.f),
                                 __builtin_amdgcn_fmed3f(__builtin_amdgcn_ldexpf(d.y, 126), -1.f, 1.f)};
                const v2f wv = {w, w};
                gr[n] = __builtin_elementwise_fma(sgn, -wv, gr[n]);
                nr = __builtin_elementwise_fma(sgn, wv, nr);
            }
        });
        float *slot = redb + ((g * WPB + wave) * 64 + lane) * NV;
#ifdef PROBE_BWD_NORED          // tuning probe (wrong results): the GN partials stay in registers (folded into GA so that they are not dead)
        gr[0] += nr; if constexpr (CPLX) gi[0] += ni;
        (void)slot;
#else
        if constexpr (CPLX) *reinterpret_cast<float4 *>(slot) = make_float4(nr.x, nr.y, ni.x, ni.y);
        else *reinterpret_cast<v2f *>(slot) = nr;
#endif
    };
    gload(min(q_lo, nq - 1));
    lstore(0, q_lo);
    // every load of the prologue (x rows included) has landed before the loop: otherwise the compiler keeps
    // "x may still be in flight" alive around the back edge and makes the first quad of every group wait for the
    // previous group's partial-sum stores
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
    __syncthreads();
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(3, 0);           // own rows and the first group of operands are staged
#endif
    int buf = 0;
    for (int qb = q_lo; qb < q_hi; qb += LC_GQ) {
        float *redb = red + (NRED == 2 ? buf : 0) * (LC_GQ * WPB * 64 * NV);
        gload(min(qb + LC_GQ, nq - 1));                          // next group (past the end: a harmless re-read)
        v2f ya, yia, yb_, yib;
        float wa0, wa1, wb0, wb1;
        lread(buf, 0, ya, yia, wa0, wa1);
        // two quads per trip, LDS operands one quad ahead.  NOT unrolled further: with the whole group in one basic
        // block the register allocation explodes (512 VGPRs + spills)
#pragma unroll 1
        for (int g = 0; g < LC_GQ; g += 2) {
            lread(buf, g + 1, yb_, yib, wb0, wb1);
            quad(g, redb, ya, yia, wa0, wa1);
            lread(buf, g + 2 < LC_GQ ? g + 2 : LC_GQ - 1, ya, yia, wa0, wa1);
            quad(g + 1, redb, yb_, yib, wb0, wb1);
        }
        lstore(buf ^ 1, qb + LC_GQ);
        __syncthreads();
#ifdef PROBE_BWD_NORED
        buf ^= 1;
        continue;
#endif
        // fixed-order sum over the 4 wavefronts: one partial per (workgroup row group, negative, column).  The
        // LDS buffers alternate, so the next group needs no second barrier.
#if !defined(KGE_PLAIN_INTERMEDIATES) && !defined(LC_ST8)
        constexpr bool ST16 = !CPLX;
#else
        constexpr bool ST16 = false;
#endif
        if constexpr (ST16) {
        // (real-valued models, write-through stores: 16 bytes each - an 8-byte sc1 store costs 2.7 x a 16-byte one per byte,
        //  MI355X_MICROARCH.md - so a thread sums the partials of TWO adjacent lanes = four consecutive columns; per element the
        //  same additions in the same wavefront order as the per-lane form.  Measured: TransE_l1 53.7 -> 53.0 us/step; RotatE
        //  (two halves per lane) is 0.5-0.9 us FASTER with the per-lane 8-byte stores and keeps them; profiles/r04_store_policy.txt)
        for (int e = tid; e < LC_GQ * (CPLX ? 64 : 32); e += LC_BLOCK) {
            const int g = CPLX ? e >> 6 : e >> 5, r = CPLX ? e & 63 : (e & 31) * 2;
            const int l0 = r & ~1, part = CPLX ? r & 1 : 0;
            const float *p = redb + (g * WPB * 64 + l0) * NV + 2 * part;
            const int sN = 4 * (qb + g) + (l0 >> 4);
            const int cl = slab * LC_CW + 2 * (l0 & 15);
            v2f s0 = *reinterpret_cast<const v2f *>(p), s1 = *reinterpret_cast<const v2f *>(p + NV);
#pragma unroll
            for (int w = 1; w < WPB; ++w) {
                s0 += *reinterpret_cast<const v2f *>(p + w * 64 * NV);
                s1 += *reinterpret_cast<const v2f *>(p + w * 64 * NV + NV);
            }
            if (sN < N && cl < K) {                              // K % 4 == 0: four columns or none
                Pack<4> o4; o4.v[0] = s0.x; o4.v[1] = s0.y; o4.v[2] = s1.x; o4.v[3] = s1.y;
                st_wt<4>(a.GNp + (((int64_t)rw * a.C + c) * N + sN) * D + cl + part * K, o4);
            }
        }
        } else {
        for (int e = tid; e < LC_GQ * 64; e += LC_BLOCK) {
            const int g = e >> 6, l = e & 63;
            const float *p = redb + (g * WPB * 64 + l) * NV;
            const int sN = 4 * (qb + g) + (l >> 4);
            const int cl = slab * LC_CW + 2 * (l & 15);
            float *o = a.GNp + (((int64_t)rw * a.C + c) * N + sN) * D + cl;
            if constexpr (CPLX) {
                // (the wavefronts' partials in wavefront order: ((v0 + v1) + v2) + v3 [+ v4 ... + v7])
                float4 pv[WPB];
#pragma unroll
                for (int w = 0; w < WPB; ++w) pv[w] = *reinterpret_cast<const float4 *>(p + w * 64 * NV);
                float4 acc = pv[0];
#pragma unroll
                for (int w = 1; w < WPB; ++w) { acc.x += pv[w].x; acc.y += pv[w].y; acc.z += pv[w].z; acc.w += pv[w].w; }
                if (sN < N && cl < K) {
                    st_v2(o, (v2f){acc.x, acc.y});
                    st_v2(o + K, (v2f){acc.z, acc.w});
                }
            } else {
                v2f acc = *reinterpret_cast<const v2f *>(p);
#pragma unroll
                for (int w = 1; w < WPB; ++w) acc += *reinterpret_cast<const v2f *>(p + w * 64 * NV);
                if (sN < N && cl < K) st_v2(o, acc);
            }
        }
        }
        buf ^= 1;
        if constexpr (NRED == 1) __syncthreads();                // the partials are read before the next group overwrites them
    }
#ifdef KGE_TL_MARKS
    KGE_TL_MARK_K(3, 1);           // groups done
#endif
    // GA: sum over the 4 lane rows (different negatives), then lane row 0 stores
#pragma unroll
    for (int n = 0; n < RT; ++n) {
        gr[n].x += __shfl_xor(gr[n].x, 16, 64); gr[n].y += __shfl_xor(gr[n].y, 16, 64);
        gr[n].x += __shfl_xor(gr[n].x, 32, 64); gr[n].y += __shfl_xor(gr[n].y, 32, 64);
        if constexpr (CPLX) {
            gi[n].x += __shfl_xor(gi[n].x, 16, 64); gi[n].y += __shfl_xor(gi[n].y, 16, 64);
            gi[n].x += __shfl_xor(gi[n].x, 32, 64); gi[n].y += __shfl_xor(gi[n].y, 32, 64);
        }
        if (sg == 0 && colok && r0 + n < rend) {
            float *o = a.GA + sp * a.ga_stride + ((int64_t)c * chunk + r0 + n) * D + col;
            st_v2(o, gr[n]);
            if constexpr (CPLX) st_v2(o + K, gi[n]);
            if (a.lc_P > 0 && sp == nsp - 1 && nsp < a.ga_parts) {       // (balanced split: this column has one part less than the consumer adds)
                st_v2(o + a.ga_stride, (v2f){0.f, 0.f});
                if constexpr (CPLX) st_v2(o + a.ga_stride + K, (v2f){0.f, 0.f});
            }
        }
    }
}

// GN[j, :] = sum over the workgroup row groups of the partials (fixed order) + regulariser of the negative row
// (body in kge_common.hpp: the fused step runs it as the second half of the edge_bwd launch, kge_rowwise.hip)
// + (ga_parts > 1, stand-alone reduction only) GA = sum of its parts, in place, in part order
__global__ __launch_bounds__(KGE_BLOCK) void gn_reduce_kernel(NegArgs a, int nrw) {
    const int64_t t = (int64_t)blockIdx.x * KGE_BLOCK + threadIdx.x;
    const int64_t n4 = (int64_t)a.C * a.N * a.d_e / 4;
    if (t < n4) { gn_reduce_body(a, nrw, t); return; }
    const int64_t u = t - n4;
    if (a.ga_parts > 1 && u < (int64_t)a.C * a.chunk * a.d_e / 4) {
        float4 acc = *reinterpret_cast<const float4 *>(a.GA + 4 * u);
        for (int q = 1; q < a.ga_parts; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(a.GA + q * a.ga_stride + 4 * u);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4 *>(a.GA + 4 * u) = acc;
    }
}

// balanced split (NegArgs::lc_P): the instances that hold four workgroups per CU (RotatE at <= 8 rows per wavefront, TransE_l1 at
// <= 16), every part >= 2 (RotatE) / 1 (TransE_l1) groups of quads, at most 8 / 4 parts.  Measured (tools/timeline.py --per-cu, rotate_fb15k): 896 equal workgroups = 3.5 per CU -
// the CUs holding four end 3.5 us after those holding three (wavefront life p50 17.6 vs 15.4 us); cfg-R: 832 = 3.25 per CU.
#ifndef LC_NO_BALANCE
#define LC_BALANCE 1
#endif
static bool lc_balance(int model, int C, int chunk, int N, int d_e, int &P, int &nB) {
    P = 0; nB = 0;
#ifdef LC_BALANCE
#ifdef LC_BALANCE_REAL       // TransE_l1: measured 52.95 -> 52.3 us/step (975 -> 1024 workgroups) and NOT validated - the oracle comparison at the
    const bool real_ok = model == KGE_TRANSE_L1;     // recipe's full shape fails with it (tests/test_gpu_parity.py SHAPES); off
#else
    const bool real_ok = false;
#endif
    if ((model != KGE_ROTATE && !real_ok) || !neg_bwd_lc_supported(model, d_e) || N % 4 || d_e % 4) return false;
    int nslab, nrw, rpw;
    lc_shape(model, C, chunk, d_e, nslab, nrw, rpw);
    if (rpw > (model == KGE_ROTATE ? 8 : 16)) return false;
    const int gq = model == KGE_ROTATE ? LC_GQ_CPLX : LC_GQ_REAL;
    const int ncol = C * nslab * nrw, ngr = ((N + 3) / 4 + gq - 1) / gq;
    const int total = 4096 / lc_wpb(model, d_e);                 // workgroups of the balanced launch: 16 wavefronts per CU
    if (ncol < 1 || ncol > total) return false;
    const int p = total / ncol, nA = total - ncol * p;
    const int minpart = model == KGE_ROTATE ? 2 : 1;
    if (nA == 0 || p + 1 > (model == KGE_ROTATE ? 8 : 4) || ngr / (p + 1) < minpart) return false;      // (nA == 0: the uniform split already fills the chip evenly)
    P = p; nB = ncol - nA;
    return true;
#else
    (void)model; (void)C; (void)chunk; (void)N; (void)d_e;
    return false;
#endif
}

// split the negatives while the launch stays within ~4 workgroups per CU and a workgroup keeps >= 2 groups of quads
int neg_bwd_lc_splits(int model, int C, int chunk, int N, int d_e) {
    { int P, nB; if (lc_balance(model, C, chunk, N, d_e, P, nB)) return P + 1; }
#ifndef LC_SPLIT_REAL
    if (model != KGE_ROTATE) return 1;
#endif
    if (!neg_bwd_lc_supported(model, d_e) || N % 4 || d_e % 4) return 1;
    int nslab, nrw, rpw;
    lc_shape(model, C, chunk, d_e, nslab, nrw, rpw);
    const int gq = model == KGE_ROTATE ? LC_GQ_CPLX : LC_GQ_REAL;
    const int ngr = ((N + 3) / 4 + gq - 1) / gq;
    int nsp = 1;
    while (nsp < 8 && (int64_t)C * nslab * nrw * (nsp + 1) <= LC_WG_CAP * 4 / lc_wpb(model, d_e) && ngr / (nsp + 1) >= 2) ++nsp;
    return nsp;
}
int neg_bwd_lc_nrw(int model, int C, int chunk, int d_e) {
    int nslab, nrw, rpw;
    lc_shape(model, C, chunk, d_e, nslab, nrw, rpw);
    return nrw;
}

template <int MODEL> static int lc_launch(const NegArgs &a, hipStream_t s) {
    int nslab, nrw, rpw;
    lc_shape(MODEL, a.C, a.chunk, a.d_e, nslab, nrw, rpw);
    if (a.ga_parts > 1 && a.d_e % 4) return KGE_ERR_ARG;
    int64_t nb = (int64_t)a.C * nslab * nrw * max(a.ga_parts, 1);
    if (nb == 0) return KGE_OK;
    NegArgs ab = a;
    ab.lc_P = 0; ab.lc_nB = 0;
    {   // the balanced split when the caller's part count is the one it asks for (neg_bwd_lc_splits)
        int P, nB;
        if (lc_balance(MODEL, a.C, a.chunk, a.N, a.d_e, P, nB) && a.ga_parts == P + 1) { ab.lc_P = P; ab.lc_nB = nB; nb = 4096 / lc_wpb(MODEL, a.d_e); }
    }
    const NegArgs &a_ = ab;
    const dim3 g((unsigned)nb), b(KGE_BLOCK);
    // rows per wavefront rounded up to the next instantiation (the padding rows carry W = 0)
    if constexpr (MODEL == KGE_ROTATE) {
        if (lc_wpb(MODEL, a.d_e) == 8) {          // wide rows: 8 wavefronts per workgroup (lc_wpb)
            const dim3 b8(512);
            if (rpw <= 8) hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, 8, 8>), g, b8, 0, s, a_, nslab, nrw, rpw);
            else if (rpw <= 12) hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, 12, 8>), g, b8, 0, s, a_, nslab, nrw, rpw);
            else hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, 16, 8>), g, b8, 0, s, a_, nslab, nrw, rpw);
            if (int rc = check_launch_b()) return rc;
            if (a.defer_reduce) return KGE_OK;
            const int64_t n4w = (int64_t)a.C * a.N * a.d_e / 4 + (a.ga_parts > 1 ? (int64_t)a.C * a.chunk * a.d_e / 4 : 0);
            hipLaunchKernelGGL(gn_reduce_kernel, dim3((unsigned)((n4w + KGE_BLOCK - 1) / KGE_BLOCK)), dim3(KGE_BLOCK), 0, s, a, nrw);
            return check_launch_b();
        }
    }
    if (rpw <= 8) hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, 8>), g, b, 0, s, a_, nslab, nrw, rpw);
    else if (rpw <= 12) hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, 12>), g, b, 0, s, a_, nslab, nrw, rpw);
    else if (rpw <= 16) hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, 16>), g, b, 0, s, a_, nslab, nrw, rpw);
    else if constexpr (MODEL != KGE_ROTATE) hipLaunchKernelGGL((neg_bwd_lc_kernel<MODEL, LC_RTMAX>), g, b, 0, s, a_, nslab, nrw, rpw);
    if (int rc = check_launch_b()) return rc;
    if (a.defer_reduce) return KGE_OK;           // the caller sums the partials in its next launch (launch_edge_bwd_with_gn_reduce)
    const int64_t n4 = (int64_t)a.C * a.N * a.d_e / 4 + (a.ga_parts > 1 ? (int64_t)a.C * a.chunk * a.d_e / 4 : 0);
    hipLaunchKernelGGL(gn_reduce_kernel, dim3((unsigned)((n4 + KGE_BLOCK - 1) / KGE_BLOCK)), dim3(KGE_BLOCK), 0, s, a, nrw);
    return check_launch_b();
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
    if (a.GNp && !a.nidx && a.N % 4 == 0 && neg_bwd_lc_supported(a.model, a.d_e))   // shared-pair kernel when the caller gave it room
        return a.model == KGE_ROTATE ? lc_launch<KGE_ROTATE>(a, s) : lc_launch<KGE_TRANSE_L1>(a, s);
    switch (a.model) {
        case KGE_TRANSE_L1: return bwd_launch<KGE_TRANSE_L1>(a, s);
        case KGE_TRANSE_L2: return bwd_launch<KGE_TRANSE_L2>(a, s);
        case KGE_DISTMULT: case KGE_COMPLEX: case KGE_SIMPLE: case KGE_RESCAL: return bwd_launch<KGE_DISTMULT>(a, s);
        case KGE_ROTATE: return bwd_launch<KGE_ROTATE>(a, s);
    }
    return KGE_ERR_ARG;
}
