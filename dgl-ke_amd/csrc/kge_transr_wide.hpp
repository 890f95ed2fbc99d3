// kge_transr_wide.hpp - TransR's three batched products on 128 x 208 workgroup tiles (included by kge_transr.hip).
//
// The 64 x 64 tiles of kge_transr.hip re-read their operands once per tile (the forward moved 1.7 GB through L2 per launch at the
// FB15k recipe and was bound by it), spend a barrier on 16 MFMAs per wavefront, and pad a 200-wide operand to 256.  Here a
// workgroup owns 128 rows x up to 208 columns (13 blocks of 16: ALL columns of a <= 208-wide operand), a wavefront 32 rows x 208
// columns = 2 x 13 accumulator blocks: up to 104 MFMAs per wavefront and barrier, each A value read from LDS feeds 13 MFMAs, each
// B value 2, and an operand row is fetched once per 128 x 208 outputs.  Used when D_e, D_r <= 208 and both % 4 == 0 (16-byte
// operand loads); everything else keeps the 64 x 64 kernels.  LDS row strides are 16 (mod 32) banks, so the four 16-lane groups
// of a wavefront's ds_read_b32 (rows q, q + 1, ..) alternate bank halves.
#pragma once

#ifndef TW_R
#define TW_R 64                     // rows of a workgroup tile: one 16-row block per wavefront (128: two - measured slower, 2 workgroups per CU)
#endif
#define TW_WR (TW_R / 4)             // rows of a wavefront
#define TW_NRB (TW_WR / 16)
#define TW_AUNITS (TW_R * TR_K / 4 / KGE_BLOCK)   // float4 units of one A slab per thread
#define TW_C 208
#define TW_NCB 13
#define TW_LDA (TW_R + 16)
#define TW_LDB TW_C
#define TW_BUNITS (TR_K * TW_C / 4)      // float4 units of one B slab (832: up to four per thread)

// NRB: 16-row blocks of this wavefront that hold real rows (0..2), NCB: 16-column blocks multiplied (the real ones, rounded up to
// an instantiated count).  AROW / BROW as in tile_sweep4: the four elements of a unit are consecutive rows (columns) at one k, or
// consecutive k of one row (column).
// BSUM (with BROW): every thread also sums the B units it moves (its column group, k-rows t / 52 + 4 e of every slab): the column sums of the B
// operand over K come out of the sweep for one LDS reduction (transr_gp_wide_kernel: dq_i = -sum_j dY_ij without a pass of its own).
template <int NRB, int NCB, bool AROW, bool BROW, bool BSUM, class LA, class LB>
__device__ __forceinline__ void wide_sweep_n(f32x4 (&acc)[TW_NRB][TW_NCB], int Ktot, LA loadA4, LB loadB4, float (*As)[TR_K][TW_LDA],
                                             float (*Bs)[TR_K][TW_LDB], float4 &bsum) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    int ai[TW_AUNITS], ak[TW_AUNITS], bi[4], bk[4];
    bool bon[4];
#pragma unroll
    for (int e = 0; e < TW_AUNITS; ++e) {
        const int u = t + KGE_BLOCK * e;
        ai[e] = AROW ? (u & (TW_R / 4 - 1)) * 4 : u >> 2;
        ak[e] = AROW ? u / (TW_R / 4) : (u & 3) * 4;
    }
    // B units: BROW - thread t < 208 owns column group t % 52 at the k-rows t / 52 + 4 e (ONE column group per thread: its column
    // sums need one float4); otherwise unit u = t + 256 e: column u / 4, k = 4 (u % 4) .. + 3
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int u = t + KGE_BLOCK * e;
        bon[e] = BROW ? t < TW_C : u < TW_BUNITS;
        bi[e] = BROW ? (t % (TW_C / 4)) * 4 : u >> 2;
        bk[e] = BROW ? t / (TW_C / 4) + 4 * e : (u & 3) * 4;
    }
    // LDS columns are rotated by one 16-column block on the k-rows with (k / 4) odd.  Row strides of 16 (mod 32) banks keep the MFMA
    // operand reads conflict-free, but a thread that scatters four consecutive k of one column (!AROW / !BROW) shares its column with
    // the three lanes holding the other k-groups: 4 lanes on one bank, 20 such stores per thread and slab in the negative-row
    // product (~1 us per slab at 4 workgroups per CU).  Rotated, the k-groups alternate bank halves: 2 lanes per bank, the floor.
    int aic[TW_AUNITS], bic[4];
#pragma unroll
    for (int e = 0; e < TW_AUNITS; ++e) aic[e] = (ai[e] + 16 * ((ak[e] >> 2) & 1)) & (TW_R - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int cc = bi[e] + 16 * ((bk[e] >> 2) & 1); bic[e] = cc >= TW_C ? cc - TW_C : cc; }
    decltype(loadA4(0, 0, 0)) ra[TW_AUNITS]{};
    decltype(loadB4(0, 0, 0)) rb[4]{};
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < TW_AUNITS; ++e) ra[e] = loadA4(ai[e], k0, ak[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (bon[e]) rb[e] = loadB4(k0, bk[e], bi[e]);
    };
    if (Ktot > 0) gload(0);
    int buf = 0;
    __syncthreads();
    for (int k0 = 0; k0 < Ktot; k0 += TR_K, buf ^= 1) {
#pragma unroll
        for (int e = 0; e < TW_AUNITS; ++e) {
            const float4 v = cvt_op(ra[e]);
            if constexpr (AROW) *reinterpret_cast<float4 *>(&As[buf][ak[e]][aic[e]]) = v;
            else { As[buf][ak[e]][aic[e]] = v.x; As[buf][ak[e] + 1][aic[e]] = v.y; As[buf][ak[e] + 2][aic[e]] = v.z; As[buf][ak[e] + 3][aic[e]] = v.w; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!bon[e]) continue;
            const float4 v = cvt_op(rb[e]);
            if constexpr (BSUM) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
            if constexpr (BROW) *reinterpret_cast<float4 *>(&Bs[buf][bk[e]][bic[e]]) = v;
            else { Bs[buf][bk[e]][bic[e]] = v.x; Bs[buf][bk[e] + 1][bic[e]] = v.y; Bs[buf][bk[e] + 2][bic[e]] = v.z; Bs[buf][bk[e] + 3][bic[e]] = v.w; }
        }
        __syncthreads();
        if (k0 + TR_K < Ktot) gload(k0 + TR_K);
        if constexpr (NRB > 0) {
            // operands of k-step s4 + 1 are requested from LDS before the MFMAs of k-step s4 are issued (two waves per SIMD at this
            // register count: an LDS round trip in front of every pair of MFMAs would not be hidden by the other wave)
            float av[2][NRB], bq[2][NCB];
            auto lds_fetch = [&](int s4, int w) {
#pragma unroll
                for (int rbk = 0; rbk < NRB; ++rbk)
                    av[w][rbk] = As[buf][4 * s4 + q][((wave * TW_NRB + rbk + (s4 & 1)) & (TW_R / 16 - 1)) * 16 + m];
#pragma unroll
                for (int ct = 0; ct < NCB; ++ct) bq[w][ct] = Bs[buf][4 * s4 + q][((ct + (s4 & 1)) % TW_NCB) * 16 + m];
            };
            lds_fetch(0, 0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (s4 < 3) lds_fetch(s4 + 1, (s4 + 1) & 1);
#pragma unroll
                for (int ct = 0; ct < NCB; ++ct)
#pragma unroll
                    for (int rbk = 0; rbk < NRB; ++rbk) acc[rbk][ct] = MFMA16(av[s4 & 1][rbk], bq[s4 & 1][ct], acc[rbk][ct]);
            }
        }
    }
}

// nrb / ncb are wavefront-uniform; every instance runs the same loads and barriers.  Column-block counts are rounded up to the
// instantiated ones (4, 7, 8, 13: operands of <= 64, 100 / 112, 128 and 200 / 208 columns).
template <bool AROW, bool BROW, bool BSUM, class LA, class LB>
__device__ __forceinline__ void wide_sweep(f32x4 (&acc)[TW_NRB][TW_NCB], int Ktot, LA la, LB lb, float (*As)[TR_K][TW_LDA],
                                           float (*Bs)[TR_K][TW_LDB], int nrb, int ncb, float4 &bsum) {
#define TW_CASE(R, Cc) wide_sweep_n<R, Cc, AROW, BROW, BSUM>(acc, Ktot, la, lb, As, Bs, bsum)
#define TW_ROWS(Cc) do { if (nrb <= 0) TW_CASE(0, Cc); else if (nrb == 1 || TW_NRB == 1) TW_CASE(1, Cc); else TW_CASE(TW_NRB, Cc); } while (0)
    if (ncb <= 4) TW_ROWS(4);
    else if (ncb <= 7) TW_ROWS(7);
    else if (ncb <= 8) TW_ROWS(8);
    else TW_ROWS(13);
#undef TW_ROWS
#undef TW_CASE
}

static inline bool transr_wide_supported(int De, int Dr) { return De % 4 == 0 && Dr % 4 == 0 && De <= TW_C && Dr <= TW_C; }

// ---------------------------------------------------------------------------------------------
// forward: workgroup = (positive i, 128 negatives): Y_i = Neg_c P_i over ALL D_r columns -> the L1 epilogue is complete per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KGE_BLOCK) void transr_fwd_wide_kernel(TransRArgs a, int nJB) {
    __shared__ float As[2][TR_K][TW_LDA], Bs[2][TR_K][TW_LDB];
    __shared__ int64_t rowoff[TW_R];
    const int i = blockIdx.x / nJB, j0 = (blockIdx.x % nJB) * TW_R;
    const int c = i / a.chunk, De = a.De, Dr = a.Dr, N = a.N;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    if (t < TW_R) rowoff[t] = (j0 + t < N) ? a.neg_ids[(int64_t)c * N + j0 + t] * (int64_t)De : -1;
    __syncthreads();
    const float *Pi = a.proj + a.rel_ids[i] * (int64_t)De * Dr;
    const float *Qi = a.Q + (int64_t)i * Dr;
    f32x4 acc[TW_NRB][TW_NCB];
#pragma unroll
    for (int rbk = 0; rbk < TW_NRB; ++rbk)
#pragma unroll
        for (int ct = 0; ct < TW_NCB; ++ct) acc[rbk][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nrb = max(0, min(TW_NRB, (N - (j0 + wave * TW_WR) + 15) / 16));
    float4 nosum;
    wide_sweep<false, true, false>(acc, De,
        [&](int row, int k0_, int kk_) { const int k = k0_ + kk_; const int64_t o = rowoff[row];
                                         return (o >= 0 && k < De) ? *reinterpret_cast<const float4 *>(a.ent + o + k) : f4zero(); },
        [&](int k0_, int kk_, int col) { const int k = k0_ + kk_;
                                         return (k < De && col < Dr) ? *reinterpret_cast<const float4 *>(Pi + (int64_t)k * Dr + col) : f4zero(); },
        As, Bs, nrb, (Dr + 15) / 16, nosum);
    // sign bytes: a workgroup's rows (i, j0 .. j0 + 127) are ONE contiguous block of Z - staged in LDS (the tile buffers are free
    // after the sweep) and written as 16-byte chunks instead of 104 one-byte stores per lane
    signed char *zs = reinterpret_cast<signed char *>(&Bs[0][0][0]);
    const int nrows = min(TW_R, N - j0);
    __syncthreads();                               // every wavefront has left the sweep's last slab
#pragma unroll
    for (int rbk = 0; rbk < TW_NRB; ++rbk) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < TW_NCB; ++ct) {
            const int col = ct * 16 + m;
            const float qv = col < Dr ? Qi[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = wave * TW_WR + rbk * 16 + 4 * q + r;
                const float v = acc[rbk][ct][r] - qv;
                if (col < Dr && jl < nrows) {
                    d[r] += fabsf(v);
                    if (a.Z) zs[jl * Dr + col] = (signed char)((v > 0.f) - (v < 0.f));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) d[r] += __shfl_xor(d[r], o, 64);
            const int j = j0 + wave * TW_WR + rbk * 16 + 4 * q + r;
            if (m == 0 && j < N) a.S[(int64_t)i * N + j] = a.gamma - d[r];
        }
    }
    if (a.Z) {
        __syncthreads();
        signed char *zg = a.Z + ((int64_t)i * N + j0) * Dr;          // (Dr % 4 == 0 and Z is 16-byte aligned: the block starts on a
        const int nbytes = nrows * Dr;                                //  4-byte boundary; chunks of 16 bytes from the first aligned one)
        const int head = (int)((16 - (reinterpret_cast<uintptr_t>(zg) & 15)) & 15);
        for (int k = t; k < min(head, nbytes); k += KGE_BLOCK) zg[k] = zs[k];
        const int nchunks = nbytes > head ? (nbytes - head) / 16 : 0;
        for (int k = t; k < nchunks; k += KGE_BLOCK) {
            const signed char *src = zs + head + 16 * k;              // (LDS side may be misaligned by `head`: four 4-byte reads)
            int4 v;
            v.x = *reinterpret_cast<const int *>(src); v.y = *reinterpret_cast<const int *>(src + 4);
            v.z = *reinterpret_cast<const int *>(src + 8); v.w = *reinterpret_cast<const int *>(src + 12);
            *reinterpret_cast<int4 *>(zg + head + 16 * k) = v;
        }
        for (int k = head + 16 * nchunks + t; k < nbytes; k += KGE_BLOCK) zg[k] = zs[k];
    }
}

// ---------------------------------------------------------------------------------------------
// GN_c[j][de] = sum_{i in group} sum_dr dY_i[j][dr] P_i[de][dr]: workgroup = (chunk, 128 negatives, group of positives), all D_e columns
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KGE_BLOCK) void transr_gn_wide_kernel(TransRArgs a, int nJB) {
    __shared__ float As[2][TR_K][TW_LDA], Bs[2][TR_K][TW_LDB];
    extern __shared__ char gnw_dyn[];            // per positive of the group: its projection matrix's offset, -W_ij of the 128 negatives
    const int g = blockIdx.x % a.nG, blk = blockIdx.x / a.nG;
    const int jb = blk % nJB, c = blk / nJB;
    const int j0 = jb * TW_R, De = a.De, Dr = a.Dr, N = a.N, chunk = a.chunk;
    const int DrP = (Dr + TR_K - 1) / TR_K * TR_K;
    const int ipg = (chunk + a.nG - 1) / a.nG, i0 = g * ipg, i1 = min(chunk, i0 + ipg);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    int64_t *s_pb = reinterpret_cast<int64_t *>(gnw_dyn);
    float *s_w = reinterpret_cast<float *>(s_pb + ipg);
    for (int k = t; k < i1 - i0; k += KGE_BLOCK) s_pb[k] = a.rel_ids[(int64_t)c * chunk + i0 + k] * (int64_t)De * Dr;
    for (int k = t; k < (i1 - i0) * TW_R; k += KGE_BLOCK) {
        const int il = k / TW_R, j = j0 + (k % TW_R);
        s_w[k] = j < N ? -a.S[((int64_t)c * chunk + i0 + il) * N + j] : 0.f;
    }
    __syncthreads();
    f32x4 acc[TW_NRB][TW_NCB];
#pragma unroll
    for (int rbk = 0; rbk < TW_NRB; ++rbk)
#pragma unroll
        for (int ct = 0; ct < TW_NCB; ++ct) acc[rbk][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nrb = max(0, min(TW_NRB, (N - (j0 + wave * TW_WR) + 15) / 16));
    float4 nosum;
    wide_sweep<false, false, false>(acc, max(0, i1 - i0) * DrP,
        [&](int row, int k0_, int kk_) {              // 4 consecutive d_r of one (positive, negative): one word of sign bytes
            const int il = k0_ / DrP, dr = k0_ % DrP + kk_, j = j0 + row;
            if (j >= N || dr >= Dr) return Sgn4{0, 0.f};
            const int64_t ij = ((int64_t)c * chunk + i0 + il) * N + j;
            return Sgn4{*reinterpret_cast<const int *>(a.Z + ij * Dr + dr), s_w[il * TW_R + row]};
        },
        [&](int k0_, int kk_, int col) {              // P_i[de][dr .. dr + 3]
            const int il = k0_ / DrP, dr = k0_ % DrP + kk_;
            if (col >= De || dr >= Dr) return f4zero();
            return *reinterpret_cast<const float4 *>(a.proj + s_pb[il] + (int64_t)col * Dr + dr);
        }, As, Bs, nrb, (De + 15) / 16, nosum);
    float *out = a.GNp + (int64_t)g * a.C * N * De;
#pragma unroll
    for (int rbk = 0; rbk < TW_NRB; ++rbk)
#pragma unroll
        for (int ct = 0; ct < TW_NCB; ++ct) {
            const int de = ct * 16 + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + wave * TW_WR + rbk * 16 + 4 * q + r;
                if (de < De && j < N) out[((int64_t)c * N + j) * De + de] = acc[rbk][ct][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// GP_i[de][dr] = sum_j Neg_j[de] dY_ij[dr] + x_i[de] dq_i[dr]: workgroup = (positive, 128 rows of D_e), all D_r columns
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KGE_BLOCK) void transr_gp_wide_kernel(TransRArgs a, int nEB) {
    __shared__ float As[2][TR_K][TW_LDA], Bs[2][TR_K][TW_LDB];
    __shared__ float red[KGE_WAVES_PER_BLOCK];
    extern __shared__ char gpw_dyn[];            // per negative of the chunk: its row's offset in the entity table, -W_ij; then dq_i [208]
#ifndef TW_GP_PER_EDGE
    const int eb_first = blockIdx.x % nEB, eb_last = eb_first + 1, i = blockIdx.x / nEB;
#else
    // (measured: one workgroup per positive walking its row tiles - ids / weights staged once, dq from the first sweep only - needs
    //  180 VGPRs, 2 wavefronts per SIMD: 334 us against 303 with a workgroup per tile)
    const int eb_first = 0, eb_last = nEB, i = blockIdx.x;
#endif
    const int De = a.De, Dr = a.Dr, N = a.N;
    const int c = i / a.chunk;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = lane & 15, q = lane >> 4;
    int64_t *s_off = reinterpret_cast<int64_t *>(gpw_dyn);
    float *s_w = reinterpret_cast<float *>(s_off + N), *s_dq = s_w + N;
    for (int k = t; k < N; k += KGE_BLOCK) {
        s_off[k] = a.neg_ids[(int64_t)c * N + k] * (int64_t)De;
        s_w[k] = -a.S[(int64_t)i * N + k];
    }
    __syncthreads();
    const float *x = a.ent + (a.neg_head ? a.t_gid[i] : a.h_gid[i]) * (int64_t)De;
    float *G = a.GP + (int64_t)i * De * Dr;
    auto loadA = [&](int de0) {
        return [&, de0](int row, int k0_, int kk_) {  // Neg_k[de0 + row .. + 3]
            const int k = k0_ + kk_;
            return (k < N && de0 + row < De) ? *reinterpret_cast<const float4 *>(a.ent + s_off[k] + de0 + row) : f4zero();
        };
    };
    auto loadB = [&](int k0_, int kk_, int col) {     // dY_ik[col .. + 3]: one word of sign bytes
        const int k = k0_ + kk_;
        if (k >= N || col >= Dr) return Sgn4{0, 0.f};
        return Sgn4{*reinterpret_cast<const int *>(a.Z + ((int64_t)i * N + k) * Dr + col), s_w[k]};
    };
    for (int eb = eb_first; eb < eb_last; ++eb) {
        const int de0 = eb * TW_R;
        f32x4 acc[TW_NRB][TW_NCB];
#pragma unroll
        for (int rbk = 0; rbk < TW_NRB; ++rbk)
#pragma unroll
            for (int ct = 0; ct < TW_NCB; ++ct) acc[rbk][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nrb = max(0, min(TW_NRB, (De - (de0 + wave * TW_WR) + 15) / 16));
        float4 bsum = f4zero();
        if (eb == eb_first) {
            wide_sweep<true, true, true>(acc, N, loadA(de0), loadB, As, Bs, nrb, (Dr + 15) / 16, bsum);
            // dq_i = -sum_j dY_ij: thread t < 208 summed the k-rows t / 52 + 4 e of every slab for its column group; the four
            // k-classes are added in a fixed order.  The tile buffers are free after the sweep.
            float *part = &Bs[0][0][0];
            __syncthreads();
            if (t < TW_C) *reinterpret_cast<float4 *>(part + (t / (TW_C / 4)) * TW_C + (t % (TW_C / 4)) * 4) = bsum;
            __syncthreads();
            if (t < TW_C) {
                float sdq = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) sdq += part[k * TW_C + t];
                s_dq[t] = -sdq;
                if (eb == 0 && t < Dr) {              // the edge's first tile also writes dq and the relation-vector gradient
                    a.DQ[(int64_t)i * Dr + t] = -sdq; // GR_i = -dp_i s_i - dq_i (+ regulariser of the traced copy)
                    float gr = -a.dpos[i] * a.SG[(int64_t)i * Dr + t] + sdq;
                    if (a.reg_coef > 0.f && a.reg_norm > 0) gr += reg_grad(a.rel[a.rel_ids[i] * (int64_t)Dr + t], a.reg_coef, a.reg_norm);
                    a.GR[(int64_t)i * Dr + t] = gr;
                }
            }
            __syncthreads();
        } else {
            wide_sweep<true, true, false>(acc, N, loadA(de0), loadB, As, Bs, nrb, (Dr + 15) / 16, bsum);
        }
        float ss = 0.f;
#pragma unroll
        for (int rbk = 0; rbk < TW_NRB; ++rbk)
#pragma unroll
            for (int ct = 0; ct < TW_NCB; ++ct) {
                const int dr = ct * 16 + m;
                const float dqv = s_dq[dr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int de = de0 + wave * TW_WR + rbk * 16 + 4 * q + r;
                    if (de < De && dr < Dr) {
                        const float v = acc[rbk][ct][r] + x[de] * dqv;
                        G[(int64_t)de * Dr + dr] = v;
                        ss = fmaf(v, v, ss);
                    }
                }
            }
        ss = block_sum_t(ss, red);
        if (t == 0) a.gs1p[(int64_t)i * nEB + eb] = ss;
    }
}
