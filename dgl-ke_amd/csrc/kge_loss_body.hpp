// kge_loss_body.hpp - LossGenerator.get_total_loss (negative half) + its gradient on ONE register-resident score row,
// shared by the stand-alone loss kernel (kge_rowwise.hip, loss_kernel_reg) and by the forward tiles' last arriver of the
// strict step's first launch (kge_neg_gemm.hip, round 4: no loss launch).  Same instructions in both places, so the two
// launch sequences give bit-identical gradients for the same scores.
// Reference: models/pytorch/loss.py:69-98 (get_total_loss), :10-38 (criteria), :76-80 (pairwise), :87-88 (-adv softmax).
#pragma once
#include "kge_common.hpp"
#include "kge_update_body.hpp"      // LANE(), acc_add

// LEAN: the common configuration (Logsigmoid, point-wise, positive part done by edge_fwd, no score clamp, no per-step
// outputs) - everything else is compiled out of the LEAN instances
__device__ __forceinline__ void loss_args_lean(LossArgs &a) {
    a.genre = KGE_LOSS_LOGSIGMOID; a.pairwise = 0; a.skip_pos = 1; a.clampv = 0.f; a.neg_copy = nullptr;
    a.row_pos = nullptr; a.row_neg = nullptr; a.diag_chunk = 0;
}

// nv[u] = score of column lane + 64 u of row i (0 beyond N and in the masked diagonal column); w = edge weight, p = positive
// score (read only by the pairwise / !skip_pos variants).  Writes dL/dn over the row (a.dneg, TransE_l2: pre-divided by the
// distance), the optional score copy, the row's loss terms and the running sums.  slot2: where this row's share of the total
// goes in the running sums (the stand-alone kernel: the row's own slot; the in-launch variant: a slot the edge half of the
// same launch does not touch)
template <int NPER>
__device__ __forceinline__ void loss_row_regs(const LossArgs &a, int64_t i, float (&nv)[NPER], float w, float p, int lane,
                                              int slot2) {
    using namespace kge;
    const int N = a.N;
    float *dn = a.dneg + i * (int64_t)N;
    float *cp = a.neg_copy ? a.neg_copy + i * (int64_t)N : nullptr;
    const int jd = a.diag_chunk > 0 ? (int)(i % a.diag_chunk) : -1;
    const float invB = 1.f / (float)a.B;
    const int slot = (int)(i & (KGE_ACC_SLOTS - 1));
    if (a.pairwise) {   // loss.py:76-80
        const float sc = w / ((float)a.B * (float)N);
        float lsum = 0.f, dsum = 0.f;
#pragma unroll
        for (int u = 0; u < NPER; ++u) {
            const int j = lane + 64 * u;
            if (j < N) {
                float val, dv;
                criterion_fast(a.genre, p - nv[u], 1.f, a.margin, val, dv);
                lsum += val * sc;
                const float dd = dv * sc;
                dsum += dd;
                if (cp) cp[j] = nv[u];
                float g = -dd;
                if (a.l2_scale) { const float d = a.gamma - nv[u]; g = d > 1e-15f ? g / d : 0.f; }
                if (a.clampv > 0.f && fabsf(nv[u]) >= a.clampv) g = 0.f;
                dn[j] = j == jd ? 0.f : g;
            }
        }
        lsum = wave_sum(lsum);
        dsum = wave_sum(dsum);
        if (lane == 0) {
            a.dpos[i] = (a.clampv > 0.f && fabsf(p) >= a.clampv) ? 0.f : dsum;
            if (a.row_pos) { a.row_pos[i] = 0.f; a.row_neg[i] = lsum; }
            if (a.acc) acc_add(&a.acc[2 * KGE_ACC_SLOTS + slot2], lsum, a.B <= KGE_ACC_SLOTS);
        }
        return;
    }
    float plw = 0.f;
    if (!a.skip_pos) {
        const float wm = mean_edge_weight(a.w, a.B, lane, a.w_mean);     // positive part: the batch's MEAN importance (kge_common.hpp)
        if (lane == 0) {
            float pl, dpl;
            criterion(a.genre, p, 1.f, a.margin, pl, dpl);
            a.dpos[i] = (a.clampv > 0.f && fabsf(p) >= a.clampv) ? 0.f : dpl * wm * 0.5f * invB;
            plw = pl * wm * invB;
            if (a.row_pos) a.row_pos[i] = plw;
        }
    }
    const float neg_label = a.genre == KGE_LOSS_BCE ? 0.f : -1.f;
    float mx = -INFINITY, Z = 1.f;
    float ex[NPER];
    if (a.adv) {   // softmax(neg * T) over the row, detached (loss.py:87-88)
#pragma unroll
        for (int u = 0; u < NPER; ++u) if (lane + 64 * u < N) mx = fmaxf(mx, nv[u] * a.adv_temp);
        mx = wave_max(mx);
        float z = 0.f;
#pragma unroll
        for (int u = 0; u < NPER; ++u) {
            ex[u] = (lane + 64 * u < N) ? __expf(nv[u] * a.adv_temp - mx) : 0.f;
            z += ex[u];
        }
        Z = wave_sum(z);
    }
    const float invZ = 1.f / Z, invN = 1.f / (float)N;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < NPER; ++u) {
        const int j = lane + 64 * u;
        if (j < N) {
            float nl, dnl;
            criterion_fast(a.genre, nv[u], neg_label, a.margin, nl, dnl);
            const float A = a.adv ? ex[u] * invZ : invN;
            acc += A * nl * w;
            float g = dnl * w * A * 0.5f * invB;
            if (cp) cp[j] = nv[u];
            if (a.l2_scale) { const float d = a.gamma - nv[u]; g = d > 1e-15f ? g / d : 0.f; }
            if (a.clampv > 0.f && fabsf(nv[u]) >= a.clampv) g = 0.f;
            dn[j] = j == jd ? 0.f : g;
        }
    }
    acc = wave_sum(acc) * invB;
    if (lane == 0) {
        if (a.row_neg) a.row_neg[i] = acc;
        if (a.acc) {
            const bool uq = a.B <= KGE_ACC_SLOTS;
            if (!a.skip_pos) acc_add(&a.acc[0 * KGE_ACC_SLOTS + slot], plw, uq);
            acc_add(&a.acc[1 * KGE_ACC_SLOTS + slot], acc, uq);
            acc_add(&a.acc[2 * KGE_ACC_SLOTS + slot2], 0.5f * (plw + acc), uq);
        }
    }
}
