// kge_sampler_tail.hpp - the on-device sampler + plan builder as TAIL WORKGROUPS of the training step's own launches (round 5).
//
// kge_sampler.hip builds a batch with one 1024-thread workgroup per plan in a launch of its own: 33 us per group of 120 batches,
// 25 us per group of 20 - serial time between two groups of steps (a second queue next to the step's graph makes every step
// 3.5 us slower, profiles/r03_merged_fwd.txt, r05_sampler_tail.txt).  Here the SAME batch (same ids from the same counter RNG and
// epoch permutation, same plan bit for bit - tests/test_gpu_sampler.py) is built by a handful of 256-thread workgroups appended to
// the grids of three launches of ONE training step, a phase per launch, the launch boundary being the only synchronisation:
//
//   phase 1 (first launch: forward tiles || edge rows || 4 tail workgroups)
//       wg 0, 1  the two halves of the batch's B edges: epoch order -> (h, t) ids, keys (id << 12 | code), dealt STABLY (block scan of
//             per-bucket counts) into ST_NBK id-range buckets;   wg 2  the C*N negatives likewise;   wg 3  relation ids + keys
//   phase 2 (backward GEMM launch: 5 tail workgroups IN FRONT of the tiles)
//       wg b < 4  bucket b: stable radix sort over the id bits (the bucket holds its keys in code order) -> sorted keys in scratch,
//             the bucket's totals (elements, unique entities, positive elements) in the header
//       wg 4  relations: sort, unique runs -> ur_id / ur_ptr / ur_edge and the counts, final (one workgroup holds them all); caches
//             the epoch constants of the next job
//   phase 3 (the NEXT step's first launch, next to that step's own phase 1; the group's last job: its own update launch)
//       wg b < 4  unique / positive flags, bucket scan, bases = totals of the buckets in front -> ue_id, CSR pointers and lists,
//             32-byte records at their final positions;   wg 4  relation records, the longest relation list, and - last batch of
//             a group - the state advance
//   (the update launch is the step's shortest, 6.9 us at cfg-T, and phase 3 under it made it 0.2 us longer; the first launch has room)
//
// Step k of group g builds batch k of group g + 1 into the other half of the sampler's slots; the batch is complete one launch
// into step k + 1 (the last one: when its own step ends).  Every phase fits under its host launch (9.3 / 9.0 / 7.8 us at cfg-T), so the group's sampling costs no time
// of its own.  Buckets hold ~ NE / 4 keys for spread ids (4 keys per thread); a bucket with more than 1024 keys - ids sorted by
// popularity - takes the 16-keys-per-thread instance (correct, slower).
#pragma once
#include <rocprim/block/block_radix_sort.hpp>
#include "kge_sampler_common.hpp"

__device__ __forceinline__ int st_bucket(int64_t id, int64_t per) { return (int)(id >= per) + (int)(id >= 2 * per) + (int)(id >= 3 * per); }

// exclusive block scan of one uint64 per thread (ST_THREADS threads); *total = the block's sum.  `tmp`: 4 LDS words of 8 bytes.
__device__ __forceinline__ uint64_t st_block_excl_u64(uint64_t v, uint64_t *tmp, uint64_t *total) {
    const int t = threadIdx.x, lane = t & 63;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    __syncthreads();                                      // (tmp may still be read from an earlier scan)
    if (lane == 63) tmp[t >> 6] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ST_THREADS / 64; ++w) { const uint64_t x = tmp[w]; if (w < (t >> 6)) base += x; tot += x; }
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------------------------------
// phase 1
// ------------------------------------------------------------------------------------------------------------------------
template <typename K>
__device__ void sampler_tail_phase1(const SmpTail &s, int wg) {
    const SamplerArgs &a = s.a;
    __shared__ uint64_t tmp[ST_THREADS / 64];
    const int t = threadIdx.x;
    const int B = a.B, CN = a.C * a.N, NE = 2 * B + CN;
    const SlotLayout L = slot_layout(B, CN);
    const TailScratch S = tail_scratch(B, CN, sizeof(K) == 8);
    char *sb = a.slots;
    int32_t *hdr = (int32_t *)(s.scratch + S.hdr);
    K *ekeys = (K *)(s.scratch + S.ekeys);
    const int64_t step = a.state[1] + s.k;                   // 1-based step number of this batch
    const int64_t nb = a.n_train / B, gk = step - 1, ep = gk / nb, pos1 = (gk % nb) * B;
    const EpochConst ec = epoch_consts_cached(a, ep);
    const int64_t per = (a.n_ent + ST_NBK - 1) / ST_NBK;
    // The loads of a workgroup are staged by hand: every epoch index first, then every permutation entry, then every triple
    // component - with indices clamped instead of predicated, so that each stage is ONE round of independent requests (a loop
    // with the loads inside its body waits for the whole chain per iteration: the edge workgroup took 9.7 us that way, longer than
    // the first launch it rides on).
    constexpr int MAXE = 8;                                  // edges per thread: ceil(2048 / 256) (relations), ceil(1024 / 256) (an edge half)
    if (wg == 3) {                                           // relation ids + keys, in edge order (thread t: edges t, t + 256, ...)
        int64_t *rel_ids = (int64_t *)(sb + L.rel_ids);
        uint64_t *rkeys = (uint64_t *)(s.scratch + S.rkeys);
        int64_t e[MAXE], r[MAXE];
#pragma unroll
        for (int q = 0; q < MAXE; ++q) e[q] = (int64_t)epoch_index_fast((uint64_t)(pos1 + min(t + ST_THREADS * q, B - 1)), ec, (uint64_t)a.n_train);
        if (a.perm && !a.preperm) {
#pragma unroll
            for (int q = 0; q < MAXE; ++q) e[q] = a.perm[e[q]];
        }
#pragma unroll
        for (int q = 0; q < MAXE; ++q) r[q] = a.R[e[q]];
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int i = t + ST_THREADS * q;
            if (i < B) { rel_ids[i] = r[q]; rkeys[i] = ((uint64_t)r[q] << SP_CODE_BITS) | (uint64_t)i; }
        }
        return;
    }
    // edges (wg 0: the first half, wg 1: the second) / negatives (wg 2): thread t owns CONSECUTIVE elements so that a bucket receives
    // its keys in ascending code order (the sort of phase 2 is stable over the id bits only)
    constexpr int MAXI = 16;                                 // keys per thread: 2 * 4 edge ends or ceil(4096 / 256) negatives
    K key[MAXI];
    int bk[MAXI];
    uint64_t cnt = 0;                                        // four 16-bit counters
    int part0;                                               // the workgroup's part of every bucket region
    if (wg < 2) {
        int64_t *h_gid = (int64_t *)(sb + L.h_gid), *t_gid = (int64_t *)(sb + L.t_gid);
        const int Bh = (B + 1) / 2;                          // edges of the first half
        const int e_lo = wg == 0 ? 0 : Bh, e_hi = wg == 0 ? Bh : B;
        const int ept = (e_hi - e_lo + ST_THREADS - 1) / ST_THREADS;          // <= 4
        part0 = 2 * e_lo;
        constexpr int ME = 4;
        int64_t e[ME], hh[ME], tt[ME];
#pragma unroll
        for (int q = 0; q < ME; ++q) {
            const int i = min(e_lo + ept * t + min(q, max(ept - 1, 0)), max(e_hi - 1, 0));
            e[q] = (int64_t)epoch_index_fast((uint64_t)(pos1 + i), ec, (uint64_t)a.n_train);
        }
        if (a.perm && !a.preperm) {
#pragma unroll
            for (int q = 0; q < ME; ++q) e[q] = a.perm[e[q]];
        }
#pragma unroll
        for (int q = 0; q < ME; ++q) { hh[q] = a.H[e[q]]; tt[q] = a.T[e[q]]; }
#pragma unroll
        for (int q = 0; q < MAXI; ++q) bk[q] = -1;
#pragma unroll
        for (int q = 0; q < ME; ++q) {
            const int i = e_lo + ept * t + q;
            if (q < ept && i < e_hi) {
                h_gid[i] = hh[q]; t_gid[i] = tt[q];
                key[2 * q] = ((K)hh[q] << SP_CODE_BITS) | (K)(2 * i);
                key[2 * q + 1] = ((K)tt[q] << SP_CODE_BITS) | (K)(2 * i + 1);
                bk[2 * q] = st_bucket(hh[q], per); bk[2 * q + 1] = st_bucket(tt[q], per);
                cnt += (1ull << (16 * bk[2 * q])) + (1ull << (16 * bk[2 * q + 1]));
            }
        }
    } else {
        int64_t *neg_ids = (int64_t *)(sb + L.neg_ids);
        const int ipt = (CN + ST_THREADS - 1) / ST_THREADS;
        part0 = 2 * B;
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
            const int j = ipt * t + q;
            bk[q] = -1;
            if (q < ipt && j < CN) {
                const int64_t id = sample_negative(a, step, j);
                neg_ids[j] = id;
                key[q] = ((K)id << SP_CODE_BITS) | (K)(2 * B + j);
                bk[q] = st_bucket(id, per);
                cnt += 1ull << (16 * bk[q]);
            }
        }
    }
    uint64_t total;
    uint64_t base = st_block_excl_u64(cnt, tmp, &total);
#pragma unroll
    for (int q = 0; q < MAXI; ++q) {
        if (bk[q] >= 0) {
            const int b = bk[q];
            const int at = (int)((base >> (16 * b)) & 0xffff);
            ekeys[(int64_t)b * NE + part0 + at] = key[q];
            base += 1ull << (16 * b);
        }
    }
    if (t < ST_NBK) hdr[ST_H_CNT(t, wg)] = (int)((total >> (16 * t)) & 0xffff);
}

// ------------------------------------------------------------------------------------------------------------------------
// phase 2
// ------------------------------------------------------------------------------------------------------------------------
// in-LDS bitonic sort of n2 (power of two) DISTINCT keys by the workgroup's ST_THREADS threads, one compare-exchange stage per
// barrier: the instance for buckets / relation lists beyond 4 keys per thread (ids sorted by popularity put most of a batch into
// one bucket) - correct and slow (~25 us for 4096 keys), a register-resident radix sort of 16 64-bit keys per thread spills
template <typename K>
__device__ void st_bitonic_lds(K *keys, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += ST_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const K x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}
// phase 2, one entity bucket: sort (IPT keys per thread, blocked arrangement; LDS carved from `lds`), sorted keys -> scratch, and the
// bucket's totals (elements, unique entities, positive elements) -> header.  The scan and the plan arrays are phase 3's.
template <typename K, int IPT>
__device__ void st_bucket_sort(const SmpTail &s, int b, int c0, int c1, int c2, char *lds) {
    typedef rocprim::block_radix_sort<K, ST_THREADS, (IPT <= 4 ? IPT : 4)> Sort;
    const SamplerArgs &a = s.a;
    const int t = threadIdx.x;
    const int B = a.B, CN = a.C * a.N, NE = 2 * B + CN, n = c0 + c1 + c2;
    const int o1 = 2 * ((B + 1) / 2), o2 = 2 * B;            // where the second edge half's / the negatives' part of the bucket region starts
    constexpr int CAP = ST_THREADS * IPT;
    const TailScratch S = tail_scratch(B, CN, sizeof(K) == 8);
    const K *in = (const K *)(s.scratch + S.ekeys) + (int64_t)b * NE;
    K *out = (K *)(s.scratch + S.esort) + (int64_t)b * NE;
    int32_t *hdr = (int32_t *)(s.scratch + S.hdr);
    K *lk = reinterpret_cast<K *>(lds);                      // [CAP] sorted keys
    uint64_t *tmp = reinterpret_cast<uint64_t *>(lds + sizeof(K) * CAP);
    if constexpr (IPT <= 4) {
        K item[IPT];
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
            const int q = IPT * t + e;
            item[e] = q < c0 ? in[q] : (q < c0 + c1 ? in[o1 + (q - c0)] : (q < n ? in[o2 + (q - c0 - c1)] : ~(K)0));
        }
        unsigned idbits = 1;
        while (idbits < 8 * sizeof(K) - SP_CODE_BITS && ((uint64_t)(a.n_ent - 1) >> idbits)) ++idbits;
        Sort().sort(item, *reinterpret_cast<typename Sort::storage_type *>(lds), SP_CODE_BITS, SP_CODE_BITS + idbits);
        __syncthreads();                                     // the sort's storage is re-used below
#pragma unroll
        for (int e = 0; e < IPT; ++e) lk[IPT * t + e] = item[e];
    } else {
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        for (int q = t; q < n2; q += ST_THREADS) lk[q] = q < c0 ? in[q] : (q < c0 + c1 ? in[o1 + (q - c0)] : (q < n ? in[o2 + (q - c0 - c1)] : ~(K)0));
        __syncthreads();
        st_bitonic_lds<K>(lk, n2);                           // (keys are distinct: (id, code) order = the stable order by id)
    }
    __syncthreads();
    uint64_t sum = 0;                                        // low 32 bits: uniques, high: positives
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int q = IPT * t + e;
        if (q < n) {
            const K kq = lk[q];
            out[q] = kq;
            const uint64_t id = (uint64_t)(kq >> SP_CODE_BITS);
            const bool uq = q == 0 || (uint64_t)(lk[q - 1] >> SP_CODE_BITS) != id;
            const bool ps = (int)(kq & ((1u << SP_CODE_BITS) - 1)) < 2 * B;
            sum += (uq ? 1ull : 0ull) + (ps ? (1ull << 32) : 0ull);
        }
    }
    uint64_t total;
    (void)st_block_excl_u64(sum, tmp, &total);
    if (t == 0) { hdr[ST_H_NB(b)] = n; hdr[ST_H_UB(b)] = (int)(total & 0xffffffffu); hdr[ST_H_PB(b)] = (int)(total >> 32); }
}

// phase 3, one entity bucket: flags, bucket scan, bases from the header -> the plan arrays and records at their final positions.
// IPT <= 4: the bucket's keys, scan words and unique starts live in LDS (16 KB); the big instance keeps the scan words and unique
// starts in the bucket's global scratch rows (written, fenced, read back through L2) and fetches the neighbour key from global
// memory - the static LDS of EVERY workgroup of the host launches pays for what the rare instance declares.
template <typename K, int IPT>
__device__ void st_bucket_finish(const SmpTail &s, int b, char *sb, char *lds) {
    const SamplerArgs &a = s.a;
    const int t = threadIdx.x;
    const int B = a.B, CN = a.C * a.N, NE = 2 * B + CN;
    constexpr int CAP = ST_THREADS * IPT;
    const SlotLayout L = slot_layout(B, CN);
    const TailScratch S = tail_scratch(B, CN, sizeof(K) == 8);
    const int32_t *hdr = (const int32_t *)(s.scratch + S.hdr);
    int K0 = 0, U0 = 0, P0 = 0, UE = 0;
#pragma unroll
    for (int j = 0; j < ST_NBK; ++j) {
        const int nj = hdr[ST_H_NB(j)], uj = hdr[ST_H_UB(j)], pj = hdr[ST_H_PB(j)];
        if (j < b) { K0 += nj; U0 += uj; P0 += pj; }
        UE += uj;
    }
    const int n = hdr[ST_H_NB(b)], Pb = hdr[ST_H_PB(b)];
    const int N0 = K0 - P0;                                  // negatives in front of the bucket
    const K *keys = (const K *)(s.scratch + S.esort) + (int64_t)b * NE;
    int64_t *ue_id = (int64_t *)(sb + L.ue_id);
    int32_t *ue_pos_ptr = (int32_t *)(sb + L.ue_pos_ptr), *ue_pos_adj = (int32_t *)(sb + L.ue_pos_adj);
    int32_t *ue_neg_ptr = (int32_t *)(sb + L.ue_neg_ptr), *ue_neg_slot = (int32_t *)(sb + L.ue_neg_slot);
    int32_t *ue_rec = (int32_t *)(sb + L.ue_rec), *counts = (int32_t *)(sb + L.counts);
    K *lk = nullptr;
    uint32_t *ls, *ust;
    uint64_t *tmp;
    if constexpr (IPT <= 4) {
        lk = reinterpret_cast<K *>(lds);
        ls = reinterpret_cast<uint32_t *>(lds + sizeof(K) * CAP);
        ust = ls + CAP;
        tmp = reinterpret_cast<uint64_t *>(ust + CAP + 1 + ((CAP + 1) & 1));
    } else {
        ls = (uint32_t *)(s.scratch + S.escan) + (int64_t)b * NE;
        ust = (uint32_t *)(s.scratch + S.ust) + (int64_t)b * (NE + 1);
        tmp = reinterpret_cast<uint64_t *>(lds);
    }
    K item[IPT];
#pragma unroll
    for (int e = 0; e < IPT; ++e) item[e] = keys[min(IPT * t + e, max(n - 1, 0))];
    K prev = item[0];                                        // the key in front of this thread's first one
    if constexpr (IPT <= 4) {
#pragma unroll
        for (int e = 0; e < IPT; ++e) lk[IPT * t + e] = item[e];
        __syncthreads();
        if (t > 0) prev = lk[IPT * t - 1];
    } else {
        if (t > 0) prev = keys[min(IPT * t - 1, max(n - 1, 0))];
    }
    uint32_t f[IPT];
    uint64_t sum = 0;
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int q = IPT * t + e;
        f[e] = 0;
        if (q < n) {
            const uint64_t id = (uint64_t)(item[e] >> SP_CODE_BITS);
            const K pk = e == 0 ? prev : item[e > 0 ? e - 1 : 0];
            const bool uq = q == 0 || (uint64_t)(pk >> SP_CODE_BITS) != id;
            const bool ps = (int)(item[e] & ((1u << SP_CODE_BITS) - 1)) < 2 * B;
            f[e] = (uq ? 1u : 0u) | (ps ? 2u : 0u);
            sum += (uq ? 1ull : 0ull) + (ps ? (1ull << 32) : 0ull);
        }
    }
    uint64_t total;
    uint64_t run = st_block_excl_u64(sum, tmp, &total);
    const int U = (int)(total & 0xffffffffu);
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int q = IPT * t + e;
        if (q < n) {
            const int u = (int)(run & 0xffffffffu), pp = (int)(run >> 32), pn = q - pp;
            const int code = (int)(item[e] & ((1u << SP_CODE_BITS) - 1));
            ls[q] = (uint32_t)pp;
            if (f[e] & 1u) {
                ust[u] = (uint32_t)q;
                ue_id[U0 + u] = (int64_t)(uint64_t)(item[e] >> SP_CODE_BITS); ue_pos_ptr[U0 + u] = P0 + pp; ue_neg_ptr[U0 + u] = N0 + pn;
            }
            if (code < 2 * B) ue_pos_adj[P0 + pp] = code; else ue_neg_slot[N0 + pn] = code - 2 * B;
            run += (f[e] & 1u ? 1ull : 0ull) + (f[e] & 2u ? (1ull << 32) : 0ull);
        }
    }
    if (t == 0) ust[U] = (uint32_t)n;
    if (b == ST_NBK - 1 && t == 0) { ue_pos_ptr[UE] = 2 * B; ue_neg_ptr[UE] = CN; counts[0] = UE; }
    if constexpr (IPT > 4) __threadfence();                  // (ls / ust live in global memory in this instance)
    __syncthreads();
    // records: {id_lo, id_hi, pos_begin, pos_end} {neg_begin, neg_end, first adj | -1, first slot | -1}
    for (int u = t; u < U; u += ST_THREADS) {
        const int q = (int)ust[u], q1 = (int)ust[u + 1];
        const int p0 = (int)ls[q], p1 = q1 < n ? (int)ls[q1] : Pb;
        const int n0 = q - p0, n1 = q1 - p1;
        K kq, kn;
        if constexpr (IPT <= 4) { kq = lk[q]; kn = lk[min(q + (p1 - p0), n - 1)]; }
        else { kq = keys[q]; kn = keys[min(q + (p1 - p0), n - 1)]; }
        const uint64_t id = (uint64_t)(kq >> SP_CODE_BITS);
        int4 r0, r1;
        r0.x = (int32_t)(id & 0xFFFFFFFF); r0.y = (int32_t)(id >> 32); r0.z = P0 + p0; r0.w = P0 + p1;
        r1.x = N0 + n0; r1.y = N0 + n1;
        r1.z = p1 > p0 ? (int)(kq & ((1u << SP_CODE_BITS) - 1)) : -1;
        r1.w = n1 > n0 ? (int)(kn & ((1u << SP_CODE_BITS) - 1)) - 2 * B : -1;
        reinterpret_cast<int4 *>(ue_rec)[2 * (U0 + u)] = r0;
        reinterpret_cast<int4 *>(ue_rec)[2 * (U0 + u) + 1] = r1;
    }
}

// the relation plan of the batch, complete (one workgroup holds all B keys): IPT keys per thread
template <int IPT>
__device__ void st_relation_plan(const SmpTail &s, char *lds) {
    typedef rocprim::block_radix_sort<uint64_t, ST_THREADS, (IPT <= 4 ? IPT : 4)> Sort;
    const SamplerArgs &a = s.a;
    const int t = threadIdx.x;
    const int B = a.B, CN = a.C * a.N;
    constexpr int CAP = ST_THREADS * IPT;
    const SlotLayout L = slot_layout(B, CN);
    const TailScratch S = tail_scratch(B, CN, false);        // (rkeys sits behind the key-size dependent arrays: resolved by the caller)
    (void)S;
    char *sb = a.slots;
    int64_t *ur_id = (int64_t *)(sb + L.ur_id);
    int32_t *ur_ptr = (int32_t *)(sb + L.ur_ptr), *ur_edge = (int32_t *)(sb + L.ur_edge), *counts = (int32_t *)(sb + L.counts);
    const uint64_t *rkeys = reinterpret_cast<const uint64_t *>(s.scratch + tail_scratch(B, CN, a.n_ent > (1ll << (32 - SP_CODE_BITS))).rkeys);
    uint64_t *lk = reinterpret_cast<uint64_t *>(lds);        // [CAP] sorted keys (after the sort)
    uint64_t *tm2 = lk + CAP;
    if constexpr (IPT <= 4) {
        uint64_t item[IPT];
        uint64_t mx = 0;
#pragma unroll
        for (int e = 0; e < IPT; ++e) {
            const int q = IPT * t + e;
            item[e] = q < B ? rkeys[q] : ~0ull;
            if (q < B) mx = mx > item[e] ? mx : item[e];
        }
        // bits of the largest relation id of the batch (the sampler does not know n_rel)
        uint64_t *tmp = reinterpret_cast<uint64_t *>(lds + sizeof(typename Sort::storage_type) + 64);
        tmp = reinterpret_cast<uint64_t *>(((uintptr_t)tmp + 7) & ~(uintptr_t)7);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint64_t v = __shfl_xor(mx, o, 64); mx = mx > v ? mx : v; }
        if ((t & 63) == 0) tmp[t >> 6] = mx;
        __syncthreads();
        mx = 0;
#pragma unroll
        for (int w = 0; w < ST_THREADS / 64; ++w) mx = mx > tmp[w] ? mx : tmp[w];
        unsigned idbits = 1;
        while (idbits < 64 - SP_CODE_BITS && ((mx >> SP_CODE_BITS) >> idbits)) ++idbits;
        __syncthreads();
        Sort().sort(item, *reinterpret_cast<typename Sort::storage_type *>(lds), SP_CODE_BITS, SP_CODE_BITS + idbits);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < IPT; ++e) lk[IPT * t + e] = item[e];
    } else {
        int n2 = 1;
        while (n2 < B) n2 <<= 1;
        for (int q = t; q < n2; q += ST_THREADS) lk[q] = q < B ? rkeys[q] : ~0ull;
        __syncthreads();
        st_bitonic_lds<uint64_t>(lk, n2);
    }
    __syncthreads();
    uint32_t f[IPT];
    uint64_t sum = 0;
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int q = IPT * t + e;
        f[e] = 0;
        if (q < B) {
            const uint64_t id = lk[q] >> SP_CODE_BITS;
            f[e] = (q == 0 || (lk[q - 1] >> SP_CODE_BITS) != id) ? 1u : 0u;
            sum += f[e];
        }
    }
    uint64_t total;
    uint64_t run = st_block_excl_u64(sum, tm2, &total);
    const int UR = (int)total;
#pragma unroll
    for (int e = 0; e < IPT; ++e) {
        const int q = IPT * t + e;
        if (q < B) {
            ur_edge[q] = (int)(lk[q] & ((1u << SP_CODE_BITS) - 1));
            if (f[e]) { ur_id[run] = (int64_t)(lk[q] >> SP_CODE_BITS); ur_ptr[run] = q; }
            run += f[e];
        }
    }
    if (t == 0) {
        int64_t *st = a.state;
        const int64_t step = st[1] + s.k;
        ur_ptr[UR] = B; counts[1] = UR; counts[2] = (int)(step & 1 ? 0 : 1);
        // the epoch constants of the NEXT job's batch (phase 1 of the next step looks them up; kge_sampler_common.hpp): batch index of
        // that job = this job's 1-based step number
        const int64_t nbat = a.n_train / B, ep_next = step / nbat;
        if (st[4] != ep_next) {
            const EpochConst c = epoch_consts_slow(a, ep_next);
            st[5] = (int64_t)c.mul; st[6] = (int64_t)c.add; st[7] = (int64_t)c.rinv;
            __threadfence();
            st[4] = ep_next;
        }
    }
}

// LDS of a phase-2 workgroup, ONE buffer for both key sizes: the larger of the sorts' storage and the arrays that replace it
// (16 keys of 64 bits per thread)
__device__ __forceinline__ char *st_phase2_lds() {
    typedef rocprim::block_radix_sort<uint64_t, ST_THREADS, 4> SortBig;
    typedef rocprim::block_radix_sort<uint64_t, ST_THREADS, 4> SortRel;
    constexpr size_t CAPB = ST_THREADS * 16, CAPS = ST_THREADS * 4;
    constexpr size_t need_a = sizeof(typename SortBig::storage_type);
    constexpr size_t need_b0 = 8 * CAPB + 8 * 8, need_b1 = 8 * CAPS + 8 * 8;
    constexpr size_t need_b = need_b0 > need_b1 ? need_b0 : need_b1;
    constexpr size_t need_c = sizeof(typename SortRel::storage_type) + 128;
    constexpr size_t need_d = 8 * ST_THREADS * 8 + 8 * 8;
    constexpr size_t m1 = need_a > need_b ? need_a : need_b, m2 = need_c > need_d ? need_c : need_d;
    __shared__ __attribute__((aligned(16))) char lds[(m1 > m2 ? m1 : m2) + 16];
    return lds;
}
template <typename K>
__device__ void sampler_tail_phase2(const SmpTail &s, int wg) {
    char *lds = st_phase2_lds();
    const SamplerArgs &a = s.a;
    const int B = a.B, CN = a.C * a.N;
    if (wg == ST_NBK) {
        if (B <= 4 * ST_THREADS) st_relation_plan<4>(s, lds); else st_relation_plan<8>(s, lds);
        return;
    }
    const TailScratch S = tail_scratch(B, CN, sizeof(K) == 8);
    const int32_t *hdr = (const int32_t *)(s.scratch + S.hdr);
    const int c0 = hdr[ST_H_CNT(wg, 0)], c1 = hdr[ST_H_CNT(wg, 1)], c2 = hdr[ST_H_CNT(wg, 2)];
    if (c0 + c1 + c2 <= 4 * ST_THREADS) st_bucket_sort<K, 4>(s, wg, c0, c1, c2, lds);
    else st_bucket_sort<K, 16>(s, wg, c0, c1, c2, lds);
}

// ------------------------------------------------------------------------------------------------------------------------
// phase 3
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ char *st_phase3_lds() {          // keys [1024] (64-bit), scan words [1024], unique starts [1025], scan scratch
    constexpr size_t CAPS = ST_THREADS * 4;
    __shared__ __attribute__((aligned(16))) char lds[8 * CAPS + 4 * CAPS + 4 * (CAPS + 2) + 8 * 8 + 16];
    return lds;
}
template <typename K>
__device__ void sampler_tail_phase3(const SmpTail &s, int wg, char *sb, bool last) {
    const SamplerArgs &a = s.a;
    const int t = threadIdx.x;
    const int B = a.B, CN = a.C * a.N, NE = 2 * B + CN;
    const SlotLayout L = slot_layout(B, CN);
    const TailScratch S = tail_scratch(B, CN, sizeof(K) == 8);
    int32_t *counts = (int32_t *)(sb + L.counts);
    if (wg == ST_NBK) {
        // relation records + the batch's longest relation list (kge_sampler.hip, part 1); the lists were written in phase 2
        const int64_t *ur_id = (const int64_t *)(sb + L.ur_id);
        const int32_t *ur_ptr = (const int32_t *)(sb + L.ur_ptr), *ur_edge = (const int32_t *)(sb + L.ur_edge);
        int32_t *ur_rec = (int32_t *)(sb + L.ur_rec);
        __shared__ int maxlen_sh;
        if (t == 0) maxlen_sh = 0;
        __syncthreads();
        const int UR = counts[1];
        int mylen = 0;
        for (int u = t; u < UR; u += ST_THREADS) {
            const int64_t id = ur_id[u];
            const int e0 = ur_ptr[u], e1 = ur_ptr[u + 1];
            int32_t *rec = ur_rec + 8 * u;
            rec[0] = (int32_t)(id & 0xFFFFFFFF); rec[1] = (int32_t)(id >> 32);
            rec[2] = e0; rec[3] = e1; rec[4] = ur_edge[e0]; rec[5] = 0; rec[6] = 0; rec[7] = 0;
            mylen = max(mylen, e1 - e0);
        }
        if (mylen > 8) atomicMax(&maxlen_sh, mylen);
        __syncthreads();
        if (t == 0) {
            counts[3] = maxlen_sh;
            if (last && s.advance > 0) {                     // the group's last batch: every phase-1 workgroup of the group has read the state
                int64_t *st = a.state;
                const int64_t p_ = st[0], s_ = st[1];
                st[0] = (p_ + (int64_t)s.advance * B) % a.n_train;
                st[1] = s_ + s.advance;
            }
        }
        return;
    }
    // entity bucket `wg`: the common instance (<= 4 keys per thread) or the big one
    const int32_t *hdr = (const int32_t *)(s.scratch + S.hdr);
    char *lds = st_phase3_lds();
    if (hdr[ST_H_NB(wg)] <= 4 * ST_THREADS) st_bucket_finish<K, 4>(s, wg, sb, lds);
    else st_bucket_finish<K, 16>(s, wg, sb, lds);
}

// dispatchers called from the host kernels: `wg` = index among the launch's tail workgroups
__device__ __forceinline__ bool st_key64(const SmpTail &s) { return s.a.n_ent > (1ll << (32 - SP_CODE_BITS)); }
__device__ __forceinline__ void sampler_tail_p1(const SmpTail &s, int wg) {       // first launch: phase 1 of this job + phase 3 of the previous one
    if (wg >= ST_P1_WGS + ST_P3_WGS) return;
    KGE_TL(6);
    if (wg >= ST_P1_WGS) {
        if (!s.slot3) return;
        if (st_key64(s)) sampler_tail_phase3<uint64_t>(s, wg - ST_P1_WGS, s.slot3, false);
        else sampler_tail_phase3<uint32_t>(s, wg - ST_P1_WGS, s.slot3, false);
        return;
    }
    if (st_key64(s)) sampler_tail_phase1<uint64_t>(s, wg); else sampler_tail_phase1<uint32_t>(s, wg);
}
__device__ __forceinline__ void sampler_tail_p2(const SmpTail &s, int wg) {
    if (wg >= ST_P2_WGS) return;
    KGE_TL(6);
    if (st_key64(s)) sampler_tail_phase2<uint64_t>(s, wg); else sampler_tail_phase2<uint32_t>(s, wg);
}
__device__ __forceinline__ void sampler_tail_p3(const SmpTail &s, int wg) {       // update launch: phase 3 of the group's LAST job
    if (wg >= ST_P3_WGS) return;
    KGE_TL(6);
    if (st_key64(s)) sampler_tail_phase3<uint64_t>(s, wg, s.a.slots, true); else sampler_tail_phase3<uint32_t>(s, wg, s.a.slots, true);
}
