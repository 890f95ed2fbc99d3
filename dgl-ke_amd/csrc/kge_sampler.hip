// kge_sampler.hip - on-device mini-batch sampler + plan builder.
//
// Restates, on the GPU, what the reference gets from DGL's C++ EdgeSampler and its Python wrappers
// (dataloader/sampler.py:376-419 call site: batch_size positives, chunked uniform negatives over
// all entities with replacement, positives not excluded; :853-859 odd steps corrupt tails / even
// steps heads; :503-504 whole batches only) and the duplicate-grouping "plan" that
// dglke_amd/plan.py builds on the host (include/kge_hip.h `kge_batch`).  DGL's RNG stream cannot
// be reproduced (no dgl here), so parity is defined structurally: tests rebuild the plan on the
// host from the ids the kernel sampled and compare every array.
//
// One workgroup (1024 threads) builds one batch entirely in LDS:
//   1. positives: whole batches of the epoch's edge order (base permutation re-keyed per epoch) -> (h, r, t);  negatives: counter-based hash
//      RNG keyed by (seed, step, j) -> uniform id in [0, n_ent)
//   2. sort of <= 4096 keys (wide instance: 8192)  (entity id << 12 (13) | element code), code = edge*2+side  (a stable block radix sort over the id bits:
//      the codes are the positions; the relation plan keeps the register bitonic network)
//      for positive edge ends, 2B + slot for negatives  -> elements grouped by entity, ascending
//      code inside a group (= the order of the host plan and of index_add_)
//   3. one block-wide scan of packed (unique, positive, negative) flags -> unique entity list,
//      CSR pointers and lists, 32-byte plan records
//   4. same for the relations (sort of rel id << 12 | edge).
// Many batches are built concurrently (grid = number of slots), ahead of the training steps, so the
// sampler is off the critical path exactly like the reference's prefetching sampler threads.
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include "kge_common.hpp"
KGE_TL_DEFINE(sampler)

#include "kge_sampler_common.hpp"

// in-LDS bitonic sort of n2 (power of two) keys by all SP_THREADS threads - generic version: one
// compare-exchange stage per barrier
template <typename K>
__device__ void bitonic_sort_lds(K *keys, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += SP_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const K a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Register-resident bitonic sort: thread t owns the E consecutive keys [E*t, E*t + E).  A stage with partner distance
// j < E exchanges inside the thread, j < 64*E inside the wavefront (cross-lane shuffles, no barrier), only j >= 64*E
// goes through LDS - 10 of the 78 stages for 4096 keys, 6 of 55 for 1024 (each __syncthreads of 16 wavefronts costs
// ~0.4 us: the barriers were most of the sampler kernel's 91 us).  Keys are distinct, so the result is THE sorted order.
// K: key type.  32-bit keys (entity id < 2^20, see launch_sample_batches) halve the cross-lane traffic of a stage (one
// ds_bpermute instead of two per key) and turn the 64-bit compare + two selects into v_min_u32 / v_max_u32: the sort is
// VALU- and LDS-crossbar-bound on ONE CU (16 wavefronts), 37 of the 52 us of an entity-plan workgroup with 64-bit keys.
template <int E, typename K>
__device__ void bitonic_sort_regs(K *keys, int n2) {
    const int t = threadIdx.x, lane = t & 63;
    (void)lane;
    K v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = keys[E * t + e];
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {                                   // partner in this thread
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = e ^ j;
                    if (p > e) {
                        const bool up = (((E * t + e) & k) == 0);
                        const K a = v[e], b = v[p];
                        const K lo = a < b ? a : b, hi = a < b ? b : a;
                        v[e] = up ? lo : hi; v[p] = up ? hi : lo;
                    }
                }
            } else if (j < 64 * E) {                       // partner in this wavefront: lane ^ (j / E), same slot
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = E * t + e;
                    const K o = __shfl_xor(v[e], j / E, 64);
                    const bool lower = (i & j) == 0, up = (i & k) == 0;
                    const bool want_min = lower == up;
                    const K lo = v[e] < o ? v[e] : o, hi = v[e] < o ? o : v[e];
                    v[e] = want_min ? lo : hi;
                }
            } else {                                       // partner in another wavefront: through LDS
                __syncthreads();                           // everybody has read what an earlier LDS stage wrote
#pragma unroll
                for (int e = 0; e < E; ++e) keys[E * t + e] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = E * t + e;
                    const K o = keys[i ^ j];
                    const bool lower = (i & j) == 0, up = (i & k) == 0;
                    const bool want_min = lower == up;
                    const K lo = v[e] < o ? v[e] : o, hi = v[e] < o ? o : v[e];
                    v[e] = want_min ? lo : hi;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) keys[E * t + e] = v[e];
    __syncthreads();
}

template <typename K>
__device__ void bitonic_sort(K *keys, int n2) {
    if (n2 == 4 * SP_THREADS) bitonic_sort_regs<4, K>(keys, n2);
    else if (n2 == 2 * SP_THREADS) bitonic_sort_regs<2, K>(keys, n2);
    else if (n2 == SP_THREADS) bitonic_sort_regs<1, K>(keys, n2);
    else bitonic_sort_lds<K>(keys, n2);
}

// block-wide exclusive scan of n (<= EPT * SP_THREADS) uint32 values in `v` (in place); returns the total.
// Each thread owns EPT consecutive elements.
template <int EPT = 4>
__device__ uint32_t block_exclusive_scan(uint32_t *v, int n, uint32_t *wsum /*[SP_THREADS/64]*/) {
    const int t = threadIdx.x;
    uint32_t loc[EPT], s = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { const int i = EPT * t + e; loc[e] = i < n ? v[i] : 0; s += loc[e]; }
    uint32_t inc = s;                        // inclusive scan of s across the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if ((t & 63) >= o) inc += up;
    }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < SP_THREADS / 64; ++w) { const uint32_t x = wsum[w]; if (w < (t >> 6)) base += x; total += x; }
    uint32_t run = base + inc - s;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { const int i = EPT * t + e; if (i < n) v[i] = run; run += loc[e]; }
    __syncthreads();
    return total;
}

// K: key type of the entity plan - uint32_t when (id << CBITS) fits, else uint64_t.  MAXE / CBITS: SP_MAXE / SP_CODE_BITS (4 keys
// per thread: every BASELINE config) or the wide instance SP_MAXE_BIG / SP_CODE_BITS_BIG (8 keys per thread; B <= MAXE / 2)
template <typename K, int MAXE = SP_MAXE, int CBITS = SP_CODE_BITS>
__global__ __launch_bounds__(SP_THREADS) void sample_plan_kernel(SamplerArgs a) {
    constexpr int EPT = MAXE / SP_THREADS;
    typedef rocprim::block_radix_sort<K, SP_THREADS, EPT> BlockSort;
    // keys: the relation plan sorts <= MAXE / 2 64-bit keys, the entity plan MAXE keys of type K.  The radix sort's scratch and the
    // scan's flags are never live together: one buffer (wide instance with 64-bit keys: 64 + 72 KB instead of 168)
    constexpr size_t KEY_BYTES = sizeof(K) * MAXE > 8 * (size_t)(MAXE / 2) ? sizeof(K) * MAXE : 8 * (size_t)(MAXE / 2);
    constexpr size_t SCR_BYTES = sizeof(typename BlockSort::storage_type) > 4 * (size_t)MAXE ? sizeof(typename BlockSort::storage_type) : 4 * (size_t)MAXE;
    __shared__ __attribute__((aligned(16))) unsigned char keys_raw[KEY_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char scr_raw[SCR_BYTES];
    uint64_t *keys = reinterpret_cast<uint64_t *>(keys_raw);
    uint32_t *scan = reinterpret_cast<uint32_t *>(scr_raw);
    __shared__ uint32_t wsum[SP_THREADS / 64];
    const int t = threadIdx.x;
    KGE_TL((int)(blockIdx.x & 1));             // developer timeline: kid 0 = entity-plan workgroups, 1 = relation-plan workgroups
    // two workgroups per batch: part 0 samples the edge ends + negatives and builds the entity plan, part 1 samples the
    // relations of the SAME edges and builds the relation plan - independent work (91 -> 72 -> ~50 us per launch)
    const int slot = blockIdx.x >> 1, part = blockIdx.x & 1;
    const int B = a.B, CN = a.C * a.N, NE = 2 * B + CN;
    // the launch's state {position, first step} is read ONCE per workgroup (thread 0 -> LDS); after that read the workgroup
    // takes a ticket, and the workgroup that takes the last ticket advances the state for the next launch - every other
    // workgroup has read it by then (no separate "advance" launch)
    __shared__ int64_t st_sh[2];
    if (t == 0) {
        const int64_t p_ = a.state[0], s_ = a.state[1];
        st_sh[0] = p_; st_sh[1] = s_;
        __threadfence();
        const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long *>(a.state + 2), 1ULL);
        if (ticket == (unsigned long long)gridDim.x - 1) {
            const int n_slots = (int)(gridDim.x >> 1);
            a.state[0] = (p_ + (int64_t)n_slots * B) % a.n_train;
            a.state[1] = s_ + n_slots;
            a.state[2] = 0;
        }
    }
    __syncthreads();
    const int64_t pos0 = st_sh[0] + (int64_t)slot * B;
    const int64_t step = st_sh[1] + slot;                     // 1-based step number of this batch
    const SlotLayout L = slot_layout(B, CN);
    char *sb = a.slots + (int64_t)slot * a.slot_bytes;
    int64_t *h_gid = (int64_t *)(sb + L.h_gid), *t_gid = (int64_t *)(sb + L.t_gid);
    int64_t *rel_ids = (int64_t *)(sb + L.rel_ids), *neg_ids = (int64_t *)(sb + L.neg_ids);
    int64_t *ue_id = (int64_t *)(sb + L.ue_id), *ur_id = (int64_t *)(sb + L.ur_id);
    int32_t *ue_pos_ptr = (int32_t *)(sb + L.ue_pos_ptr), *ue_pos_adj = (int32_t *)(sb + L.ue_pos_adj);
    int32_t *ue_neg_ptr = (int32_t *)(sb + L.ue_neg_ptr), *ue_neg_slot = (int32_t *)(sb + L.ue_neg_slot);
    int32_t *ur_ptr = (int32_t *)(sb + L.ur_ptr), *ur_edge = (int32_t *)(sb + L.ur_edge);
    int32_t *ue_rec = (int32_t *)(sb + L.ue_rec), *ur_rec = (int32_t *)(sb + L.ur_rec);
    int32_t *counts = (int32_t *)(sb + L.counts);

    // ---- 1. sample ----
    // Whole batches only (the reference drops the trailing partial batch of an epoch, dataloader/sampler.py:503-504)
    // and a NEW edge order every epoch (shuffle=True): batch k of the run is batch k % nb of epoch k / nb, whose
    // order is the base permutation composed with an affine bijection of [0, n_train) keyed by (seed, epoch) -
    // a fresh permutation without regenerating n_train indices; epoch 0 uses the base permutation as it is.
    const int64_t nb = a.n_train / B;
    const int64_t gk = step - 1, ep = gk / nb, pos1 = (gk % nb) * B;
    // (multiplier AND offset are hashed from (seed, epoch): a table of eight multipliers made epochs e and e + 8 walk the same
    //  cyclic sequence, and graphs of >= 2^31 triples got mul = 1 - every epoch a rotation of epoch 0)
    uint64_t mul, add;
    epoch_affine(a, ep, mul, add);
    (void)pos0;
    if (part == 1) {
        // ---- relations of the batch's edges (part 1) ----
        int b2 = 1;
        while (b2 < B) b2 <<= 1;
        for (int i = t; i < b2; i += SP_THREADS) {
            uint64_t key = ~0ULL;
            if (i < B) {
                int64_t e = (int64_t)epoch_index((uint64_t)(pos1 + i), mul, add, (uint64_t)a.n_train);
                if (a.perm) e = a.perm[e];
                const int64_t r = a.R[e];
                rel_ids[i] = r;
                key = ((uint64_t)r << CBITS) | (uint64_t)i;
            }
            keys[i] = key;
        }
        __syncthreads();
        bitonic_sort<uint64_t>(keys, b2);
        for (int k = t; k < B; k += SP_THREADS) {
            const uint64_t id = keys[k] >> CBITS;
            scan[k] = (k == 0 || (keys[k - 1] >> CBITS) != id) ? 1u : 0u;
        }
        __syncthreads();
        const int UR = (int)block_exclusive_scan<EPT>(scan, B, wsum);
        for (int k = t; k < B; k += SP_THREADS) {
            const uint64_t id = keys[k] >> CBITS;
            const bool uniq = k == 0 || (keys[k - 1] >> CBITS) != id;
            const int u = (int)scan[k];
            ur_edge[k] = (int)(keys[k] & ((1u << CBITS) - 1));
            if (uniq) { ur_id[u] = (int64_t)id; ur_ptr[u] = k; }
        }
        __shared__ int maxlen_sh;
        if (t == 0) { ur_ptr[UR] = B; counts[1] = UR; counts[2] = (int)(step & 1 ? 0 : 1); maxlen_sh = 0; }
        __syncthreads();
        __threadfence_block();
        int mylen = 0;
        for (int u = t; u < UR; u += SP_THREADS) {
            const int64_t id = ur_id[u];
            int32_t *rec = ur_rec + 8 * u;
            rec[0] = (int32_t)(id & 0xFFFFFFFF); rec[1] = (int32_t)(id >> 32);
            rec[2] = ur_ptr[u]; rec[3] = ur_ptr[u + 1]; rec[4] = ur_edge[ur_ptr[u]]; rec[5] = 0; rec[6] = 0; rec[7] = 0;
            mylen = max(mylen, rec[3] - rec[2]);
        }
        // counts[3] = edges of the batch's most frequent relation: the update kernel shares lists of that length between the
        // wavefronts of a workgroup - and skips the workgroup barrier that decision needs when no list of the batch is long
        if (mylen > 8) atomicMax(&maxlen_sh, mylen);
        __syncthreads();
        if (t == 0) counts[3] = maxlen_sh;
        return;
    }
    K *ek = reinterpret_cast<K *>(keys);        // the entity plan's keys
    for (int i = t; i < B; i += SP_THREADS) {
        int64_t e = (int64_t)epoch_index((uint64_t)(pos1 + i), mul, add, (uint64_t)a.n_train);
        if (a.perm) e = a.perm[e];
        const int64_t h = a.H[e], tl = a.T[e];
        h_gid[i] = h; t_gid[i] = tl;
        ek[2 * i] = ((K)h << CBITS) | (K)(2 * i);
        ek[2 * i + 1] = ((K)tl << CBITS) | (K)(2 * i + 1);
    }
    for (int j = t; j < CN; j += SP_THREADS) {
        const int64_t id = sample_negative(a, step, j);
        neg_ids[j] = id;
        ek[2 * B + j] = ((K)id << CBITS) | (K)(2 * B + j);
    }
#ifdef SP_BITONIC
    int n2 = 1;
    while (n2 < NE) n2 <<= 1;
    for (int i = NE + t; i < n2; i += SP_THREADS) ek[i] = ~(K)0;
    __syncthreads();
#ifdef KGE_TL_MARKS
    KGE_TL_MARK(0);              // ids sampled, keys in LDS
#endif
    // ---- 2. sort by (entity, code) ----
    bitonic_sort<K>(ek, n2);
#else
    for (int i = NE + t; i < MAXE; i += SP_THREADS) ek[i] = ~(K)0;
    __syncthreads();
#ifdef KGE_TL_MARKS
    KGE_TL_MARK(0);              // ids sampled, keys in LDS
#endif
    // ---- 2. sort by (entity, code): the codes ARE the positions, so a STABLE sort by the id bits alone gives that order -
    // a block radix sort (rocPRIM: 8 bits per pass, ranks by wavefront matching) over the bits of n_ent - 1: 2 passes for
    // FB15k's 14 951 entities, 4 for Freebase's 86 M, instead of the 78 compare-exchange stages of the bitonic network
    {
        typename BlockSort::storage_type &sort_tmp = *reinterpret_cast<typename BlockSort::storage_type *>(scr_raw);
        K item[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) item[e] = ek[(EPT) * t + e];
        unsigned idbits = 1;
        while (idbits < 8 * sizeof(K) - CBITS && ((uint64_t)(a.n_ent - 1) >> idbits)) ++idbits;
        BlockSort().sort(item, sort_tmp, CBITS, CBITS + idbits);
#pragma unroll
        for (int e = 0; e < EPT; ++e) ek[(EPT) * t + e] = item[e];
        __syncthreads();
    }
#endif
#ifdef KGE_TL_MARKS
    KGE_TL_MARK(1);              // sorted
#endif
    // ---- 3. packed flags: bit fields {unique: 0..15, positive: 16..31}; negative rank = k - positive rank ----
    for (int k = t; k < NE; k += SP_THREADS) {
        const uint64_t id = ek[k] >> CBITS, code = ek[k] & ((1u << CBITS) - 1);
        const bool uniq = k == 0 || ((uint64_t)(ek[k - 1] >> CBITS)) != id;
        scan[k] = (uniq ? 1u : 0u) | (code < (uint64_t)(2 * B) ? (1u << 16) : 0u);
    }
    __syncthreads();
    const uint32_t tot = block_exclusive_scan<EPT>(scan, NE, wsum);
    const int UE = (int)(tot & 0xFFFF);
    for (int k = t; k < NE; k += SP_THREADS) {
        const uint64_t id = ek[k] >> CBITS;
        const int code = (int)(ek[k] & ((1u << CBITS) - 1));
        const bool uniq = k == 0 || ((uint64_t)(ek[k - 1] >> CBITS)) != id;
        const uint32_t ex = scan[k];
        const int u = (int)(ex & 0xFFFF), pp = (int)(ex >> 16), pn = k - pp;
        if (uniq) { ue_id[u] = (int64_t)id; ue_pos_ptr[u] = pp; ue_neg_ptr[u] = pn; }
        if (code < 2 * B) ue_pos_adj[pp] = code; else ue_neg_slot[pn] = code - 2 * B;
    }
    if (t == 0) { ue_pos_ptr[UE] = 2 * B; ue_neg_ptr[UE] = CN; counts[0] = UE; }
    __syncthreads();
    __threadfence_block();
#ifdef KGE_TL_MARKS
    KGE_TL_MARK(2);              // lists written
#endif
    for (int u = t; u < UE; u += SP_THREADS) {
        const int64_t id = ue_id[u];
        const int p0 = ue_pos_ptr[u], p1 = ue_pos_ptr[u + 1], n0 = ue_neg_ptr[u], n1 = ue_neg_ptr[u + 1];
        int32_t *rec = ue_rec + 8 * u;
        rec[0] = (int32_t)(id & 0xFFFFFFFF); rec[1] = (int32_t)(id >> 32);
        rec[2] = p0; rec[3] = p1; rec[4] = n0; rec[5] = n1;
        rec[6] = p1 > p0 ? ue_pos_adj[p0] : -1;
        rec[7] = n1 > n0 ? ue_neg_slot[n0] : -1;
    }
}

int launch_sample_batches(const SamplerArgs &a, int n_slots, hipStream_t s) {
    if (n_slots <= 0) return KGE_OK;
    if (2 * a.B + a.C * a.N > SP_MAXE || a.B > SP_MAXE / 2) {      // the wide instance: 8 keys per thread, 13 code bits
        if (a.n_ent <= (1ll << (32 - SP_CODE_BITS_BIG)))
            hipLaunchKernelGGL((sample_plan_kernel<uint32_t, SP_MAXE_BIG, SP_CODE_BITS_BIG>), dim3(2 * n_slots), dim3(SP_THREADS), 0, s, a);
        else hipLaunchKernelGGL((sample_plan_kernel<uint64_t, SP_MAXE_BIG, SP_CODE_BITS_BIG>), dim3(2 * n_slots), dim3(SP_THREADS), 0, s, a);
    } else if (a.n_ent <= (1ll << (32 - SP_CODE_BITS))) hipLaunchKernelGGL(sample_plan_kernel<uint32_t>, dim3(2 * n_slots), dim3(SP_THREADS), 0, s, a);
    else hipLaunchKernelGGL(sample_plan_kernel<uint64_t>, dim3(2 * n_slots), dim3(SP_THREADS), 0, s, a);
    return hipGetLastError() == hipSuccess ? KGE_OK : KGE_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

size_t kge_sampler_slot_bytes(int B, int C, int N) {
    return (size_t)slot_layout(B, C * N).total;
}

int kge_sample_batches(const int64_t *heads, const int64_t *rels, const int64_t *tails, const int64_t *perm,
                       int64_t n_train, int64_t n_ent, int B, int C, int chunk, int N, uint64_t seed,
                       int64_t *state, void *slots, size_t slot_bytes, int n_slots, void *stream) {
    if (n_train < B) return kge_fail(KGE_ERR_ARG, "kge_sample_batches: fewer training triples than one batch (n_train < batch)");
    if (!heads || !rels || !tails || !state || !slots || n_train <= 0 || n_ent <= 0 || B <= 0 || C <= 0 ||
        chunk <= 0 || N <= 0 || C * chunk != B)
        return KGE_ERR_ARG;
    if (2 * B + C * N > SP_MAXE_BIG || B > SP_MAXE_BIG / 2 || n_ent >= ((int64_t)1 << (64 - SP_CODE_BITS_BIG)))
        return KGE_ERR_ARG;                      // larger batches: build the plan on the host
    if (slot_bytes < kge_sampler_slot_bytes(B, C, N)) return KGE_ERR_WORKSPACE;
    SamplerArgs a{};
    a.H = heads; a.R = rels; a.T = tails; a.perm = perm; a.n_train = n_train; a.n_ent = n_ent;
    a.B = B; a.C = C; a.chunk = chunk; a.N = N; a.seed = seed; a.state = state;
    a.slots = (char *)slots; a.slot_bytes = (int64_t)slot_bytes;
    return launch_sample_batches(a, n_slots, (hipStream_t)stream);
}

// fill a kge_batch whose arrays live in sampler slot `slot` (host-side pointer arithmetic only).
// UE / UR are set to their upper bounds; the kernels read the actual counts from counts_dev.
int kge_batch_from_slot(void *slots, size_t slot_bytes, int slot, int B, int C, int chunk, int N, int neg_head,
                        kge_batch *out) {
    if (!slots || !out || C * chunk != B) return KGE_ERR_ARG;
    const SlotLayout L = slot_layout(B, C * N);
    char *sb = (char *)slots + (size_t)slot * slot_bytes;
    memset(out, 0, sizeof(*out));
    out->B = B; out->C = C; out->chunk = chunk; out->N = N; out->neg_head = neg_head;
    out->U = 0; out->UE = 2 * B + C * N; out->UR = B;
    out->h_gid = (const int64_t *)(sb + L.h_gid); out->t_gid = (const int64_t *)(sb + L.t_gid);
    out->rel_ids = (const int64_t *)(sb + L.rel_ids); out->neg_ids = (const int64_t *)(sb + L.neg_ids);
    out->edge_w = nullptr;
    out->ue_id = (const int64_t *)(sb + L.ue_id);
    out->ue_pos_ptr = (const int32_t *)(sb + L.ue_pos_ptr); out->ue_pos_adj = (const int32_t *)(sb + L.ue_pos_adj);
    out->ue_neg_ptr = (const int32_t *)(sb + L.ue_neg_ptr); out->ue_neg_slot = (const int32_t *)(sb + L.ue_neg_slot);
    out->ur_id = (const int64_t *)(sb + L.ur_id);
    out->ur_ptr = (const int32_t *)(sb + L.ur_ptr); out->ur_edge = (const int32_t *)(sb + L.ur_edge);
    out->ue_rec = (const int32_t *)(sb + L.ue_rec); out->ur_rec = (const int32_t *)(sb + L.ur_rec);
    out->counts_dev = (const int32_t *)(sb + L.counts);
    return KGE_OK;
}

}  // extern "C"
