"""CPU tests of the host logic: the batch plan, the C-ABI surface of libkge_hip.so (symbols only -
no compute without a GPU), the torch-CPU port used as cpu_baseline, the no-CPU-fallback rule."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from golden_util import golden_names, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from dglke_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "kge_hip.h")).read()
    declared = set(re.findall(r"\b(kge_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(h, name), "libkge_hip.so does not export " + name
    assert declared == set(_lib.EXPORTED_SYMBOLS), (declared ^ set(_lib.EXPORTED_SYMBOLS))
    assert _lib.lib().kge_abi_version() == _lib.KGE_ABI_VERSION == 8


def test_argument_errors_are_reported_without_a_gpu():
    from dglke_amd import _lib
    h = _lib.lib()
    rc = h.kge_gather_rows(None, 0, 4, None, 0, None, None)
    assert rc == -1 and b"kge_gather_rows" in h.kge_last_error()
    rc = h.kge_score_pos(3, 1, 1, 1, 2, 6, 4, 1.0, 1.0, 1, None)   # ComplEx with d_r != d_e
    assert rc == -1 and b"ComplEx" in h.kge_last_error()
    rc = h.kge_score_pos(9, 1, 1, 1, 2, 4, 4, 1.0, 1.0, 1, None)
    assert rc == -1 and b"unknown model" in h.kge_last_error()


def test_product_has_no_cpu_fallback():
    from dglke_amd import _lib, ops
    from dglke_amd.engine import StepEngine
    with pytest.raises(_lib.KgeError):
        ops.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int64))
    with pytest.raises(_lib.KgeError):
        StepEngine("TransE_l2", 10, 2, 8, 12.0, 0.1, "cpu")
    # nothing in the product imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dgl-ke_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def _check_plan(p):
    B, C, N = p["B"], p["C"], p["N"]
    ue = p["ue_id"]
    assert np.all(np.diff(ue) > 0)
    assert set(ue) == set(p["nid"]) | set(p["neg_ids"])
    # every edge end appears exactly once, under the right entity
    seen = np.zeros(2 * B, bool)
    for u in range(p["UE"]):
        for k in range(p["ue_pos_ptr"][u], p["ue_pos_ptr"][u + 1]):
            code = p["ue_pos_adj"][k]
            e, side = code >> 1, code & 1
            assert (p["t_gid"][e] if side else p["h_gid"][e]) == ue[u]
            assert not seen[code]
            seen[code] = True
        slots = p["ue_neg_slot"][p["ue_neg_ptr"][u]:p["ue_neg_ptr"][u + 1]]
        assert np.all(p["neg_ids"][slots] == ue[u]) and np.all(np.diff(slots) > 0)
    assert seen.all() and p["ue_neg_ptr"][-1] == C * N
    seen_e = np.zeros(B, bool)
    for u in range(p["UR"]):
        es = p["ur_edge"][p["ur_ptr"][u]:p["ur_ptr"][u + 1]]
        assert np.all(p["rel_ids"][es] == p["ur_id"][u]) and np.all(np.diff(es) > 0)
        seen_e[es] = True
    assert seen_e.all()
    assert np.array_equal(p["nid"][p["h_local"]], p["h_gid"])
    assert np.array_equal(p["nid"][p["t_local"]], p["t_gid"])


@pytest.mark.parametrize("n_ent,n_rel,B,N,chunk", [(50, 3, 16, 4, 4), (5, 1, 12, 3, 6), (14951, 1345, 1000, 200, 200)])
def test_plan_groups_every_row_once(n_ent, n_rel, B, N, chunk):
    from dglke_amd import plan
    rng = np.random.RandomState(0)
    h, t = rng.randint(0, n_ent, B), rng.randint(0, n_ent, B)
    r = rng.randint(0, n_rel, B)
    neg = rng.randint(0, n_ent, (B // chunk) * N)
    p = plan.build_plan(h, t, r, neg, chunk, N, True)
    _check_plan(p)
    with pytest.raises(ValueError):
        plan.build_plan(h[:-1], t[:-1], r[:-1], neg, chunk, N, True)     # ragged batch is rejected
    with pytest.raises(ValueError):
        plan.build_plan(h, t, r, neg[:-1], chunk, N, True)


def test_sampler_alternates_corruption_side_and_covers_epochs():
    """dataloader/sampler.py:853-859: step 1 corrupts tails, step 2 heads, ..."""
    from dglke_amd.dataloader import UniformChunkedSampler
    rng = np.random.RandomState(0)
    n = 100
    s = UniformChunkedSampler(rng.randint(0, 30, n), rng.randint(0, 4, n), rng.randint(0, 30, n), 30,
                              batch_size=16, neg_sample_size=4, device="cpu", seed=1)
    plans = s.next_plans(12)
    assert [p["neg_head"] for p in plans[:4]] == [0, 1, 0, 1]
    assert all(p["B"] == 16 and p["C"] == 4 and p["neg_ids"].shape[0] == 16 for p in plans)
    for p in plans:
        _check_plan(p)


@pytest.mark.parametrize("name", [n for n in golden_names(transr=False) if not any(
    k in n for k in ("logistic", "hinge", "bce", "impts"))])
def test_torch_port_matches_reference(name):
    """the cpu_baseline port reproduces the reference step (same torch ops)."""
    from oracle import torch_port
    z, case = load_golden(name)
    m = torch_port.TorchPort(case["model"], case["n_ent"], case["n_rel"], case["hidden"], case["gamma"],
                             case["lr"], case["de"], case["dr"], case["adv"], case["adv_temp"],
                             case["reg_coef"], case["reg_norm"])
    m.ent = torch.from_numpy(z["init_entity"].copy())
    m.rel = torch.from_numpy(z["init_relation"].copy())
    torch.set_num_threads(1)
    for s in range(1, case["steps"] + 1):
        p = "s%d_" % s
        pl = dict(nid=z[p + "nid"], rel_ids=z[p + "r"], neg_ids=z[p + "neg"], h_local=z[p + "h_local"],
                  t_local=z[p + "t_local"], C=case["B"] // case["chunk"], chunk=case["chunk"],
                  N=case["N"], neg_head=int(z[p + "neg_head"]))
        out = m.step(pl)
        np.testing.assert_allclose(out["neg_score"], z[p + "neg_score"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out["g_neg"], z[p + "g_neg"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(out["log"][2], z[p + "log"][2], rtol=1e-5)
    np.testing.assert_allclose(m.ent.numpy(), z["final_entity"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.rel.numpy(), z["final_relation"], rtol=1e-4, atol=1e-5)


def test_dist_make_comm_without_gpu_is_the_c10d_wrapper_and_stream_override_nests():
    """dist.make_comm(): no GPU (or KGE_DIST_COMM=torch) -> the torch.distributed wrappers; _lib.use_stream() returns the
    previous override so that sections nest"""
    from dglke_amd import dist as kd, _lib
    import torch
    if not torch.cuda.is_available():
        assert isinstance(kd.make_comm(), kd.TorchComm)
    assert isinstance(kd.make_comm(kind="torch"), kd.TorchComm)
    assert _lib.use_stream(123) is None
    assert _lib.stream_ptr() == 123
    assert _lib.use_stream(456) == 123 and _lib.stream_ptr() == 456
    assert _lib.use_stream(None) == 456
    assert kd.default_cap(3000, 8, 1.5) % 64 == 0 and kd.default_cap(3000, 8, 1.5) >= 3000 * 1.5 / 8


def test_workspace_accounts_for_the_gradient_parts_of_the_shared_pair_backward():
    """kge_step_workspace_bytes (host arithmetic only): RotatE / TransE_l1 reserve room for GA in parts (the shared-pair backward
    splits a chunk's negatives over workgroups, kge_neg_bcast.hip neg_bwd_lc_splits) - exactly (parts - 1) more [B, d_e] blocks
    than with the two-pass kernels, a whole number of them and at most 7; the matrix-core models reserve none"""
    import ctypes as C
    from dglke_amd import _lib
    L = _lib.lib()

    def ws(model, d_e, d_r, B, Cn, chunk, N, flags):
        hp = _lib.KgeHParams()
        hp.model, hp.d_e, hp.d_r, hp.flags = _lib.model_id(model), d_e, d_r, flags
        hp.gamma, hp.lr, hp.adv_temp, hp.reg_norm = 12.0, 0.1, 1.0, 3
        return L.kge_step_workspace_bytes(C.byref(hp), B, Cn, chunk, N, 2 * B + Cn * N, B)
    for model, d_e, d_r, B, Cn, chunk, N in (("RotatE", 400, 200, 1024, 4, 256, 256), ("RotatE", 800, 400, 1024, 4, 256, 256),
                                           ("TransE_l1", 400, 400, 1000, 5, 200, 200)):
        full, two_pass = ws(model, d_e, d_r, B, Cn, chunk, N, 0), ws(model, d_e, d_r, B, Cn, chunk, N, _lib.FLAG_TWO_PASS_PAIR)
        extra = full - two_pass
        block = B * d_e * 4
        assert extra > 0 and extra % block == 0 and 1 <= extra // block <= 7, (model, extra, block)
    for model in ("TransE_l2", "DistMult", "ComplEx"):
        d_e = 400
        assert ws(model, d_e, d_e if model != "ComplEx" else d_e, 1000, 5, 200, 200, 0) == \
               ws(model, d_e, d_e if model != "ComplEx" else d_e, 1000, 5, 200, 200, _lib.FLAG_TWO_PASS_PAIR)


def test_inline_assembly_wide_stores_carry_their_wait_states():
    """a 128-bit global store written as inline assembly must be followed by two wait states before its data registers may be
    overwritten (gfx940+ VMEM store-data hazard) - the compiler does not see inside the asm statement, so the statement itself
    carries the s_nop (kge_common.hpp st_wt<4>; found in round 3 when three such stores ran back to back)"""
    import re
    src_dir = os.path.join(ROOT, "dgl-ke_amd", "csrc")
    n = 0
    for f in os.listdir(src_dir):
        if not f.endswith((".hip", ".hpp")):
            continue
        for m in re.finditer(r'asm\s+volatile\(\s*"((?:[^"\\]|\\.)*)"', open(os.path.join(src_dir, f)).read()):
            text = m.group(1)
            if re.search(r"(global|flat|buffer)_store_dwordx[34]", text):
                n += 1
                assert "s_nop" in text, "%s: wide store without wait states: %s" % (f, text)
    assert n >= 1


def _lc_balanced_map(ncol, ngr, P, nB):
    """Python statement of the block -> (column, part, groups) map of the shared-pair backward's balanced split
    (dgl-ke_amd/csrc/kge_neg_bcast.hip, neg_bwd_lc_kernel, NegArgs::lc_P): 1024 block ids, rounds of 256, odd rounds reversed,
    classes in descending part size."""
    nA = ncol - nB
    remB, remA = ngr % P, ngr % (P + 1)
    n1, n2, n3 = nB * remB, nB * (P - remB), nA * remA
    out = []
    for b in range(1024):
        k, rnd = b & 255, b >> 8
        t = (rnd << 8) + ((255 - k) if (rnd & 1) else k)
        if t < n1:
            col, j, nsp = t // remB, t % remB, P
        elif t < n1 + n2:
            t -= n1
            col, j, nsp = t // (P - remB), remB + t % (P - remB), P
        elif t < n1 + n2 + n3:
            t -= n1 + n2
            col, j, nsp = nB + t // remA, t % remA, P + 1
        else:
            t -= n1 + n2 + n3
            col, j, nsp = nB + t // (P + 1 - remA), remA + t % (P + 1 - remA), P + 1
        base, rem = ngr // nsp, ngr % nsp
        g_lo = j * base + min(j, rem)
        out.append((col, j, nsp, g_lo, g_lo + base + (1 if j < rem else 0)))
    return out


def test_balanced_split_map_covers_every_group_once_and_balances_the_cus():
    """every column's quad groups are covered exactly once by its P or P + 1 parts, and the four block ids that share a CU
    (k, k + 256, k + 512, k + 768) carry almost equal work - the property the dealing order exists for"""
    for ncol, ngr in [(224, 16), (416, 16), (208, 16), (320, 8), (192, 16), (832, 16), (325, 7), (513, 9), (1023, 4)]:
        P = 1024 // ncol
        nA = 1024 - ncol * P
        if nA == 0:
            continue
        nB = ncol - nA
        m = _lc_balanced_map(ncol, ngr, P, nB)
        cover = {}
        for col, j, nsp, lo, hi in m:
            assert 0 <= col < ncol and 0 <= j < nsp and nsp == (P if col < nB else P + 1)
            assert 0 <= lo <= hi <= ngr
            cover.setdefault(col, []).append((lo, hi, j))
        assert len(cover) == ncol
        for col, parts in cover.items():
            parts.sort()
            assert len(parts) == (P if col < nB else P + 1) and sorted(p[2] for p in parts) == list(range(len(parts)))
            assert parts[0][0] == 0 and parts[-1][1] == ngr
            assert all(parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))
        per_cu = [sum(m[k + 256 * r][4] - m[k + 256 * r][3] for r in range(4)) for k in range(256)]
        sizes = sorted(hi - lo for _, _, _, lo, hi in m)
        assert max(per_cu) - min(per_cu) <= 2 * (sizes[-1] - sizes[0]), (ncol, ngr, min(per_cu), max(per_cu))


def test_device_filter_lists_equal_the_host_lists():
    """eval.build_filter_device (one composite key through torch.unique - runs on any torch device, here the CPU) returns the lists
    of eval.build_filter (np.lexsort): unique (key, entity) pairs in the same order, the same [left, right) range per test triple -
    duplicates among the known triples, test triples without any known match, both corruption sides; None when the composite
    key would not fit int64."""
    import numpy as np
    import torch
    from dglke_amd import eval as E
    rng = np.random.RandomState(3)
    n_ent, n_rel = 97, 7
    known = tuple(rng.randint(0, n, 4000) for n in (n_ent, n_rel, n_ent))           # many duplicates
    test = (np.concatenate([known[0][:300], rng.randint(0, n_ent, 50)]), np.concatenate([known[1][:300], rng.randint(0, n_rel, 50)]),
            np.concatenate([known[2][:300], rng.randint(0, n_ent, 50)]))
    for neg_head in (True, False):
        want = E.build_filter(known[0], known[1], known[2], test[0], test[1], test[2], neg_head, n_rel)
        got = E.build_filter_device(known, test, neg_head, n_rel, n_ent, torch.device("cpu"))
        assert got[0].dtype == torch.int64 and got[1].dtype == torch.int64
        assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1])
    # Freebase-sized id spaces (the composite key would overflow): two stable sorts give the same lists
    E._FORCE_TWO_KEY_SORT = True
    try:
        for neg_head in (True, False):
            want = E.build_filter(known[0], known[1], known[2], test[0], test[1], test[2], neg_head, n_rel)
            got = E.build_filter_device(known, test, neg_head, n_rel, n_ent, torch.device("cpu"))
            assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1])
    finally:
        E._FORCE_TWO_KEY_SORT = False
    big = E.build_filter_device(tuple(np.asarray(x[:50]) for x in known), tuple(np.asarray(x[:5]) for x in test), True, 14824, 86054151,
                                torch.device("cpu"))
    assert big is not None and big[0].shape == (5, 2)
    assert E.build_filter_device(known, test, True, 1 << 40, 1 << 30, torch.device("cpu")) is None
