"""Helpers shared by the golden-vector tests (test infrastructure)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def async_golden_names():
    """--async_update goldens (gen_golden.py cases with async=True: the reference's own async_update body, one step late)"""
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "async_*.npz"))))


def golden_names(transr=None, nd=False):
    """training-step goldens (gen_golden.py).  transr: None = all, False = without the TransR cases (three
    tables: the generic two-table harnesses skip them), True = only TransR.  nd: False = without the
    --neg_deg_sample cases (nd_*), True = only those, None = both.  (The async_* goldens have their own list.)"""
    names = sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                   if not n.startswith("eval_") and not n.startswith("async_"))
    if nd is not None:
        names = [n for n in names if n.startswith("nd_") == bool(nd)]
    if transr is None:
        return names
    return [n for n in names if ("transr_" in n) == bool(transr)]          # (transr_*, nd_transr_*)


def eval_golden_names():
    """ranking-evaluation goldens (gen_golden_eval.py)"""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "eval_*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    case = json.loads(str(z["case_json"]))
    return z, case


def oracle_config(case):
    from oracle import kge_oracle as O
    return O.Config(case["model"], case["gamma"], case["hidden"], case["lr"], adv=case["adv"],
                    adv_temp=case["adv_temp"], reg_coef=case["reg_coef"],
                    reg_norm=case["reg_norm"], loss_genre=case.get("loss_genre", "Logsigmoid"),
                    pairwise=case.get("pairwise", False), margin=case.get("margin", 1.0),
                    double_ent=case["de"], double_rel=case["dr"], neg_deg=case.get("neg_deg", False))
