#!/usr/bin/env python3
"""Stage the reference's own hot-path modules into oracle/_ref so that the GPU box can TIME THE REFERENCE ITSELF.

*** TEST / MEASUREMENT INFRASTRUCTURE ONLY. ***  Recipe, not sources: the six files SURVEY.md 8(a) names are COMPILED at build()
time (py_compile: CPython bytecode, the only "build" a pure-Python reference has) from where they lie under /root/reference into
oracle/_ref/ - binaries only, like a C reference's .so: no reference source text is written anywhere in this repository, and the
directory is git-ignored (but NOT gpurun-ignored: it travels to the GPU box next to the built libkge_hip.so; same image, same
interpreter).  /root/reference does not exist on the GPU box; there `stage()` is a no-op and `available()` says whether an earlier
build() staged the files.

The reference is pure Python on torch + DGL: no native code to compile.  The package skeleton around the six files (the `__init__.py`
files and the two modules train_pytorch.py imports but the step never calls - `dglke.utils`, `dglke.dataloader`) is written
here as stubs; DGL itself is replaced by oracle/ref_stub.py.  bench.py's `cpu_baseline` then reports `kind: "reference"`
(oracle/ref_baseline.py); without the staged files it falls back to the torch-CPU port (`kind: "port"`).
"""
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "_ref")
REF = os.environ.get("DGLKE_REFERENCE", "/root/reference/python")

# the reference files on the hot path (relative to python/dglke/)
FILES = [
    "models/general_models.py",          # KEModel.forward / update / predict_neg_score
    "models/base_loss.py",               # loss genres
    "models/pytorch/score_fun.py",       # edge_func / create_neg of the eight score functions
    "models/pytorch/loss.py",            # LossGenerator
    "models/pytorch/tensor_models.py",   # ExternalEmbedding (gather, trace, row-sparse Adagrad)
    "train_pytorch.py",                  # the step loop (train(): forward -> backward -> update with its four timers)
]

SKELETON = {
    "__init__.py": "# staged by oracle/make_ref.py (the reference's own __init__ reads pkg_resources metadata)\n",
    "models/__init__.py": "# staged by oracle/make_ref.py: `from .general_models import KEModel` as the reference's does\n"
                          "from .general_models import KEModel\n",
    "models/pytorch/__init__.py": "",
    # train_pytorch.py imports these at module level; the training step never calls them
    "utils.py": "def save_model(*a, **k):\n    raise RuntimeError('stub (oracle/make_ref.py)')\n\n\n"
                "def get_compatible_batch_size(batch_size, neg_sample_size):\n    return batch_size\n",
    "dataloader/__init__.py": "class EvalDataset(object):\n    pass\n\n\n"
                              "def get_dataset(*a, **k):\n    raise RuntimeError('stub (oracle/make_ref.py)')\n",
}


def _pyc(f):
    return os.path.join(DST, "dglke", f[:-3] + ".pyc")          # sourceless module: <name>.pyc next to the package's __init__.py


def available():
    return all(os.path.exists(_pyc(f)) for f in FILES)


def stage(force=False):
    """compile the six files (when the reference is present: the build container) and write the skeleton; returns True if oracle/_ref
    is usable afterwards."""
    src_root = os.path.join(REF, "dglke")
    if not os.path.isdir(src_root):
        return available()                     # GPU box: use what an earlier build() staged, or nothing
    if available() and not force:
        fresh = all(os.path.getmtime(_pyc(f)) >= os.path.getmtime(os.path.join(src_root, f)) for f in FILES)
        if fresh:
            return True
    for f in FILES:
        d = _pyc(f)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        stale_src = os.path.join(DST, "dglke", f)               # (an earlier version of this recipe staged the .py itself)
        if os.path.exists(stale_src):
            os.remove(stale_src)
        py_compile.compile(os.path.join(src_root, f), cfile=d, dfile="dglke/" + f, doraise=True)
    for f, text in SKELETON.items():
        d = os.path.join(DST, "dglke", f)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        with open(d, "w") as fh:
            fh.write(text)
    return available()


if __name__ == "__main__":
    ok = stage(force="--force" in sys.argv)
    print("oracle/_ref:", "staged" if ok else "not available (no reference here)")
