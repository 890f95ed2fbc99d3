"""On-disk knowledge-graph formats of the reference (dataloader/KGDataset.py:73-145 `KGDataset`,
:186-330 built-in FB15k / FB15k-237 / wn18 / wn18rr layouts, :497-620 `raw_udd_*`, :622-737 `udd_*`,
:739-780 `get_dataset`) so that `--data_path/--dataset/--format/--data_files/--delimiter` mean the same.

Differences: there is no network here, so a built-in dataset must already be unpacked under
`<data_path>/<name>/` (the reference downloads it); files are parsed with pandas instead of
per-line Python loops (FB15k loads in well under a second)."""
import os
import re

import numpy as np
import pandas as pd

BUILT_IN = ("FB15k", "FB15k-237", "wn18", "wn18rr")


def _order(fmt):
    """'raw_udd_hrt' / 'udd_trh' ... -> column index of (head, relation, tail) (KGDataset.py:43-65)."""
    key = fmt.split("_")[-1]
    if sorted(key) != ["h", "r", "t"]:
        raise ValueError("unsupported triple format %r: expected a permutation of h, r, t" % fmt)
    return [key.index("h"), key.index("r"), key.index("t")]


def _read_table(path, delimiter, ncols=None):
    # a multi-character separator goes to the python engine, which treats it as a REGEX: escape it so that
    # '||' or '.'-style delimiters split literally (like str.split in the reference's readers)
    sep = delimiter if len(delimiter) == 1 else re.escape(delimiter)
    df = pd.read_csv(path, sep=sep, header=None, dtype=str, keep_default_na=False, quoting=3,
                     engine="c" if len(delimiter) == 1 else "python")
    if ncols is not None and df.shape[1] < ncols:
        raise ValueError("%s: expected at least %d columns separated by %r" % (path, ncols, delimiter))
    return df


class KGDataset(object):
    """n_entities, n_relations, train/valid/test = (heads, rels, tails[, importance]) int64 arrays,
    entity2id / relation2id (None for id-based formats), emap_fname / rmap_fname."""

    def __init__(self):
        self.entity2id = self.relation2id = None
        self.n_entities = self.n_relations = 0
        self.train = self.valid = self.test = None
        self.emap_fname = self.rmap_fname = None
        self.has_edge_importance = False
        self.name = None

    def _triples(self, path, order, delimiter, by_name, what):
        if path is None:
            return None
        print('Reading {} triples....'.format(what))
        want_w = self.has_edge_importance and what == "train"
        df = None
        if not by_name and len(delimiter) == 1:
            # id triples: parsed as integers straight away (8 bytes per id instead of a Python string each - Freebase's 338 M
            # training triples are 8 GB this way and would not fit as strings); anything that is not a plain integer column
            # falls through to the string path below, which reports it like the reference does
            try:
                dt = {0: np.int64, 1: np.int64, 2: np.int64}
                if want_w:
                    dt[3] = np.float32
                df = pd.read_csv(path, sep=delimiter, header=None, usecols=list(range(4 if want_w else 3)), dtype=dt,
                                 engine="c", quoting=3)
            except (ValueError, TypeError, OverflowError):
                df = None
        if df is None:
            df = _read_table(path, delimiter, 4 if want_w else 3)
        cols = [df[order[0]], df[order[1]], df[order[2]]]
        if by_name:
            h = cols[0].map(self.entity2id)
            r = cols[1].map(self.relation2id)
            t = cols[2].map(self.entity2id)
            if h.isna().any() or r.isna().any() or t.isna().any():
                raise KeyError("%s mentions an entity / relation that is not in the id maps" % path)
        else:
            try:
                h, r, t = (c.astype(np.int64) for c in cols)
            except ValueError:
                raise ValueError("For User Defined Dataset, both node ids and relation ids in the triplets "
                                 "should be int (%s)" % path)
        out = [np.asarray(x, np.int64) for x in (h, r, t)]
        if len(out[0]):
            if out[0].min() < 0 or out[2].min() < 0 or out[1].min() < 0:
                raise ValueError("negative id in " + path)
            if max(out[0].max(), out[2].max()) >= self.n_entities:
                raise ValueError("node id in %s exceeds the number of entities %d" % (path, self.n_entities))
            if out[1].max() >= self.n_relations:
                raise ValueError("relation id in %s exceeds the number of relations %d" % (path, self.n_relations))
        print('Finished. Read {} {} triples.'.format(len(out[0]), what))
        if self.has_edge_importance and what == "train":
            w = df[3].astype(np.float32).to_numpy()
            if len(w) and w.min() <= 0:
                raise ValueError("Edge importance score should > 0")
            out.append(w)
        return tuple(out)


def _built_in(path, name):
    d = os.path.join(path, name)
    need = ["entities.dict", "relations.dict", "train.txt", "valid.txt", "test.txt"]
    missing = [f for f in need if not os.path.exists(os.path.join(d, f))]
    if missing:
        raise FileNotFoundError("built-in dataset %s: %s not found under %s (no network here: unpack the "
                                "reference's %s.zip there)" % (name, ", ".join(missing), d, name))
    ds = KGDataset()
    ds.name = name
    e = _read_table(os.path.join(d, "entities.dict"), "\t", 2)          # 'id\tname'
    r = _read_table(os.path.join(d, "relations.dict"), "\t", 2)
    ds.entity2id = dict(zip(e[1], e[0].astype(np.int64)))
    ds.relation2id = dict(zip(r[1], r[0].astype(np.int64)))
    ds.n_entities, ds.n_relations = len(ds.entity2id), len(ds.relation2id)
    ds.train = ds._triples(os.path.join(d, "train.txt"), [0, 1, 2], "\t", True, "train")
    ds.valid = ds._triples(os.path.join(d, "valid.txt"), [0, 1, 2], "\t", True, "valid")
    ds.test = ds._triples(os.path.join(d, "test.txt"), [0, 1, 2], "\t", True, "test")
    ds.emap_fname, ds.rmap_fname = "entities.dict", "relations.dict"
    return ds


def _raw_udd(path, name, fmt, delimiter, files, has_edge_importance):
    """triples by NAME; the id maps are built here in order of first appearance over all files and
    written to entities.tsv / relations.tsv as 'id<delim>name' (KGDataset.py:586-609)."""
    if not files or len(files) not in (1, 3):
        raise ValueError("raw_udd_{htr} format requires 1 or 3 input files: train_file [valid_file test_file]")
    order = _order(fmt)
    ds = KGDataset()
    ds.name, ds.has_edge_importance = name, bool(has_edge_importance)
    ent, rel = {}, {}
    for f in files:
        df = _read_table(os.path.join(path, f), delimiter, 3)
        for hname, rname, tname in zip(df[order[0]], df[order[1]], df[order[2]]):
            if hname not in ent:
                ent[hname] = len(ent)
            if tname not in ent:
                ent[tname] = len(ent)
            if rname not in rel:
                rel[rname] = len(rel)
    with open(os.path.join(path, "entities.tsv"), "w") as f:
        f.writelines("%d%s%s\n" % (v, delimiter, k) for k, v in ent.items())
    with open(os.path.join(path, "relations.tsv"), "w") as f:
        f.writelines("%d%s%s\n" % (v, delimiter, k) for k, v in rel.items())
    ds.entity2id, ds.relation2id = ent, rel
    ds.n_entities, ds.n_relations = len(ent), len(rel)
    ds.train = ds._triples(os.path.join(path, files[0]), order, delimiter, True, "train")
    if len(files) == 3:
        ds.valid = ds._triples(os.path.join(path, files[1]), order, delimiter, True, "valid")
        ds.test = ds._triples(os.path.join(path, files[2]), order, delimiter, True, "test")
    ds.emap_fname, ds.rmap_fname = "entities.tsv", "relations.tsv"
    return ds


def _udd(path, name, fmt, delimiter, files, has_edge_importance):
    """triples by integer ID; entity / relation files only give the counts (KGDataset.py:672-684)."""
    if not files or len(files) not in (3, 5):
        raise ValueError("udd_{htr} format requires 3 or 5 input files: entity2id, relation2id, train_file "
                         "[valid_file test_file]")
    order = _order(fmt)
    ds = KGDataset()
    ds.name, ds.has_edge_importance = name, bool(has_edge_importance)
    with open(os.path.join(path, files[0])) as f:
        ds.n_entities = sum(1 for _ in f)
    with open(os.path.join(path, files[1])) as f:
        ds.n_relations = sum(1 for _ in f)
    ds.train = ds._triples(os.path.join(path, files[2]), order, delimiter, False, "train")
    if len(files) == 5:
        ds.valid = ds._triples(os.path.join(path, files[3]), order, delimiter, False, "valid")
        ds.test = ds._triples(os.path.join(path, files[4]), order, delimiter, False, "test")
    ds.emap_fname, ds.rmap_fname = files[0], files[1]
    return ds


def get_dataset(data_path, data_name, format_str, delimiter='\t', files=None, has_edge_importance=False):
    if format_str == 'built_in':
        if data_name not in BUILT_IN:
            raise ValueError("Unknown / unsupported built-in dataset %s (supported from local files: %s; Freebase, "
                             "wikikg2, biokg and wikikg90M need their own loaders)" % (data_name, ", ".join(BUILT_IN)))
        return _built_in(data_path, data_name)
    if format_str.startswith('raw_udd'):
        if data_name == 'FB15k':
            raise ValueError('You should provide the dataset name for raw_udd format.')
        return _raw_udd(data_path, data_name, format_str, delimiter, files, has_edge_importance)
    if format_str.startswith('udd'):
        if data_name == 'FB15k':
            raise ValueError('You should provide the dataset name for udd format.')
        return _udd(data_path, data_name, format_str, delimiter, files, has_edge_importance)
    raise ValueError("Unknown format {}".format(format_str))


def write_built_in_layout(path, name, n_entities, n_relations, train, valid, test):
    """write id triples in the built-in layout (entities.dict / relations.dict 'id\\tname', triples
    by name) - used by the synthetic-KG generators so that they feed the same loader as FB15k."""
    d = os.path.join(path, name)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "entities.dict"), "w") as f:
        f.writelines("%d\t/e/%d\n" % (i, i) for i in range(n_entities))
    with open(os.path.join(d, "relations.dict"), "w") as f:
        f.writelines("%d\t/r/%d\n" % (i, i) for i in range(n_relations))
    for fname, trip in (("train.txt", train), ("valid.txt", valid), ("test.txt", test)):
        h, r, t = (np.asarray(x) for x in trip)
        pd.DataFrame({"h": ["/e/%d" % x for x in h], "r": ["/r/%d" % x for x in r],
                      "t": ["/e/%d" % x for x in t]}).to_csv(os.path.join(d, fname), sep="\t", header=False, index=False)
    return d
