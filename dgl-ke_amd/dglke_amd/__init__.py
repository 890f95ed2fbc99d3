"""dglke_amd - MI355X (gfx950) implementation of the DGL-KE training hot path behind the
reference's KEModel / score_func / ExternalEmbedding plugin surface.  All arithmetic is in
libkge_hip.so (dgl-ke_amd/csrc, C ABI in include/kge_hip.h)."""
from ._lib import KgeError, LIB_PATH, lib  # noqa: F401

__version__ = "0.1.0"
