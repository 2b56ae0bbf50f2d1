"""Loaders for tests/golden (fixtures generated from the real reference by tools/make_golden.py)."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_png(path=None):
    from PIL import Image
    a = np.array(Image.open(path or os.path.join(G, "original.png")).convert("RGBA"))
    return np.ascontiguousarray(a).view(np.uint32).reshape(a.shape[0], a.shape[1])


def cases():
    idx = json.load(open(os.path.join(G, "cases.json")))
    z = np.load(os.path.join(G, "cases.npz"))
    return idx, z


def hashes():
    return json.load(open(os.path.join(G, "hashes.json")))


def chain():
    return json.load(open(os.path.join(G, "chain.json")))


def blocks():
    return np.load(os.path.join(G, "blocks.npz"))


def big_input(name, orc):
    """Rebuild the input of a hashes.json entry (generators are integer-defined; the input hash is checked by the caller)."""
    if name.startswith("original"):
        return load_png()
    if name.startswith("rga1024"):
        return orc.random_gradient(1024, 1024, 1, False)
    if name.startswith("rg1024"):
        return orc.random_gradient(1024, 1024, 1, True)
    if name.startswith("pn1024"):
        return orc.photo_noise(1024, 1024, 1)
    raise KeyError(name)
