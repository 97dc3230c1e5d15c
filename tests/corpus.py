"""Loader of tests/golden/corpus/ (made by scripts/gen_corpus_fixture.py): every LP of the reference's Netlib directory as the
standardised `MatrixData` the pivot engine is handed, with the objective's fixed part, the reference's pin where it holds one
and the optimum HiGHS finds for the same standardised LP (an independent check, not the reference: "parity unpinned")."""
import json
import os

import numpy as np

from rust_lp_amd import MatrixData

CORPUS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus")


def index():
    with open(os.path.join(CORPUS, "index.json")) as f:
        return {r["name"]: r for r in json.load(f)}


def load(name):
    """(MatrixData, fixed cost)"""
    z = np.load(os.path.join(CORPUS, name + ".npz"))
    nn, ne, nr, nl, ng = (int(v) for v in z["counts"])
    md = MatrixData(nr_normal=nn, nr_eq=ne, nr_range=nr, nr_le=nl, nr_ge=ng, b=z["b"], cost=z["cost"], upper_bound=z["upper_bound"],
                    ranges=z["ranges"], col_ptr=z["col_ptr"], row_idx=z["row_idx"], values=z["values"])
    return md, float(z["fixed_cost"][0])
