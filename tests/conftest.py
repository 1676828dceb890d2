import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def demo():
    """ibrm(T1 ~ 1) slice of the reference's inst/extdata demo (R/bayes.r:80-97): genotypes of the
    300 individuals that have both a genotype and a T1 record."""
    import numpy as np
    import hibayes_amd as H

    d = os.path.join(ROOT, "tests", "golden", "demo", "demo")
    pl = H.read_plink(d)
    phe = H.read_table(d + ".phe")
    ids = [r[1] for r in pl["fam"]]
    pos = {v: i for i, v in enumerate(phe["id"])}
    rows = [i for i, v in enumerate(ids) if v in pos and phe["T1"][pos[v]] is not None]
    y = np.array([float(phe["T1"][pos[ids[i]]]) for i in rows])
    return {"y": y, "M": np.asfortranarray(pl["geno"][rows, :]), "rows": rows, "plink": pl, "phe": phe,
            "ids": ids, "prefix": d}
