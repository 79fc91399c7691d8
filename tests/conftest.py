import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference, when oracle/_ref/libgms_ref.so exists (built here from /root/reference; travels prebuilt)."""
    from oracle import bindings
    if not bindings.have_ref():
        pytest.skip("oracle/_ref/libgms_ref.so not built")
    try:
        return bindings.Reference()
    except OSError as e:  # e.g. an ISA mismatch on a different host
        pytest.skip(f"compiled reference not loadable here: {e}")


@pytest.fixture(scope="session")
def capi():
    from gms_amd import capi as c
    c.lib()  # raises if libgmsx.so has not been built: there is no fallback
    return c


@pytest.fixture(scope="session")
def gpu(capi):
    capi.init(0)  # raises GmsxError(-6) without a GPU
    return capi


_GRAPH_CACHE = {}


def host_graph(capi, generator, scale, degree=16, relabel=True):
    key = (generator, scale, degree, relabel)
    if key not in _GRAPH_CACHE:
        if generator == "rmat":  # the com-Orkut-shaped family of BASELINE configs[3]: own generator, A=.45 B=C=.22 (graphs.json "rmat-*" records)
            _GRAPH_CACHE[key] = capi.HostCSR.generate_rmat(scale, degree, 0.45, 0.22, 0.22, capi.RELABEL_AUTO if relabel else capi.RELABEL_NEVER)
        else:
            _GRAPH_CACHE[key] = capi.HostCSR.generate(generator, scale, degree, capi.RELABEL_AUTO if relabel else capi.RELABEL_NEVER)
    return _GRAPH_CACHE[key]


def edges_to_csr(capi, edges, n=-1):
    e = np.asarray(edges, dtype=np.int32).reshape(-1, 2)
    return capi.HostCSR.from_edges(e[:, 0], e[:, 1], num_nodes=n)
