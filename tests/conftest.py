import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mesh_ops():
    from cape_amd.load_data import load_graph_mtx, load_pack
    L, D, U, p, L_d, D_d, U_d = load_graph_mtx(None, load_for_demo=True)
    return dict(L=L, D=D, U=U, p=p, L_d=L_d, D_d=D_d, U_d=U_d, pack=load_pack())
