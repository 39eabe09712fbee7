import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's intra-op pool: the GPU hosts expose 128+ hardware threads, and torch's default of one worker per thread makes the oracle's
    # many small operators pay the pool's wake-up -- measured on such a host: the same five oracle-heavy tests 174 s at the default, 29 s with 32
    # threads, 23 s with 8.  (Affects test infrastructure only: the product path runs no CPU arithmetic.)
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), 8))
    except Exception:
        pass


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _knobs_do_not_leak():
    """module-level A/B switches must leave every test the way they entered it (a test that wants another route uses
    monkeypatch.setattr): a leaked switch silently changes which kernels the REST of the suite exercises"""
    import importlib
    mods = {}
    for name, attrs in (("ap_adapter_amd.processors", ("USE_FUSED_XATTN", "USE_XATTN_ROWS")), ("ap_adapter_amd.ops", ("XATTN_MAXL", "MLP_C", "XROWS_C", "RP_K", "FUSED_DTYPES", "HS_ATTN", "HS_FF", "HS_FF2", "SATTN_FUSED", "MLP_PACKED", "MLP_PACKED_MIN_M", "GEGLU_PACKED_MIN_M", "HCONV")),
                        ("ap_adapter_amd.unet", ("CFG_SHARED_PREFIX", "NO_CAT"))):
        try:
            m = importlib.import_module(name)
        except Exception:  # the CPU suite may run without the built library
            continue
        mods[m] = {a: getattr(m, a) for a in attrs if hasattr(m, a)}
    yield
    for m, saved in mods.items():
        for a, v in saved.items():
            assert getattr(m, a) == v, f"{m.__name__}.{a} leaked out of a test: {getattr(m, a)!r} (was {v!r})"
