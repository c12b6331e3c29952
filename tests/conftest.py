import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
import timetuning_amd  # noqa: E402,F401  (before the first HIP call - torch.cuda.is_available() below: it switches ROCm's hipGraph packet capture off)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests never run by accident on the CPU box (and fail loudly, not skip, on a GPU box
    whose HIP library is missing)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def rel_err(a, b):
    """max |a-b| / max |b| - the "relative fp32" measure used throughout (north_star: <= 1e-3)."""
    # (features returned by FeatureExtractor.forward carry grad, as in the reference: detach before converting)
    a = np.asarray(a.detach().cpu() if hasattr(a, "detach") else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if hasattr(b, "detach") else b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 - the twin of ``rel_err`` that a single large element of b cannot flatter (assertions on
    embeddings / logits use both)."""
    a = np.asarray(a.detach().cpu() if hasattr(a, "detach") else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if hasattr(b, "detach") else b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def tail_err(a, b, frac=0.1):
    """Relative L2 error restricted to the ``frac`` smallest-magnitude elements of b: what a max-normalised bound hides."""
    a = np.asarray(a.detach().cpu() if hasattr(a, "detach") else a, dtype=np.float64).ravel()
    b = np.asarray(b.detach().cpu() if hasattr(b, "detach") else b, dtype=np.float64).ravel()
    idx = np.argsort(np.abs(b))[: max(1, int(frac * b.size))]
    return float(np.linalg.norm(a[idx] - b[idx]) / max(np.linalg.norm(b[idx]), 1e-30))


@pytest.fixture(params=["f32", "f16x3", "bf16x6"])
def accurate_precision(request):
    """The arithmetic modes that must meet the fp32 contract: exact f32 MFMA, and the fp32-accurate split modes - "f16x3" (fp16
    pairs, three products; round 4) and "bf16x6" (three bf16 planes, six products).  Tests that take this fixture run at their
    UNCHANGED fp32 tolerances in all of them."""
    from timetuning_amd import hip_ops

    hip_ops.set_gemm_precision(request.param)
    keep, hip_ops.PAIRS_MIN_ROWS = hip_ops.PAIRS_MIN_ROWS, 0   # the pair kernels at EVERY size: the tiny golden models would otherwise dispatch to f32
    yield request.param
    hip_ops.PAIRS_MIN_ROWS = keep
    hip_ops.set_gemm_precision("f32")
