"""examples/dlrm_main.py's evaluation meter (the torchmetrics AUROC / Accuracy pair of recsys/dlrm_main.py:303-304,
written out in stock torch) against scikit-learn on the CPU."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "examples"))


def _dm():
    import importlib
    return importlib.import_module("dlrm_main")


@pytest.mark.parametrize("n,levels", [(1, None), (2, None), (1000, None), (5000, 7), (4096, 2)])
def test_auroc_and_accuracy_equal_sklearn(n, levels):
    from sklearn.metrics import accuracy_score, roc_auc_score
    dm = _dm()
    rng = np.random.default_rng(n + (levels or 0))
    labels = rng.integers(0, 2, n).astype(np.int32)
    if n >= 2:
        labels[0], labels[1] = 0, 1
    scores = rng.random(n).astype(np.float32) * 0.6 + labels * 0.25
    if levels:                                                  # heavy ties: a handful of distinct scores
        scores = (np.floor(scores * levels) / levels).astype(np.float32)
    meter = dm.BinaryMetrics()
    for lo in range(0, n, 300):                                 # accumulated over batches, computed once
        meter(torch.from_numpy(scores[lo:lo + 300]), torch.from_numpy(labels[lo:lo + 300]))
    auroc, acc = meter.compute()
    if n < 2:
        assert np.isnan(auroc)                                  # one class only
    else:
        assert abs(auroc - roc_auc_score(labels, scores)) < 1e-12
    assert abs(acc - accuracy_score(labels, scores >= 0.5)) < 1e-12


def test_auroc_corner_cases():
    dm = _dm()
    m = dm.BinaryMetrics()
    assert all(np.isnan(v) for v in m.compute())                # nothing accumulated
    p = torch.tensor([0.1, 0.2, 0.8, 0.9])
    assert dm.BinaryMetrics.auroc(p, torch.tensor([0, 0, 1, 1])) == 1.0
    assert dm.BinaryMetrics.auroc(p, torch.tensor([1, 1, 0, 0])) == 0.0
    assert dm.BinaryMetrics.auroc(torch.full((6,), 0.5), torch.tensor([0, 1, 0, 1, 1, 0])) == 0.5   # all tied
    assert np.isnan(dm.BinaryMetrics.auroc(p, torch.ones(4, dtype=torch.int32)))
