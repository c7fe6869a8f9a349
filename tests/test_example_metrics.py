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


def test_evaluate_loop_on_the_cpu_with_a_stand_in_model(capsys):
    """examples/dlrm_main.py::_evaluate (recsys/dlrm_main.py:300-333) end to end on the CPU: eval mode, no gradients,
    every batch of the loader once, the set's AUROC / accuracy printed under the reference's wording and returned."""
    from sklearn.metrics import accuracy_score, roc_auc_score
    dm = _dm()

    class Model(torch.nn.Module):                       # logits = a fixed function of the batch: dense[:, 0] + parity of an id
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(3.0))
            self.calls = []

        def forward(self, dense, sparse):
            self.calls.append((self.training, torch.is_grad_enabled()))
            values, offsets, stride = sparse
            return (self.w * (dense[:, 0] - 0.5) + 0.3 * (values[:stride] % 2).float()).unsqueeze(-1)

    g = torch.Generator().manual_seed(3)
    B = 50
    loader = []
    for _ in range(7):
        dense = torch.rand(B, 4, generator=g)
        values = torch.randint(0, 1000, (3 * B,), generator=g)
        labels = ((dense[:, 0] + 0.2 * torch.rand(B, generator=g)) > 0.6).float()
        loader.append(dict(dense=dense, sparse=[values, torch.arange(3 * B + 1, dtype=torch.int32), B], labels=labels))
    args = dm.parse_args(["--use_cache", "--eval_acc"])                # use_overlap off: a plain iterator, no side stream
    model = Model().train()
    auroc, acc = dm._evaluate(model, loader, "val", args, torch.device("cpu"), 0, 1)
    assert dm._evaluate.batches == 7 and len(model.calls) == 7
    assert all(c == (False, False) for c in model.calls)               # eval mode, gradients off, for every batch
    with torch.no_grad():
        preds = torch.cat([torch.sigmoid(model(b["dense"], b["sparse"]).squeeze(-1)) for b in loader]).numpy()
    labels = torch.cat([b["labels"] for b in loader]).numpy().astype(int)
    assert abs(auroc - roc_auc_score(labels, preds)) < 1e-12 and abs(acc - accuracy_score(labels, preds >= 0.5)) < 1e-12
    out = capsys.readouterr().out
    assert f"AUROC over val set: {auroc}" in out and f"Accuracy over val set: {acc}" in out
