"""egomcq_accuracy_metrics (reference model/metric.py:218-234) against a literal loop restatement of the reference lines."""
import torch

from egovlp_amd.model.metric import egomcq_accuracy_metrics


def reference_loop(preds, labels, types):          # model/metric.py:218-234, line by line
    metrics = {}
    type_list = torch.unique(types)
    group_list = ["Intra-video", "Inter-video"]
    for type_i, group_i in zip(type_list, group_list):
        correct = 0
        total = 0
        for pred, label, type in zip(preds, labels, types):
            if type == type_i:
                pred_ = torch.argmax(pred)
                if pred_.item() == label.item():
                    correct += 1
                total += 1
        metrics[group_i] = correct / total * 100
    return metrics


def test_egomcq_accuracy_matches_reference_loop():
    g = torch.Generator().manual_seed(3)
    for q in (1, 7, 200):
        preds = torch.randn(q, 5, generator=g)
        labels = torch.randint(0, 5, (q,), generator=g)
        types = torch.randint(1, 3, (q,), generator=g)
        labels[: q // 2] = preds[: q // 2].argmax(1)          # make about half of them hits
        got, want = egomcq_accuracy_metrics(preds, labels, types), reference_loop(preds, labels, types)
        assert got.keys() == want.keys()
        for k in want:
            assert abs(got[k] - want[k]) < 1e-9
    only = egomcq_accuracy_metrics(torch.eye(5)[:3], torch.tensor([0, 1, 4]), torch.tensor([2, 2, 2]))
    assert list(only) == ["Intra-video"] and abs(only["Intra-video"] - 200.0 / 3) < 1e-9   # one type id present: first name (reference zip)
