"""EgoMCQ accuracy -- drop-in for the one metric the EgoClip validation pass uses (reference model/metric.py:218-234,
named by configs/pt/egoclip.json `metrics: ["egomcq_accuracy_metrics"]`).

preds [Q, 5] (text-to-video similarities of the five candidate clips), labels [Q] (index of the correct clip), types [Q]
(1 = inter-video, 2 = intra-video in the EgoMCQ json).  The reference pairs the SORTED unique type ids with the fixed name
list ["Intra-video", "Inter-video"] (its zip, :221-223) -- reproduced literally, including the fact that the names follow
the sort order of the ids present, not their meaning.  Vectorised: no Python loop over the questions."""
import torch


def egomcq_accuracy_metrics(preds, labels, types):
    metrics = {}
    preds, labels, types = torch.as_tensor(preds), torch.as_tensor(labels).reshape(-1), torch.as_tensor(types).reshape(-1)
    hit = (preds.reshape(labels.shape[0], -1).argmax(dim=1) == labels.to(preds.device)).double()
    group_list = ["Intra-video", "Inter-video"]
    for type_i, group_i in zip(torch.unique(types), group_list):
        sel = (types == type_i).to(hit.device)
        metrics[group_i] = float(hit[sel].sum() / sel.sum()) * 100
    return metrics
