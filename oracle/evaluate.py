"""Oracle for the validation bookkeeping (test infrastructure).

Restates the validation loop of reference training_scripts/supervised_training.py:177-215 on
CPU torch tensors: per-class correct / total counts from `torch.max(pred, 1)[1]` (:193-198) and
the cross-entropy of the logits against the labels (`nn.CrossEntropyLoss`, :191), here summed
over frames and divided once at the end instead of the script's per-batch means.  The reference
holds no fixture for this loop (it is a script without tests): parity unpinned beyond torch's own
cross_entropy and max."""
import torch


def validate(logit_batches, label_batches, n_classes=3):
    correct = torch.zeros([n_classes])
    total = torch.zeros([n_classes])
    loss_sum, frames = 0.0, 0
    for pred, labels in zip(logit_batches, label_batches):
        pred = torch.as_tensor(pred, dtype=torch.float32)
        labels = torch.as_tensor(labels, dtype=torch.int64).reshape(-1)
        loss_sum += float(torch.nn.functional.cross_entropy(pred.double(), labels, reduction="sum"))
        pc = torch.max(pred, dim=1)[1]
        for i in range(n_classes):
            correct[i] += torch.sum(pc[labels == i] == labels[labels == i])
            total[i] += torch.sum(labels == i)
        frames += pred.shape[0]
    return {"correct": [int(c) for c in correct], "total": [int(t) for t in total],
            "accuracy": [(c / t).item() if t else float("nan") for c, t in zip(correct, total)],
            "loss": loss_sum / frames, "frames": frames}
