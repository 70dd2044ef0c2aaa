"""The epoch loops around the batch processors (reference torch_src/session/session.py:161-205, ``Session.train_epoch`` /
``Session.validate_epoch``): same arguments and order of operations -- batch to the device as float32 / int64, ``zero_grad``,
process the batch, optimizer step, progress -- with the copy left to ``data.ClipBatches`` when it already delivers device tensors
(its pinned double-buffered H2D pipeline).  Everything else of the reference's Session (config, checkpoints, logging, metrics
classes) is control plane and out of scope; ``metrics`` / ``progress`` are duck-typed and optional."""
from __future__ import annotations

import torch

from .procedures.batch_train import BatchProcessor


def _to_device(features_batch, label_batch, device):
    move = lambda t, dt: t if (t.device == device and t.dtype == dt) else t.to(device=device, dtype=dt, non_blocking=True)   # noqa: E731
    with torch.no_grad():
        if isinstance(features_batch, dict):
            features = {k: move(v, torch.float32) for k, v in features_batch.items()}
        else:
            features = move(features_batch, torch.float32)
        return features, move(label_batch, torch.int64)


class Session:
    @staticmethod
    def train_epoch(batch_processor: BatchProcessor, model: torch.nn.Module, loss_function, dataset, optimizer, progress=None,
                    metrics=None) -> None:
        model.train()
        device = next(model.parameters()).device
        for features_batch, label_batch, indices in dataset:
            features, label = _to_device(features_batch, label_batch, device)
            optimizer.zero_grad()
            batch_processor.process_single_batch(model, loss_function, features, label, indices,
                                                 metrics.update_training if metrics is not None else None)
            batch_processor.run_optimizer_step(optimizer)
            if progress:
                progress.update_epoch_mode(0, metrics=metrics.format_training() if metrics is not None else None)

    @staticmethod
    def validate_epoch(batch_processor: BatchProcessor, model: torch.nn.Module, loss_function, dataset, progress=None, metrics=None,
                       mode: int = 1) -> None:
        model.eval()
        device = next(model.parameters()).device
        with torch.no_grad():
            for features_batch, label_batch, indices in dataset:
                features, label = _to_device(features_batch, label_batch, device)
                batch_processor.process_single_batch(model, loss_function, features, label, indices,
                                                     metrics.update_validation if metrics is not None else None)
                if progress:
                    progress.update_epoch_mode(mode, metrics=metrics.format_all() if metrics is not None else None)
