"""How a batch is fed through a ``Step`` (reference torch_src/session/procedures/batch_train.py:9-117): whole, or in micro-batches
whose gradients accumulate (each micro-batch's loss divided by its size, as the reference does)."""
from __future__ import annotations

import abc
from typing import Callable, Dict, Optional, Union

import torch

from .step import DefaultStep, GraphStep, MixedPrecisionStep, Step

Features = Union[torch.Tensor, Dict[str, torch.Tensor]]
# update_metrics_function(loss, (y_pred, y_true), model, indices)
MetricsHook = Optional[Callable]


class BatchProcessor(abc.ABC):
    def __init__(self, step_function: Step):
        self._step_function = step_function

    @abc.abstractmethod
    def process_single_batch(self, model: torch.nn.Module, loss_function, features: Features, label: torch.Tensor,
                             indices: torch.Tensor, update_metrics_function: MetricsHook = None) -> None:
        """Forward (+ backward when ``model.training``) of one batch; ``features`` is a tensor or one tensor per modality."""

    def _micro_batch(self, model, loss_function, features, label, indices, update_metrics_function, loss_quotient: int = 1):
        y_pred, loss = self._step_function.forward(model, loss_function, features, label, loss_quotient)
        if model.training:
            self._step_function.backward(loss)
        if update_metrics_function:
            update_metrics_function(loss, (y_pred, label), model, indices)

    def run_optimizer_step(self, optimizer):
        return self._step_function.run_optimizer_step(optimizer)

    def reset(self) -> None:
        self._step_function.reset()

    def get_state_dict_objects(self, object_container: dict) -> None:
        self._step_function.get_state_dict_objects(object_container)

    def __str__(self):
        return str(self.__class__)


class DefaultBatchProcessor(BatchProcessor):
    def process_single_batch(self, model, loss_function, features, label, indices, update_metrics_function=None):
        self._micro_batch(model, loss_function, features, label, indices, update_metrics_function)


class GradientAccumulationBatchProcessor(BatchProcessor):
    def __init__(self, step_function: Step, batch_size: int, gradient_accumulation_batch_size: int):
        super().__init__(step_function)
        if batch_size % gradient_accumulation_batch_size:
            raise AssertionError(f"batch size {batch_size} is not a multiple of the accumulation size {gradient_accumulation_batch_size}")
        self._steps = batch_size // gradient_accumulation_batch_size
        self._gradient_accumulation_batch_size = gradient_accumulation_batch_size

    def process_single_batch(self, model, loss_function, features, label, indices, update_metrics_function=None):
        size = self._gradient_accumulation_batch_size
        for lo in range(0, self._steps * size, size):
            cut = slice(lo, lo + size)
            x = {k: v[cut] for k, v in features.items()} if isinstance(features, dict) else features[cut]
            y_true = label[cut]
            self._micro_batch(model, loss_function, x, y_true, indices[cut], update_metrics_function, loss_quotient=len(y_true))


def get_batch_processor_from_config(base_args, config: dict) -> BatchProcessor:
    """Same selection as the reference (:107-117) from its argparse namespace + session config: ``mixed_precision`` picks the
    bf16 step, ``batch_size != grad_accum_step`` the accumulating processor.  Additionally ``hip_graph`` (attribute of ``base_args``
    or key of ``config``) records the step into a HIP graph (``GraphStep``; with ``mixed_precision`` in the bf16 math mode)."""
    batch_size = config.get("batch_size", base_args.batch_size)
    grad_accum_step = config.get("grad_accum_step", base_args.grad_accum_step)
    graphed = config.get("hip_graph", getattr(base_args, "hip_graph", False))
    if graphed:
        step = GraphStep(math=MixedPrecisionStep.MODE if base_args.mixed_precision else None)
    else:
        step = MixedPrecisionStep() if base_args.mixed_precision else DefaultStep()
    if base_args.batch_size != base_args.grad_accum_step:
        return GradientAccumulationBatchProcessor(step, batch_size, grad_accum_step)
    return DefaultBatchProcessor(step)
