"""The pretraining losses of models/train_origin.py:62-63,120-126 as ONE call.

    mlm_loss = criterion(mlm_output.transpose(1, 2), txt_labels)        # nn.CrossEntropyLoss(ignore_index=-100)
    itm_loss = criterion(itm_output, is_aligned)                        # nn.CrossEntropyLoss()
    loss = itm_loss + mlm_loss

becomes `loss = medvill_amd.losses.mlm_itm_loss(mlm_output, itm_output, txt_labels, is_aligned)`.  With plain tensors it computes exactly the
two torch cross-entropies above.  With `model.lazy_logits = True` the model's forward returns a `LazyLogits` handle instead of the
[B, L, V] tensor and this call runs the fused MLM head of the training step on the labelled rows only (cxrbert.LazyLogits): the literal
drop-in path (model swapped, loop kept) then costs about what the fused `TrainStep` costs per sample instead of 15x.
"""
from __future__ import annotations

import torch

from .cxrbert import LazyLogits


def mlm_itm_loss(mlm, itm, txt_labels, is_aligned, mlm_task=True, itm_task=True):
    """-> scalar loss tensor (mean MLM NLL over the labelled positions + mean ITM NLL over the batch; a task switched off contributes 0,
    as train_origin.py:108-118 does).  `mlm`: [B, L, V] logits or the LazyLogits handle of a `lazy_logits` forward."""
    if isinstance(mlm, LazyLogits):
        return mlm.loss(txt_labels, is_aligned, mlm_task=mlm_task, itm_task=itm_task)
    loss = itm.new_zeros(())
    if mlm_task:
        loss = loss + torch.nn.functional.cross_entropy(mlm.transpose(1, 2).float(), txt_labels.to(mlm.device), ignore_index=-100)
    if itm_task:
        loss = loss + torch.nn.functional.cross_entropy(itm.float(), is_aligned.to(itm.device))
    return loss
