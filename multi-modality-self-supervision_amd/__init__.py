"""MI355X-native (gfx950) implementation of MedViLL's cross-modal BERT pretraining hot path.

Mirrors the reference's Python interface for that path and nothing else:
    CXRBERT(config, args).forward(...)            models/cxrbert_origin.py:132-149
    CXRBERT_Trainer(args, train_dl, test_dl)      models/train_origin.py:19-266
on top of hand-written HIP kernels reached through the C ABI of include/medvill.h.
"""
import os as _os

# The engine overlaps the weight-gradient GEMMs with the backward's main chain on a second HIP stream.  HIP maps streams onto a
# small pool of hardware queues (4 by default); once RCCL has created its own streams the engine's second stream lands on the SAME
# hardware queue as the first and everything serialises (measured on a one-rank RCCL group: 26.6 -> 29.5 ms/step; with 8 queues
# 26.9).  The variable is read when the HIP runtime initialises, i.e. at the first device call -- importing torch is not one.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .engine import Engine, ModelConfig, param_layout  # noqa: F401
from .cxrbert import CXRBERT  # noqa: F401
from .trainer import CXRBERT_Trainer, TrainStep  # noqa: F401
from .retrieval import CXRBertForRetrieval  # noqa: F401
from .image import ImageEncoder_cnn  # noqa: F401
from . import checkpoint, data  # noqa: F401

__all__ = ["Engine", "ModelConfig", "param_layout", "CXRBERT", "CXRBERT_Trainer", "TrainStep", "CXRBertForRetrieval", "ImageEncoder_cnn", "data"]
