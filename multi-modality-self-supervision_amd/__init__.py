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
_hwq_preset = "GPU_MAX_HW_QUEUES" in _os.environ          # exported by the caller (bench.py does) before anything could start HIP
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory instead of host-coherent memory (also read when the HIP runtime initialises): a dependent kernel starts
# sooner, and a step is ~390 mostly dependent launches -- 24.56 / 24.72 ms without against 24.31 / 24.29 ms with it, alternating runs on one
# box (profiles/r03_notes.txt).  AMD's MI300-series tuning notes recommend the setting.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def _warn_if_hip_started():
    # main_origin.py imports torch / wandb and may touch the device before it imports the trainer: the variable above is then
    # too late, the engine's side stream can share the main stream's hardware queue and the step serialises (26.6 -> 29.5 ms)
    import sys as _sys
    _torch = _sys.modules.get("torch")
    try:
        started = _torch is not None and _torch.cuda.is_initialized()
    except Exception:
        started = False
    if started and not _hwq_preset:
        import warnings as _w
        _w.warn("medvill_amd was imported after the HIP runtime had started: GPU_MAX_HW_QUEUES=8 could not take effect in this "
                "process.  With fewer hardware queues the backward's side stream may serialise behind the main stream (about 10 % "
                "slower steps).  Export GPU_MAX_HW_QUEUES=8 before starting Python, or import medvill_amd before the first CUDA call.",
                RuntimeWarning, stacklevel=3)


_warn_if_hip_started()

from .engine import Engine, ModelConfig, param_layout  # noqa: F401
from .cxrbert import CXRBERT  # noqa: F401
from .trainer import CXRBERT_Trainer, TrainStep  # noqa: F401
from .retrieval import CXRBertForRetrieval  # noqa: F401
from .image import ImageEncoder_cnn  # noqa: F401
from . import checkpoint, data, dist, hip_ops, losses, optim  # noqa: F401

__all__ = ["Engine", "ModelConfig", "param_layout", "CXRBERT", "CXRBERT_Trainer", "TrainStep", "CXRBertForRetrieval", "ImageEncoder_cnn", "data"]
