"""`models.train_origin` of the reference (models/train_origin.py:19-266), served by medvill_amd: same class name, constructor
signature `CXRBERT_Trainer(args, train_dataloader, test_dataloader=None)` and `train(epoch)` / `save(epoch, file_path)`."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)

import medvill_amd as _mv  # noqa: E402

CXRBERT_Trainer = _mv.CXRBERT_Trainer
TrainStep = _mv.TrainStep
CXRBERT = _mv.CXRBERT

__all__ = ["CXRBERT_Trainer", "TrainStep", "CXRBERT"]
