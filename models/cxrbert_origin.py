"""`models.cxrbert_origin` of the reference (models/cxrbert_origin.py:132-149 CXRBERT; Retrieval/retrieval.py:26-31 imports it
the same way), served by medvill_amd."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)

import medvill_amd as _mv  # noqa: E402

CXRBERT = _mv.CXRBERT
CXRBertForRetrieval = _mv.CXRBertForRetrieval

__all__ = ["CXRBERT", "CXRBertForRetrieval"]
