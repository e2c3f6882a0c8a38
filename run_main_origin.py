#!/usr/bin/env python3
"""Run the reference's own driver, UNCHANGED, on the MI355X implementation.

    cd /path/to/Multi-modality-Self-supervision          # the reference checkout: its data/, utils/, main_origin.py are used as they are
    python /path/to/this/repo/run_main_origin.py [main_origin.py's flags ...]

`main_origin.py:19` says `from models.train_origin import CXRBERT_Trainer`.  Python resolves `models` to the first package of that
name on sys.path, and a script's own directory always comes first -- so exporting PYTHONPATH is not enough.  This launcher puts the
directory that holds the builder-written `models/` package (re-exports of medvill_amd) in front and then executes
main_origin.py as `__main__` from the current directory.  medvill_amd is imported first, before anything touches the GPU
(GPU_MAX_HW_QUEUES)."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import medvill_amd  # noqa: E402,F401  (before the first device call)

target = os.path.join(os.getcwd(), "main_origin.py")
if not os.path.exists(target):
    raise SystemExit("run this from the reference checkout (main_origin.py not found in the current directory)")
sys.argv = [target] + sys.argv[1:]
sys.path.insert(1, os.getcwd())            # the reference's data/ and utils/ packages
runpy.run_path(target, run_name="__main__")
