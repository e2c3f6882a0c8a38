"""Import-path shim: with this directory ahead of the reference's own `models/` on sys.path,
`main_origin.py:19` -- `from models.train_origin import CXRBERT_Trainer` -- resolves to the MI355X implementation without an
edit.  Nothing of the reference is copied here: the two modules only re-export `medvill_amd` symbols."""
