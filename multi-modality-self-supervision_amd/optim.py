"""The optimizer line of the reference's trainer on the engine's flat buffers.

The reference builds `AdamW(self.model.parameters(), lr=args.lr)` from transformers.optimization (train_origin.py:15, 60) and calls
`optimizer.zero_grad(); loss.backward(); optimizer.step()` (train_origin.py:129-131).  `medvill_amd.optim.AdamW` is that class for a
medvill_amd.CXRBERT: the same constructor arguments and update rule (HF AdamW: eps outside the square root, decoupled weight decay,
`correct_bias`), one fused kernel over the flat fp32 master / moment buffers that also writes the 16-bit weight copies the MFMA kernels read
-- so the next forward has nothing to convert (a torch optimizer on the Parameters makes every forward refresh 110 M weights).

    from medvill_amd.optim import AdamW
    self.optimizer = AdamW(self.model.parameters(), lr=args.lr)          # train_origin.py:60, unchanged otherwise

It follows torch.optim.Optimizer's protocol (param_groups for schedulers, zero_grad, state_dict / load_state_dict) with ONE parameter group: the
update runs over the whole flat buffer, so per-group hyper-parameters and partial parameter sets are refused.
"""
from __future__ import annotations

import torch


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, overlap=False):
        """overlap=True: the update runs per parameter range on the engine's side stream and the next forward waits range by range (what the
        fused training step does: the HBM-bound update hides under the next step's first layers).  Anything ELSE that reads the Parameters on
        the current stream right after step() must then call `model.engine.wait_optimizer()` first -- hence opt-in."""
        params = list(params)
        if not params or isinstance(params[0], dict):
            raise ValueError("medvill_amd.optim.AdamW takes model.parameters() of ONE medvill_amd.CXRBERT (a single parameter group)")
        ref = getattr(params[0], "_medvill_model", None)
        model = ref() if ref is not None else None
        if model is None:
            raise ValueError("these are not the Parameters of a medvill_amd.CXRBERT: use a torch optimizer")
        own = {id(p) for p in model._plist}
        extra = [p for p in params if id(p) not in own]
        if len({id(p) for p in params} & own) != len(own):
            raise ValueError("medvill_amd.optim.AdamW updates the model's whole flat parameter buffer: pass ALL of model.parameters() "
                             "(freeze by other means, or use a torch optimizer for a subset)")
        if extra and any(p.requires_grad for p in extra):
            # e.g. a trainable region encoder: not in the flat buffer
            raise ValueError(f"{len(extra)} parameters do not belong to the CXRBERT's flat buffer (a trainable image encoder?): give those to a "
                             "torch optimizer of their own")
        super().__init__([p for p in params if id(p) in own], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._model, self._t, self.overlap = model, 0, bool(overlap)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self._model
        eng = model.engine
        grads = [p.grad for p in model._plist]
        if all(g is None for g in grads):
            return loss                                   # nothing was back-propagated since zero_grad(): like torch, no update
        if any(g is None for g in grads):
            raise RuntimeError("some Parameters have a gradient and some have none: the flat update cannot skip individual tensors")
        eng.ensure_grad()
        for n, g in zip(model._param_names, grads):       # normally every .grad IS the view of the flat buffer (CXRBERT.grad_views)
            if g.data_ptr() != eng.g[n].data_ptr():
                eng.g[n].copy_(g)
        hp = self.param_groups[0]
        self._t += 1
        eng.adamw_step(self._t, lr=float(hp["lr"]), betas=tuple(hp["betas"]), eps=float(hp["eps"]), weight_decay=float(hp["weight_decay"]),
                       correct_bias=bool(hp["correct_bias"]), overlap=self.overlap)
        # the kernel has written the 16-bit copies: until somebody else modifies a Parameter in place (version counters), forwards need not
        model._opt_versions = sum(p._version for p in model._plist)
        return loss

    def state_dict(self):
        eng = self._model.engine
        eng.wait_optimizer()
        eng.ensure_opt()
        return {"step": self._t, "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "flat_m": eng.flat_m.detach().cpu(), "flat_v": eng.flat_v.detach().cpu()}

    def load_state_dict(self, sd):
        eng = self._model.engine
        eng.wait_optimizer()
        eng.ensure_opt()
        if tuple(sd["flat_m"].shape) != tuple(eng.flat_m.shape):
            raise ValueError("optimizer state of a different model configuration")
        eng.flat_m.copy_(sd["flat_m"].to(eng.device))
        eng.flat_v.copy_(sd["flat_v"].to(eng.device))
        self._t = int(sd["step"])
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: v for k, v in s.items() if k != "params"})
