"""Pin for SURVEY 8(f) rank 4 (the ResNet-50 region encoder, models/image.py:46-53 = torchvision.models.resnet50 children()[:-2]).

TEST INFRASTRUCTURE ONLY.  torchvision is in neither /root/reference nor this image, so oracle/resnet_oracle.py is a restatement
checked against nothing ("parity unpinned").  Run this script ANYWHERE torchvision is installed:

    python oracle/gen_resnet_golden.py          # writes tests/golden/resnet50.npz (~0.3 MB)

It builds `torchvision.models.resnet50(weights=None)`, fills every parameter / buffer with deterministic splitmix64 values (the scheme
of oracle/cxrbert_oracle.splitmix_uniform: nothing depends on torch's RNG), runs the reference's trunk `nn.Sequential(*children()[:-2])`
on a seeded [2, 3, 64, 64] input in eval() and in train() mode (models/image.py keeps the CNN in train(): batch statistics), and stores
the input seed, the output feature maps and the updated running statistics of three BatchNorm layers.  tests/test_image_gpu.py and
tests/test_oracle_golden.py pick the file up when it exists; until then they report the row as unpinned."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.cxrbert_oracle import splitmix_uniform      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "resnet50.npz")
PROBES = ("1", "5.0.bn2", "7.2.bn3")                 # BatchNorm layers whose running statistics are recorded after the train() pass


def fill_state(sd: dict, seed: int = 4100) -> dict:
    """Deterministic values for a torchvision-layout state dict (keys as given): conv weights ~ U(-a, a) with a = sqrt(3 / fan_in) (He-like,
    keeps activations O(1) through 53 convolutions), BatchNorm weight in [0.5, 1.5], bias in [-0.1, 0.1], running_mean in [-0.1, 0.1],
    running_var in [0.5, 1.5].  Used by the generator here AND by the tests to rebuild the same weights without torchvision."""
    out = {}
    for i, (k, v) in enumerate(sorted(sd.items())):
        shape = tuple(v.shape)
        n = int(np.prod(shape)) if shape else 1
        u = splitmix_uniform(seed + i, n).astype(np.float32).reshape(shape) if n else np.zeros(shape, np.float32)     # U(-1, 1)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            out[k] = torch.from_numpy(u * np.float32(np.sqrt(3.0 / (shape[1] * shape[2] * shape[3]))))
        elif k.endswith(("running_var", ".weight")):
            out[k] = torch.from_numpy(1.0 + 0.5 * u)
        else:
            out[k] = torch.from_numpy(0.1 * u)
    return out


def make_input(seed: int = 4242, B: int = 2, S: int = 64) -> torch.Tensor:
    return torch.from_numpy(splitmix_uniform(seed, B * 3 * S * S).astype(np.float32).reshape(B, 3, S, S))


def main():
    try:
        import torchvision
    except ImportError:
        print("torchvision is not installed here: nothing written (the region encoder's parity stays UNPINNED; see the module docstring)")
        return 2
    net = torchvision.models.resnet50(weights=None) if hasattr(torchvision.models, "get_model") else torchvision.models.resnet50(pretrained=False)
    trunk = torch.nn.Sequential(*list(net.children())[:-2])           # models/image.py:50-52
    sd = fill_state({k: v for k, v in trunk.state_dict().items()})
    trunk.load_state_dict(sd)
    x = make_input()
    rec = {"in_seed": np.int64(4242), "w_seed": np.int64(4100), "torchvision": np.array(torchvision.__version__)}
    trunk.eval()
    with torch.no_grad():
        rec["out_eval"] = trunk(x).numpy()
    trunk.train()
    with torch.no_grad():
        rec["out_train"] = trunk(x).numpy()
    after = trunk.state_dict()
    for p in PROBES:
        rec[f"rm_{p}"] = after[p + ".running_mean"].numpy()
        rec[f"rv_{p}"] = after[p + ".running_var"].numpy()
    rec["keys"] = np.array(sorted(sd))
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, {k: getattr(v, "shape", ()) for k, v in rec.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
