#!/usr/bin/env python3
"""Fixtures for the model-level convert and the checkpoint reader.  RUNS ONLY IN THE BUILD CONTAINER.

Rebuilds exactly the reference `conv_resnet_bbb` of make_golden.py (same seeds, same calibration) and records
  * resnet_bbb_prepared_a7w8.npz : the flat state dict of the PREPARED model (quant_utils.prepare_model, calibrated: fp32 mu / rho,
    BatchNorm statistics, every observer's min / max) right before the reference's own `quant_utils.convert` runs.  The
    converted state it must turn into is `state/` of resnet_bbb_a7w8.npz (written by make_golden.py from the same model).
  * resnet_bbb_a7w8_weights.pt   : the converted model's checkpoint exactly as the reference writes it --
    `utils.save_model` = torch.save(model.state_dict()) with qint8 tensors (src/utils.py:84-93) -- with the `module.` prefix
    a DataParallel-trained run would carry, for the reader's prefix stripping (src/utils.py:112-123).
  * ensemble_ckpt/weights_{1,2}.pt: the two members of ensemble_resnet_a7w8.npz saved one file per member, the way
    `models_sgld.Network.load_ensemble` (models_sgld.py:245-261) expects to find them.
Data only: tensors produced by the reference; no reference source.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402

import make_golden as mg  # noqa: E402


def flat_prepared(model):
    out = {}
    for k, v in model.state_dict().items():
        if v is None or "fake_quant_enabled" in k or "observer_enabled" in k or k.endswith("num_batches_tracked"):
            continue
        out[k] = v.detach().numpy().copy()
    return out


def main():
    grabbed = {}
    model, args = mg.build_reference_model(7, 8, 4, before_convert=lambda m: grabbed.update(flat_prepared(m)))
    path = os.path.join(HERE, "resnet_bbb_prepared_a7w8.npz")
    np.savez_compressed(path, **{"state/" + k: v for k, v in grabbed.items()})
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB,", len(grabbed), "entries")
    # the converted state of THIS model must be the committed int8 fixture's
    ref = np.load(os.path.join(HERE, "resnet_bbb_a7w8.npz"))
    conv = mg.flat_state(model)
    bad = [k for k, v in conv.items() if not np.array_equal(np.asarray(v), ref["state/" + k])]
    assert not bad, bad[:5]
    # the reference's own checkpoint writer (src/utils.py:84-93)
    import src.utils as ru
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        sargs = types.SimpleNamespace(**vars(args), save=td)
        ru.save_model(model, sargs)
        sd = torch.load(os.path.join(td, "weights.pt"), map_location="cpu", weights_only=False)
    torch.save({"module." + k: v for k, v in sd.items()}, os.path.join(HERE, "resnet_bbb_a7w8_weights.pt"))
    print("wrote resnet_bbb_a7w8_weights.pt", round(os.path.getsize(os.path.join(HERE, "resnet_bbb_a7w8_weights.pt")) / 1e6, 2), "MB")


if __name__ == "__main__":
    main()
