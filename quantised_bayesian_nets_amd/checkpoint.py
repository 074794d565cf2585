"""Reference checkpoint ingestion: the wire format is `torch.save(model.state_dict())`.

What the reference writes and reads:
  * `utils.save_model` (src/utils.py:84-93): `weights<special>.pt` = torch.save of the state dict.  For a converted BBB model
    the stochastic layers' custom `_save_to_state_dict` (conv_q.py:72-78, linear_q.py:40-46) puts per-tensor-affine **qint8
    tensors** under `<p>weight` / `<p>std`, plain tensors under `<p>scale`, `<p>zero_point`, `<p>bias_`, plus the
    `QFunctional`s' `<p>add_weight.scale/zero_point`, `<p>mul_noise.scale/zero_point`, the stub's `quant.scale/zero_point`
    and every block's `add.add.scale/zero_point`.  Standard quantised layers (MC-Dropout nets, SGHMC members) store
    `<p>weight` (qint8), `<p>bias`, `<p>scale`, `<p>zero_point`; a quantised Linear keeps its pair in
    `<p>_packed_params._packed_params`.
  * `utils.load_model` (src/utils.py:112-123): torch.load on the CPU, strips `module.` (DataParallel) and `main_net.`
    (the SGHMC training wrapper) from the keys, drops keys the model does not have.
  * `sgld.Network.load_ensemble` (models_sgld.py:245-261): one file per member, `weights_<special><n>.pt`, natural order,
    the last `args.samples` of them.

The build's models ingest a FLAT numpy dict (`load_reference_state`): a quantised tensor `k` becomes `k` (int8 values),
`k.q_scale`, `k.q_zero_point`.  This module converts between the two; it does host-side file I/O only.
"""
import os
import re

import numpy as np
import torch

STRIP_PREFIXES = ("module.", "main_net.")      # src/utils.py:118
_PACKED = "_packed_params._packed_params"


def flat_state_from_torch(state_dict, replace=True):
    """torch state dict (as saved by the reference) -> the flat numpy dict `load_reference_state` takes."""
    out = {}

    def put(k, v):
        if v is None:
            return
        if isinstance(v, torch.Tensor) and v.is_quantized:
            if v.qscheme() not in (torch.per_tensor_affine, torch.per_tensor_symmetric):
                raise RuntimeError("Unsupported qscheme: the reference quantises per tensor only")     # cf. conv_q.py:117
            out[k] = v.int_repr().cpu().numpy()
            out[k + ".q_scale"] = np.float64(v.q_scale())
            out[k + ".q_zero_point"] = np.int64(v.q_zero_point())
        elif isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)

    for k, v in state_dict.items():
        if replace:
            for p in STRIP_PREFIXES:
                k = k.replace(p, "")
        if k.endswith(_PACKED):                      # torch.ao.nn.quantized.Linear: (qint8 weight, fp32 bias or None)
            base = k[:-len(_PACKED)]
            w, b = v
            put(base + "weight", w)
            put(base + "bias", b)
        elif k.endswith("_packed_params.dtype"):
            continue
        else:
            put(k, v)
    return out


def torch_state_from_flat(flat, prefix=""):
    """Inverse of flat_state_from_torch: the reference's wire format (qint8 tensors under the reference's keys)."""
    out = {}
    for k, v in flat.items():
        if k.endswith(".q_scale") or k.endswith(".q_zero_point"):
            continue
        a = np.asarray(v)
        if (k + ".q_scale") in flat:
            out[prefix + k] = torch._make_per_tensor_quantized_tensor(torch.from_numpy(np.ascontiguousarray(a, np.int8)),
                                                                       float(flat[k + ".q_scale"]), int(flat[k + ".q_zero_point"]))
        else:
            out[prefix + k] = torch.from_numpy(np.array(a, copy=True)) if a.ndim else torch.tensor(a.item())
    return out


def load_state(model_path, replace=True):
    """Flat numpy state of a reference checkpoint file (CPU)."""
    sd = torch.load(model_path, map_location=torch.device("cpu"), weights_only=False)    # the reference's own files: qint8 tensors / tuples
    return flat_state_from_torch(sd, replace)


def load_model(model, model_path, replace=True):
    """reference `utils.load_model(model, model_path, replace=True)` (src/utils.py:112-123) for the build's models."""
    model.load_reference_state(load_state(model_path, replace))
    return model


def save_model(model, path, prefix=""):
    """Write `model` (a converted int8 BBB model of this package) as a reference-format checkpoint: the file the
    reference's `utils.load_model` reads back into its own converted model (src/utils.py:84-93)."""
    flat = {}
    for name, layer in zip(model.stochastic_layer_names(), model.stochastic_layers()):
        flat.update(layer.reference_state(name + "."))
        flat[name + ".std_prior"] = layer.std_prior.detach().cpu().numpy()
    for n, m in model.named_modules():
        if type(m).__name__ == "Add" and hasattr(m, "add"):
            flat[n + ".add.scale"] = np.float32(m.add.scale)
            flat[n + ".add.zero_point"] = np.int64(m.add.zero_point)
    flat["quant.scale"] = np.asarray([model.quant.scale], np.float32)
    flat["quant.zero_point"] = np.asarray([model.quant.zero_point], np.int64)
    torch.save(torch_state_from_flat(flat, prefix), path)
    return path


def _natural_keys(text):
    return [int(c) if c.isdigit() else c for c in re.split(r"(-?\d+)", text)]     # src/utils.py:57-61 (its regex, verbatim)


def ensemble_files(path, samples, special_info=""):
    """The member files `load_ensemble` picks (models_sgld.py:246-257): `weights_<special><n>.pt`, natural order, last `samples`."""
    names = []
    for _root, _dirs, files in os.walk(path):
        for f in files:
            if ".pt" in f:
                m = re.findall("weights_" + special_info + "[0-9]*.pt", f)
                if len(m) >= 1:
                    names.append(m[0])
    names.sort(key=_natural_keys)
    return names[-samples:]
