"""Compatibility shim that lets the upstream reference (torch 1.7.1 era, at
/root/reference) import under torch 2.10 in THIS container.

Used ONLY by tests/golden/make_golden.py (the fixture generator).  Nothing in the
product, the -m gpu tests, smoke() or bench.py imports this file: /root/reference
does not exist on the GPU box.  See SURVEY.md section 8(c) for the list of
incompatibilities this papers over.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def install():
    import torch
    import torch.ao.nn.quantized.modules.conv as _aoconv

    # (1) stub modules imported at module top by src/utils.py, src/metrics.py, src/data.py
    def _stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    if "torch.utils.tensorboard" not in sys.modules:
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            tb = _stub("torch.utils.tensorboard", SummaryWriter=type("SummaryWriter", (), {}))
            torch.utils.tensorboard = tb
    if "torchmetrics" not in sys.modules:
        _stub("torchmetrics", Metric=type("Metric", (torch.nn.Module,), {}),
              CalibrationError=type("CalibrationError", (), {}))
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.datasets = _stub("torchvision.datasets")
        tv.transforms = _stub("torchvision.transforms")

    # (2) renamed private base class
    import torch.nn.quantized.modules.conv as _oldconv
    if not hasattr(_oldconv, "_ConvNd"):
        _oldconv._ConvNd = _aoconv._ConvNd

    # (3) renamed mapping symbols
    import torch.quantization.quantization_mappings as qm
    if not hasattr(qm, "QAT_MODULE_MAPPINGS"):
        qm.QAT_MODULE_MAPPINGS = qm.DEFAULT_QAT_MODULE_MAPPINGS
    if not hasattr(qm, "STATIC_QUANT_MODULE_MAPPINGS"):
        qm.STATIC_QUANT_MODULE_MAPPINGS = qm.DEFAULT_STATIC_QUANT_MODULE_MAPPINGS
    if not hasattr(qm, "get_qconfig_propagation_list"):
        qm.get_qconfig_propagation_list = qm.get_default_qconfig_propagation_list
    if hasattr(qm, "__all__"):
        for n in ("QAT_MODULE_MAPPINGS", "STATIC_QUANT_MODULE_MAPPINGS", "get_qconfig_propagation_list"):
            if n not in qm.__all__:
                qm.__all__.append(n)

    # (4) swap_module grew a mandatory argument
    import importlib
    qz = importlib.import_module("torch.quantization.quantize")
    if not getattr(qz.swap_module, "_qbnn_shim", False):
        _orig_swap = qz.swap_module

        def swap_module(mod, mapping, custom_module_class_mapping=None, *a, **k):
            if custom_module_class_mapping is None:
                custom_module_class_mapping = {}
            return _orig_swap(mod, mapping, custom_module_class_mapping, *a, **k)

        swap_module._qbnn_shim = True
        qz.swap_module = swap_module

    # (5) fuse_modules: user fuser_func(mod_list) vs new (mod_list, is_qat, extra) signature
    import torch.quantization as tq
    if not getattr(tq.fuse_modules, "_qbnn_shim", False):
        fm = importlib.import_module("torch.ao.quantization.fuse_modules")

        def fuse_modules(model, modules_to_fuse, inplace=False, fuser_func=None, **kw):
            if fuser_func is not None:
                user = fuser_func

                def _wrapped(mod_list, is_qat=None, additional_fuser_method_mapping=None):
                    return user(mod_list)

                return fm._fuse_modules(model, modules_to_fuse, is_qat=False, inplace=inplace,
                                        fuser_func=_wrapped, fuse_custom_config_dict=None)
            # default fuser: train-mode fusion goes to the qat variant
            training = any(m.training for m in model.modules())
            if training:
                return fm.fuse_modules_qat(model, modules_to_fuse, inplace=inplace)
            return fm.fuse_modules(model, modules_to_fuse, inplace=inplace)

        fuse_modules._qbnn_shim = True
        tq.fuse_modules = fuse_modules

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
