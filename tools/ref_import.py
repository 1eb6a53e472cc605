"""Import the upstream reference (/root/reference) on CPU in THIS container only.

The reference needs `timm` (only trunc_normal_) and `torchvision` (only for
classes that are out of scope) at import time; neither is installed here, so
two throw-away stub modules are registered before the import.  Nothing in this
file (or anything it imports from /root/reference) travels to the GPU box: it
is used by tools/make_goldens.py to produce tests/golden/*.npz and by
tools/check_oracle_vs_reference.py.
"""
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _install_stubs():
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        wi = types.ModuleType("timm.models.layers.weight_init")
        wi.trunc_normal_ = torch.nn.init.trunc_normal_
        layers.weight_init = wi
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        models.layers = layers
        timm.models = models
        sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers,
                            "timm.models.layers.weight_init": wi})
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvu = types.ModuleType("torchvision.models._utils")

        class _Missing:  # any use of a torchvision model is out of scope
            def __init__(self, *a, **k):
                raise RuntimeError("torchvision is not available in this image")

        tvm.resnet18 = _Missing
        tvm.vgg19 = _Missing
        tvu.IntermediateLayerGetter = _Missing
        tvm._utils = tvu
        tv.models = tvm
        sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models._utils": tvu})


def import_reference():
    """Returns the reference's `modules` package (modules.raft, modules.dense_motion, ...)."""
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # modules/util.py in the reference is missing `TPS` unless the full file is read; import as-is.
    import modules.util  # noqa: F401
    import modules.generator  # noqa: F401
    import modules.dense_motion  # noqa: F401
    import modules.kp_detector  # noqa: F401
    import modules.raft  # noqa: F401
    import modules
    return modules
