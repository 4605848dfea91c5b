"""Import the real reference model class in the BUILD container.  TEST INFRASTRUCTURE ONLY.

/root/reference does not exist on the GPU box, so this module is used only by
``oracle/gen_golden.py`` and by CPU tests that skip themselves when the path is
absent.  Recipe: the reference imports ``timm`` and
``torchvision`` at module import time (loadImageModelClassifier.py:3-4), neither
of which is installed; empty stand-in *module objects* are registered for the
import statement only -- no backbone arithmetic of those libraries is emulated,
and only ``cnn_model_name='custom-cnn'`` (defined inside the reference itself)
is ever built through this path.
"""
import os
import sys
import types

REF_MODELS = "/root/reference/src/scripts/benchmark/models"


def available():
    return os.path.isdir(REF_MODELS)


def load():
    """-> dict with the reference classes."""
    import transformers  # noqa: F401  (must probe torchvision availability before the stub)
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm.list_models = lambda pretrained=True: []
        sys.modules["timm"] = timm
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tv.models = tvm
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
    if REF_MODELS not in sys.path:
        sys.path.insert(0, REF_MODELS)
    import importlib
    mm = importlib.import_module("multimodalIntraInterModal")
    return {
        "MultimodalModel": mm.MultimodalModel,
        "MetaBlock": importlib.import_module("metablock").MetaBlock,
        "GatedAlteredResidualBlock": importlib.import_module("gatedResidualBlock").GatedAlteredResidualBlock,
        "TabTransformer": importlib.import_module("tab_transformer").TabTransformer,
    }
