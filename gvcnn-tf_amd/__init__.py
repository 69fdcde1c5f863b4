"""gvcnn-tf_amd — MI355X-native GVCNN hot path (per-view backbone + grouping module).

The directory name carries a hyphen (it is the name the build contract asks for); import it
as `gvcnn_tf_amd` through the shim module of that name at the repository root.

Importing this package loads libgvcnn_hip.so; a missing library is an ImportError — the HIP
library is the only compute path.
"""
from . import _lib

_lib.load()

from . import backbones, params  # noqa: E402
from . import model  # noqa: E402
from .model import (AUTO_REUSE, GVCNN, basic, configure, group_fusion, group_scheme, group_weight,  # noqa: E402,F401
                    grouping_module, gvcnn, gvcnn_fused, view_pooling)

__all__ = ["GVCNN", "gvcnn", "basic", "group_scheme", "group_weight", "view_pooling", "group_fusion",
           "grouping_module", "gvcnn_fused", "configure", "AUTO_REUSE", "backbones", "params", "model"]
