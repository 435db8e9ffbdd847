"""MI355X-native implementation of RCF's per-frame training hot path.

Public surface mirrors the reference's: `RCFModel` (models/rcf_model.py), the component registry
(`ResNet`, `FCNHead`, `FlowAggregationHeadWithResidual`, `CRFHead`, `CompactnessHead`), the warp
helpers of utils/warp_utils.py and a `torchcrf_cpp`-shaped `crf_soft` / `crf_hard`.
"""
__version__ = "0.1.0"

from . import _lib, data_pipeline, evaluate, ncut, offline, ops, synth, vit  # noqa: F401
from .backbone import FCNHead, ResNet  # noqa: F401
from .evaluate import Evaluator  # noqa: F401
from .crf import CRFHead, crf_hard, crf_soft  # noqa: F401
from .flow_head import CompactnessHead, FlowAggregationHeadWithResidual  # noqa: F401
from .model import RCFModel  # noqa: F401
from .ops import flow_warp, occu_mask_backward as get_occu_mask_backward  # noqa: F401
from .ops import occu_mask_bidirection as get_occu_mask_bidirection  # noqa: F401
from .trainer import Trainer, poly_lr_factor  # noqa: F401
