"""Host-side mirrors of the reference's trainer / calibrator plugin classes for the hot path.

Same class names, constructor roles, ``forward`` / ``model_inference`` return contracts as
``trainers/classification/{zsclip,coop,cocoop,kgcoop,maple,proda,prograd,promptsrc,vpt,clip_adapter,taskres}.py`` and ``trainers/calibration/{tempscaling,
distanse_aware_calibration,vl_calibrator}.py``; the Dassl engine, datasets and the training loops around them are
out of scope.  Class names are tokenised upstream (tokenizer = SURVEY f-3), so constructors take token
ids where the reference takes class-name strings.
"""
from .zsclip import ZeroshotCLIP  # noqa: F401
from .coop import CustomCLIP as CoOpCLIP, PromptLearner, TextEncoder  # noqa: F401
from .kgcoop import CustomCLIP as KgCoOpCLIP  # noqa: F401
from .maple import CustomCLIP as MaPLeCLIP, MultiModalPromptLearner  # noqa: F401
from .promptsrc import CustomCLIP as PromptSRCCLIP  # noqa: F401
from .vpt import CustomCLIP as VPTCLIP  # noqa: F401
from .cocoop import CustomCLIP as CoCoOpCLIP  # noqa: F401
from .prograd import CustomCLIP as ProGradCLIP  # noqa: F401
from .proda import CustomCLIP as ProDACLIP  # noqa: F401
from .clip_adapter import CustomCLIP as CLIPAdapterCLIP  # noqa: F401
from .taskres import CustomCLIP as TaskResCLIP  # noqa: F401
from .tempscaling import CustomCLIPCalibration, ScaleLearner  # noqa: F401
