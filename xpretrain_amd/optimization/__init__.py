from .loss import NCELearnableTempLoss, NCELearnableTempLoss_vsc_fc, build_loss_func  # noqa: F401
from .adamw import AdamW  # noqa: F401
from .sched import get_lr_sched  # noqa: F401
from .utils import build_e2e_optimizer_w_lr_mul, setup_e2e_optimizer  # noqa: F401
