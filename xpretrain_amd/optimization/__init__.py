from .loss import NCELearnableTempLoss, build_loss_func  # noqa: F401
