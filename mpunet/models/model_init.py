"""init_model / model_initializer call sites of the reference (mpunet/models/model_init.py:5-59)."""
import os


def init_model(build_hparams, logger=None):
    """models.__dict__[cls_name](logger=logger, **build_hparams): extra YAML keys are ignored by UNet."""
    from mpunet import models
    hp = dict(build_hparams)
    cls_name = hp.pop("model_class_name", "UNet")
    if cls_name not in ("UNet",):
        raise NotImplementedError("only model_class_name='UNet' is on the MI355X path")
    return getattr(models, cls_name)(logger=logger, **hp)


def model_initializer(hparams, continue_training, project_dir, initialize_from=None, logger=None):
    """Build the model and optionally load weights (by layer name) to resume / initialise from."""
    model = init_model(hparams["build"], logger)
    if continue_training:
        path = os.path.join(project_dir, "model", "model_weights.npz")
        if os.path.exists(path):
            model.load_weights(path, by_name=True)
    elif initialize_from:
        model.load_weights(initialize_from, by_name=True)
    return model
