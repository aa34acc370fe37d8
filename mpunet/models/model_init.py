"""init_model / model_initializer call sites of the reference (mpunet/models/model_init.py:5-59)."""


def init_model(build_hparams, logger=None):
    """models.__dict__[cls_name](logger=logger, **build_hparams): extra YAML keys are ignored by UNet."""
    from mpunet import models
    hp = dict(build_hparams)
    cls_name = hp.pop("model_class_name", "UNet")
    if cls_name not in ("UNet",):
        raise NotImplementedError("only model_class_name='UNet' is on the MI355X path")
    return getattr(models, cls_name)(logger=logger, **hp)


def model_initializer(hparams, continue_training, project_dir, initialize_from=None, logger=None):
    """Build the model; with continue_training resume as the reference does (newest `@epoch_` checkpoint by layer name,
    hparams["fit"]["init_epoch"], the learning rate logged for that epoch, logs/training.csv cut back:
    multiplanarunet_amd/resume.py), else optionally initialise from a weights file."""
    model = init_model(hparams["build"], logger)
    hparams.setdefault("fit", {})
    if continue_training:
        if initialize_from:
            raise ValueError("Failed to initialize model with both continue_training and initialize_from set.")
        from multiplanarunet_amd.resume import resume_state, apply_resume
        log = logger or print
        apply_resume(model, hparams, resume_state(project_dir, log), log)
    else:
        hparams["fit"]["init_epoch"] = 0
        if initialize_from:
            model.load_weights(initialize_from, by_name=True)
            (logger or print)("[NOTICE] Initializing parameters from:\n{}".format(initialize_from))
    return model
