from multiplanarunet_amd.cli.train_fusion import entry_func, get_argparser  # noqa: F401
