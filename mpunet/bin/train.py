from multiplanarunet_amd.cli.train import entry_func, get_argparser, validate_args  # noqa: F401
