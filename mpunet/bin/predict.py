from multiplanarunet_amd.cli.predict import entry_func, get_argparser  # noqa: F401
