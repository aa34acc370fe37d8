"""`mp <script> [args]` dispatcher (mpunet/bin/mp.py:45-55): only the hot-path scripts exist here."""
import sys
import importlib

SCRIPTS = ("train", "train_fusion", "predict")


def entry_func(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print("usage: mp [--help] script [script args...]\n\nAvailable scripts on the MI355X hot path:\n- "
              + "\n- ".join(SCRIPTS))
        return 0
    script, rest = argv[0], argv[1:]
    if script not in SCRIPTS:
        raise SystemExit("mp: script '%s' is outside the accelerated path (available: %s)" % (script, ", ".join(SCRIPTS)))
    importlib.import_module("multiplanarunet_amd.cli." + script).entry_func(rest)
    return 0


if __name__ == "__main__":
    entry_func()
