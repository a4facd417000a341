"""The real reference, where it exists (the build container: /root/reference), with the GUI / audio modules the image
lacks replaced by inert stand-ins.  TEST INFRASTRUCTURE; never used by `-m gpu` tests: the
reference does not travel to the GPU box."""
import os
import sys
import types
from unittest import mock

REF = os.environ.get("SSDR_REFERENCE_DIR", "/root/reference")
_cache = {}


def available():
    return os.path.isfile(os.path.join(REF, "utils_supersdr.py"))


def load():
    """-> (utils_supersdr module, kiwi.worker module, kiwi.client module) or None"""
    if "mods" in _cache:
        return _cache["mods"]
    if not available():
        _cache["mods"] = None
        return None
    sys.dont_write_bytecode = True               # the tree is read-only by convention: no __pycache__ into it
    for name in ("pygame", "pygame.font", "pygame.event", "pygame.draw", "pygame.freetype", "sounddevice", "xmltodict", "requests"):
        sys.modules.setdefault(name, mock.MagicMock())
    loc = types.ModuleType("pygame.locals")
    for k in (["K_%d" % i for i in range(10)] + ["K_KP%d" % i for i in range(10)] + ["K_BACKSPACE", "K_RETURN", "K_ESCAPE", "K_KP_ENTER"]):
        setattr(loc, k, hash(k) & 0xFFFF)
    sys.modules.setdefault("pygame.locals", loc)
    if "tkinter" not in sys.modules:
        tk = types.ModuleType("tkinter")
        tk.__all__ = []
        sys.modules["tkinter"] = tk
        sys.modules["tkinter.ttk"] = mock.MagicMock()
        sys.modules["tkinter.messagebox"] = mock.MagicMock()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)                                # the reference opens its .ttf fonts relative to CWD at import time
    try:
        import utils_supersdr as U
        from kiwi import worker as KW
        from kiwi import client as KC
    finally:
        os.chdir(cwd)
    _cache["mods"] = (U, KW, KC)
    return _cache["mods"]
