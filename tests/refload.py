"""Load single files of the reference checkout as modules (TEST INFRASTRUCTURE ONLY; /root/reference exists in the build
container, not on the GPU box - callers skip when it is absent)."""
import importlib.util
import os

REF = "/root/reference"


def have_reference():
    return os.path.isdir(os.path.join(REF, "utils"))


def load(relpath, name=None, absent=()):
    """Execute /root/reference/<relpath> as module `name` without touching sys.path (its package names - `utils`, `models` -
    collide with ours).  `absent`: top-of-file imports of packages that are not installed here and that the functions under
    test do not use (cv2, torchvision, imageio in utils/common.py) are satisfied with empty modules for the duration of the load."""
    import sys
    import types
    path = os.path.join(REF, relpath)
    name = name or "reference_" + relpath.replace("/", "_").replace(".py", "")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    added = []
    for a in absent:
        if a not in sys.modules:
            sys.modules[a] = types.ModuleType(a)
            added.append(a)
    try:
        spec.loader.exec_module(mod)
    finally:
        for a in added:
            del sys.modules[a]
    return mod
