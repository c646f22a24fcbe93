"""Load single files of the reference checkout as modules (TEST INFRASTRUCTURE ONLY; /root/reference exists in the build
container, not on the GPU box - callers skip when it is absent)."""
import importlib.util
import os

REF = "/root/reference"


def have_reference():
    return os.path.isdir(os.path.join(REF, "utils"))


def load(relpath, name=None):
    """Execute /root/reference/<relpath> as module `name` without touching sys.path (its package names - `utils`, `models` -
    collide with ours)."""
    path = os.path.join(REF, relpath)
    name = name or "reference_" + relpath.replace("/", "_").replace(".py", "")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
