"""vt355 -- import name of the MI355X-native CogVideoX finetune engine.

The sources live in ``videotuna-dev_amd/`` (a directory name Python cannot import directly because of
the hyphen); this shim only redirects the package search path there.
"""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "videotuna-dev_amd")
__path__.insert(0, _src)

from ._lib import lib_path, load_library, VtError   # noqa: E402,F401
