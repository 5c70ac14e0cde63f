"""Import alias: the package lives in ``pl-bert_amd/`` (a directory name Python cannot import).

``import plbert_amd`` resolves submodules from that directory.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pl-bert_amd"))

from ._api import *  # noqa: F401,F403,E402
from ._api import __all__  # noqa: E402
