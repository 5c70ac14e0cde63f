"""MI355X-native PL-BERT pre-training hot path.  Import it as ``plbert_amd`` (see ../plbert_amd)."""
from ._api import *  # noqa: F401,F403
from ._api import __all__  # noqa: F401
