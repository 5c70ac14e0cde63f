"""Import alias: the package lives in ``pl-bert_amd/`` (a directory name Python cannot import).

``import plbert_amd`` resolves submodules from that directory.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pl-bert_amd"))

from . import _api  # noqa: E402
from ._api import (AlbertConfig, albert_config_from_yaml, load_config, CharacterIndexer, symbols, PAD_ID, MASK_ID,  # noqa: E402,F401
                   SEPARATOR_ID, UNKNOWN_ID, param_shapes, deterministic_state_dict, reference_init_state_dict,
                   MaskedPhonemeDataset, PhonemeOnlyCollater, Collater, build_dataloader, length_to_mask,
                   masked_indices_to_csr, synthetic_batch, seed_reference_streams)

__all__ = _api.__all__


def __getattr__(name):  # lazily resolved GPU-backed names (AlbertModel, PhonemeOnlyModel, PLBertTrainer, ...)
    return _api.__getattr__(name)
