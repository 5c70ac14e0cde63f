"""Public names of the package (host-side mirror of the reference interface for the hot path)."""
from .config import AlbertConfig, albert_config_from_yaml, load_config
from .symbols import CharacterIndexer, symbols, PAD_ID, MASK_ID, SEPARATOR_ID, UNKNOWN_ID
from .init import param_shapes, deterministic_state_dict, reference_init_state_dict
from .data import (MaskedPhonemeDataset, PhonemeOnlyCollater, Collater, build_dataloader,
                   length_to_mask, masked_indices_to_csr, synthetic_batch, seed_reference_streams)

__all__ = [
    "AlbertConfig", "albert_config_from_yaml", "load_config",
    "CharacterIndexer", "symbols", "PAD_ID", "MASK_ID", "SEPARATOR_ID", "UNKNOWN_ID",
    "param_shapes", "deterministic_state_dict", "reference_init_state_dict",
    "MaskedPhonemeDataset", "PhonemeOnlyCollater", "Collater", "build_dataloader",
    "length_to_mask", "masked_indices_to_csr", "synthetic_batch", "seed_reference_streams",
]
