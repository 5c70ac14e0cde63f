"""MI355X-native PL-BERT pre-training hot path: host-side mirror of the reference interface
(model.py / train.py / dataloader.py / char_indexer.py names) over libplbert_hip.so (include/plbert.h)."""
from .config import AlbertConfig, albert_config_from_yaml, load_config
from .symbols import CharacterIndexer, symbols, PAD_ID, MASK_ID, SEPARATOR_ID, UNKNOWN_ID
from .init import param_shapes, deterministic_state_dict, reference_init_state_dict
from .data import (MaskedPhonemeDataset, PhonemeOnlyCollater, Collater, build_dataloader,
                   length_to_mask, masked_indices_to_csr, synthetic_batch, seed_reference_streams, collate_decisions, seed_worker, DecisionsDataset)



def __getattr__(name):
    # GPU-backed classes import torch.cuda-facing modules lazily so that the host-only parts of the
    # package (data path, config, layout) stay importable on machines without the HIP library.
    if name in ("AlbertModel", "PhonemeOnlyModel", "MultiTaskModel", "BaseModelOutputWithPooling"):
        from . import model
        return getattr(model, name)
    if name in ("PLBertTrainer", "process_batch", "AdamW", "StagedBatch", "stage_reference_batch", "validate_batch",
                "device_mask_batch", "device_apply_mask"):
        from . import train
        return getattr(train, name)
    if name in ("save_checkpoint", "load_checkpoint", "find_latest_checkpoint"):
        from . import checkpoint
        return getattr(checkpoint, name)
    if name in ("export_pretrained", "encoder_state_dict", "load_pl_bert_model"):
        from . import export
        return getattr(export, name)
    if name == "DeviceFeeder":
        from .pipeline import DeviceFeeder
        return DeviceFeeder
    if name == "HipEngine":
        from .engine import HipEngine
        return HipEngine
    raise AttributeError(name)


__all__ = [
    "AlbertModel", "PhonemeOnlyModel", "MultiTaskModel", "PLBertTrainer", "process_batch", "AdamW", "HipEngine",
    "save_checkpoint", "load_checkpoint", "find_latest_checkpoint", "export_pretrained", "encoder_state_dict",
    "load_pl_bert_model", "device_mask_batch", "device_apply_mask",
    "AlbertConfig", "albert_config_from_yaml", "load_config",
    "CharacterIndexer", "symbols", "PAD_ID", "MASK_ID", "SEPARATOR_ID", "UNKNOWN_ID",
    "param_shapes", "deterministic_state_dict", "reference_init_state_dict",
    "MaskedPhonemeDataset", "PhonemeOnlyCollater", "Collater", "build_dataloader",
    "length_to_mask", "masked_indices_to_csr", "synthetic_batch", "seed_reference_streams", "collate_decisions", "seed_worker", "DecisionsDataset", "DeviceFeeder",
]
