"""MI355X-native PL-BERT pre-training hot path.  Import it as ``plbert_amd`` (see ../plbert_amd)."""
