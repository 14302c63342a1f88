"""Model modules mirroring lidbox.models for the hot path: `xvector`, `cnn` (create / as_embedding_extractor)."""
