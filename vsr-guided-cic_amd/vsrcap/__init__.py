"""vsrcap: host-side plumbing of the MI355X VSR captioning decoder (ctypes binding, engine, synthetic data)."""
