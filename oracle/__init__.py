"""CPU oracle for the OD-VAE hot path -- test infrastructure only (see oracle/ldm_model.py header)."""
