"""Import paths of the reference (`src.*`) resolved onto the MI355X implementation, so the untouched
configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml `target:` strings instantiate the HIP-backed classes."""
