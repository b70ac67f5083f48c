"""Mirror of the reference's `model` package for the hot path (disentangled_vae.py, variational_base_vae.py)."""
