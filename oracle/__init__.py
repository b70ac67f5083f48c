"""TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32) of the disentangled-VAE training step of
v-manhlt3/Disentangle-VAE-for-VC.  It is the checker for the HIP path; the
product package never imports it.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import anything from here.

Parity status: PINNED against the imported reference (see
tests/golden/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).
"""
