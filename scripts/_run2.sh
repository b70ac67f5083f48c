B=128 T=256 DVAE_COMPUTE_DTYPE=bf16 python3 scripts/gemm_shapes.py 2>&1 | grep -v amdgpu | head -8
