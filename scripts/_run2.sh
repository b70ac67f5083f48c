python3 scripts/skinny_wgrad_sweep.py 2>&1 | grep -v amdgpu | tail -8
