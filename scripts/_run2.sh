python3 scripts/wgrad_split_sweep.py 2>&1 | grep -v "amdgpu.ids" | tail -12
