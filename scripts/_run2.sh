python3 scripts/_dbg8.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -24
