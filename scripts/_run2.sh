PERS=1 python3 scripts/train_soak.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -2
PERS=0 python3 scripts/train_soak.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -2
PERS=1 GRAPH=0 python3 scripts/train_soak.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -2
