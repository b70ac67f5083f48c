#!/bin/bash
# Piggy-backed on GPU calls (round 5): does THIS box show the 16-row forward failure?  The sentinel form (dev build) is the more
# sensitive probe (failed 7 / 200 on one chip); if it fails, the dump variants run right away on the same chip.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/hunt
export DVAE_LIB_PATH=$PWD/disentangle-vae-for-vc_amd/libdvae_dev.so
tag=$(date +%H%M%S)
log=gpurun_out/hunt/$tag.log
python3 - > $log 2>&1 <<'PY'
import glob
for p in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
    try:
        kv = dict(l.split()[:2] for l in open(p) if len(l.split()) >= 2)
        if int(kv.get("simd_count", "0")) > 0: print("box gpu unique_id", hex(int(kv.get("unique_id", "0"))))
    except Exception as e:
        print("no topology:", e)
PY
DVAE_PERS_SENT=1 DVAE_PERS_X3_MT1=3 timeout 300 python scripts/x3_handoff_stress.py 512 96 128 300 2>&1 | grep -v amdgpu.ids | tail -4 >> $log
DVAE_PERS_SENT=0 DVAE_PERS_X3_MT1=3 timeout 300 python scripts/x3_handoff_stress.py 512 96 128 300 2>&1 | grep -v amdgpu.ids | tail -2 >> $log
if grep -q "[1-9][0-9]* bad rounds" $log; then
  echo "THIS BOX FAILS: dumps" >> $log
  for v in "--sent 1 --fresh" "--sent 1 --fresh --nslot 96" "--sent 1 --fresh --uncached" "--sent 0 --fresh" "--sent 0 --fresh --poison"; do
    echo "== $v" >> $log
    timeout 300 python scripts/x3_fwd16_diag2.py --rounds 500 --maxbad 5 $v 2>&1 | grep -v amdgpu.ids | tail -40 >> $log
  done
fi
echo "hunt16: $(grep -c 'bad rounds' $log) runs, $(grep 'bad rounds' $log | tr '\n' ' ')"
