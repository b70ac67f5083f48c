#!/bin/bash
# GPU box, round 6: the round-5 tree and HEAD on the SAME box, back to back.  Prepare in the build container first:
#   mkdir -p build_exp/r5 && git archive 0a3c193 | tar -x -C build_exp/r5 && (cd build_exp/r5 && bash disentangle-vae-for-vc_amd/csrc/build.sh)
# (build_exp/ is git-ignored but travels with gpurun)
(cd build_exp/r5 && timeout 600 python bench.py --no-cpu-baseline > ../../gpurun_out/bench_r5.json 2> ../../gpurun_out/bench_r5.err)
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_r5.json", "gpurun_out/bench_b.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line:", e); continue
    print(f, round(d["ms_per_step"], 3), round(d["value"], 1), round(d["roofline"]["frac"], 4), round(d["roofline"]["kernel_ms_per_step"], 3),
          d["roofline"]["launches_per_step"], "lstm", d.get("roofline_lstm", {}).get("kernel_ms_per_step"))
    for i in d["roofline"]["instantiations"]:
        print("   ", round(i["ms_per_step"], 3), round(i["tflops"], 1), i["launches_per_step"], i["kernel"])
    for i in d.get("roofline_lstm", {}).get("instantiations", []):
        print("   L ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in i.items() if k in ("kernel", "ms_per_step", "us_per_frame", "launches_per_step")})
    for oc in d.get("other_configs", []):
        print(oc["config"], oc.get("ms_per_step"), oc.get("roofline", {}).get("frac"), oc.get("roofline", {}).get("kernel_ms_per_step"),
              "lstm", oc.get("roofline_lstm", {}).get("kernel_ms_per_step"))
        for i in oc.get("roofline", {}).get("instantiations", []):
            print("   ", round(i["ms_per_step"], 3), round(i["tflops"], 1), i["kernel"])
PY
timeout 600 python scripts/g256_check.py time 60 2>&1 | grep -i "wgrad"
