"""Dev tool: print the stage timings of bench runs for the library selected by GS2D_LIB_PATH (two runs: the second one's
numbers are the ones to read, the first also warms the box).  usage: stage_ms.py [runs] [extra bench.py arguments ...]"""
import json, os, subprocess, sys
extra = sys.argv[2:]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-extra-legs"] + extra,
                         capture_output=True, text=True).stdout
    j = json.loads(out.strip().splitlines()[-1])
    sm = (j.get("roofline") or {}).get("stage_ms") or (j.get("config") or {}).get("stage_ms") or {}
    print(os.path.basename(os.environ.get("GS2D_LIB_PATH", "product")), j["value"], j["ms_per_step"],
          " ".join(f"{k} {v}" for k, v in sm.items() if v), flush=True)
