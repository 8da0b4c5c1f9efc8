"""Dev tool: print the stage timings of bench runs for the library selected by GS2D_LIB_PATH (two runs: the second one's
numbers are the ones to read, the first also warms the box)."""
import json, os, subprocess, sys
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-extra-legs"],
                         capture_output=True, text=True).stdout
    j = json.loads(out.strip().splitlines()[-1])
    sm = j["roofline"]["stage_ms"]
    print(os.path.basename(os.environ.get("GS2D_LIB_PATH", "product")), j["value"], j["ms_per_step"], "fwd", sm["blend_fwd"], "bwd", sm["blend_bwd"],
          "sort", sm["sort"], flush=True)
