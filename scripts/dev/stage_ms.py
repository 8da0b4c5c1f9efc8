"""Dev tool: print the stage timings of one bench run for the library selected by GS2D_LIB_PATH."""
import json, os, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True).stdout
j = json.loads(out.strip().splitlines()[-1])
print(os.environ.get("GS2D_LIB_PATH", "product"), j["ms_per_step"], j["roofline"]["stage_ms"])
